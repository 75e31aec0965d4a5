// sortscan.hip -- exclusive prefix sum and stable LSD radix sort for the preparation ops (grid.hip: cell keys of
// grid_subsampling.cpp:51-54 sorted so that a cell's points are contiguous AND in input order; volume.hip: compaction of the
// voxels dataPrepareBraTS keeps).  HBM-bound integer work: every pass streams its arrays once, 16 bytes per lane where the
// layout allows, and the only cross-workgroup communication is a table of per-tile totals scanned by the next launch.
#include "sortscan.h"

namespace ps {

namespace {

constexpr int kTile = 2048;  // elements per workgroup: 256 threads x 8

__device__ __forceinline__ unsigned wave_inclusive_sum(unsigned v, int lane)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned u = (unsigned)__shfl_up((int)v, o);
        if (lane >= o) v += u;
    }
    return v;
}

// out = exclusive scan of the tile (without the tiles before it), sums[tile] = the tile's total; thread t owns eight
// CONSECUTIVE elements (two 16-byte accesses when the tile is full)
__global__ __launch_bounds__(256) void scan_tile_kernel(const unsigned* in, unsigned* out, size_t n, unsigned* __restrict__ sums)
{
    __shared__ unsigned s_w[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t base = blockIdx.x * (size_t)kTile + (size_t)threadIdx.x * 8;
    unsigned v[8];
    if (base + 8 <= n && ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0) {
        const uint4 a = *reinterpret_cast<const uint4*>(in + base), b = *reinterpret_cast<const uint4*>(in + base + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = base + j < n ? in[base + j] : 0u;
    }
    unsigned t = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const unsigned x = v[j];
        v[j] = t;
        t += x;
    }
    const unsigned inc = wave_inclusive_sum(t, lane);
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    unsigned woff = 0;
    for (int w = 0; w < wave; ++w) woff += s_w[w];
    const unsigned excl = woff + inc - t;
    if (base + 8 <= n && ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0) {
        *reinterpret_cast<uint4*>(out + base) = make_uint4(v[0] + excl, v[1] + excl, v[2] + excl, v[3] + excl);
        *reinterpret_cast<uint4*>(out + base + 4) = make_uint4(v[4] + excl, v[5] + excl, v[6] + excl, v[7] + excl);
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (base + j < n) out[base + j] = v[j] + excl;
    }
    if (threadIdx.x == 255) sums[blockIdx.x] = woff + inc;
}

__global__ __launch_bounds__(256) void scan_add_kernel(unsigned* __restrict__ out, size_t n, const unsigned* __restrict__ sums)
{
    const unsigned add = sums[blockIdx.x];
    const size_t base = blockIdx.x * (size_t)kTile + (size_t)threadIdx.x * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (base + j < n) out[base + j] += add;
}

size_t tiles_of(size_t n) { return (n + kTile - 1) / kTile; }
size_t pad64(size_t n) { return (n + 63) & ~size_t(63); }

// digit counts of one tile -> table[digit * ntiles + tile] (digit-major: its exclusive scan is, for every (digit, tile), the
// number of keys with a smaller digit plus those with the same digit in earlier tiles = where the tile's run of that digit starts)
template <class KeyT>
__global__ __launch_bounds__(256) void radix_hist_kernel(const KeyT* __restrict__ keys, size_t n, int shift, unsigned* __restrict__ table,
                                                         unsigned ntiles)
{
    __shared__ unsigned h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const size_t t0 = blockIdx.x * (size_t)kTile;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const size_t i = t0 + j * 256 + threadIdx.x;
        if (i < n) atomicAdd(&h[(unsigned)(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    table[(size_t)threadIdx.x * ntiles + blockIdx.x] = h[threadIdx.x];
}

// Stable scatter of one tile.  Wave w owns elements [512 w, 512 w + 512) of the tile and walks them in order, 64 at a time:
// a key's position = start of (digit, tile) + the same-digit keys of the earlier waves + those this wave has already placed +
// its rank among the same-digit lanes below it (match mask from eight ballots).
template <class KeyT>
__global__ __launch_bounds__(256) void radix_scatter_kernel(const KeyT* __restrict__ kin, const unsigned* __restrict__ vin,
                                                            KeyT* __restrict__ kout, unsigned* __restrict__ vout, size_t n, int shift,
                                                            const unsigned* __restrict__ start /* scanned table */, unsigned ntiles)
{
    __shared__ unsigned s_cnt[4][256], s_off[4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t w0 = blockIdx.x * (size_t)kTile + (size_t)wave * 512;
    KeyT k[8];
    unsigned v[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const size_t i = w0 + s * 64 + lane;
        k[s] = i < n ? kin[i] : KeyT(0);
        v[s] = i < n ? vin[i] : 0u;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) s_cnt[wave][j * 64 + lane] = 0;
    // (a wave's LDS operations execute in order: its own zeroes precede its own atomics)
#pragma unroll
    for (int s = 0; s < 8; ++s)
        if (w0 + s * 64 + lane < n) atomicAdd(&s_cnt[wave][(unsigned)(k[s] >> shift) & 255u], 1u);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int d = j * 64 + lane;
        unsigned off = start[(size_t)d * ntiles + blockIdx.x];
        for (int w = 0; w < wave; ++w) off += s_cnt[w][d];
        s_off[wave][d] = off;
    }
    // (s_off[wave] is private to the wave from here on)
    const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const bool valid = w0 + s * 64 + lane < n;
        const unsigned d = (unsigned)(k[s] >> shift) & 255u;
        unsigned long long same = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long has = __ballot((d >> b) & 1u);
            same &= ((d >> b) & 1u) ? has : ~has;
        }
        const unsigned rank = (unsigned)__popcll(same & below);
        const unsigned pos = s_off[wave][d] + rank;
        if (valid) {
            kout[pos] = k[s];
            vout[pos] = v[s];
            if (rank == 0) s_off[wave][d] = pos + (unsigned)__popcll(same);  // the lowest lane of the group moves the cursor on
        }
    }
}

}  // namespace

size_t scan_workspace_words(size_t n)
{
    size_t words = 64;
    for (size_t m = tiles_of(n); ; m = tiles_of(m)) {
        words += pad64(m);
        if (m <= 1) break;
    }
    return words;
}

void exclusive_scan_u32(hipStream_t st, const unsigned* in, unsigned* out, size_t n, unsigned* work)
{
    if (n == 0) return;
    const size_t nt = tiles_of(n);
    hipLaunchKernelGGL(scan_tile_kernel, dim3((unsigned)nt), dim3(256), 0, st, in, out, n, work);
    if (nt > 1) {
        exclusive_scan_u32(st, work, work, nt, work + pad64(nt));
        hipLaunchKernelGGL(scan_add_kernel, dim3((unsigned)nt), dim3(256), 0, st, out, n, work);
    }
}

size_t sort_workspace_words(size_t n)
{
    const size_t table = pad64(256 * tiles_of(n));
    return table + scan_workspace_words(table);
}

template <class KeyT>
static int radix_sort_pairs(hipStream_t st, KeyT* k0, KeyT* k1, unsigned* v0, unsigned* v1, size_t n, int bits, unsigned* work)
{
    if (n == 0) return 0;
    const size_t nt = tiles_of(n), table_n = 256 * nt;
    unsigned* table = work;
    unsigned* scan_work = work + pad64(table_n);
    KeyT* k[2] = {k0, k1};
    unsigned* v[2] = {v0, v1};
    const int passes = bits <= 8 ? 1 : (bits + 7) / 8;
    int cur = 0;
    for (int p = 0; p < passes; ++p) {
        hipLaunchKernelGGL(radix_hist_kernel<KeyT>, dim3((unsigned)nt), dim3(256), 0, st, k[cur], n, 8 * p, table, (unsigned)nt);
        exclusive_scan_u32(st, table, table, table_n, scan_work);
        hipLaunchKernelGGL(radix_scatter_kernel<KeyT>, dim3((unsigned)nt), dim3(256), 0, st, k[cur], v[cur], k[cur ^ 1], v[cur ^ 1], n, 8 * p, table,
                           (unsigned)nt);
        cur ^= 1;
    }
    return cur;
}

int radix_sort_pairs_u64(hipStream_t st, unsigned long long* k0, unsigned long long* k1, unsigned* v0, unsigned* v1, size_t n, int bits, unsigned* work)
{
    return radix_sort_pairs<unsigned long long>(st, k0, k1, v0, v1, n, bits, work);
}

int radix_sort_pairs_u32(hipStream_t st, unsigned* k0, unsigned* k1, unsigned* v0, unsigned* v1, size_t n, int bits, unsigned* work)
{
    return radix_sort_pairs<unsigned>(st, k0, k1, v0, v1, n, bits, work);
}

}  // namespace ps
