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

constexpr int kSortTile = 8192;  // elements per workgroup of the radix passes: 256 threads x 32 (a wave owns 2048 consecutive ones)
size_t sort_tiles_of(size_t n) { return (n + kSortTile - 1) / kSortTile; }

// digit counts of one tile -> table[digit * ntiles + tile] (digit-major: its exclusive scan is, for every (digit, tile), the
// number of keys with a smaller digit plus those with the same digit in earlier tiles = where the tile's run of that digit starts)
template <class KeyT>
__global__ __launch_bounds__(256) void radix_hist_kernel(const KeyT* __restrict__ keys, size_t n, int shift, unsigned* __restrict__ table,
                                                         unsigned ntiles)
{
    __shared__ unsigned h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const size_t t0 = blockIdx.x * (size_t)kSortTile;
#pragma unroll 8
    for (int j = 0; j < kSortTile / 256; ++j) {
        const size_t i = t0 + j * 256 + threadIdx.x;
        if (i < n) atomicAdd(&h[(unsigned)(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    table[(size_t)threadIdx.x * ntiles + blockIdx.x] = h[threadIdx.x];
}

// Stable scatter of one tile, in two steps so that the global writes are RUNS instead of single words:
//   1. the tile is sorted by digit into LDS.  Wave w owns elements [2048 w, 2048 w + 2048) and walks them in order, 64 at a time: a
//      key's place = start of its digit within the tile + the same-digit keys of the earlier waves + those this wave has already
//      placed + its rank among the same-digit lanes below it (match mask from eight ballots);
//   2. thread t copies LDS slots t, t + 256, ... to where the tile's run of that digit starts in the output (scanned table): with 8192
//      keys and 256 digits a run is ~32 keys = 128 contiguous bytes per array.
// (The first form wrote every key straight to its global place: 64 different cache lines per store instruction, 1.2 TB/s per pass.)
template <class KeyT>
__global__ __launch_bounds__(256) void radix_scatter_kernel(const KeyT* __restrict__ kin, const unsigned* __restrict__ vin,
                                                            KeyT* __restrict__ kout, unsigned* __restrict__ vout, size_t n, int shift,
                                                            const unsigned* __restrict__ start /* scanned table */, unsigned ntiles)
{
    constexpr int ROUNDS = kSortTile / 256;  // 64-key rounds per wave
    __shared__ unsigned s_cnt[4][256], s_off[4][256], s_delta[256], s_w[4];
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    KeyT* sk = reinterpret_cast<KeyT*>(s_raw);
    unsigned* sv = reinterpret_cast<unsigned*>(s_raw + sizeof(KeyT) * kSortTile);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t t0 = blockIdx.x * (size_t)kSortTile;
    const size_t w0 = t0 + (size_t)wave * (kSortTile / 4);
    const unsigned tile_n = (unsigned)(n - t0 < (size_t)kSortTile ? n - t0 : (size_t)kSortTile);
    KeyT k[ROUNDS];
#pragma unroll
    for (int s = 0; s < ROUNDS; ++s) {
        const size_t i = w0 + s * 64 + lane;
        k[s] = i < n ? kin[i] : KeyT(0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) s_cnt[wave][j * 64 + lane] = 0;
    // (a wave's LDS operations execute in order: its own zeroes precede its own atomics)
#pragma unroll
    for (int s = 0; s < ROUNDS; ++s)
        if (w0 + s * 64 + lane < n) atomicAdd(&s_cnt[wave][(unsigned)(k[s] >> shift) & 255u], 1u);
    __syncthreads();
    {   // exclusive scan of the tile's digit totals (thread = digit), then every wave's first slot per digit
        const int d = threadIdx.x;
        const unsigned c0 = s_cnt[0][d], c1 = s_cnt[1][d], c2 = s_cnt[2][d], c3 = s_cnt[3][d];
        const unsigned tot = c0 + c1 + c2 + c3;
        const unsigned inc = wave_inclusive_sum(tot, lane);
        if (lane == 63) s_w[wave] = inc;
        __syncthreads();
        unsigned woff = 0;
        for (int w = 0; w < wave; ++w) woff += s_w[w];
        const unsigned loc = woff + inc - tot;  // where digit d starts inside the sorted tile
        s_off[0][d] = loc;
        s_off[1][d] = loc + c0;
        s_off[2][d] = loc + c0 + c1;
        s_off[3][d] = loc + c0 + c1 + c2;
        s_delta[d] = start[(size_t)d * ntiles + blockIdx.x] - loc;  // global position = delta + slot (unsigned wrap-around is fine)
    }
    __syncthreads();
    // (s_off[wave] is private to the wave from here on)
    const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll 4
    for (int s = 0; s < ROUNDS; ++s) {
        const size_t i = w0 + s * 64 + lane;
        const bool valid = i < n;
        const unsigned d = (unsigned)(k[s] >> shift) & 255u;
        unsigned long long same = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long has = __ballot((d >> b) & 1u);
            same &= ((d >> b) & 1u) ? has : ~has;
        }
        const unsigned rank = (unsigned)__popcll(same & below);
        if (valid) {
            const unsigned slot = s_off[wave][d] + rank;
            sk[slot] = k[s];
            sv[slot] = vin[i];
            if (rank == 0) s_off[wave][d] = slot + (unsigned)__popcll(same);  // the lowest lane of the group moves the cursor on
        }
    }
    __syncthreads();
    for (unsigned slot = threadIdx.x; slot < tile_n; slot += 256) {
        const KeyT key = sk[slot];
        const unsigned pos = s_delta[(unsigned)(key >> shift) & 255u] + slot;
        kout[pos] = key;
        vout[pos] = sv[slot];
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
    const size_t table = pad64(256 * sort_tiles_of(n));
    return table + scan_workspace_words(table);
}

template <class KeyT>
static int radix_sort_pairs(hipStream_t st, KeyT* k0, KeyT* k1, unsigned* v0, unsigned* v1, size_t n, int bits, unsigned* work)
{
    if (n == 0) return 0;
    const size_t nt = sort_tiles_of(n), table_n = 256 * nt;
    unsigned* table = work;
    unsigned* scan_work = work + pad64(table_n);
    KeyT* k[2] = {k0, k1};
    unsigned* v[2] = {v0, v1};
    const int passes = bits <= 8 ? 1 : (bits + 7) / 8;
    const size_t stage_bytes = (sizeof(KeyT) + sizeof(unsigned)) * kSortTile;  // the tile sorted in LDS: 64 KB (u32 keys) / 96 KB (u64)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(radix_scatter_kernel<KeyT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)stage_bytes);
    // (a refusal shows up as a launch error, which the callers pick up with hipGetLastError)
    int cur = 0;
    for (int p = 0; p < passes; ++p) {
        hipLaunchKernelGGL(radix_hist_kernel<KeyT>, dim3((unsigned)nt), dim3(256), 0, st, k[cur], n, 8 * p, table, (unsigned)nt);
        exclusive_scan_u32(st, table, table, table_n, scan_work);
        hipLaunchKernelGGL(radix_scatter_kernel<KeyT>, dim3((unsigned)nt), dim3(256), stage_bytes, st, k[cur], v[cur], k[cur ^ 1], v[cur ^ 1], n, 8 * p, table,
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
