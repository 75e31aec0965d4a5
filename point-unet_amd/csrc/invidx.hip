// invidx.hip -- inverse index of a gather table and the gather-reductions built on it: the DETERMINISTIC backward of
// gather_neighbour / nearest_interpolation / random_sample (tf.batch_gather and tf.reduce_max in PointSegment/RandLANet.py:345-386).
//
// The backward of `out[r, :] = pc[b(r) * N + idx[r], :]` is a scatter-add, dpc[b * N + idx[r], :] += dout[r, :].  With float atomics
// the order of the additions -- and with it the last bits of every gradient -- changes from run to run.  Here the table is inverted once
// per pyramid level and step: offsets[j] .. offsets[j + 1] delimit the rows r that read source row j, in ASCENDING r (count with integer
// atomics, exclusive scan, fill, then a per-segment sort that removes the fill order), and every backward becomes a gather-reduction
// over that segment in that fixed order -- one coalesced read of every gradient row, one write per source row, no atomics.  Segments are
// short (a point sits in about K neighbour lists), wide rows (d >= 64 floats at levels 2-4) make the gather itself efficient.
#include "common.h"
#include "sortscan.h"
#include "bf16_io.h"

#include <algorithm>
#include <cstdlib>

namespace ps {

__global__ __launch_bounds__(256) void inv_count_kernel(const int32_t* __restrict__ idx, int64_t rows, int rows_per_cloud, int n_cloud,
                                                        unsigned* __restrict__ cnt)
{
    for (int64_t r = blockIdx.x * (int64_t)256 + threadIdx.x; r < rows; r += (int64_t)gridDim.x * 256)
        atomicAdd(&cnt[(r / rows_per_cloud) * n_cloud + idx[r]], 1u);  // (integer atomics: the counts do not depend on the order)
}

__global__ __launch_bounds__(256) void inv_fill_kernel(const int32_t* __restrict__ idx, int64_t rows, int rows_per_cloud, int n_cloud,
                                                       const unsigned* __restrict__ offsets, unsigned* __restrict__ cursor, int32_t* __restrict__ src)
{
    for (int64_t r = blockIdx.x * (int64_t)256 + threadIdx.x; r < rows; r += (int64_t)gridDim.x * 256) {
        const int64_t j = (r / rows_per_cloud) * n_cloud + idx[r];
        const unsigned slot = atomicAdd(&cursor[j], 1u);
        src[offsets[j] + slot] = (int32_t)r;
    }
}

// one thread per segment: ascending order (the fill order is whatever the atomics made it).  Segments of up to 32 entries are sorted in
// registers, longer ones (duplicate points, degenerate clouds) by insertion in place.
__global__ __launch_bounds__(256) void inv_sort_kernel(const unsigned* __restrict__ offsets, int64_t n_dst, int32_t* __restrict__ src)
{
    const int64_t j = blockIdx.x * (int64_t)256 + threadIdx.x;
    if (j >= n_dst) return;
    const unsigned lo = offsets[j], hi = offsets[j + 1];
    const int n = (int)(hi - lo);
    if (n < 2) return;
    int32_t* s = src + lo;
    for (int i = 1; i < n; ++i) {
        const int32_t v = s[i];
        int k = i - 1;
        while (k >= 0 && s[k] > v) {
            s[k + 1] = s[k];
            --k;
        }
        s[k + 1] = v;
    }
}

// ---- the same inverse index by a stable radix sort of (destination, row) pairs: for the multi-million-row tables of the shallow levels the
// two random-access atomics per row above cost 3 ms for 23 M rows; three 8-bit passes of the sort stream the pairs instead ----
__global__ __launch_bounds__(256) void inv_keys_kernel(const int32_t* __restrict__ idx, int64_t rows, int rows_per_cloud, int n_cloud,
                                                       unsigned* __restrict__ keys, unsigned* __restrict__ vals)
{
    for (int64_t r = blockIdx.x * (int64_t)256 + threadIdx.x; r < rows; r += (int64_t)gridDim.x * 256) {
        keys[r] = (unsigned)((r / rows_per_cloud) * n_cloud + idx[r]);
        vals[r] = (unsigned)r;
    }
}
// ---- the inverse index in ONE stable bucket pass + a sort inside every bucket (tables of >= kInvBucketRows rows) ----
// The radix sort above moves (key, row) pairs three times (8 B read + 8 B written per pass, a histogram read in front of each).  It does not
// use what the table is: rows come cloud by cloud (the high part of the key is sorted already), a key is < N, and a row number inside its
// cloud needs far fewer than 32 bits.  Here a destination id splits as (bucket = id >> lo, low = id & (2^lo - 1)) with at most 512 buckets
// per cloud:
//   1. bk_hist:    per 8192-row tile of a cloud, rows per bucket -> table[(cloud, bucket), tile]; an exclusive scan of the table is where
//                  every tile's run of every bucket starts (and, at tile 0, where the bucket itself starts);
//   2. bk_scatter: a tile is sorted by bucket in LDS (stable: wave-ordered ranks from ballot match masks, as radix_scatter_kernel does) and
//                  leaves as runs of ONE word per row, (row inside the cloud) << lo | low;
//   3. bk_local:   a workgroup owns a bucket (<= 512 destinations, ~8 K rows at level 0: L2 resident), counts per destination, scans,
//                  writes the destinations' offsets, places the rows -- stable again, so ascending inside a segment -- through an LDS image
//                  of the bucket and streams `src` out.
// 20 bytes per row instead of 72, four launches + the table scan instead of fifteen.  Identical output (offsets and src) to the two other
// forms.  Measured, batch 8 x 180 000 points: DESIGN 4.2.
constexpr int kBkTile = 4096;    // rows per workgroup of passes 1 and 2 (a wave owns 1024 consecutive rows).  PS_INV_TILE = 8192 | 6144 | 4096: measured 0.303 /
                                 // 0.267 / 0.250 ms for the 23 M-row table of level 0 (LDS per workgroup 58 / 46 / 34 KB: 2 / 3 / 4 workgroups per CU)
constexpr int kBkDigits = 512;   // buckets per cloud (pass 2) and destinations per bucket (pass 3) the LDS counters hold
constexpr int kBkStage = 10240;  // rows of a bucket sorted through LDS; a larger bucket (skewed tables) writes its rows straight to `src`

struct BkPlan {
    int lo, nb, tpc, nb_bits;
    size_t table;  // entries (the scan runs over table + 1)
};

__device__ __forceinline__ unsigned bk_wave_inclusive_sum(unsigned v, int lane)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned u = (unsigned)__shfl_up((int)v, o);
        if (lane >= o) v += u;
    }
    return v;
}

// (kBkHistTiles consecutive tiles per workgroup; 4 measured slower than 1 -- 76 against 50 us at level 0: fewer, longer workgroups)
constexpr int kBkHistTiles = 1;
template <int TILE>
__global__ __launch_bounds__(256) void bk_hist_kernel(const int32_t* __restrict__ idx, int rpc, int tpc, int lo, int nb, unsigned* __restrict__ table,
                                                      size_t table_n)
{
    __shared__ unsigned h[kBkHistTiles][kBkDigits];
#pragma unroll
    for (int j = 0; j < kBkHistTiles * kBkDigits / 256; ++j) (&h[0][0])[j * 256 + threadIdx.x] = 0;
    __syncthreads();
    const int groups = (tpc + kBkHistTiles - 1) / kBkHistTiles;
    const int b = blockIdx.x / groups, t0 = (blockIdx.x - b * groups) * kBkHistTiles;
    const int tiles = min(kBkHistTiles, tpc - t0);
    for (int q = 0; q < tiles; ++q) {
        const int r0 = (t0 + q) * TILE;
        const int n = min(TILE, rpc - r0);
        const int32_t* p = idx + (size_t)b * rpc + r0;
#pragma unroll 8
        for (int j = 0; j < TILE / 256; ++j) {
            const int i = j * 256 + threadIdx.x;
            if (i < n) atomicAdd(&h[q][min((unsigned)p[i] >> lo, (unsigned)nb - 1u)], 1u);  // (clamped: a value outside [0, N) stays memory-safe)
        }
    }
    __syncthreads();
    for (int d = threadIdx.x; d < nb; d += 256) {
        unsigned* o = table + ((size_t)b * nb + d) * tpc + t0;
        for (int q = 0; q < tiles; ++q) o[q] = h[q][d];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) table[table_n] = 0;  // scanned: the total, where the bucket after the last one would start
}

// exclusive scan over the 512 per-digit totals of a workgroup's four waves (thread t owns digits 2t, 2t+1), IN PLACE: s[w][d] = rows of digit
// d in wave w's share -> first slot of those rows inside the sorted tile / bucket; returns the slot where digit 2t starts (2t+1: + tot0)
__device__ __forceinline__ unsigned bk_scan_digits(unsigned (*s)[kBkDigits], unsigned* s_w, unsigned& tot0)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int d0 = 2 * threadIdx.x;
    uint2 c[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) c[w] = *reinterpret_cast<const uint2*>(&s[w][d0]);
    tot0 = c[0].x + c[1].x + c[2].x + c[3].x;
    const unsigned tot1 = c[0].y + c[1].y + c[2].y + c[3].y;
    const unsigned inc = bk_wave_inclusive_sum(tot0 + tot1, lane);
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    unsigned woff = 0;
    for (int w = 0; w < wave; ++w) woff += s_w[w];
    const unsigned loc0 = woff + inc - (tot0 + tot1);
    unsigned a0 = loc0, a1 = loc0 + tot0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        *reinterpret_cast<uint2*>(&s[w][d0]) = make_uint2(a0, a1);
        a0 += c[w].x;
        a1 += c[w].y;
    }
    return loc0;
}

template <int TILE>
__global__ __launch_bounds__(256) void bk_scatter_kernel(const int32_t* __restrict__ idx, int rpc, int tpc, int lo, int nb, int nb_bits,
                                                         const unsigned* __restrict__ start /* scanned table */, unsigned* __restrict__ payload)
{
    constexpr int ROUNDS = TILE / 256;
    __shared__ __attribute__((aligned(8))) unsigned s_off[4][kBkDigits];
    __shared__ unsigned s_delta[kBkDigits], s_w[4];
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    unsigned* sv = reinterpret_cast<unsigned*>(s_raw);
    unsigned short* sd = reinterpret_cast<unsigned short*>(s_raw + sizeof(unsigned) * TILE);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x / tpc, t = blockIdx.x - b * tpc;
    const int r0 = t * TILE;
    const int tile_n = min(TILE, rpc - r0);
    const int w0 = wave * (TILE / 4);
    const int32_t* p = idx + (size_t)b * rpc + r0;
    unsigned k[ROUNDS];
#pragma unroll
    for (int s = 0; s < ROUNDS; ++s) {
        const int i = w0 + s * 64 + lane;
        k[s] = i < tile_n ? (unsigned)p[i] : 0u;
    }
    // this thread's two entries of the scanned table (used after the digit scan: requested now, under the counting)
    const int d0 = 2 * threadIdx.x;
    const unsigned g0 = d0 < nb ? start[((size_t)b * nb + d0) * tpc + t] : 0u, g1 = d0 + 1 < nb ? start[((size_t)b * nb + d0 + 1) * tpc + t] : 0u;
#pragma unroll
    for (int j = 0; j < kBkDigits / 64; ++j) s_off[wave][j * 64 + lane] = 0;
    // (a wave's LDS operations execute in order: its own zeroes precede its own atomics)
#pragma unroll
    for (int s = 0; s < ROUNDS; ++s)
        if (w0 + s * 64 + lane < tile_n) atomicAdd(&s_off[wave][min(k[s] >> lo, (unsigned)nb - 1u)], 1u);
    __syncthreads();
    {
        unsigned tot0;
        const unsigned loc0 = bk_scan_digits(s_off, s_w, tot0);
        // global position = delta + slot (unsigned wrap-around is fine)
        s_delta[d0] = g0 - loc0;
        s_delta[d0 + 1] = g1 - (loc0 + tot0);
    }
    __syncthreads();
    // (s_off[wave] is private to the wave from here on)
    const unsigned long long below = (1ull << lane) - 1ull;
    const unsigned low_mask = (1u << lo) - 1u;
#pragma unroll 4
    for (int s = 0; s < ROUNDS; ++s) {
        const int i = w0 + s * 64 + lane;
        const bool valid = i < tile_n;
        const unsigned d = min(k[s] >> lo, (unsigned)nb - 1u);
        unsigned long long same = __ballot(valid);
        for (int bit = 0; bit < nb_bits; ++bit) {
            const unsigned long long has = __ballot((d >> bit) & 1u);
            same &= ((d >> bit) & 1u) ? has : ~has;
        }
        const unsigned rank = (unsigned)__popcll(same & below);
        if (valid) {
            const unsigned slot = s_off[wave][d] + rank;
            sv[slot] = ((unsigned)(r0 + i) << lo) | (k[s] & low_mask);
            sd[slot] = (unsigned short)d;
            if (rank == 0) s_off[wave][d] = slot + (unsigned)__popcll(same);  // the lowest lane of the group moves the cursor on
        }
    }
    __syncthreads();
    if (tile_n == TILE) {
#pragma unroll 8
        for (int j = 0; j < ROUNDS; ++j) {
            const int slot = j * 256 + threadIdx.x;
            payload[s_delta[sd[slot]] + slot] = sv[slot];
        }
    } else {
        for (int slot = threadIdx.x; slot < tile_n; slot += 256) payload[s_delta[sd[slot]] + slot] = sv[slot];
    }
}

__global__ __launch_bounds__(256) void bk_local_kernel(const unsigned* __restrict__ payload, const unsigned* __restrict__ start, int rpc, int tpc, int lo,
                                                       int nb, int n_cloud, int64_t n_dst, unsigned rows, unsigned* __restrict__ offsets,
                                                       int32_t* __restrict__ src)
{
    constexpr int MAXR = kBkStage / 256;  // rounds of 64 a wave's share of a staged bucket has at most
    __shared__ __attribute__((aligned(8))) unsigned s_off[4][kBkDigits];
    __shared__ unsigned s_w[4];
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    unsigned* sv = reinterpret_cast<unsigned*>(s_raw);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x / nb, d = blockIdx.x - b * nb;
    const unsigned first = start[(size_t)blockIdx.x * tpc], n = start[((size_t)blockIdx.x + 1) * tpc] - first;
    const unsigned mask = (1u << lo) - 1u;
    const unsigned* pl = payload + first;
    const bool staged = n <= (unsigned)kBkStage;
    const unsigned chunk = ((n + 255u) / 256u) * 64u;  // a wave's share of the bucket: consecutive rows, whole rounds of 64
    const unsigned w_lo = min(n, (unsigned)wave * chunk), w_hi = min(n, w_lo + chunk);
    // a staged bucket's rows are read ONCE, all loads in flight together (a loop of dependent loads per round is one memory latency per 64 rows)
    unsigned k[MAXR];
    if (staged) {
#pragma unroll
        for (int r = 0; r < MAXR; ++r) {
            const unsigned i = w_lo + r * 64 + lane;
            k[r] = i < w_hi ? pl[i] : 0u;
        }
    }
#pragma unroll
    for (int j = 0; j < kBkDigits / 64; ++j) s_off[wave][j * 64 + lane] = 0;
    if (staged) {
#pragma unroll
        for (int r = 0; r < MAXR; ++r)
            if (w_lo + r * 64 + lane < w_hi) atomicAdd(&s_off[wave][k[r] & mask], 1u);
    } else {
        for (unsigned i = w_lo + lane; i < w_hi; i += 64) atomicAdd(&s_off[wave][pl[i] & mask], 1u);
    }
    __syncthreads();
    {
        unsigned tot0;
        const unsigned loc0 = bk_scan_digits(s_off, s_w, tot0);
        const int id0 = (d << lo) + 2 * (int)threadIdx.x;  // (digits >= 2^lo hold nothing: their ids belong to the next bucket, which writes them)
        if (2 * threadIdx.x < (1u << lo)) {
            unsigned* o = offsets + (size_t)b * n_cloud;
            if (id0 < n_cloud) o[id0] = first + loc0;
            if (id0 + 1 < n_cloud && 2 * threadIdx.x + 1 < (1u << lo)) o[id0 + 1] = first + loc0 + tot0;
        }
        if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) offsets[n_dst] = rows;
    }
    __syncthreads();
    const unsigned long long below = (1ull << lane) - 1ull;
    const unsigned row0 = (unsigned)b * (unsigned)rpc;
    auto place = [&](unsigned pv, bool valid, bool to_lds) {
        const unsigned key = pv & mask;
        unsigned long long same = __ballot(valid);
        for (int bit = 0; bit < lo; ++bit) {
            const unsigned long long has = __ballot((key >> bit) & 1u);
            same &= ((key >> bit) & 1u) ? has : ~has;
        }
        const unsigned rank = (unsigned)__popcll(same & below);
        if (valid) {
            const unsigned slot = s_off[wave][key] + rank;
            const unsigned val = row0 + (pv >> lo);
            if (to_lds)
                sv[slot] = val;
            else
                src[first + slot] = (int32_t)val;
            if (rank == 0) s_off[wave][key] = slot + (unsigned)__popcll(same);
        }
    };
    if (!staged) {
        for (unsigned base = w_lo; base < w_hi; base += 64) {
            const unsigned i = base + lane;
            place(i < w_hi ? pl[i] : 0u, i < w_hi, false);
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < MAXR; ++r) {
        if (w_lo + r * 64 >= w_hi) break;  // (uniform per wave)
        place(k[r], w_lo + r * 64 + lane < w_hi, true);
    }
    __syncthreads();
#pragma unroll 4
    for (unsigned i = threadIdx.x; i < n; i += 256) src[first + i] = (int32_t)sv[i];
}

// sorted keys -> offsets: position i opens the segments of every destination in (keys[i-1], keys[i]]; the last position closes the rest
__global__ __launch_bounds__(256) void inv_offsets_kernel(const unsigned* __restrict__ keys, int64_t rows, int64_t n_dst, unsigned* __restrict__ offsets)
{
    for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < rows; i += (int64_t)gridDim.x * 256) {
        const long long key = (long long)keys[i], prev = i > 0 ? (long long)keys[i - 1] : -1;
        for (long long q = prev + 1; q <= key; ++q) offsets[q] = (unsigned)i;
        if (i == rows - 1)
            for (long long q = key + 1; q <= n_dst; ++q) offsets[q] = (unsigned)rows;
    }
}

// dst[j, :] (+)= sum over the segment of j of rows[src, :], in segment order.  One lane group of d/4 (VEC) or d lanes per destination row.
// A lane walks its segment alone, so the loop is written four sources at a time: four independent index loads, then four independent row
// loads, then the four additions in segment order (a one-source-per-iteration loop is a chain of dependent loads: 2 x 16 round trips per
// destination row at a few waves per SIMD).
template <bool VEC>
__global__ __launch_bounds__(256) void gather_reduce_kernel(const float* __restrict__ rows, int64_t ldr, const unsigned* __restrict__ offsets,
                                                            const int32_t* __restrict__ src, int64_t n_dst, int d, float* __restrict__ dst, int64_t ldd,
                                                            int accumulate)
{
    const unsigned per = VEC ? d / 4 : d;
    const int64_t total = n_dst * per;
    for (int64_t t = blockIdx.x * (int64_t)256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int64_t j = total < (1ll << 32) ? (int64_t)((unsigned)t / per) : t / per;
        const int q = (int)(t - j * per);
        const unsigned lo = offsets[j], hi = offsets[j + 1];
        if (VEC) {
            const float* base = rows + 4 * q;
            float4 acc = accumulate ? *reinterpret_cast<const float4*>(dst + j * ldd + 4 * q) : float4{0.f, 0.f, 0.f, 0.f};
            unsigned s = lo;
            for (; s + 4 <= hi; s += 4) {
                const int32_t i0 = src[s], i1 = src[s + 1], i2 = src[s + 2], i3 = src[s + 3];
                const float4 v0 = *reinterpret_cast<const float4*>(base + (int64_t)i0 * ldr);
                const float4 v1 = *reinterpret_cast<const float4*>(base + (int64_t)i1 * ldr);
                const float4 v2 = *reinterpret_cast<const float4*>(base + (int64_t)i2 * ldr);
                const float4 v3 = *reinterpret_cast<const float4*>(base + (int64_t)i3 * ldr);
                acc.x += v0.x; acc.y += v0.y; acc.z += v0.z; acc.w += v0.w;
                acc.x += v1.x; acc.y += v1.y; acc.z += v1.z; acc.w += v1.w;
                acc.x += v2.x; acc.y += v2.y; acc.z += v2.z; acc.w += v2.w;
                acc.x += v3.x; acc.y += v3.y; acc.z += v3.z; acc.w += v3.w;
            }
            for (; s < hi; ++s) {
                const float4 v = *reinterpret_cast<const float4*>(base + (int64_t)src[s] * ldr);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
            *reinterpret_cast<float4*>(dst + j * ldd + 4 * q) = acc;
        } else {
            float acc = accumulate ? dst[j * ldd + q] : 0.f;
            for (unsigned s = lo; s < hi; ++s) acc += rows[(int64_t)src[s] * ldr + q];
            dst[j * ldd + q] = acc;
        }
    }
}

// The same reduction with the destinations walked in a SPATIALLY COHERENT order (order[b, t] = row of cloud b's t-th point in kd-tree leaf
// order, ps_pyramid.order), one contiguous eighth of that order per XCD.  A destination's segment lists rows (q, k) of its spatial
// neighbours q; the K rows of one q form one 64 x d / 16-byte block whose other rows belong to destinations next to this one -- walked in
// cloud order (a shuffled cloud: runBraTS.py:114) every 32-byte row of level 0 costs its own memory transaction, walked in leaf order the
// block is fetched once into the XCD's L2 and its rows are consumed by the neighbouring lanes and workgroups.  Same sums, same order.
// RB: the rows are bfloat16 (ldr in elements): the gathered half's gradient rows of the bf16-MLP mode (ps_set_train_act_bf16)
template <bool RB>
__global__ __launch_bounds__(256) void gather_reduce_ordered_kernel(const float* __restrict__ rows, int64_t ldr, const unsigned* __restrict__ offsets,
                                                                    const int32_t* __restrict__ src, const int32_t* __restrict__ order, unsigned n_dst,
                                                                    unsigned n_cloud, unsigned per /* d / 4 lanes per destination */,
                                                                    float* __restrict__ dst, int64_t ldd, int accumulate)
{
    const unsigned dpw = 256u / per;                                      // destinations per workgroup and pass (per divides 256: host)
    const unsigned per_xcd = (((n_dst + 7u) >> 3) + dpw - 1u) / dpw * dpw;
    const unsigned xcd = blockIdx.x & 7u, slots = gridDim.x >> 3;         // (the host launches a multiple of 8 workgroups)
    const unsigned end = min(n_dst, (xcd + 1u) * per_xcd);
    const unsigned sub = threadIdx.x / per, q = threadIdx.x - sub * per;
    for (unsigned t = xcd * per_xcd + (blockIdx.x >> 3) * dpw + sub; t < end; t += slots * dpw) {
        const unsigned b = t / n_cloud;
        const int64_t j = order ? (int64_t)b * n_cloud + order[t] : (int64_t)t;
        const unsigned lo = offsets[j], hi = offsets[j + 1];
        float4 acc = accumulate ? *reinterpret_cast<const float4*>(dst + j * ldd + 4 * q) : float4{0.f, 0.f, 0.f, 0.f};
        unsigned s = lo;
        for (; s + 4 <= hi; s += 4) {
            const int32_t i0 = src[s], i1 = src[s + 1], i2 = src[s + 2], i3 = src[s + 3];
            const float4 v0 = load4_any(rows, (size_t)((int64_t)i0 * ldr) + 4 * q, RB);
            const float4 v1 = load4_any(rows, (size_t)((int64_t)i1 * ldr) + 4 * q, RB);
            const float4 v2 = load4_any(rows, (size_t)((int64_t)i2 * ldr) + 4 * q, RB);
            const float4 v3 = load4_any(rows, (size_t)((int64_t)i3 * ldr) + 4 * q, RB);
            acc.x += v0.x; acc.y += v0.y; acc.z += v0.z; acc.w += v0.w;
            acc.x += v1.x; acc.y += v1.y; acc.z += v1.z; acc.w += v1.w;
            acc.x += v2.x; acc.y += v2.y; acc.z += v2.z; acc.w += v2.w;
            acc.x += v3.x; acc.y += v3.y; acc.z += v3.z; acc.w += v3.w;
        }
        for (; s < hi; ++s) {
            const float4 v = load4_any(rows, (size_t)((int64_t)src[s] * ldr) + 4 * q, RB);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        *reinterpret_cast<float4*>(dst + j * ldd + 4 * q) = acc;
    }
}

// random_sample backward, step 1: share[m, c] = dout[m, c] / (number of the K gathered rows that attain the maximum) -- ties share the
// gradient evenly like tf.reduce_max's
__global__ __launch_bounds__(256) void maxpool_share_kernel(const float* __restrict__ dout, const float* __restrict__ out, const float* __restrict__ feat,
                                                            const int32_t* __restrict__ idx, size_t rows, int m_cloud, int n_cloud, int K, int d,
                                                            float* __restrict__ share)
{
    const size_t t = blockIdx.x * (size_t)256 + threadIdx.x;
    if (t >= rows * d) return;
    const size_t row = t / d;
    const int ch = (int)(t - row * d);
    const size_t base = (row / m_cloud) * n_cloud;
    const int32_t* ix = idx + row * K;
    const float mx = out[t];
    int ties = 0;
    for (int k = 0; k < K; ++k) ties += feat[(base + ix[k]) * d + ch] == mx;
    share[t] = dout[t] / (float)ties;
}
// step 2: dfeat[j, c] += sum over the (m, k) that gathered row j, in ascending order, of share[m, c] where feat[j, c] attains out[m, c].
// The inverse index is that of the level's NEIGHBOUR table [B, N, K]: the pooling table is its first M rows per cloud (sub_idx =
// neigh_idx[:, :M], runBraTS.py:150), a segment lists its rows in ascending order and all of them belong to the cloud of j -- so the
// pooling rows of a segment are a PREFIX of it and no second index is needed.
template <bool TIES>
__global__ __launch_bounds__(256) void maxpool_bwd_inv_kernel(const float* __restrict__ share, const unsigned char* __restrict__ ties,
                                                              const float* __restrict__ out, const float* __restrict__ feat,
                                                              const unsigned* __restrict__ offsets, const int32_t* __restrict__ src, int64_t n_dst, int n_cloud,
                                                              int m_cloud, int K, int d, float* __restrict__ dfeat)
{
    // TIES: `share` is dout itself and ties[m, c] (written by the forward, ps_op_random_sample_ties) the number of rows that attain the
    // maximum -- no separate pass that re-gathers the K rows of every (m, c) to count them
    const int64_t total = n_dst * d;
    for (int64_t t = blockIdx.x * (int64_t)256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int64_t j = t / d;
        const int ch = (int)(t - j * d);
        const unsigned lo = offsets[j], hi = offsets[j + 1];
        if (lo == hi) continue;
        const int64_t b = j / n_cloud;
        const float f = feat[t];
        float acc = 0.f;
        for (unsigned s = lo; s < hi; ++s) {
            const int64_t n = src[s] / K - b * n_cloud;  // src holds flat (point * K + k) positions of the neighbour table
            if (n >= m_cloud) break;                      // (ascending: the rest of the segment is not part of the pooling table)
            const int64_t m = b * m_cloud + n;
            if (out[m * d + ch] == f) acc += TIES ? share[m * d + ch] / (float)ties[m * d + ch] : share[m * d + ch];
        }
        dfeat[t] += acc;
    }
}

// The same walk with four channels per lane (d % 4 == 0, 32-bit index arithmetic): the segment bookkeeping -- two divisions per lane in the
// scalar form above -- is shared by four channels, rows travel as 16-byte loads and the tie counts as one 4-byte load.
__global__ __launch_bounds__(256) void maxpool_bwd_inv4_kernel(const float* __restrict__ dout, const unsigned char* __restrict__ ties,
                                                               const float* __restrict__ out, const float* __restrict__ feat,
                                                               const unsigned* __restrict__ offsets, const int32_t* __restrict__ src, unsigned n_dst,
                                                               unsigned n_cloud, unsigned m_cloud, unsigned K, unsigned d4, float* __restrict__ dfeat,
                                                               int overwrite)
{
    const unsigned total = n_dst * d4;
    const float4* out4 = reinterpret_cast<const float4*>(out);
    const float4* dout4 = reinterpret_cast<const float4*>(dout);
    const uchar4* ties4 = reinterpret_cast<const uchar4*>(ties);
    for (unsigned t = blockIdx.x * 256u + threadIdx.x; t < total; t += gridDim.x * 256u) {
        const unsigned j = t / d4, q = t - j * d4;
        const unsigned lo = offsets[j], hi = offsets[j + 1];
        if (lo == hi) {
            if (overwrite) reinterpret_cast<float4*>(dfeat)[t] = make_float4(0.f, 0.f, 0.f, 0.f);
            continue;
        }
        const unsigned b = j / n_cloud;
        const unsigned first = b * n_cloud, mbase = b * m_cloud;
        const float4 f = reinterpret_cast<const float4*>(feat)[t];
        float4 acc = {0.f, 0.f, 0.f, 0.f};
        for (unsigned s = lo; s < hi; ++s) {
            const unsigned n = (unsigned)src[s] / K - first;  // src holds flat (point * K + k) positions of the neighbour table
            if (n >= m_cloud) break;                           // (ascending: the rest of the segment is not part of the pooling table)
            const unsigned e = (mbase + n) * d4 + q;
            const float4 o = out4[e];
            if (o.x == f.x || o.y == f.y || o.z == f.z || o.w == f.w) {
                const float4 g = dout4[e];
                const uchar4 c = ties4[e];
                if (o.x == f.x) acc.x += g.x / (float)c.x;
                if (o.y == f.y) acc.y += g.y / (float)c.y;
                if (o.z == f.z) acc.z += g.z / (float)c.z;
                if (o.w == f.w) acc.w += g.w / (float)c.w;
            }
        }
        float4* dst = reinterpret_cast<float4*>(dfeat) + t;
        if (overwrite) {
            *dst = acc;
        } else {
            float4 h = *dst;
            h.x += acc.x; h.y += acc.y; h.z += acc.z; h.w += acc.w;
            *dst = h;
        }
    }
}

// maxpool_bwd_inv4_kernel with (a) the destinations walked in a spatially coherent order, one contiguous eighth per XCD (the rows of `out` /
// `dout` a destination compares against are those of the pooled points around it, each shared by its K gatherers: L2 hits instead of one
// memory transaction per compare) and (b) four segment entries per iteration: their index loads, then their `out` loads, are independent --
// the one-entry loop is a chain of three dependent round trips per entry.  Same additions in the same order.
__global__ __launch_bounds__(256) void maxpool_bwd_inv4_ordered_kernel(const float* __restrict__ dout, const unsigned char* __restrict__ ties,
                                                                       const float* __restrict__ out, const float* __restrict__ feat,
                                                                       const unsigned* __restrict__ offsets, const int32_t* __restrict__ src,
                                                                       const int32_t* __restrict__ order, unsigned n_dst, unsigned n_cloud, unsigned m_cloud,
                                                                       unsigned K, unsigned d4, float* __restrict__ dfeat, int overwrite)
{
    const float4* out4 = reinterpret_cast<const float4*>(out);
    const float4* dout4 = reinterpret_cast<const float4*>(dout);
    const uchar4* ties4 = reinterpret_cast<const uchar4*>(ties);
    const unsigned dpw = 256u / d4;
    const unsigned per_xcd = (((n_dst + 7u) >> 3) + dpw - 1u) / dpw * dpw;
    const unsigned xcd = blockIdx.x & 7u, slots = gridDim.x >> 3;
    const unsigned end = min(n_dst, (xcd + 1u) * per_xcd);
    const unsigned sub = threadIdx.x / d4, q = threadIdx.x - sub * d4;
    for (unsigned w = xcd * per_xcd + (blockIdx.x >> 3) * dpw + sub; w < end; w += slots * dpw) {
        const unsigned b = w / n_cloud;
        const unsigned first = b * n_cloud, mbase = b * m_cloud;
        const unsigned j = order ? first + (unsigned)order[w] : w;
        const unsigned t = j * d4 + q;
        const unsigned lo = offsets[j], hi = offsets[j + 1];
        float4 acc = {0.f, 0.f, 0.f, 0.f};
        if (lo != hi) {
            const float4 f = reinterpret_cast<const float4*>(feat)[t];
            for (unsigned s = lo; s < hi; s += 4) {
                unsigned n[4];
                float4 o[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) n[u] = s + u < hi ? (unsigned)src[s + u] / K - first : 0xffffffffu;  // src: flat (point * K + k) positions
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (n[u] < m_cloud) o[u] = out4[(mbase + n[u]) * d4 + q];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (n[u] >= m_cloud) break;  // (ascending: the rest of the segment is not part of the pooling table)
                    if (o[u].x == f.x || o[u].y == f.y || o[u].z == f.z || o[u].w == f.w) {
                        const unsigned e = (mbase + n[u]) * d4 + q;
                        const float4 g = dout4[e];
                        const uchar4 c = ties4[e];
                        if (o[u].x == f.x) acc.x += g.x / (float)c.x;
                        if (o[u].y == f.y) acc.y += g.y / (float)c.y;
                        if (o[u].z == f.z) acc.z += g.z / (float)c.z;
                        if (o[u].w == f.w) acc.w += g.w / (float)c.w;
                    }
                }
                if (n[3] >= m_cloud) break;
            }
        }
        float4* dst = reinterpret_cast<float4*>(dfeat) + t;
        if (overwrite) {
            *dst = acc;
        } else if (lo != hi) {
            float4 h = *dst;
            h.x += acc.x; h.y += acc.y; h.z += acc.z; h.w += acc.w;
            *dst = h;
        }
    }
}

static inline unsigned iv_grid(int64_t n)
{
    const int64_t b = (n + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

}  // namespace ps

using namespace ps;

extern "C" {

static constexpr int64_t kInvSortRows = 1 << 11;  // tables of at least this many rows are inverted by the bucket pass (or, when its plan does not fit, the radix sort)

// the bucket form's shape for a table, or false: more than 512 x 512 destinations per cloud, or (row inside the cloud, low id bits) past 32 bits
static bool bk_plan(int64_t B, int64_t N, int64_t rpc, int tile, BkPlan* pl)
{
    int lo = 0;
    while (((N + (1ll << lo) - 1) >> lo) > kBkDigits) ++lo;  // at most 512 buckets per cloud
    // ... of ~1024 rows or more each where the table has them (a bucket is a workgroup of pass 3), within the 512 destinations a bucket may hold
    while (lo < 9 && (rpc << lo) < 1024 * N) ++lo;
    if (lo > 9) return false;
    int rbits = 0;
    while ((1ll << rbits) < rpc) ++rbits;
    if (rbits + lo > 32) return false;
    pl->lo = lo;
    pl->nb = (int)((N + (1ll << lo) - 1) >> lo);
    pl->nb_bits = 0;
    while ((1 << pl->nb_bits) < pl->nb) ++pl->nb_bits;
    pl->tpc = (int)((rpc + tile - 1) / tile);
    pl->table = (size_t)B * pl->nb * pl->tpc;
    return B * (int64_t)pl->tpc < (1ll << 31) && B * (int64_t)pl->nb < (1ll << 31);
}
static size_t bk_workspace_words(const BkPlan& pl, int64_t rows) { return (size_t)rows + ((pl.table + 1 + 63) & ~size_t(63)) + scan_workspace_words(pl.table + 1); }

int64_t ps_op_inverse_index_workspace(int64_t n_dst, int64_t rows)
{
    if (n_dst < 0 || rows < 0) return -1;
    if (rows >= kInvSortRows) return (int64_t)(2 * rows + rows + sort_workspace_words((size_t)rows) + 64);  // two key arrays, one value array
    return (int64_t)(n_dst + 1 + scan_workspace_words((size_t)n_dst + 1) + 64);
}

int ps_op_inverse_index(ps_context* c, const int32_t* idx, int64_t B, int64_t N, int64_t rows_per_cloud, int32_t* offsets, int32_t* src, int32_t* workspace)
{
    PS_CHECK(c && idx && offsets && src && workspace, "ps_op_inverse_index: NULL argument");
    PS_CHECK(B >= 1 && N >= 1 && rows_per_cloud >= 0 && B * N < (1ll << 31) && B * rows_per_cloud < (1ll << 31), "ps_op_inverse_index: bad sizes");
    PS_HIP(hipSetDevice(c->device));
    const int64_t n_dst = B * N, rows = B * rows_per_cloud;
    unsigned* off = reinterpret_cast<unsigned*>(offsets);
    BkPlan pl;
    const bool bucket_on = c->tune.inv_bucket;  // (A/B knob: off = the radix-sort form)
    static_assert(kBkTile == 4096, "common.h: Tuning::inv_tile defaults to kBkTile");
    const int tile = c->tune.inv_tile;          // (4 096 = kBkTile, or 8 192)
    if (rows >= kInvSortRows && bucket_on && bk_plan(B, N, rows_per_cloud, tile, &pl) &&
        (int64_t)bk_workspace_words(pl, rows) + 64 <= ps_op_inverse_index_workspace(n_dst, rows)) {
        Stage st(c, "train_inverse_index", 6);
        unsigned* payload = reinterpret_cast<unsigned*>(workspace);
        unsigned* table = payload + rows;
        unsigned* scan_ws = table + ((pl.table + 1 + 63) & ~size_t(63));
        const int stage2 = (int)((sizeof(unsigned) + sizeof(unsigned short)) * tile), stage3 = (int)(sizeof(unsigned) * kBkStage);
        static const bool attr = [&] {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(bk_scatter_kernel<8192>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(6 * 8192));
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(bk_local_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, stage3);
            return true;
        }();
        (void)attr;
        const unsigned tiles = (unsigned)(B * pl.tpc), buckets = (unsigned)(B * pl.nb);
        const unsigned hgroups = (unsigned)(B * ((pl.tpc + kBkHistTiles - 1) / kBkHistTiles));
#define PS_BK_PASSES(T)                                                                                                                                   \
    hipLaunchKernelGGL(bk_hist_kernel<T>, dim3(hgroups), dim3(256), 0, c->stream, idx, (int)rows_per_cloud, pl.tpc, pl.lo, pl.nb, table, pl.table);      \
    exclusive_scan_u32(c->stream, table, table, pl.table + 1, scan_ws);                                                                                   \
    hipLaunchKernelGGL(bk_scatter_kernel<T>, dim3(tiles), dim3(256), stage2, c->stream, idx, (int)rows_per_cloud, pl.tpc, pl.lo, pl.nb, pl.nb_bits, table, \
                       payload)
        if (tile == 4096) {
            PS_BK_PASSES(4096);
        } else if (tile == 6144) {
            PS_BK_PASSES(6144);
        } else {
            PS_BK_PASSES(8192);
        }
#undef PS_BK_PASSES
        hipLaunchKernelGGL(bk_local_kernel, dim3(buckets), dim3(256), stage3, c->stream, payload, table, (int)rows_per_cloud, pl.tpc, pl.lo, pl.nb, (int)N, n_dst,
                           (unsigned)rows, off, src);
        PS_HIP(hipGetLastError());
        return PS_OK;
    }
    if (rows >= kInvSortRows) {
        Stage st(c, "train_inverse_index", 12);
        PS_CHECK((reinterpret_cast<uintptr_t>(workspace) & 7) == 0, "ps_op_inverse_index: workspace must be 8-byte aligned");
        unsigned* k0 = reinterpret_cast<unsigned*>(workspace);
        unsigned* k1 = k0 + rows;
        unsigned* vtmp = k1 + rows;
        unsigned* sort_ws = vtmp + rows;
        int bits = 1;
        while ((1ll << bits) < n_dst) ++bits;
        const int passes = (bits + 7) / 8;
        // the sorted values must end in `src`: an odd number of passes ends in the second pair of arrays
        unsigned* v0 = (passes & 1) ? vtmp : reinterpret_cast<unsigned*>(src);
        unsigned* v1 = (passes & 1) ? reinterpret_cast<unsigned*>(src) : vtmp;
        hipLaunchKernelGGL(inv_keys_kernel, dim3(iv_grid(rows)), dim3(256), 0, c->stream, idx, rows, (int)rows_per_cloud, (int)N, k0, v0);
        const int cur = radix_sort_pairs_u32(c->stream, k0, k1, v0, v1, (size_t)rows, bits, sort_ws);
        hipLaunchKernelGGL(inv_offsets_kernel, dim3(iv_grid(rows)), dim3(256), 0, c->stream, cur ? k1 : k0, rows, n_dst, off);
        PS_HIP(hipGetLastError());
        return PS_OK;
    }
    Stage st(c, "train_inverse_index", 5);
    unsigned* cursor = reinterpret_cast<unsigned*>(workspace);
    unsigned* scan_ws = cursor + n_dst + 1;
    PS_HIP(hipMemsetAsync(off, 0, sizeof(unsigned) * (size_t)(n_dst + 1), c->stream));
    PS_HIP(hipMemsetAsync(cursor, 0, sizeof(unsigned) * (size_t)(n_dst + 1), c->stream));
    if (rows > 0) hipLaunchKernelGGL(inv_count_kernel, dim3(iv_grid(rows)), dim3(256), 0, c->stream, idx, rows, (int)rows_per_cloud, (int)N, off);
    exclusive_scan_u32(c->stream, off, off, (size_t)n_dst + 1, scan_ws);
    if (rows > 0) {
        hipLaunchKernelGGL(inv_fill_kernel, dim3(iv_grid(rows)), dim3(256), 0, c->stream, idx, rows, (int)rows_per_cloud, (int)N, off, cursor, src);
        hipLaunchKernelGGL(inv_sort_kernel, dim3(ceil_div(n_dst, 256)), dim3(256), 0, c->stream, off, n_dst, src);
    }
    PS_HIP(hipGetLastError());
    return PS_OK;
}

int ps_op_gather_reduce_rows_ordered(ps_context* c, const float* rows, int64_t ldr, const int32_t* offsets, const int32_t* src, int64_t n_dst, int64_t d,
                                     float* dst, int64_t ldd, int accumulate, const int32_t* order, int64_t n_cloud)
{
    PS_CHECK(c && rows && offsets && src && dst, "ps_op_gather_reduce_rows: NULL argument");
    PS_CHECK(n_dst >= 0 && d >= 1 && ldr >= d && ldd >= d, "ps_op_gather_reduce_rows: bad shape");
    PS_CHECK(!order || (n_cloud >= 1 && n_dst % n_cloud == 0), "ps_op_gather_reduce_rows_ordered: n_dst must be a multiple of n_cloud");
    if (!n_dst) return PS_OK;
    PS_HIP(hipSetDevice(c->device));
    Stage st(c, "train_scatter_add", 1);
    const bool vec = (d % 4) == 0 && (ldr % 4) == 0 && (ldd % 4) == 0 && ((reinterpret_cast<uintptr_t>(rows) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
    const unsigned* off = reinterpret_cast<const unsigned*>(offsets);
    const bool ordered_on = c->tune.gather_reduce_ordered;  // (A/B knob)
    const int64_t per = d / 4;
    // rows of bfloat16 (ps_set_train_act_bf16 inside the bf16-MLP mode; ldr in elements, 8-byte aligned rows): the ordered kernel only
    const bool rb = c->train_act_bf16 && c->train_bf16;
    const bool can_walk = per >= 1 && per <= 256 && 256 % per == 0 && n_dst < (1ll << 31) && (d % 4) == 0 && (ldd % 4) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0;
    PS_CHECK(!rb || (can_walk && ldr % 4 == 0 && (reinterpret_cast<uintptr_t>(rows) & 7) == 0),
             "ps_op_gather_reduce_rows: bfloat16 rows need d %% 4 == 0, d / 4 a divisor of 256, ldr %% 4 == 0 and 8-byte aligned rows");
    if (rb || (order && ordered_on && vec && can_walk)) {
        const int64_t dpw = 256 / per;
        int64_t wgs = ceil_div(n_dst, dpw);
        wgs = std::min<int64_t>((wgs + 7) / 8 * 8, 8 * 2048);
        if (rb)
            hipLaunchKernelGGL(gather_reduce_ordered_kernel<true>, dim3((unsigned)wgs), dim3(256), 0, c->stream, rows, ldr, off, src, order, (unsigned)n_dst,
                               (unsigned)(order ? n_cloud : n_dst), (unsigned)per, dst, ldd, accumulate);
        else
            hipLaunchKernelGGL(gather_reduce_ordered_kernel<false>, dim3((unsigned)wgs), dim3(256), 0, c->stream, rows, ldr, off, src, order, (unsigned)n_dst,
                               (unsigned)n_cloud, (unsigned)per, dst, ldd, accumulate);
    } else if (vec) {
        hipLaunchKernelGGL(gather_reduce_kernel<true>, dim3(iv_grid(n_dst * (d / 4))), dim3(256), 0, c->stream, rows, ldr, off, src, n_dst, (int)d, dst, ldd,
                           accumulate);
    } else {
        hipLaunchKernelGGL(gather_reduce_kernel<false>, dim3(iv_grid(n_dst * d)), dim3(256), 0, c->stream, rows, ldr, off, src, n_dst, (int)d, dst, ldd,
                           accumulate);
    }
    PS_HIP(hipGetLastError());
    return PS_OK;
}

int ps_op_gather_reduce_rows(ps_context* c, const float* rows, int64_t ldr, const int32_t* offsets, const int32_t* src, int64_t n_dst, int64_t d, float* dst,
                             int64_t ldd, int accumulate)
{
    return ps_op_gather_reduce_rows_ordered(c, rows, ldr, offsets, src, n_dst, d, dst, ldd, accumulate, nullptr, 0);
}

int ps_op_random_sample_bwd_inv(ps_context* c, const float* dout, const float* out, const float* feature, const int32_t* pool_idx, const int32_t* offsets,
                                const int32_t* src, int64_t B, int64_t N, int64_t M, int64_t K, int64_t d, const uint8_t* ties, float* share_ws,
                                float* dfeature)
{
    PS_CHECK(c && dout && out && feature && pool_idx && offsets && src && (ties || share_ws) && dfeature, "ps_op_random_sample_bwd_inv: NULL argument");
    const size_t rows = (size_t)B * M;
    if (!rows) return PS_OK;
    PS_HIP(hipSetDevice(c->device));
    Stage st(c, "train_maxpool_bwd", ties ? 1 : 2);
    const bool al16 = ((reinterpret_cast<uintptr_t>(dout) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(feature) |
                        reinterpret_cast<uintptr_t>(dfeature)) & 15) == 0 && (reinterpret_cast<uintptr_t>(ties) & 3) == 0;
    if (ties && d % 4 == 0 && al16 && B * N * d < (1ll << 32) && B * N * K < (1ll << 31)) {
        const int mode = c->tune.maxpool_bwd_ordered;  // (A/B knob: 0 = the one-entry cloud-order walk)
        const int64_t d4 = d / 4;
        if (mode && d4 <= 256 && 256 % d4 == 0) {
            // (c->walk_order: the trainer's hint for the op it is about to enqueue -- the leaf order of the N points of this level, or NULL)
            const int32_t* order = c->walk_order && c->walk_order_n == N ? c->walk_order : nullptr;
            const int64_t dpw = 256 / d4;
            const int64_t wgs = std::min<int64_t>((ceil_div(B * N, dpw) + 7) / 8 * 8, 8 * 4096);
            hipLaunchKernelGGL(maxpool_bwd_inv4_ordered_kernel, dim3((unsigned)wgs), dim3(256), 0, c->stream, dout, ties, out, feature,
                               reinterpret_cast<const unsigned*>(offsets), src, order, (unsigned)(B * N), (unsigned)N, (unsigned)M, (unsigned)K, (unsigned)d4,
                               dfeature, c->pool_bwd_overwrite ? 1 : 0);
        } else {
            hipLaunchKernelGGL(maxpool_bwd_inv4_kernel, dim3(iv_grid(B * N * (d / 4))), dim3(256), 0, c->stream, dout, ties, out, feature,
                               reinterpret_cast<const unsigned*>(offsets), src, (unsigned)(B * N), (unsigned)N, (unsigned)M, (unsigned)K, (unsigned)(d / 4),
                               dfeature, c->pool_bwd_overwrite ? 1 : 0);
        }
        PS_HIP(hipGetLastError());
        return PS_OK;
    }
    PS_CHECK(!c->pool_bwd_overwrite, "ps_op_random_sample_bwd_inv: the overwriting form needs tie counts, d % 4 == 0 and 16-byte aligned rows");
    if (ties) {
        hipLaunchKernelGGL(maxpool_bwd_inv_kernel<true>, dim3(iv_grid(B * N * d)), dim3(256), 0, c->stream, dout, ties, out, feature,
                           reinterpret_cast<const unsigned*>(offsets), src, B * N, (int)N, (int)M, (int)K, (int)d, dfeature);
        PS_HIP(hipGetLastError());
        return PS_OK;
    }
    hipLaunchKernelGGL(maxpool_share_kernel, dim3(ceil_div(rows * d, 256)), dim3(256), 0, c->stream, dout, out, feature, pool_idx, rows, (int)M, (int)N, (int)K,
                       (int)d, share_ws);
    hipLaunchKernelGGL(maxpool_bwd_inv_kernel<false>, dim3(iv_grid(B * N * d)), dim3(256), 0, c->stream, share_ws, nullptr, out, feature,
                       reinterpret_cast<const unsigned*>(offsets), src, B * N, (int)N, (int)M, (int)K, (int)d, dfeature);
    PS_HIP(hipGetLastError());
    return PS_OK;
}

}  // extern "C"
