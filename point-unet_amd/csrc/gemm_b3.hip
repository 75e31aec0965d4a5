// gemm_b3.hip -- Y[R, N] (+)= act(X[R, K] . W[K, N] + b) for the large fp32 products of the training step on
// v_mfma_f32_32x32x16_bf16 over exact three-way bfloat16 splits of both operands (see attpool32b.hip: six piece products per fp32
// product, fp32 accumulation, fp32-level error).
//
// The products in question are att_pooling's score GEMMs at the levels whose attention is not fused (RandLANet.py:394-395,
// [B*N*K, d] x [d, d], d = 128 / 256 / 512: 47 GFLOP each at batch 8) and their input-gradient twins: 45-55 % of the fp32 MFMA peak in
// rowgemm.hip, i.e. bound by the matrix pipe, which runs fp32 at 1/16 of the bf16 rate.
//   * a workgroup owns 256 rows x 128 columns, a wave 64 rows x 128 columns (2 x 4 accumulator tiles: a weight fragment read from LDS
//     feeds two row tiles -- one row tile per wave was bound by the LDS reads): an activation is read from HBM
//     once per column panel, by exactly one wave -- straight from global memory, 32 bytes per lane and 16-K chunk, split in registers;
//   * the weight planes (packed per call by gemm_b3_pack_kernel: the weights change every step) go through LDS, 24 KB per 32-K step,
//     double buffered, shared by the four waves;
//   * per wave and step: 96 MFMAs (3 072 cycles) against ~180 VALU (the splits), 24 ds_read_b128, 14 global accesses.
// Row counts need not be multiples of 128 (clamped loads, predicated stores); K % 32 == 0, N % 128 == 0.
#include "common.h"

#include <cstdlib>
#include "mfma_tile.h"
#include "b3_ops.h"

namespace ps {

struct GemmB3Args {
    const float* x; int ldx;
    const uint4* wp;    // planes: [K/16][N/32][3][64] uint4
    const float* bias;  // [N] or nullptr
    float* y; int ldy;
    int R, K, N, leaky, accum;
};

// W[K, N] (row-major, ldw) -> planes [K/16][N/32][3][64] x 8 bfloat16: lane l of (chunk q, column tile cb) holds
// W[16 q + 8 (l >> 5) + j][32 cb + (l & 31)], j = 0..7.  One thread per (q, cb, lane).
template <int P>
__global__ __launch_bounds__(256) void gemm_b3_pack_kernel(const float* __restrict__ w, int64_t sk, int64_t sn, int K, int N, uint4* __restrict__ out)
{
    // element (k, n) of the [K, N] matrix sits at w[k * sk + n * sn] (sk = N, sn = 1 row-major; sk = 1, sn = K for a matrix stored [N, K])
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int ncb = N / 32;
    if (i >= (K / 16) * ncb * 64) return;
    const int lane = i & 63, cb = (i >> 6) % ncb, q = (i >> 6) / ncb;
    const float* src = w + (int64_t)(16 * q + 8 * (lane >> 5)) * sk + (int64_t)(32 * cb + (lane & 31)) * sn;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = src[(int64_t)j * sk];
    const BPlanes<P> p = b3_split8<P>(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]));
    uint4* dst = out + ((size_t)(q * ncb + cb) * P) * 64 + lane;
#pragma unroll
    for (int pl = 0; pl < P; ++pl) dst[64 * pl] = p.p[pl];
}

// RT = 32-row tiles per wave: a weight fragment read from LDS feeds RT x 6 MFMAs (RT = 1 was LDS-bandwidth bound: 24 KB of fragments
// per wave and step against 1 536 cycles of products, twelve waves per CU)
constexpr int kB3RT = 2;
constexpr int64_t kB3SmallRows = 16384;  // below: one row tile per wave (gemm_b3)

template <int P, bool ACC, int RT = kB3RT>
__global__ __launch_bounds__(256) void gemm_b3_kernel(GemmB3Args a)
{
    // LDS: two buffers of one 32-K step of the workgroup's column panel: [2 chunks][4 column tiles][P planes][64 lanes] uint4 = 24 KB each (P = 3)
    __shared__ uint4 Bs[2][2 * 4 * P * 64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int hl = lane >> 5, c32 = lane & 31;
    const int ncb = a.N / 32, panels = a.N / 128;
    const int panel = blockIdx.x % panels;
    const int64_t r0 = (int64_t)(blockIdx.x / panels) * (128 * RT) + wave * (32 * RT);
    const int steps = a.K / 32;
    const float* xr[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        const int row = (int)min<int64_t>(r0 + 32 * i + c32, a.R - 1);  // clamped: rows past the end are computed and not stored
        xr[i] = a.x + (size_t)row * a.ldx + 8 * hl;
    }
    // a step's B sub-tile in the global image: chunk q = 2 s + u, column tiles 4 panel .. 4 panel + 3 (contiguous: 4 * 3 * 64 uint4)
    auto bsrc = [&](int s, int i) {  // i in [0, 512 P): (u, rest)
        const int u = i / (256 * P), rest = i - u * (256 * P);
        return a.wp + ((size_t)((2 * s + u) * ncb + 4 * panel) * P) * 64 + rest;
    };
    // (six named registers, not an array behind a lambda: the array form stayed an alloca -- 112 bytes of private segment -- and every
    //  prefetched weight plane went global -> scratch -> LDS with an s_waitcnt in front of each scratch store)
    uint4 breg0, breg1, breg2, breg3, breg4, breg5;
    float4 areg[RT][4];
#define PS_B3_LOAD_B(s_)                                  \
    do {                                                  \
        breg0 = *bsrc((s_), 0 * 256 + threadIdx.x);       \
        breg1 = *bsrc((s_), 1 * 256 + threadIdx.x);       \
        if constexpr (P == 3) {                           \
            breg2 = *bsrc((s_), 2 * 256 + threadIdx.x);   \
            breg3 = *bsrc((s_), 3 * 256 + threadIdx.x);   \
            breg4 = *bsrc((s_), 4 * 256 + threadIdx.x);   \
            breg5 = *bsrc((s_), 5 * 256 + threadIdx.x);   \
        }                                                 \
    } while (0)
#define PS_B3_STORE_B(buf_)                               \
    do {                                                  \
        Bs[(buf_)][0 * 256 + threadIdx.x] = breg0;        \
        Bs[(buf_)][1 * 256 + threadIdx.x] = breg1;        \
        if constexpr (P == 3) {                           \
            Bs[(buf_)][2 * 256 + threadIdx.x] = breg2;    \
            Bs[(buf_)][3 * 256 + threadIdx.x] = breg3;    \
            Bs[(buf_)][4 * 256 + threadIdx.x] = breg4;    \
            Bs[(buf_)][5 * 256 + threadIdx.x] = breg5;    \
        }                                                 \
    } while (0)
    // one 16-K half (u) of a step's activations: 2 x 16 bytes per row tile and lane
    auto load_a_half = [&](int s, int u) {
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            areg[i][2 * u] = *reinterpret_cast<const float4*>(xr[i] + 32 * s + 16 * u);
            areg[i][2 * u + 1] = *reinterpret_cast<const float4*>(xr[i] + 32 * s + 16 * u + 4);
        }
    };
    f32x16 acc[RT][4];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.f;

    PS_B3_LOAD_B(0);
    load_a_half(0, 0);
    load_a_half(0, 1);
    PS_B3_STORE_B(0);
    __syncthreads();
    // Register budget (256 VGPRs, two waves per SIMD): 128 accumulators + 32 (activations in flight) + 24 (weight planes in flight) + 24
    // (one half's split planes) + 12 (a weight fragment): each half is split right before its products, its registers take the next
    // step's loads as soon as they are dead, and the weight planes are fetched under the second half.  (Round 2's form kept the
    // prefetched planes in scratch -- see breg above -- and the matrix pipe idled 74 % of the time: rocprofv3 SQ_VALU_MFMA_BUSY_CYCLES,
    // [360k, 256] x [256, 128] 0.248 ms.)
    for (int s = 0; s < steps; ++s) {
        const int buf = s & 1;
        const bool more = s + 1 < steps;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            BPlanes<P> ap[RT];
#pragma unroll
            for (int i = 0; i < RT; ++i) ap[i] = b3_split8<P>(areg[i][2 * u], areg[i][2 * u + 1]);
            __builtin_amdgcn_sched_barrier(0);
            if (more) {  // the next step's operands travel under this step's products
                load_a_half(s + 1, u);
                if (u == 1) PS_B3_LOAD_B(s + 1);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                BPlanes<P> bp;
#pragma unroll
                for (int pl = 0; pl < P; ++pl) bp.p[pl] = Bs[buf][((u * 4 + t) * P + pl) * 64 + lane];
#pragma unroll
                for (int i = 0; i < RT; ++i) acc[i][t] = b3_mfma6<P>(ap[i], bp, acc[i][t]);
            }
        }
        if (more) PS_B3_STORE_B(buf ^ 1);  // (the other buffer: its last readers passed the barrier at the end of step s - 1)
        __syncthreads();
    }
#undef PS_B3_LOAD_B
#undef PS_B3_STORE_B
    // accumulator register r of tile (i, t) = row 32 i + (r & 3) + 8 (r >> 2) + 4 hl of the wave's rows, column 128 panel + 32 t + c32
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int col = 128 * panel + 32 * t + c32;
            const float bb = a.bias ? a.bias[col] : 0.f;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                float old[8];
                if constexpr (ACC) {  // eight reads of the tile in flight before the first store (stores to y would otherwise fence them)
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int r = 8 * half + q;
                        const int64_t rr = min<int64_t>(r0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * hl, a.R - 1);
                        old[q] = a.y[(size_t)rr * a.ldy + col];
                    }
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int r = 8 * half + q;
                    const int64_t rr = r0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * hl;
                    if (rr < a.R) {
                        float v = acc[i][t][r] + bb;
                        if (a.leaky) v = leaky02(v);
                        if constexpr (ACC) v += old[q];  // (y += act(x . W + b): the op's accumulate epilogue, as rowgemm.hip)
                        a.y[(size_t)rr * a.ldy + col] = v;
                    }
                }
            }
        }
}

// ---- weight gradient on the same split-bf16 products: dW[cin, cout] = X^T . dY with the ROW axis as K -------------------------------------
// (RandLANet.py's shared MLPs under tf.gradients: the third of the training step's FLOPs that ops_train.hip's wgrad_kernel runs on the fp32
// MFMA at 45-50 % of its peak.)  A workgroup owns a 128 x 128 block of dW (2 x 2 waves, each 64 x 64 = four 32x32 accumulator tiles) and a
// slab of rows; its partial goes to part[slab] with plain stores and wgrad_finish adds the slabs up in slab order (deterministic, as the
// fp32 kernel).  Both operands need "eight consecutive k (= rows) per lane": a loader thread fetches an 8-row x 4-column patch (eight
// 16-byte loads, rows coalesced across the wave), so every one of its four columns IS one lane's k-group -- it splits the eight values
// into the three bfloat16 planes once and writes them to LDS in MFMA operand order (ds_write_b128); the four waves then read ready
// operands (ds_read_b128) -- nothing is split twice and no operand is read column-wise.  Waves 0-1 load X, waves 2-3 load dY (one
// 16-row k-step each); the next chunk's patches are in flight under the 48 MFMAs of the current one.  The bias gradient (column sums of
// dY) rides on the dY loaders.
struct WgradB3Args {
    const float* x; int ldx;
    const float* dy; int lddy;
    int64_t R, rows_per_slab;
    int cin, cout;
    float* part;    // [slabs][cin][cout]
    float* dbpart;  // [slabs][cout] or nullptr
    // split-source X (the neighbour set of att_pooling without its concat buffer, RandLANet.py:326-333): columns [0, xh) of row r are
    // xl[(cloud(r) * n_src + xidx[r]) * ldxl + column], columns [xh, cin) are x[r * ldx + column - xh]; xidx == nullptr: X is x
    const float* xl; int ldxl;
    const int32_t* xidx;
    int64_t n_src, rows_per_cloud;
    int xh;
};

template <int P>
__global__ __launch_bounds__(256) void wgrad_b3_kernel(WgradB3Args a)
{
    // [operand: 0 = X (A), 1 = dY (B)][k-step][plane][32-column tile][lane] : 48 KB (P = 3)
    __shared__ uint4 Ops[2][2][P][4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int hl = lane >> 5, c32 = lane & 31;
    const int wi = wave & 1, wj = wave >> 1;  // this wave's 64 x 64 block: cin tiles 2 wi, 2 wi + 1; cout tiles 2 wj, 2 wj + 1
    const int c0 = blockIdx.y * 128, n0 = blockIdx.z * 128;
    const int64_t r0 = (int64_t)blockIdx.x * a.rows_per_slab;
    const int64_t r1 = r0 + a.rows_per_slab < a.R ? r0 + a.rows_per_slab : a.R;
    // loader role of this wave
    const int op = wave >> 1, ks = wave & 1;
    // (this thread's four columns of X lie in one half of a split source: xh is a multiple of 4)
    const bool gathered = op == 0 && a.xidx != nullptr && c0 + 4 * c32 < a.xh;
    const float* src = op == 0 ? (gathered ? a.xl + c0 + 4 * c32 : a.x + c0 + 4 * c32 - (a.xidx ? a.xh : 0)) : a.dy + n0 + 4 * c32;
    const int ld = op == 0 ? (gathered ? a.ldxl : a.ldx) : a.lddy;
    float4 pre[8];
    // gathered half: the source rows of the NEXT chunk are looked up while this chunk's rows travel (the index load and the row load it
    // addresses were one dependent chain per chunk: 0.70 against 0.40 ms for the materialised operand at [1.44 M, 128])
    int at[8];
    auto load_sources = [&](int64_t rbase) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t row = rbase + 16 * ks + 8 * hl + j;
            at[j] = row < r1 ? (int)((row / a.rows_per_cloud) * a.n_src) + a.xidx[row] : -1;
        }
    };
    if (gathered) load_sources(r0);
    auto load_patch = [&](int64_t rbase) {
        if (gathered) {
#pragma unroll
            for (int j = 0; j < 8; ++j) pre[j] = at[j] >= 0 ? *reinterpret_cast<const float4*>(src + (size_t)at[j] * ld) : float4{0.f, 0.f, 0.f, 0.f};
            load_sources(rbase + 32);
            return;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t row = rbase + 16 * ks + 8 * hl + j;
            pre[j] = row < r1 ? *reinterpret_cast<const float4*>(src + (size_t)row * ld) : float4{0.f, 0.f, 0.f, 0.f};
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};

    load_patch(r0);
    for (int64_t r = r0; r < r1; r += 32) {
        __syncthreads();  // the previous chunk's operands have been read
        {
            // column 4 c32 + e of the patch = lane (c32' = (4 c32 + e) & 31, hl) of tile (4 c32 + e) >> 5: four consecutive lanes of one tile
            const int t = (4 * c32) >> 5, l0 = ((4 * c32) & 31) + 32 * hl;
            const float px[8] = {pre[0].x, pre[1].x, pre[2].x, pre[3].x, pre[4].x, pre[5].x, pre[6].x, pre[7].x};
            const float py[8] = {pre[0].y, pre[1].y, pre[2].y, pre[3].y, pre[4].y, pre[5].y, pre[6].y, pre[7].y};
            const float pz[8] = {pre[0].z, pre[1].z, pre[2].z, pre[3].z, pre[4].z, pre[5].z, pre[6].z, pre[7].z};
            const float pw[8] = {pre[0].w, pre[1].w, pre[2].w, pre[3].w, pre[4].w, pre[5].w, pre[6].w, pre[7].w};
            const float* cols[4] = {px, py, pz, pw};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float* v = cols[e];
                const BPlanes<P> p = b3_split8<P>(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]));
#pragma unroll
                for (int pl = 0; pl < P; ++pl) Ops[op][ks][pl][t][l0 + e] = p.p[pl];
                bsum[e] += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
            }
        }
        if (r + 32 < r1) load_patch(r + 32);  // the next chunk travels under this chunk's products
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            BPlanes<P> av[2], bv[2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int pl = 0; pl < P; ++pl) {
                    av[i].p[pl] = Ops[0][k][pl][2 * wi + i][lane];
                    bv[i].p[pl] = Ops[1][k][pl][2 * wj + i][lane];
                }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = b3_mfma6<P>(av[i], bv[j], acc[i][j]);
        }
    }
    // accumulator register r of tile (i, j) = dW row 32 (2 wi + i) + (r & 3) + 8 (r >> 2) + 4 hl, column 32 (2 wj + j) + c32
    float* out = a.part + (size_t)blockIdx.x * a.cin * a.cout;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = c0 + 32 * (2 * wi + i) + (r & 3) + 8 * (r >> 2) + 4 * hl, n = n0 + 32 * (2 * wj + j) + c32;
                out[(size_t)ci * a.cout + n] = acc[i][j][r];
            }
    if (a.dbpart != nullptr && blockIdx.y == 0) {
        // the dY loaders hold the column sums of their row groups: four partials per column, added in a fixed order
        __syncthreads();
        float* red = reinterpret_cast<float*>(&Ops[0][0][0][0][0]);
        if (op == 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) red[(2 * ks + hl) * 128 + 4 * c32 + e] = bsum[e];
        }
        __syncthreads();
        if (threadIdx.x < 128)
            a.dbpart[(size_t)blockIdx.x * a.cout + n0 + threadIdx.x] =
                (red[threadIdx.x] + red[128 + threadIdx.x]) + (red[256 + threadIdx.x] + red[384 + threadIdx.x]);
    }
}

bool wgrad_b3_fits(const Tuning& tn, int64_t R, int64_t cin, int64_t cout, const float* x, int64_t ldx, const float* dy, int64_t lddy, bool one_plane)
{
    // (three planes -- the fp32 step -- from 4 096 rows: batch 8 37.32 -> 37.15 ms; the one-plane products of the bf16-MLP mode from 16 384)
    const int64_t env_rows = tn.wgrad_b3_min_rows;  // (experiment knob)
    const int64_t min_rows = env_rows > 0 ? env_rows : (one_plane ? 16384 : 4096);
    return R >= min_rows && R < (1ll << 40) && cin % 128 == 0 && cout % 128 == 0 && ldx % 4 == 0 && lddy % 4 == 0 &&
           ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15) == 0;
}

static void wgrad_b3_plan(const Tuning& tn, int64_t R, int64_t cin, int64_t cout, int64_t& rows_per_slab, int64_t& slabs)
{
    const int64_t blocks = (cin / 128) * (cout / 128);
    const int64_t total = tn.wgrad_wgs;  // 512 (768 / 512 / 384 / 256 measured: one-cloud step 8.03 / 7.95 / 7.96 / 8.17 ms, batch 8: 39.0 / 38.1 / 38.8 / 39.3 -- every slab is a partial the finish reads back)
    int64_t want = total / blocks;  // ~2 workgroups per CU
    want = want < 1 ? 1 : want;
    rows_per_slab = (R + want - 1) / want;
    rows_per_slab = ((rows_per_slab + 31) / 32) * 32;
    rows_per_slab = rows_per_slab < 256 ? 256 : rows_per_slab;
    slabs = (R + rows_per_slab - 1) / rows_per_slab;
}

int64_t wgrad_b3_slabs(const Tuning& tn, int64_t R, int64_t cin, int64_t cout)
{
    int64_t rps, slabs;
    wgrad_b3_plan(tn, R, cin, cout, rps, slabs);
    return slabs;
}

int wgrad_b3_partial(ps_context* c, const float* x, int64_t ldx, const float* dy, int64_t lddy, int64_t R, int64_t cin, int64_t cout, float* part,
                     float* dbpart)
{
    WgradB3Args a = {};
    a.x = x; a.ldx = (int)ldx; a.dy = dy; a.lddy = (int)lddy; a.R = R; a.cin = (int)cin; a.cout = (int)cout; a.part = part; a.dbpart = dbpart;
    int64_t slabs;
    wgrad_b3_plan(c->tune, R, cin, cout, a.rows_per_slab, slabs);
    const dim3 grid((unsigned)slabs, (unsigned)(cin / 128), (unsigned)(cout / 128));
    if (c->train_bf16) hipLaunchKernelGGL(wgrad_b3_kernel<1>, grid, dim3(256), 0, c->stream, a);  // bf16-MLP mode: operands rounded, one product
    else hipLaunchKernelGGL(wgrad_b3_kernel<3>, grid, dim3(256), 0, c->stream, a);
    PS_HIP(hipGetLastError());
    return PS_OK;
}

// X = [xl[xidx] | xr] (split source, see WgradB3Args): the weight gradient of the fused attentive pooling of the wide levels
bool wgrad_b3_split_fits(const Tuning& tn, int64_t R, int64_t cin, int64_t cout, const float* xl, int64_t ldxl, const int32_t* xidx, const float* xr, int64_t ldxr, const float* dy,
                         int64_t lddy)
{
    return wgrad_b3_fits(tn, R, cin, cout, xr, ldxr, dy, lddy, /*one_plane: the split form's own floor*/ true) && xl && xidx && ldxl % 4 == 0 && (reinterpret_cast<uintptr_t>(xl) & 15) == 0 && (cin / 2) % 4 == 0;
}

int wgrad_b3_partial_split(ps_context* c, const float* xl, int64_t ldxl, const int32_t* xidx, int64_t n_src, int64_t rows_per_cloud, const float* xr, int64_t ldxr,
                           const float* dy, int64_t lddy, int64_t R, int64_t cin, int64_t cout, float* part)
{
    WgradB3Args a = {};
    a.x = xr; a.ldx = (int)ldxr; a.dy = dy; a.lddy = (int)lddy; a.R = R; a.cin = (int)cin; a.cout = (int)cout; a.part = part; a.dbpart = nullptr;
    a.xl = xl; a.ldxl = (int)ldxl; a.xidx = xidx; a.n_src = n_src; a.rows_per_cloud = rows_per_cloud; a.xh = (int)(cin / 2);
    int64_t slabs;
    wgrad_b3_plan(c->tune, R, cin, cout, a.rows_per_slab, slabs);
    const dim3 grid((unsigned)slabs, (unsigned)(cin / 128), (unsigned)(cout / 128));
    if (c->train_bf16) hipLaunchKernelGGL(wgrad_b3_kernel<1>, grid, dim3(256), 0, c->stream, a);
    else hipLaunchKernelGGL(wgrad_b3_kernel<3>, grid, dim3(256), 0, c->stream, a);
    PS_HIP(hipGetLastError());
    return PS_OK;
}

bool gemm_b3_fits(const Tuning& tn, int64_t R, int64_t K, int64_t N, const float* x, int64_t ldx, bool one_plane)
{
    // Measured against rowgemm.hip (profiles/tools/gemm_shapes_ab.py, round 3, after the prefetched weight planes left scratch memory):
    //   [360k, 256] x [256, 128] 0.177 vs 0.340 ms, [90k, 512] x [512, 256] 0.148 vs 0.279, [90k, 256] x [256, 512] 0.164 vs 0.298,
    //   [22k, 256] x [256, 512] 0.056 vs 0.085; K = 128: [1.44M, 128] x [128, 128] 0.457 vs 0.530 (1.5 GB of rows: HBM bound),
    //   [360k, 128] x [128, 256] 0.219 vs 0.289.
    // R >= 16384 (the measurements above): a workgroup owns 256 rows x 128 columns, so the few-row GEMMs of the deep levels leave most CUs without one ([5624, 1536]
    // x [1536, 512] 0.276 ms here against 0.125 on the split-K fp32 kernel, [5624, 512] x [512, 256] 0.078 against 0.024)
    // (8 192 <= R < 16 384 runs the 128-row workgroup form: one-cloud step 8.05 -> 7.97 ms; 4 096: 7.99, 2 048: 8.12)
    //  Batch 8: from 4 096 rows the three-plane (fp32) products gain 0.2 ms per step -- 38.6 -> 38.4 --, the one-plane products of the bf16-MLP
    //  mode lose 0.1 (what they replace there is a bf16-MFMA kernel already).)
    const int64_t env_rows = tn.gemm_b3_min_rows;  // (experiment knob)
    const int64_t min_rows = env_rows > 0 ? env_rows : (one_plane ? 8192 : 4096);
    return R >= min_rows && K >= 128 && K % 32 == 0 && N % 128 == 0 && ldx % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && R < (1ll << 31);
}

size_t gemm_b3_plane_bytes(int64_t K, int64_t N) { return (size_t)K * N * 6; }

// kinds (common.h): 3 = three planes, 4 = one RNE plane, 5 / 6 = the same in accumulator K order
__global__ __launch_bounds__(256) void pack_batch_b3_kernel(const PackJob* __restrict__ jobs)
{
    const PackJob j = jobs[blockIdx.y];
    for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < j.total; i += (int64_t)gridDim.x * 256) {
        b3_pack_elem_any(j, i);
    }
}
int pack_batch_b3(ps_context* c, const PackJob* table, int n)
{
    if (n <= 0) return PS_OK;
    hipLaunchKernelGGL(pack_batch_b3_kernel, dim3(32, (unsigned)n), dim3(256), 0, c->stream, table);
    PS_HIP(hipGetLastError());
    return PS_OK;
}

int gemm_b3(ps_context* c, const float* x, int64_t ldx, const float* w, int64_t sk, int64_t sn, const float* bias, int64_t R, int64_t K, int64_t N, int leaky,
            int accumulate, float* y, int64_t ldy, void* planes, int pack)
{
    const bool one = c->train_bf16;  // bf16-MLP mode: one plane of RNE-rounded operands (the buffer is sized for three)
    if (!pack) {
        // (the planes were packed at the start of the step: PackCache)
    } else if (one)
        hipLaunchKernelGGL(gemm_b3_pack_kernel<1>, dim3(ceil_div((K / 16) * (N / 32) * 64, 256)), dim3(256), 0, c->stream, w, sk, sn, (int)K, (int)N,
                           static_cast<uint4*>(planes));
    else
        hipLaunchKernelGGL(gemm_b3_pack_kernel<3>, dim3(ceil_div((K / 16) * (N / 32) * 64, 256)), dim3(256), 0, c->stream, w, sk, sn, (int)K, (int)N,
                           static_cast<uint4*>(planes));
    GemmB3Args a;
    a.x = x; a.ldx = (int)ldx; a.wp = static_cast<const uint4*>(planes); a.bias = bias; a.y = y; a.ldy = (int)ldy;
    a.R = (int)R; a.K = (int)K; a.N = (int)N; a.leaky = leaky; a.accum = accumulate ? 1 : 0;
    // few rows (the deep levels of a one-cloud step): 128-row workgroups -- twice as many of them; with many rows that form is bound by the
    // LDS reads of the weight fragments (kB3RT)
    const bool small = R < kB3SmallRows;
    const int64_t rows_wg = small ? 128 : 128 * kB3RT;
    const int64_t blocks = ((R + rows_wg - 1) / rows_wg) * (N / 128);
#define PS_B3_LAUNCH(P_, ACC_)                                                                                                          \
    do {                                                                                                                                \
        if (small) hipLaunchKernelGGL((gemm_b3_kernel<P_, ACC_, 1>), dim3((unsigned)blocks), dim3(256), 0, c->stream, a);               \
        else hipLaunchKernelGGL((gemm_b3_kernel<P_, ACC_, kB3RT>), dim3((unsigned)blocks), dim3(256), 0, c->stream, a);                 \
    } while (0)
    if (one) {
        if (a.accum) PS_B3_LAUNCH(1, true);
        else PS_B3_LAUNCH(1, false);
    } else {
        if (a.accum) PS_B3_LAUNCH(3, true);
        else PS_B3_LAUNCH(3, false);
    }
#undef PS_B3_LAUNCH
    PS_HIP(hipGetLastError());
    return PS_OK;
}

}  // namespace ps
