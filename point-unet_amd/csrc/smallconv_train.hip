// smallconv_train.hip -- conv2d(c -> c) + batch_normalization(training=True) + LeakyReLU of the TRAINING step on [N*K]-row tensors, fused.
//
//   LFA mlp2 of building_block (PointSegment/RandLANet.py:331; helper_tf_util.conv2d :115-170): z = lrelu(BN(x . W + b)), x = f_xyz rows,
//   c = d_out / 2 = 8 / 32 / 64 at encoder levels 0-2 of a batch of 8 x 180 000 points: 23 M / 5.8 M / 1.4 M rows.
//
// Op by op the pre-BatchNorm product y is written, read for the statistics, read again and written normalised; and in the backward
// BatchNorm's two passes read (dz, y) each, write dy, which the weight-gradient GEMM and the input-gradient GEMM read again: 14 passes
// over [rows, c] tensors.  Here y is never stored: a 16-row tile of x goes to LDS and y is recomputed from it on the fp32 MFMA wherever it
// is needed (c^2 MACs per row: nothing against the memory passes).
//   forward   sums  : per-channel sum y, sum y^2 (fp64 accumulators: the variance is a difference of nearly equal sums), sum x
//             apply : z = lrelu((y - mean) gamma invstd + beta) -> rows
//   backward  sums  : with xh = (y - mean) invstd, g = dz lrelu'(.):  S1 = sum g, S2 = sum g xh, XS = sum xh, A = x^T g, G = x^T xh
//                     (the two c x c products on the MFMA with the 16 rows of a tile as K, accumulated in registers per wave)
//             apply : dy = gamma invstd (g - S1/M - xh S2/M);  dx = dy . W^T -> rows (or added to an existing gradient)
//   and the caller finishes  dW = gamma invstd (A - (sum x) x S1/M - G . S2/M),  db, dgamma = S2, dbeta = S1  on the [c, c] sums
//   (train.py; under SyncBN that is also where the sums of all ranks meet).
// 8 passes instead of 14.  Sums are per-workgroup partials merged in a fixed order (deterministic).  c in {8, 16, 32, 64}.
#include "common.h"
#include "bf16_io.h"
#include "reduce_partials.h"
#include "mfma_tile.h"

namespace ps {

struct ScArgs {
    const float* x;      // [R, C] rows (ldx)
    const float* w;      // [C, C] row-major
    const float* b;      // [C]
    const float* mean; const float* invstd; const float* scale; const float* beta;  // [C]; scale = gamma invstd
    const float* m1; const float* m2;  // [C] S1 / M, S2 / M (backward apply)
    float mscale;        // m1 / m2 hold the raw sums S1 / S2 of all ranks: multiplied by this (1 / rows of all ranks); 1 when they are means
    const float* dz;     // [R, C] (lddz)
    float* out;          // apply: z rows (ldo);  backward apply: dx rows (ldo)
    void* part;          // per-workgroup partial sums
    int64_t R;
    int ldx, lddz, ldo, accum;
    int vec;             // dz / out rows are 16-byte aligned with pitches % 4 == 0: tiles travel as float4 through LDS
    int x_bf16;          // x rows (and apply's out rows) are STORED as bfloat16 (ps_set_train_act_bf16): ldx / ldo in elements
    int bf16;            // bf16-MLP mode: the operands of the three products (x and w; dy and w^T; x and dy) are rounded to bfloat16 (RNE) first and
                         // multiplied on the fp32 MFMA (exact products, fp32 accumulation: the values of a bf16 MFMA with fp32 accumulate)
};

template <int C>
struct ScGeom {
    static constexpr int CP = C < 16 ? 16 : C;                       // channels padded to a tile
    static constexpr int NT = CP / 16;
    static constexpr int PW = CP + 16 + (CP % 32 == 16 ? 16 : 0);   // weight pitch = 16 (mod 32): conflict-free B-fragment reads
    static constexpr int PA = CP + 2;                                // tile pitch = 2 (mod 32): conflict-free A-fragment reads
};

__device__ __forceinline__ float sc_round_bf16(float x)
{
    unsigned u = __float_as_uint(x);
    u += 0x7fffu + ((u >> 16) & 1u);  // round to nearest even (finite inputs)
    return __uint_as_float(u & 0xffff0000u);
}

// W (and optionally W^T) -> LDS, zero-padded to CP x CP
template <int C, int THREADS>
__device__ __forceinline__ void sc_stage_w(const float* __restrict__ w, float* W, float* WT, bool bf16 = false)
{
    using G = ScGeom<C>;
    for (int i = threadIdx.x; i < G::CP * G::CP; i += THREADS) {
        const int r = i / G::CP, c = i - r * G::CP;
        float v = (r < C && c < C) ? w[r * C + c] : 0.f;
        if (bf16) v = sc_round_bf16(v);
        W[r * G::PW + c] = v;
        if (WT) WT[c * G::PW + r] = v;
    }
}

// 16 rows of a [R, C] tensor starting at row r0 as registers (lane e of pass i: float4 q of row (64 i + e) / (C/4); rows past R are zero).
// fetch = global -> registers, issued one tile AHEAD of its use (a wave works on one tile at a time: nothing else covers the latency),
// commit = registers -> LDS tile (pitch PA; padding columns zeroed), take = LDS tile -> registers (8-byte aligned rows), put = 16-byte stores.
template <int C>
struct ScTile {
    using G = ScGeom<C>;
    static constexpr int Q = C / 4, TOT = 16 * Q, NV = (TOT + 63) / 64;
    float4 v[NV];
    __device__ __forceinline__ void fetch(const float* __restrict__ x, int ldx, int64_t r0, int64_t R, int lane)
    {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = 64 * i + lane;
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (TOT % 64 == 0 || e < TOT) {
                const int row = e / Q, q = e - row * Q;
                if (r0 + row < R) v[i] = *reinterpret_cast<const float4*>(x + (size_t)r0 * ldx + (unsigned)(row * ldx + 4 * q));  // (wave-uniform base, 32-bit lane offset)
            }
        }
    }
    // the same from / to rows of bfloat16 (8 bytes per lane and pass; converted at the register)
    template <bool B16>
    __device__ __forceinline__ void fetch_any(const float* __restrict__ x, int ldx, int64_t r0, int64_t R, int lane)
    {
        if constexpr (!B16) return fetch(x, ldx, r0, R, lane);
        const unsigned short* xb = reinterpret_cast<const unsigned short*>(x) + (size_t)r0 * ldx;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = 64 * i + lane;
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (TOT % 64 == 0 || e < TOT) {
                const int row = e / Q, q = e - row * Q;
                if (r0 + row < R) {  // (the raw 8 bytes: converting here would make the wave wait for its own prefetch)
                    const uint2 u = *reinterpret_cast<const uint2*>(xb + (unsigned)(row * ldx + 4 * q));
                    v[i].x = __uint_as_float(u.x);
                    v[i].y = __uint_as_float(u.y);
                }
            }
        }
    }
    // ... expanded to fp32 where the tile is consumed (in front of commit)
    template <bool B16>
    __device__ __forceinline__ void expand()
    {
        if constexpr (B16) {
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i] = unpack_bf16x4(make_uint2(__float_as_uint(v[i].x), __float_as_uint(v[i].y)));
        }
    }
    template <bool B16>
    __device__ __forceinline__ void put_any(float* __restrict__ out, int ldo, int64_t r0, int64_t R, int lane) const
    {
        if constexpr (!B16) return put(out, ldo, r0, R, lane);
        unsigned short* ob = reinterpret_cast<unsigned short*>(out) + (size_t)r0 * ldo;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = 64 * i + lane;
            if (TOT % 64 == 0 || e < TOT) {
                const int row = e / Q, q = e - row * Q;
                if (r0 + row < R) *reinterpret_cast<uint2*>(ob + (unsigned)(row * ldo + 4 * q)) = pack_bf16x4(v[i]);
            }
        }
    }
    __device__ __forceinline__ void commit(float* A, int lane, bool bf16 = false) const
    {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = 64 * i + lane;
            if (TOT % 64 == 0 || e < TOT) {
                const int row = e / Q, q = e - row * Q;
                float* dst = A + row * G::PA + 4 * q;
                if (bf16) {
                    dst[0] = sc_round_bf16(v[i].x); dst[1] = sc_round_bf16(v[i].y); dst[2] = sc_round_bf16(v[i].z); dst[3] = sc_round_bf16(v[i].w);
                } else {
                    dst[0] = v[i].x; dst[1] = v[i].y; dst[2] = v[i].z; dst[3] = v[i].w;
                }
            }
        }
        if constexpr (C < 16) {  // padding columns (read as A operands of the x^T products)
            for (int e = lane; e < 16 * (16 - C); e += 64) A[(e / (16 - C)) * G::PA + C + e % (16 - C)] = 0.f;
        }
    }
    __device__ __forceinline__ void take(const float* S, int lane)
    {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = 64 * i + lane;
            if (TOT % 64 == 0 || e < TOT) {
                const int row = e / Q, q = e - row * Q;
                const float2 lo = *reinterpret_cast<const float2*>(S + row * G::PA + 4 * q);
                const float2 hi = *reinterpret_cast<const float2*>(S + row * G::PA + 4 * q + 2);
                v[i] = make_float4(lo.x, lo.y, hi.x, hi.y);
            }
        }
    }
    __device__ __forceinline__ void add(const ScTile& o)
    {
#pragma unroll
        for (int i = 0; i < NV; ++i) { v[i].x += o.v[i].x; v[i].y += o.v[i].y; v[i].z += o.v[i].z; v[i].w += o.v[i].w; }
    }
    __device__ __forceinline__ void put(float* __restrict__ out, int ldo, int64_t r0, int64_t R, int lane) const
    {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = 64 * i + lane;
            if (TOT % 64 == 0 || e < TOT) {
                const int row = e / Q, q = e - row * Q;
                if (r0 + row < R) *reinterpret_cast<float4*>(out + (size_t)r0 * ldo + (unsigned)(row * ldo + 4 * q)) = v[i];
            }
        }
    }
};

// column tile ct of y = X . W: lane (c16, g) gets rows 4 g + r (r = 0..3) of column 16 ct + c16
template <int C>
__device__ __forceinline__ f32x4 sc_y_tile(const float* X, const float* W, int ct, int lane)
{
    using G = ScGeom<C>;
    const float* xa = X + (lane & 15) * G::PA + (lane >> 4);
    const float* wb = W + (lane >> 4) * G::PW + ct * 16 + (lane & 15);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < G::CP / 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[4 * s], wb[4 * s * G::PW], acc, 0, 0, 0);
    return acc;
}

// per-channel constants of this lane's columns
template <int NT>
struct ScCols {
    float v[NT];
    __device__ __forceinline__ ScCols(const float* p, int c16, int C)
    {
#pragma unroll
        for (int t = 0; t < NT; ++t) v[t] = (t * 16 + c16 < C && p) ? p[t * 16 + c16] : 0.f;
    }
};

constexpr int kScWaves = 4;

// ---- forward: statistics (sum y, sum y^2 in fp64; sum x in fp32) ---------------------------------------------------------------
// partial layout per workgroup (doubles): sy[CP] | sq[CP] | sx[CP]
template <int C, bool XB>  // XB: x rows (and apply's output rows) are stored as bfloat16 -- a compile-time switch: a run-time branch around the
                           // tile loads made the compiler drain every load in flight at the join (bwd apply 0.66 -> 0.80 ms)
__global__ __launch_bounds__(kScWaves * 64) void sc_sums_kernel(ScArgs a)
{
    using G = ScGeom<C>;
    constexpr int NT = G::NT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* W = smem;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, g = lane >> 4, c16 = lane & 15;
    float* A = smem + G::CP * G::PW + wave * 16 * G::PA;
    double* red = reinterpret_cast<double*>(smem + G::CP * G::PW + kScWaves * 16 * G::PA);  // [kScWaves][3 CP]
    sc_stage_w<C, kScWaves * 64>(a.w, W, nullptr, a.bf16 != 0);
    __syncthreads();
    const ScCols<NT> bias(a.b, c16, C);
    double sy[NT], sq[NT], sx[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { sy[t] = 0.; sq[t] = 0.; sx[t] = 0.; }
    const int64_t tiles = (a.R + 15) / 16, tstride = (int64_t)gridDim.x * kScWaves;
    int64_t tl = (int64_t)blockIdx.x * kScWaves + wave;
    ScTile<C> xr;
    if (tl < tiles) xr.template fetch_any<XB>(a.x, a.ldx, tl * 16, a.R, lane);
    for (; tl < tiles; tl += tstride) {
        const int64_t r0 = tl * 16;
        xr.template expand<XB>(), xr.commit(A, lane, a.bf16 != 0);
        wave_lds_sync();
        if (tl + tstride < tiles) xr.template fetch_any<XB>(a.x, a.ldx, (tl + tstride) * 16, a.R, lane);
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
            const f32x4 y = sc_y_tile<C>(A, W, ct, lane);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (r0 + 4 * g + r < a.R) {
                    const double yd = (double)(y[r] + bias.v[ct]);
                    sy[ct] += yd;
                    sq[ct] = __builtin_fma(yd, yd, sq[ct]);
                    sx[ct] += (double)A[(4 * g + r) * G::PA + ct * 16 + c16];
                }
            }
        }
        wave_lds_sync();
    }
    // lanes g = 0..3 hold the same columns: butterfly over g, then the waves through LDS in order
    auto gsum = [&](double v) {
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        return v;
    };
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const double v1 = gsum(sy[t]), v2 = gsum(sq[t]), v3 = gsum(sx[t]);
        if (g == 0) {
            red[wave * 3 * G::CP + t * 16 + c16] = v1;
            red[wave * 3 * G::CP + G::CP + t * 16 + c16] = v2;
            red[wave * 3 * G::CP + 2 * G::CP + t * 16 + c16] = v3;
        }
    }
    __syncthreads();
    double* dst = static_cast<double*>(a.part) + (size_t)blockIdx.x * 3 * G::CP;
    for (int i = threadIdx.x; i < 3 * G::CP; i += kScWaves * 64) {
        double s = 0.;
        for (int w = 0; w < kScWaves; ++w) s += red[w * 3 * G::CP + i];
        dst[i] = s;
    }
}

// ---- forward: normalise + LeakyReLU -> rows --------------------------------------------------------------------------------------
template <int C, bool XB>
__global__ __launch_bounds__(kScWaves * 64) void sc_apply_kernel(ScArgs a)
{
    using G = ScGeom<C>;
    constexpr int NT = G::NT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* W = smem;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, g = lane >> 4, c16 = lane & 15;
    float* A = smem + G::CP * G::PW + wave * 16 * G::PA;
    sc_stage_w<C, kScWaves * 64>(a.w, W, nullptr, a.bf16 != 0);
    __syncthreads();
    const ScCols<NT> bias(a.b, c16, C), mu(a.mean, c16, C), sc(a.scale, c16, C), be(a.beta, c16, C);
    const int64_t tiles = (a.R + 15) / 16, tstride = (int64_t)gridDim.x * kScWaves;
    int64_t tl = (int64_t)blockIdx.x * kScWaves + wave;
    ScTile<C> xr;
    if (tl < tiles) {
        xr.template fetch_any<XB>(a.x, a.ldx, tl * 16, a.R, lane);
        xr.template expand<XB>(), xr.commit(A, lane, a.bf16 != 0);
    }
    wave_lds_sync();
    for (; tl < tiles; tl += tstride) {
        const int64_t r0 = tl * 16;
        const bool more = tl + tstride < tiles;
        if (more) xr.template fetch_any<XB>(a.x, a.ldx, (tl + tstride) * 16, a.R, lane);
        f32x4 z[NT];
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
            const f32x4 y = sc_y_tile<C>(A, W, ct, lane);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float t = __builtin_fmaf((y[r] + bias.v[ct]) - mu.v[ct], sc.v[ct], be.v[ct]);
                z[ct][r] = t < 0.f ? 0.2f * t : t;
            }
        }
        if (a.vec) {  // z staged in the (now dead) x tile, 16-byte stores
            wave_lds_sync();
#pragma unroll
            for (int ct = 0; ct < NT; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) A[(4 * g + r) * G::PA + ct * 16 + c16] = z[ct][r];
            wave_lds_sync();
            ScTile<C> o;
            o.take(A, lane);
            wave_lds_sync();
            if (more) xr.template expand<XB>(), xr.commit(A, lane, a.bf16 != 0);
            o.template put_any<XB>(a.out, a.ldo, r0, a.R, lane);
        } else {
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) {
                const int col = ct * 16 + c16;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (col < C && r0 + 4 * g + r < a.R) a.out[(size_t)(r0 + 4 * g + r) * a.ldo + col] = z[ct][r];
            }
            wave_lds_sync();
            if (more) xr.template expand<XB>(), xr.commit(A, lane, a.bf16 != 0);
        }
        wave_lds_sync();
    }
}

// ---- backward: sums --------------------------------------------------------------------------------------------------------------
// partial layout per workgroup (floats): S1[CP] | S2[CP] | XS[CP] | A[CP][CP] | G[CP][CP]
// FULL = false: only S1 | S2 | XS (the weight gradient then comes out of sc_bwd_apply_kernel<C, true, XB> as x^T dy)
template <int C, bool FULL, bool XB>
__global__ __launch_bounds__(kScWaves * 64) void sc_bwd_sums_kernel(ScArgs a)
{
    using G = ScGeom<C>;
    constexpr int NT = G::NT, CP = G::CP, NV = FULL ? 3 * CP + 2 * CP * CP : 3 * CP;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* W = smem;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, g = lane >> 4, c16 = lane & 15;
    float* A = smem + CP * G::PW + wave * 3 * 16 * G::PA;  // x tile | g tile | xh tile
    float* T1 = A + 16 * G::PA;
    float* T2 = T1 + 16 * G::PA;
    sc_stage_w<C, kScWaves * 64>(a.w, W, nullptr, a.bf16 != 0);
    __syncthreads();
    const ScCols<NT> bias(a.b, c16, C), mu(a.mean, c16, C), is(a.invstd, c16, C), sc(a.scale, c16, C), be(a.beta, c16, C);
    float s1[NT], s2[NT], xs[NT];
    f32x4 aw[NT][NT], gw[NT][NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        s1[t] = 0.f; s2[t] = 0.f; xs[t] = 0.f;
#pragma unroll
        for (int u = 0; u < NT; ++u) { aw[t][u] = f32x4{0.f, 0.f, 0.f, 0.f}; gw[t][u] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    }
    // vec: the dz tile travels like the x tile (16-byte loads one tile ahead, through LDS tile T1) instead of as 4-byte loads in the
    // accumulator layout
    const int64_t tiles = (a.R + 15) / 16, tstride = (int64_t)gridDim.x * kScWaves;
    int64_t tl = (int64_t)blockIdx.x * kScWaves + wave;
    ScTile<C> xr, zr;
    if (tl < tiles) {
        xr.template fetch_any<XB>(a.x, a.ldx, tl * 16, a.R, lane);
        if (a.vec) zr.template fetch_any<XB>(a.dz, a.lddz, tl * 16, a.R, lane);  // (gradient rows in the format of the activation rows)
    }
    for (; tl < tiles; tl += tstride) {
        const int64_t r0 = tl * 16;
        xr.template expand<XB>(), xr.commit(A, lane, a.bf16 != 0);
        if (a.vec) zr.template expand<XB>(), zr.commit(T1, lane);
        wave_lds_sync();
        if (tl + tstride < tiles) {
            xr.template fetch_any<XB>(a.x, a.ldx, (tl + tstride) * 16, a.R, lane);
            if (a.vec) zr.template fetch_any<XB>(a.dz, a.lddz, (tl + tstride) * 16, a.R, lane);
        }
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
            const f32x4 y = sc_y_tile<C>(A, W, ct, lane);
            const int col = ct * 16 + c16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool live = col < C && r0 + 4 * g + r < a.R;
                const float yc = (y[r] + bias.v[ct]) - mu.v[ct];
                const float xh = live ? yc * is.v[ct] : 0.f;
                float gv = !live ? 0.f : a.vec ? T1[(4 * g + r) * G::PA + col] : a.dz[(size_t)(r0 + 4 * g + r) * a.lddz + col];
                if (__builtin_fmaf(yc, sc.v[ct], be.v[ct]) < 0.f) gv *= 0.2f;
                s1[ct] += gv;
                s2[ct] = __builtin_fmaf(gv, xh, s2[ct]);
                xs[ct] += xh;
                if (FULL) {
                    T1[(4 * g + r) * G::PA + col] = gv;
                    T2[(4 * g + r) * G::PA + col] = xh;
                }
            }
        }
        wave_lds_sync();
        // A += x^T g, G += x^T xh: contraction over the tile's 16 rows (four MFMA steps per tile pair)
#pragma unroll
        for (int s = 0; FULL && s < 4; ++s) {
            float fa[NT], f1[NT], f2[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                fa[t] = A[(4 * s + g) * G::PA + t * 16 + c16];   // A operand: x^T[i = 16 t + c16][k = 4 s + g]
                f1[t] = T1[(4 * s + g) * G::PA + t * 16 + c16];  // B operands
                f2[t] = T2[(4 * s + g) * G::PA + t * 16 + c16];
            }
#pragma unroll
            for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                for (int tj = 0; tj < NT; ++tj) {
                    aw[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[ti], f1[tj], aw[ti][tj], 0, 0, 0);
                    gw[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[ti], f2[tj], gw[ti][tj], 0, 0, 0);
                }
        }
        wave_lds_sync();
    }
    // the workgroup's partial: waves add up through LDS in order (weights and tiles are dead: the buffer is sized for kScWaves x NV)
    __syncthreads();
    float* red = smem;
    float* r = red + (size_t)wave * NV;
    auto gsum = [&](float v) {
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        return v;
    };
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float v1 = gsum(s1[t]), v2 = gsum(s2[t]), v3 = gsum(xs[t]);
        if (g == 0) { r[t * 16 + c16] = v1; r[CP + t * 16 + c16] = v2; r[2 * CP + t * 16 + c16] = v3; }
#pragma unroll
        for (int u = 0; FULL && u < NT; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                r[3 * CP + (t * 16 + 4 * g + q) * CP + u * 16 + c16] = aw[t][u][q];
                r[3 * CP + CP * CP + (t * 16 + 4 * g + q) * CP + u * 16 + c16] = gw[t][u][q];
            }
    }
    __syncthreads();
    float* dst = static_cast<float*>(a.part) + (size_t)blockIdx.x * NV;
    for (int i = threadIdx.x; i < NV; i += kScWaves * 64) {
        float s = 0.f;
        for (int w = 0; w < kScWaves; ++w) s += red[(size_t)w * NV + i];
        dst[i] = s;
    }
}

// ---- backward: input gradient ----------------------------------------------------------------------------------------------------
// WG = true: also the weight / bias gradient dW = x^T dy, db = sum dy as per-workgroup partials (dW[CP][CP] | db[CP]) in a.part
template <int C, bool WG, bool XB>
__global__ __launch_bounds__(kScWaves * 64) void sc_bwd_apply_kernel(ScArgs a)
{
    using G = ScGeom<C>;
    constexpr int NT = G::NT, CP = G::CP;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* W = smem;
    float* WT = smem + CP * G::PW;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, g = lane >> 4, c16 = lane & 15;
    float* A = smem + 2 * CP * G::PW + wave * 2 * 16 * G::PA;  // x tile | dy tile
    float* T1 = A + 16 * G::PA;
    sc_stage_w<C, kScWaves * 64>(a.w, W, WT, a.bf16 != 0);
    __syncthreads();
    const ScCols<NT> bias(a.b, c16, C), mu(a.mean, c16, C), is(a.invstd, c16, C), sc(a.scale, c16, C), be(a.beta, c16, C);
    ScCols<NT> m1(a.m1, c16, C), m2(a.m2, c16, C);
#pragma unroll
    for (int t = 0; t < NT; ++t) { m1.v[t] *= a.mscale; m2.v[t] *= a.mscale; }
    f32x4 dw[NT][NT];
    float dbs[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        dbs[t] = 0.f;
#pragma unroll
        for (int u = 0; u < NT; ++u) dw[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // vec: x, dz and (accumulate) the old dx rows of the NEXT tile are requested before this tile's work starts; the dz tile sits in T1 and is
    // replaced in place by dy; dx is staged in the x tile once the weight-gradient product has read it, and leaves as 16-byte stores
    const int64_t tiles = (a.R + 15) / 16, tstride = (int64_t)gridDim.x * kScWaves;
    int64_t tl = (int64_t)blockIdx.x * kScWaves + wave;
    const bool vec = a.vec != 0, acc_old = vec && a.accum;
    ScTile<C> xr, zr, old_next;
    if (tl < tiles) {
        xr.template fetch_any<XB>(a.x, a.ldx, tl * 16, a.R, lane);
        if (vec) zr.template fetch_any<XB>(a.dz, a.lddz, tl * 16, a.R, lane);  // (gradient rows dz / dx: the format of the activation rows)
        if (acc_old) old_next.template fetch_any<XB>(a.out, a.ldo, tl * 16, a.R, lane);
        xr.template expand<XB>(), xr.commit(A, lane, a.bf16 != 0);
        if (vec) zr.template expand<XB>(), zr.commit(T1, lane);
    }
    wave_lds_sync();
    for (; tl < tiles; tl += tstride) {
        const int64_t r0 = tl * 16;
        const bool more = tl + tstride < tiles;
        ScTile<C> old_cur = old_next;
        if (more) {
            xr.template fetch_any<XB>(a.x, a.ldx, (tl + tstride) * 16, a.R, lane);
            if (vec) zr.template fetch_any<XB>(a.dz, a.lddz, (tl + tstride) * 16, a.R, lane);
            if (acc_old) old_next.template fetch_any<XB>(a.out, a.ldo, (tl + tstride) * 16, a.R, lane);
        }
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
            const f32x4 y = sc_y_tile<C>(A, W, ct, lane);
            const int col = ct * 16 + c16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool live = col < C && r0 + 4 * g + r < a.R;
                const float yc = (y[r] + bias.v[ct]) - mu.v[ct];
                const float xh = yc * is.v[ct];
                float gv = !live ? 0.f : vec ? T1[(4 * g + r) * G::PA + col] : a.dz[(size_t)(r0 + 4 * g + r) * a.lddz + col];
                if (__builtin_fmaf(yc, sc.v[ct], be.v[ct]) < 0.f) gv *= 0.2f;
                const float dyv = live ? sc.v[ct] * (gv - m1.v[ct] - xh * m2.v[ct]) : 0.f;
                T1[(4 * g + r) * G::PA + col] = a.bf16 ? sc_round_bf16(dyv) : dyv;  // (operand of the two products; the bias gradient sums the unrounded value)
                if (WG) dbs[ct] += dyv;
            }
        }
        wave_lds_sync();
        if (WG) {  // dW += x^T dy: contraction over the tile's 16 rows
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                float fa[NT], fb[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    fa[t] = A[(4 * s + g) * G::PA + t * 16 + c16];
                    fb[t] = T1[(4 * s + g) * G::PA + t * 16 + c16];
                }
#pragma unroll
                for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                    for (int tj = 0; tj < NT; ++tj) dw[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[ti], fb[tj], dw[ti][tj], 0, 0, 0);
            }
        }
        if (vec) wave_lds_sync();  // the x tile is dead from here on: it stages dx
        // dx = dy . W^T
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) {
            const float* xa = T1 + c16 * G::PA + g;
            const float* wb = WT + g * G::PW + tj * 16 + c16;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < CP / 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[4 * s], wb[4 * s * G::PW], acc, 0, 0, 0);
            const int col = tj * 16 + c16;
            if (vec) {
#pragma unroll
                for (int r = 0; r < 4; ++r) A[(4 * g + r) * G::PA + col] = acc[r];
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (col < C && r0 + 4 * g + r < a.R) {
                        float* dst = a.out + (size_t)(r0 + 4 * g + r) * a.ldo + col;
                        *dst = a.accum ? *dst + acc[r] : acc[r];
                    }
            }
        }
        wave_lds_sync();
        ScTile<C> o;
        if (vec) {
            o.take(A, lane);
            if (acc_old) old_cur.template expand<XB>(), o.add(old_cur);
            wave_lds_sync();
        }
        if (more) {
            xr.template expand<XB>(), xr.commit(A, lane, a.bf16 != 0);
            if (vec) zr.template expand<XB>(), zr.commit(T1, lane);
        }
        if (vec) o.template put_any<XB>(a.out, a.ldo, r0, a.R, lane);
        wave_lds_sync();
    }
    if (WG) {  // the workgroup's partial: waves add up through LDS in order (weights and tiles are dead)
        constexpr int NV = CP * CP + CP;
        __syncthreads();
        float* red = smem;
        float* r = red + (size_t)wave * NV;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float v = dbs[t];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (g == 0) r[CP * CP + t * 16 + c16] = v;
#pragma unroll
            for (int u = 0; u < NT; ++u)
#pragma unroll
                for (int q = 0; q < 4; ++q) r[(t * 16 + 4 * g + q) * CP + u * 16 + c16] = dw[t][u][q];
        }
        __syncthreads();
        float* dst = static_cast<float*>(a.part) + (size_t)blockIdx.x * NV;
        for (int i = threadIdx.x; i < NV; i += kScWaves * 64) {
            float s = 0.f;
            for (int w = 0; w < kScWaves; ++w) s += red[(size_t)w * NV + i];
            dst[i] = s;
        }
    }
}

static bool sc_ok(int64_t C) { return C == 8 || C == 16 || C == 32 || C == 64; }

// what: 0 forward sums, 1 forward apply, 2 backward sums (S1 | S2 | XS | A | G), 3 backward apply, 4 backward sums (S1 | S2 | XS only),
//       5 backward apply + weight / bias gradient (result = dW, result2 = db)
template <int C, bool XB>
static int sc_launch_x(ps_context* c, ScArgs a, int what, void* result, void* result2)
{
    using G = ScGeom<C>;
    constexpr int CP = G::CP, NVB = 3 * CP + 2 * CP * CP, NVW = CP * CP + CP;
    const size_t wts = sizeof(float) * CP * G::PW, tile = sizeof(float) * 16 * G::PA;
    size_t smem = 0;
    if (what == 0) smem = wts + kScWaves * tile + sizeof(double) * kScWaves * 3 * CP;
    if (what == 1) smem = wts + kScWaves * tile;
    if (what == 2) smem = std::max(wts + kScWaves * 3 * tile, sizeof(float) * (size_t)kScWaves * NVB);
    if (what == 4) smem = std::max(wts + kScWaves * 3 * tile, sizeof(float) * (size_t)kScWaves * 3 * CP);
    if (what == 3) smem = 2 * wts + kScWaves * 2 * tile;
    if (what == 5) smem = std::max(2 * wts + kScWaves * 2 * tile, sizeof(float) * (size_t)kScWaves * NVW);
    PS_CHECK(smem <= 160 * 1024, "smallconv_train: %zu bytes of LDS needed", smem);
    const int64_t tiles = (a.R + 15) / 16;
    const int per_cu = std::max(1, std::min(4, (int)(160 * 1024 / smem)));
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((tiles + kScWaves - 1) / kScWaves, 256 * per_cu));
    const void* kern = what == 0 ? reinterpret_cast<const void*>(sc_sums_kernel<C, XB>)
                     : what == 1 ? reinterpret_cast<const void*>(sc_apply_kernel<C, XB>)
                     : what == 2 ? reinterpret_cast<const void*>(sc_bwd_sums_kernel<C, true, XB>)
                     : what == 3 ? reinterpret_cast<const void*>(sc_bwd_apply_kernel<C, false, XB>)
                     : what == 4 ? reinterpret_cast<const void*>(sc_bwd_sums_kernel<C, false, XB>)
                                 : reinterpret_cast<const void*>(sc_bwd_apply_kernel<C, true, XB>);
    if (smem > 48 * 1024) PS_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    if (what == 0) {
        PS_TRY(c->red_ws.reserve(sizeof(double) * (size_t)blocks * 3 * CP + 256));
        a.part = c->red_ws.as<void>();
        hipLaunchKernelGGL((sc_sums_kernel<C, XB>), dim3(blocks), dim3(kScWaves * 64), smem, c->stream, a);
        hipLaunchKernelGGL(reduce_partials_kernel<double>, dim3(ceil_div(3 * CP, 16)), dim3(256), 0, c->stream, static_cast<const double*>(a.part), blocks, 3 * CP,
                           static_cast<double*>(result));
    } else if (what == 1) {
        hipLaunchKernelGGL((sc_apply_kernel<C, XB>), dim3(blocks), dim3(kScWaves * 64), smem, c->stream, a);
    } else if (what == 2 || what == 4) {
        const int nv = what == 2 ? NVB : 3 * CP;
        PS_TRY(c->red_ws.reserve(sizeof(float) * (size_t)blocks * nv + 256));
        a.part = c->red_ws.as<void>();
        if (what == 2) hipLaunchKernelGGL((sc_bwd_sums_kernel<C, true, XB>), dim3(blocks), dim3(kScWaves * 64), smem, c->stream, a);
        else hipLaunchKernelGGL((sc_bwd_sums_kernel<C, false, XB>), dim3(blocks), dim3(kScWaves * 64), smem, c->stream, a);
        hipLaunchKernelGGL(reduce_partials_kernel<float>, dim3(ceil_div(nv, 16)), dim3(256), 0, c->stream, static_cast<const float*>(a.part), blocks, nv,
                           static_cast<float*>(result));
    } else if (what == 3) {
        hipLaunchKernelGGL((sc_bwd_apply_kernel<C, false, XB>), dim3(blocks), dim3(kScWaves * 64), smem, c->stream, a);
    } else {
        PS_TRY(c->red_ws.reserve(sizeof(float) * (size_t)blocks * NVW + 256));
        a.part = c->red_ws.as<void>();
        hipLaunchKernelGGL((sc_bwd_apply_kernel<C, true, XB>), dim3(blocks), dim3(kScWaves * 64), smem, c->stream, a);
        hipLaunchKernelGGL(reduce_partials2_kernel<float>, dim3(ceil_div(NVW, 16)), dim3(256), 0, c->stream, static_cast<const float*>(a.part), blocks, NVW,
                           CP * CP, static_cast<float*>(result), static_cast<float*>(result2));
    }
    PS_HIP(hipGetLastError());
    return PS_OK;
}

template <int C>
static int sc_launch(ps_context* c, const ScArgs& a, int what, void* result, void* result2 = nullptr)
{
    return a.x_bf16 ? sc_launch_x<C, true>(c, a, what, result, result2) : sc_launch_x<C, false>(c, a, what, result, result2);
}

static int sc_dispatch(ps_context* c, int64_t C, const ScArgs& a_in, int what, void* result, void* result2 = nullptr)
{
    ScArgs a = a_in;
    auto al = [](const void* q, int ld) { return !q || ((reinterpret_cast<uintptr_t>(q) & 15) == 0 && ld % 4 == 0); };
    a.vec = al(a.dz, a.lddz) && al(a.out, a.ldo) ? 1 : 0;
    PS_CHECK(!a.x_bf16 || a.vec, "ps_op_conv_bn_train_*: bfloat16 rows (x, the output, dz, dx: ps_set_train_act_bf16) need 16-byte aligned bases and pitches %% 4 == 0");
    a.bf16 = c->train_bf16 && C % 16 == 0 ? 1 : 0;  // (ps_set_train_gemm_bf16; the rule of ps_op_conv1x1_ex: an 8-channel product stays fp32)
    switch (C) {
        case 8: return sc_launch<8>(c, a, what, result, result2);
        case 16: return sc_launch<16>(c, a, what, result, result2);
        case 32: return sc_launch<32>(c, a, what, result, result2);
        default: return sc_launch<64>(c, a, what, result, result2);
    }
}

// one thread per row at C = 8 (convbn_rows.hip)
int convbn_rows_sums(ps_context* c, const float* x, int64_t ldx, const float* w, const float* b, int64_t R, double* sums);
int convbn_rows_apply(ps_context* c, const float* x, int64_t ldx, const float* w, const float* b, int64_t R, const float* mean, const float* scale,
                      const float* beta, float* out, int64_t ldo);
int convbn_rows_bwd_sums(ps_context* c, const float* x, int64_t ldx, const float* w, const float* b, int64_t R, const float* mean, const float* invstd,
                         const float* scale, const float* beta, const float* dz, int64_t lddz, float* s12);
int convbn_rows_bwd_apply(ps_context* c, const float* x, int64_t ldx, const float* w, const float* b, int64_t R, const float* mean, const float* invstd,
                          const float* scale, const float* beta, const float* s12, float inv_rows, const float* dz, int64_t lddz, int accumulate, float* dx,
                          int64_t lddx, float* dw, float* db);

static bool sc_rows_ok(const float* p, int64_t ld, int64_t C) { return p && ld >= C && ld % 4 == 0 && (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace ps

using namespace ps;

extern "C" int ps_op_conv_bn_train_supported(int64_t C) { return sc_ok(C) ? 1 : 0; }

extern "C" int ps_op_conv_bn_train_sums(ps_context* c, const float* x, int64_t ldx, const float* w, const float* b, int64_t R, int64_t C, double* sums)
{
    PS_CHECK(c && w && b && sums && sc_ok(C) && sc_rows_ok(x, ldx, C), "ps_op_conv_bn_train_sums: C in {8, 16, 32, 64}, rows 16-byte aligned");
    PS_HIP(hipSetDevice(c->device));
    const int64_t CP = C < 16 ? 16 : C;
    if (R <= 0) {
        PS_HIP(hipMemsetAsync(sums, 0, sizeof(double) * 3 * CP, c->stream));
        return PS_OK;
    }
    Stage st(c, "train_convbn_fwd", 2);
    if (C == 8) return convbn_rows_sums(c, x, ldx, w, b, R, sums);
    ScArgs a = {};
    a.x_bf16 = c->train_act_bf16 && c->train_bf16 ? 1 : 0;  // (x, apply's out and the gradient rows dz / dx as bfloat16: ps_set_train_act_bf16)
    a.x = x; a.ldx = (int)ldx; a.w = w; a.b = b; a.R = R;
    return sc_dispatch(c, C, a, 0, sums);
}

extern "C" int ps_op_conv_bn_train_apply(ps_context* c, const float* x, int64_t ldx, const float* w, const float* b, int64_t R, int64_t C, const float* mean,
                                         const float* scale, const float* beta, float* out, int64_t ldo)
{
    PS_CHECK(c && w && b && mean && scale && beta && out && sc_ok(C) && sc_rows_ok(x, ldx, C) && ldo >= C,
             "ps_op_conv_bn_train_apply: C in {8, 16, 32, 64}, rows 16-byte aligned");
    if (R <= 0) return PS_OK;
    PS_HIP(hipSetDevice(c->device));
    Stage st(c, "train_convbn_fwd", 1);
    if (C == 8 && ldo % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) return convbn_rows_apply(c, x, ldx, w, b, R, mean, scale, beta, out, ldo);
    ScArgs a = {};
    a.x_bf16 = c->train_act_bf16 && c->train_bf16 ? 1 : 0;  // (x, apply's out and the gradient rows dz / dx as bfloat16: ps_set_train_act_bf16)
    a.x = x; a.ldx = (int)ldx; a.w = w; a.b = b; a.R = R; a.mean = mean; a.scale = scale; a.beta = beta; a.out = out; a.ldo = (int)ldo;
    return sc_dispatch(c, C, a, 1, nullptr);
}

extern "C" int ps_op_conv_bn_train_bwd_sums(ps_context* c, const float* x, int64_t ldx, const float* w, const float* b, int64_t R, int64_t C,
                                            const float* mean, const float* invstd, const float* scale, const float* beta, const float* dz, int64_t lddz,
                                            float* sums)
{
    PS_CHECK(c && w && b && mean && invstd && scale && beta && dz && sums && sc_ok(C) && sc_rows_ok(x, ldx, C) && lddz >= C,
             "ps_op_conv_bn_train_bwd_sums: C in {8, 16, 32, 64}, rows 16-byte aligned");
    PS_HIP(hipSetDevice(c->device));
    const int64_t CP = C < 16 ? 16 : C;
    if (R <= 0) {
        PS_HIP(hipMemsetAsync(sums, 0, sizeof(float) * (3 * CP + 2 * CP * CP), c->stream));
        return PS_OK;
    }
    Stage st(c, "train_convbn_bwd", 2);
    ScArgs a = {};
    a.x_bf16 = c->train_act_bf16 && c->train_bf16 ? 1 : 0;  // (x, apply's out and the gradient rows dz / dx as bfloat16: ps_set_train_act_bf16)
    a.x = x; a.ldx = (int)ldx; a.w = w; a.b = b; a.R = R; a.mean = mean; a.invstd = invstd; a.scale = scale; a.beta = beta; a.dz = dz; a.lddz = (int)lddz;
    return sc_dispatch(c, C, a, 2, sums);
}

extern "C" int ps_op_conv_bn_train_bwd_apply(ps_context* c, const float* x, int64_t ldx, const float* w, const float* b, int64_t R, int64_t C,
                                             const float* mean, const float* invstd, const float* scale, const float* beta, const float* m1,
                                             const float* m2, const float* dz, int64_t lddz, int accumulate, float* dx, int64_t lddx)
{
    PS_CHECK(c && w && b && mean && invstd && scale && beta && m1 && m2 && dz && dx && sc_ok(C) && sc_rows_ok(x, ldx, C) && lddz >= C && lddx >= C,
             "ps_op_conv_bn_train_bwd_apply: C in {8, 16, 32, 64}, rows 16-byte aligned");
    if (R <= 0) return PS_OK;
    PS_HIP(hipSetDevice(c->device));
    Stage st(c, "train_convbn_bwd", 1);
    ScArgs a = {};
    a.x_bf16 = c->train_act_bf16 && c->train_bf16 ? 1 : 0;  // (x, apply's out and the gradient rows dz / dx as bfloat16: ps_set_train_act_bf16)
    a.x = x; a.ldx = (int)ldx; a.w = w; a.b = b; a.R = R; a.mean = mean; a.invstd = invstd; a.scale = scale; a.beta = beta; a.m1 = m1; a.m2 = m2;
    a.dz = dz; a.lddz = (int)lddz; a.out = dx; a.ldo = (int)lddx; a.accum = accumulate ? 1 : 0; a.mscale = 1.f;
    return sc_dispatch(c, C, a, 3, nullptr);
}

extern "C" int ps_op_conv_bn_train_bwd_sums2(ps_context* c, const float* x, int64_t ldx, const float* w, const float* b, int64_t R, int64_t C,
                                             const float* mean, const float* invstd, const float* scale, const float* beta, const float* dz, int64_t lddz,
                                             float* s12)
{
    PS_CHECK(c && w && b && mean && invstd && scale && beta && dz && s12 && sc_ok(C) && sc_rows_ok(x, ldx, C) && lddz >= C,
             "ps_op_conv_bn_train_bwd_sums2: C in {8, 16, 32, 64}, rows 16-byte aligned");
    PS_HIP(hipSetDevice(c->device));
    if (R <= 0) {
        PS_HIP(hipMemsetAsync(s12, 0, sizeof(float) * 3 * C, c->stream));
        return PS_OK;
    }
    Stage st(c, "train_convbn_bwd", 2);
    if (C == 8 && sc_rows_ok(dz, lddz, C)) return convbn_rows_bwd_sums(c, x, ldx, w, b, R, mean, invstd, scale, beta, dz, lddz, s12);
    PS_CHECK(C >= 16, "ps_op_conv_bn_train_bwd_sums2: C = 8 needs 16-byte aligned dz rows");
    ScArgs a = {};
    a.x_bf16 = c->train_act_bf16 && c->train_bf16 ? 1 : 0;  // (x, apply's out and the gradient rows dz / dx as bfloat16: ps_set_train_act_bf16)
    a.x = x; a.ldx = (int)ldx; a.w = w; a.b = b; a.R = R; a.mean = mean; a.invstd = invstd; a.scale = scale; a.beta = beta; a.dz = dz; a.lddz = (int)lddz;
    return sc_dispatch(c, C, a, 4, s12);
}

extern "C" int ps_op_conv_bn_train_bwd_apply_w(ps_context* c, const float* x, int64_t ldx, const float* w, const float* b, int64_t R, int64_t C,
                                               const float* mean, const float* invstd, const float* scale, const float* beta, const float* s12,
                                               float inv_rows, const float* dz, int64_t lddz, int accumulate, float* dx, int64_t lddx, float* dw, float* db)
{
    PS_CHECK(c && w && b && mean && invstd && scale && beta && s12 && dz && dx && dw && db && sc_ok(C) && sc_rows_ok(x, ldx, C) && lddz >= C && lddx >= C,
             "ps_op_conv_bn_train_bwd_apply_w: C in {8, 16, 32, 64}, rows 16-byte aligned");
    PS_HIP(hipSetDevice(c->device));
    if (R <= 0) {
        PS_HIP(hipMemsetAsync(dw, 0, sizeof(float) * C * C, c->stream));
        PS_HIP(hipMemsetAsync(db, 0, sizeof(float) * C, c->stream));
        return PS_OK;
    }
    Stage st(c, "train_convbn_bwd", 2);
    if (C == 8 && sc_rows_ok(dz, lddz, C) && sc_rows_ok(dx, lddx, C))
        return convbn_rows_bwd_apply(c, x, ldx, w, b, R, mean, invstd, scale, beta, s12, inv_rows, dz, lddz, accumulate, dx, lddx, dw, db);
    PS_CHECK(C >= 16, "ps_op_conv_bn_train_bwd_apply_w: C = 8 needs 16-byte aligned dz / dx rows");
    ScArgs a = {};
    a.x_bf16 = c->train_act_bf16 && c->train_bf16 ? 1 : 0;  // (x, apply's out and the gradient rows dz / dx as bfloat16: ps_set_train_act_bf16)
    a.x = x; a.ldx = (int)ldx; a.w = w; a.b = b; a.R = R; a.mean = mean; a.invstd = invstd; a.scale = scale; a.beta = beta; a.m1 = s12; a.m2 = s12 + C;
    a.mscale = inv_rows; a.dz = dz; a.lddz = (int)lddz; a.out = dx; a.ldo = (int)lddx; a.accum = accumulate ? 1 : 0;
    return sc_dispatch(c, C, a, 5, dw, db);
}
