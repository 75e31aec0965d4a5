// rectconv_train.hip -- conv2d(cin -> cout) + batch_normalization(training=True) [+ LeakyReLU] of the TRAINING step on [N]-row tensors with the
// pre-BatchNorm product recomputed instead of stored: the rectangular sibling of smallconv_train.hip.
//
//   The shared MLPs that WIDEN their rows -- Encoder mlp2 (d -> 2d) and shortcut (d_in -> 2d) of dilated_res_block (RandLANet.py:312-321, no
//   activation: the sum gets it), fc1 of the head (32 -> 64, :145) -- at the point counts of levels 0-1 of a batch of 8 x 180 000 points.
//   Op by op such a layer makes 3 passes over its input width and 11 over its OUTPUT width (product written, read for the statistics, read
//   and written normalised; BatchNorm backward 2 + 3; input gradient 1; weight gradient 1); recomputed from 16-row tiles of x on the fp32
//   MFMA it is 5 passes over the input width and 3 over the output width -- for 16 -> 32 channels 11 units instead of 25.
//     forward   sums      : sum y, sum y^2 per output channel (fp64)                                            reads x
//               apply     : z = act((y - mean) gamma invstd + beta)                                              reads x, writes z
//     backward  sums      : g = dz act'(.), xh = (y - mean) invstd:  S1 = sum g, S2 = sum g xh                   reads x, dz
//               apply     : dy = gamma invstd (g - S1/M - xh S2/M);  dx (+)= dy . W^T;  dW = x^T dy, db = sum dy  reads x, dz, writes dx
//   Tiles travel as 16-byte accesses fetched one tile ahead and staged through LDS (the lesson of smallconv_train.hip's first form).  Sums are
//   per-workgroup partials merged in a fixed order (deterministic).  bf16-MLP mode: the operands of the three products are rounded to
//   bfloat16 first (cin % 16 == 0, the rule of ps_op_conv1x1_ex).  Compiled pairs: ps_op_convbn_train_supported.
#include "common.h"
#include "reduce_partials.h"
#include "mfma_tile.h"

namespace ps {

struct RcArgs {
    const float* x;      // [R, CI] rows (ldx)
    const float* w;      // [CI, CO] row-major
    const float* b;      // [CO]
    const float* mean; const float* invstd; const float* scale; const float* beta;  // [CO]; scale = gamma invstd
    const float* s12;    // [2 CO] S1 | S2 of all ranks (backward apply)
    float inv_rows;      // 1 / rows of all ranks
    const float* dz;     // [R, CO] (lddz)
    float* out;          // apply: z rows [R, CO];  backward apply: dx rows [R, CI]  (ldo)
    void* part;          // per-workgroup partial sums
    int64_t R;
    int ldx, lddz, ldo, accum, leaky, bf16;
    const float* addend;  // apply: non-null -> out = LeakyReLU(BN(x . W + b) + addend) (rows [R, CO], ld_add): the residual sum of dilated_res_block
    int ld_add;          // (RandLANet.py:306-307) in the pass that writes the second summand instead of a pass of its own
};

template <int CI, int CO>
struct RcGeom {
    static constexpr int CIP = CI < 16 ? 16 : CI, COP = CO < 16 ? 16 : CO;  // channels padded to a tile
    static constexpr int NTI = CIP / 16, NTO = COP / 16;
    static constexpr int PWO = COP + 16 + (COP % 32 == 16 ? 16 : 0);  // pitch of W  [CIP][.]: 16 (mod 32), conflict-free B-fragment reads
    static constexpr int PWI = CIP + 16 + (CIP % 32 == 16 ? 16 : 0);  // pitch of W^T [COP][.]
    static constexpr int PX = CIP + 2, PZ = COP + 2;                  // tile pitches: 2 (mod 32), conflict-free A-fragment reads
    static constexpr int WAVES = CIP * COP >= 4096 ? 4 : 8;           // waves per workgroup (LDS: both weight orientations + two tiles per wave)
};

__device__ __forceinline__ float rc_round_bf16(float x)
{
    unsigned u = __float_as_uint(x);
    u += 0x7fffu + ((u >> 16) & 1u);  // round to nearest even (finite inputs)
    return __uint_as_float(u & 0xffff0000u);
}

// W -> LDS [CIP][PWO] and optionally W^T -> [COP][PWI], zero-padded
template <int CI, int CO>
__device__ __forceinline__ void rc_stage_w(const float* __restrict__ w, float* W, float* WT, bool bf16)
{
    using G = RcGeom<CI, CO>;
    constexpr int kRcWaves = G::WAVES;
    for (int i = threadIdx.x; i < G::CIP * G::COP; i += kRcWaves * 64) {
        const int r = i / G::COP, c = i - r * G::COP;
        float v = (r < CI && c < CO) ? w[r * CO + c] : 0.f;
        if (bf16) v = rc_round_bf16(v);
        W[r * G::PWO + c] = v;
        if (WT) WT[c * G::PWI + r] = v;
    }
}

// 16 rows of a [R, C] tensor as registers / as an LDS tile of pitch P (see ScTile in smallconv_train.hip)
template <int C, int P>
struct RcTile {
    static constexpr int CP = C < 16 ? 16 : C, Q = C / 4, TOT = 16 * Q, NV = (TOT + 63) / 64;
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
    __device__ __forceinline__ void commit(float* A, int lane, bool bf16) const
    {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = 64 * i + lane;
            if (TOT % 64 == 0 || e < TOT) {
                const int row = e / Q, q = e - row * Q;
                float* dst = A + row * P + 4 * q;
                if (bf16) {
                    dst[0] = rc_round_bf16(v[i].x); dst[1] = rc_round_bf16(v[i].y); dst[2] = rc_round_bf16(v[i].z); dst[3] = rc_round_bf16(v[i].w);
                } else {
                    dst[0] = v[i].x; dst[1] = v[i].y; dst[2] = v[i].z; dst[3] = v[i].w;
                }
            }
        }
        if constexpr (C < 16) {  // padding columns (read as operands of the products)
            for (int e = lane; e < 16 * (16 - C); e += 64) A[(e / (16 - C)) * P + C + e % (16 - C)] = 0.f;
        }
    }
    __device__ __forceinline__ void take(const float* S, int lane)
    {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = 64 * i + lane;
            if (TOT % 64 == 0 || e < TOT) {
                const int row = e / Q, q = e - row * Q;
                const float2 lo = *reinterpret_cast<const float2*>(S + row * P + 4 * q);
                const float2 hi = *reinterpret_cast<const float2*>(S + row * P + 4 * q + 2);
                v[i] = make_float4(lo.x, lo.y, hi.x, hi.y);
            }
        }
    }
    __device__ __forceinline__ void add(const RcTile& o)
    {
#pragma unroll
        for (int i = 0; i < NV; ++i) { v[i].x += o.v[i].x; v[i].y += o.v[i].y; v[i].z += o.v[i].z; v[i].w += o.v[i].w; }
    }
    __device__ __forceinline__ void lrelu()
    {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            v[i].x = v[i].x < 0.f ? 0.2f * v[i].x : v[i].x; v[i].y = v[i].y < 0.f ? 0.2f * v[i].y : v[i].y;
            v[i].z = v[i].z < 0.f ? 0.2f * v[i].z : v[i].z; v[i].w = v[i].w < 0.f ? 0.2f * v[i].w : v[i].w;
        }
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

// output-column tile ct of y = X . W: lane (c16, g) gets rows 4 g + r (r = 0..3) of column 16 ct + c16
template <int CI, int CO>
__device__ __forceinline__ f32x4 rc_y_tile(const float* X, const float* W, int ct, int lane)
{
    using G = RcGeom<CI, CO>;
    const float* xa = X + (lane & 15) * G::PX + (lane >> 4);
    const float* wb = W + (lane >> 4) * G::PWO + ct * 16 + (lane & 15);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < G::CIP / 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[4 * s], wb[4 * s * G::PWO], acc, 0, 0, 0);
    return acc;
}

template <int NT>
struct RcCols {
    float v[NT];
    __device__ __forceinline__ RcCols(const float* p, int c16, int C, float mul = 1.f)
    {
#pragma unroll
        for (int t = 0; t < NT; ++t) v[t] = (t * 16 + c16 < C && p) ? p[t * 16 + c16] * mul : 0.f;
    }
};

// ---- forward: statistics.  partial per workgroup (doubles): sy[COP] | sq[COP]
template <int CI, int CO>
__global__ __launch_bounds__((RcGeom<CI, CO>::WAVES * 64)) void rc_sums_kernel(RcArgs a)
{
    using G = RcGeom<CI, CO>;
    constexpr int kRcWaves = G::WAVES;
    constexpr int NT = G::NTO;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* W = smem;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, g = lane >> 4, c16 = lane & 15;
    float* X = smem + G::CIP * G::PWO + wave * 16 * G::PX;
    double* red = reinterpret_cast<double*>(smem + G::CIP * G::PWO + kRcWaves * 16 * G::PX);  // [kRcWaves][2 COP]
    rc_stage_w<CI, CO>(a.w, W, nullptr, a.bf16 != 0);
    __syncthreads();
    const RcCols<NT> bias(a.b, c16, CO);
    double sy[NT], sq[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { sy[t] = 0.; sq[t] = 0.; }
    const int64_t tiles = (a.R + 15) / 16, tstride = (int64_t)gridDim.x * kRcWaves;
    int64_t tl = (int64_t)blockIdx.x * kRcWaves + wave;
    RcTile<CI, G::PX> xr;
    if (tl < tiles) xr.fetch(a.x, a.ldx, tl * 16, a.R, lane);
    for (; tl < tiles; tl += tstride) {
        const int64_t r0 = tl * 16;
        xr.commit(X, lane, a.bf16 != 0);
        wave_lds_sync();
        if (tl + tstride < tiles) xr.fetch(a.x, a.ldx, (tl + tstride) * 16, a.R, lane);
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
            const f32x4 y = rc_y_tile<CI, CO>(X, W, ct, lane);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r0 + 4 * g + r < a.R) {
                    const double yd = (double)(y[r] + bias.v[ct]);
                    sy[ct] += yd;
                    sq[ct] = __builtin_fma(yd, yd, sq[ct]);
                }
        }
        wave_lds_sync();
    }
    auto gsum = [&](double v) {
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        return v;
    };
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const double v1 = gsum(sy[t]), v2 = gsum(sq[t]);
        if (g == 0) {
            red[wave * 2 * G::COP + t * 16 + c16] = v1;
            red[wave * 2 * G::COP + G::COP + t * 16 + c16] = v2;
        }
    }
    __syncthreads();
    double* dst = static_cast<double*>(a.part) + (size_t)blockIdx.x * 2 * G::COP;
    for (int i = threadIdx.x; i < 2 * G::COP; i += kRcWaves * 64) {
        double s = 0.;
        for (int w = 0; w < kRcWaves; ++w) s += red[w * 2 * G::COP + i];
        dst[i] = s;
    }
}

// ---- forward: normalise (+ LeakyReLU) -> rows.  ADD: + addend rows, then LeakyReLU (a template parameter: a run-time branch around the addend
// tile's prefetch would make the compiler wait for every load in flight at the join)
template <int CI, int CO, bool ADD>
__global__ __launch_bounds__((RcGeom<CI, CO>::WAVES * 64)) void rc_apply_kernel(RcArgs a)
{
    using G = RcGeom<CI, CO>;
    constexpr int kRcWaves = G::WAVES;
    constexpr int NT = G::NTO;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* W = smem;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, g = lane >> 4, c16 = lane & 15;
    float* X = smem + G::CIP * G::PWO + wave * 16 * (G::PX + G::PZ);
    float* Z = X + 16 * G::PX;
    rc_stage_w<CI, CO>(a.w, W, nullptr, a.bf16 != 0);
    __syncthreads();
    const RcCols<NT> bias(a.b, c16, CO), mu(a.mean, c16, CO), sc(a.scale, c16, CO), be(a.beta, c16, CO);
    const int64_t tiles = (a.R + 15) / 16, tstride = (int64_t)gridDim.x * kRcWaves;
    int64_t tl = (int64_t)blockIdx.x * kRcWaves + wave;
    RcTile<CI, G::PX> xr;
    RcTile<CO, G::PZ> ad;
    if (tl < tiles) {
        xr.fetch(a.x, a.ldx, tl * 16, a.R, lane);
        if constexpr (ADD) ad.fetch(a.addend, a.ld_add, tl * 16, a.R, lane);
    }
    for (; tl < tiles; tl += tstride) {
        const int64_t r0 = tl * 16;
        xr.commit(X, lane, a.bf16 != 0);
        wave_lds_sync();
        [[maybe_unused]] const RcTile<CO, G::PZ> ad_cur = ad;
        if (tl + tstride < tiles) {
            xr.fetch(a.x, a.ldx, (tl + tstride) * 16, a.R, lane);
            if constexpr (ADD) ad.fetch(a.addend, a.ld_add, (tl + tstride) * 16, a.R, lane);
        }
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
            const f32x4 y = rc_y_tile<CI, CO>(X, W, ct, lane);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float t = __builtin_fmaf((y[r] + bias.v[ct]) - mu.v[ct], sc.v[ct], be.v[ct]);
                if (a.leaky && t < 0.f) t *= 0.2f;
                Z[(4 * g + r) * G::PZ + ct * 16 + c16] = t;
            }
        }
        wave_lds_sync();
        RcTile<CO, G::PZ> o;
        o.take(Z, lane);
        if constexpr (ADD) {
            o.add(ad_cur);
            o.lrelu();
        }
        o.put(a.out, a.ldo, r0, a.R, lane);
        wave_lds_sync();
    }
}

// ---- backward: S1 = sum g, S2 = sum g xh.  partial per workgroup (floats): S1[COP] | S2[COP]
template <int CI, int CO>
__global__ __launch_bounds__((RcGeom<CI, CO>::WAVES * 64)) void rc_bwd_sums_kernel(RcArgs a)
{
    using G = RcGeom<CI, CO>;
    constexpr int kRcWaves = G::WAVES;
    constexpr int NT = G::NTO, COP = G::COP;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* W = smem;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, g = lane >> 4, c16 = lane & 15;
    float* X = smem + G::CIP * G::PWO + wave * 16 * (G::PX + G::PZ);
    float* Z = X + 16 * G::PX;
    float* red = smem + G::CIP * G::PWO + kRcWaves * 16 * (G::PX + G::PZ);  // [kRcWaves][2 COP]
    rc_stage_w<CI, CO>(a.w, W, nullptr, a.bf16 != 0);
    __syncthreads();
    const RcCols<NT> bias(a.b, c16, CO), mu(a.mean, c16, CO), is(a.invstd, c16, CO), sc(a.scale, c16, CO), be(a.beta, c16, CO);
    float s1[NT], s2[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { s1[t] = 0.f; s2[t] = 0.f; }
    const int64_t tiles = (a.R + 15) / 16, tstride = (int64_t)gridDim.x * kRcWaves;
    int64_t tl = (int64_t)blockIdx.x * kRcWaves + wave;
    RcTile<CI, G::PX> xr;
    RcTile<CO, G::PZ> zr;
    if (tl < tiles) {
        xr.fetch(a.x, a.ldx, tl * 16, a.R, lane);
        zr.fetch(a.dz, a.lddz, tl * 16, a.R, lane);
    }
    for (; tl < tiles; tl += tstride) {
        const int64_t r0 = tl * 16;
        xr.commit(X, lane, a.bf16 != 0);
        zr.commit(Z, lane, false);
        wave_lds_sync();
        if (tl + tstride < tiles) {
            xr.fetch(a.x, a.ldx, (tl + tstride) * 16, a.R, lane);
            zr.fetch(a.dz, a.lddz, (tl + tstride) * 16, a.R, lane);
        }
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
            const f32x4 y = rc_y_tile<CI, CO>(X, W, ct, lane);
            const int col = ct * 16 + c16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool live = col < CO && r0 + 4 * g + r < a.R;
                const float yc = (y[r] + bias.v[ct]) - mu.v[ct];
                const float xh = live ? yc * is.v[ct] : 0.f;
                float gv = live ? Z[(4 * g + r) * G::PZ + col] : 0.f;
                if (a.leaky && __builtin_fmaf(yc, sc.v[ct], be.v[ct]) < 0.f) gv *= 0.2f;
                s1[ct] += gv;
                s2[ct] = __builtin_fmaf(gv, xh, s2[ct]);
            }
        }
        wave_lds_sync();
    }
    auto gsum = [&](float v) {
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        return v;
    };
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float v1 = gsum(s1[t]), v2 = gsum(s2[t]);
        if (g == 0) {
            red[wave * 2 * COP + t * 16 + c16] = v1;
            red[wave * 2 * COP + COP + t * 16 + c16] = v2;
        }
    }
    __syncthreads();
    float* dst = static_cast<float*>(a.part) + (size_t)blockIdx.x * 2 * COP;
    for (int i = threadIdx.x; i < 2 * COP; i += kRcWaves * 64) {
        float s = 0.f;
        for (int w = 0; w < kRcWaves; ++w) s += red[w * 2 * COP + i];
        dst[i] = s;
    }
}

// ---- backward: input gradient rows + weight / bias gradient.  partial per workgroup (floats): dW[CIP][COP] | db[COP]
template <int CI, int CO>
__global__ __launch_bounds__((RcGeom<CI, CO>::WAVES * 64)) void rc_bwd_apply_kernel(RcArgs a)
{
    using G = RcGeom<CI, CO>;
    constexpr int kRcWaves = G::WAVES;
    constexpr int NTI = G::NTI, NTO = G::NTO, CIP = G::CIP, COP = G::COP, NV = CIP * COP + COP;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* W = smem;
    float* WT = smem + CIP * G::PWO;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, g = lane >> 4, c16 = lane & 15;
    float* X = smem + CIP * G::PWO + COP * G::PWI + wave * 16 * (G::PX + G::PZ);
    float* Z = X + 16 * G::PX;  // dz, replaced in place by dy
    rc_stage_w<CI, CO>(a.w, W, WT, a.bf16 != 0);
    __syncthreads();
    const RcCols<NTO> bias(a.b, c16, CO), mu(a.mean, c16, CO), is(a.invstd, c16, CO), sc(a.scale, c16, CO), be(a.beta, c16, CO);
    const RcCols<NTO> m1(a.s12, c16, CO, a.inv_rows), m2(a.s12 + CO, c16, CO, a.inv_rows);
    f32x4 dw[NTI][NTO];
    float dbs[NTO];
#pragma unroll
    for (int u = 0; u < NTO; ++u) {
        dbs[u] = 0.f;
#pragma unroll
        for (int t = 0; t < NTI; ++t) dw[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int64_t tiles = (a.R + 15) / 16, tstride = (int64_t)gridDim.x * kRcWaves;
    int64_t tl = (int64_t)blockIdx.x * kRcWaves + wave;
    const bool want_dx = a.out != nullptr, acc_old = want_dx && a.accum;
    RcTile<CI, G::PX> xr, old_next;
    RcTile<CO, G::PZ> zr;
    if (tl < tiles) {
        xr.fetch(a.x, a.ldx, tl * 16, a.R, lane);
        zr.fetch(a.dz, a.lddz, tl * 16, a.R, lane);
        if (acc_old) old_next.fetch(a.out, a.ldo, tl * 16, a.R, lane);
        xr.commit(X, lane, a.bf16 != 0);
        zr.commit(Z, lane, false);
    }
    wave_lds_sync();
    for (; tl < tiles; tl += tstride) {
        const int64_t r0 = tl * 16;
        const bool more = tl + tstride < tiles;
        RcTile<CI, G::PX> old_cur = old_next;
        if (more) {
            xr.fetch(a.x, a.ldx, (tl + tstride) * 16, a.R, lane);
            zr.fetch(a.dz, a.lddz, (tl + tstride) * 16, a.R, lane);
            if (acc_old) old_next.fetch(a.out, a.ldo, (tl + tstride) * 16, a.R, lane);
        }
#pragma unroll
        for (int ct = 0; ct < NTO; ++ct) {
            const f32x4 y = rc_y_tile<CI, CO>(X, W, ct, lane);
            const int col = ct * 16 + c16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool live = col < CO && r0 + 4 * g + r < a.R;
                const float yc = (y[r] + bias.v[ct]) - mu.v[ct];
                const float xh = yc * is.v[ct];
                float gv = live ? Z[(4 * g + r) * G::PZ + col] : 0.f;
                if (a.leaky && __builtin_fmaf(yc, sc.v[ct], be.v[ct]) < 0.f) gv *= 0.2f;
                const float dyv = live ? sc.v[ct] * (gv - m1.v[ct] - xh * m2.v[ct]) : 0.f;
                Z[(4 * g + r) * G::PZ + col] = a.bf16 ? rc_round_bf16(dyv) : dyv;  // (operand of the two products; db sums the unrounded value)
                dbs[ct] += dyv;
            }
        }
        wave_lds_sync();
        // dW += x^T dy: contraction over the tile's 16 rows
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float fa[NTI], fb[NTO];
#pragma unroll
            for (int t = 0; t < NTI; ++t) fa[t] = X[(4 * s + g) * G::PX + t * 16 + c16];
#pragma unroll
            for (int u = 0; u < NTO; ++u) fb[u] = Z[(4 * s + g) * G::PZ + u * 16 + c16];
#pragma unroll
            for (int t = 0; t < NTI; ++t)
#pragma unroll
                for (int u = 0; u < NTO; ++u) dw[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[t], fb[u], dw[t][u], 0, 0, 0);
        }
        RcTile<CI, G::PX> o;
        if (want_dx) {
            wave_lds_sync();  // the x tile is dead from here on: it stages dx
#pragma unroll
            for (int ti = 0; ti < NTI; ++ti) {  // dx = dy . W^T
                const float* za = Z + c16 * G::PZ + g;
                const float* wb = WT + g * G::PWI + ti * 16 + c16;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < COP / 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(za[4 * s], wb[4 * s * G::PWI], acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) X[(4 * g + r) * G::PX + ti * 16 + c16] = acc[r];
            }
            wave_lds_sync();
            o.take(X, lane);
            if (acc_old) o.add(old_cur);
        }
        wave_lds_sync();
        if (more) {
            xr.commit(X, lane, a.bf16 != 0);
            zr.commit(Z, lane, false);
        }
        if (want_dx) o.put(a.out, a.ldo, r0, a.R, lane);
        wave_lds_sync();
    }
    // the workgroup's partial: waves add up through LDS in order (weights and tiles are dead)
    __syncthreads();
    float* red = smem;
    float* r = red + (size_t)wave * NV;
#pragma unroll
    for (int u = 0; u < NTO; ++u) {
        float v = dbs[u];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (g == 0) r[CIP * COP + u * 16 + c16] = v;
#pragma unroll
        for (int t = 0; t < NTI; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) r[(t * 16 + 4 * g + q) * COP + u * 16 + c16] = dw[t][u][q];
    }
    __syncthreads();
    float* dst = static_cast<float*>(a.part) + (size_t)blockIdx.x * NV;
    for (int i = threadIdx.x; i < NV; i += kRcWaves * 64) {
        float s = 0.f;
        for (int w = 0; w < kRcWaves; ++w) s += red[(size_t)w * NV + i];
        dst[i] = s;
    }
}

// dW partial [CIP][COP] | db[COP] (padded) -> dw [CI, CO], db [CO]
__global__ __launch_bounds__(256) void rc_unpad_kernel(const float* __restrict__ full, int CI, int CO, int COP, int CIPxCOP, float* __restrict__ dw,
                                                       float* __restrict__ db)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < CI * CO) dw[i] = full[(i / CO) * COP + i % CO];
    if (i < CO) db[i] = full[CIPxCOP + i];
}

// what: 0 forward sums, 1 forward apply, 2 backward sums, 3 backward apply (+ weight / bias gradient)
template <int CI, int CO>
static int rc_launch(ps_context* c, RcArgs a, int what, void* result, void* result2)
{
    using G = RcGeom<CI, CO>;
    constexpr int CIP = G::CIP, COP = G::COP, NVW = CIP * COP + COP, kRcWaves = G::WAVES;
    const size_t w1 = sizeof(float) * CIP * G::PWO, w2 = sizeof(float) * COP * G::PWI;
    const size_t tx = sizeof(float) * 16 * G::PX, tz = sizeof(float) * 16 * G::PZ;
    size_t smem = 0;
    if (what == 0) smem = w1 + kRcWaves * tx + sizeof(double) * kRcWaves * 2 * COP;
    if (what == 1) smem = w1 + kRcWaves * (tx + tz);
    if (what == 2) smem = w1 + kRcWaves * (tx + tz) + sizeof(float) * kRcWaves * 2 * COP;
    if (what == 3) smem = std::max(w1 + w2 + kRcWaves * (tx + tz), sizeof(float) * (size_t)kRcWaves * NVW);
    PS_CHECK(smem <= 160 * 1024, "rectconv_train: %zu bytes of LDS needed", smem);
    const int64_t tiles = (a.R + 15) / 16;
    const int per_cu = std::max(1, std::min(4, (int)(160 * 1024 / smem)));
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((tiles + kRcWaves - 1) / kRcWaves, 256 * per_cu));
    const void* kern = what == 0 ? reinterpret_cast<const void*>(rc_sums_kernel<CI, CO>)
                     : what == 1 ? (a.addend ? reinterpret_cast<const void*>(rc_apply_kernel<CI, CO, true>) : reinterpret_cast<const void*>(rc_apply_kernel<CI, CO, false>))
                     : what == 2 ? reinterpret_cast<const void*>(rc_bwd_sums_kernel<CI, CO>)
                                 : reinterpret_cast<const void*>(rc_bwd_apply_kernel<CI, CO>);
    if (smem > 48 * 1024) PS_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    if (what == 0) {
        PS_TRY(c->red_ws.reserve(sizeof(double) * (size_t)blocks * 2 * COP + 256));
        a.part = c->red_ws.as<void>();
        hipLaunchKernelGGL((rc_sums_kernel<CI, CO>), dim3(blocks), dim3(kRcWaves * 64), smem, c->stream, a);
        hipLaunchKernelGGL(reduce_partials_kernel<double>, dim3(ceil_div(2 * COP, 16)), dim3(256), 0, c->stream, static_cast<const double*>(a.part), blocks,
                           2 * COP, static_cast<double*>(result));
    } else if (what == 1) {
        if (a.addend) hipLaunchKernelGGL((rc_apply_kernel<CI, CO, true>), dim3(blocks), dim3(kRcWaves * 64), smem, c->stream, a);
        else hipLaunchKernelGGL((rc_apply_kernel<CI, CO, false>), dim3(blocks), dim3(kRcWaves * 64), smem, c->stream, a);
    } else if (what == 2) {
        PS_TRY(c->red_ws.reserve(sizeof(float) * (size_t)blocks * 2 * COP + 256));
        a.part = c->red_ws.as<void>();
        hipLaunchKernelGGL((rc_bwd_sums_kernel<CI, CO>), dim3(blocks), dim3(kRcWaves * 64), smem, c->stream, a);
        hipLaunchKernelGGL(reduce_partials_kernel<float>, dim3(ceil_div(2 * COP, 16)), dim3(256), 0, c->stream, static_cast<const float*>(a.part), blocks, 2 * COP,
                           static_cast<float*>(result));
    } else {
        // partials, then their sum behind them in the same workspace, then the unpadded copy into dw / db
        PS_TRY(c->red_ws.reserve(sizeof(float) * ((size_t)blocks + 1) * NVW + 256));
        a.part = c->red_ws.as<void>();
        float* full = c->red_ws.as<float>() + (size_t)blocks * NVW;
        hipLaunchKernelGGL((rc_bwd_apply_kernel<CI, CO>), dim3(blocks), dim3(kRcWaves * 64), smem, c->stream, a);
        hipLaunchKernelGGL(reduce_partials_kernel<float>, dim3(ceil_div(NVW, 16)), dim3(256), 0, c->stream, static_cast<const float*>(a.part), blocks, NVW, full);
        hipLaunchKernelGGL(rc_unpad_kernel, dim3(ceil_div(std::max(CI * CO, CO), 256)), dim3(256), 0, c->stream, static_cast<const float*>(full), CI, CO, COP,
                           CIP * COP, static_cast<float*>(result), static_cast<float*>(result2));
    }
    PS_HIP(hipGetLastError());
    return PS_OK;
}

static bool rc_ok(int64_t ci, int64_t co)
{
    return (ci == 8 && co == 32) || (ci == 16 && co == 32) || (ci == 32 && co == 64) || (ci == 32 && co == 128) || (ci == 64 && co == 128);
}

static int rc_dispatch(ps_context* c, int64_t ci, int64_t co, RcArgs a, int what, void* result, void* result2 = nullptr)
{
    a.bf16 = c->train_bf16 && ci % 16 == 0 ? 1 : 0;  // (ps_set_train_gemm_bf16; the rule of ps_op_conv1x1_ex)
    if (ci == 8 && co == 32) return rc_launch<8, 32>(c, a, what, result, result2);
    if (ci == 16 && co == 32) return rc_launch<16, 32>(c, a, what, result, result2);
    if (ci == 32 && co == 64) return rc_launch<32, 64>(c, a, what, result, result2);
    if (ci == 32 && co == 128) return rc_launch<32, 128>(c, a, what, result, result2);
    return rc_launch<64, 128>(c, a, what, result, result2);
}

static bool rc_rows_ok(const float* p, int64_t ld, int64_t C) { return p && ld >= C && ld % 4 == 0 && (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace ps

using namespace ps;

extern "C" int ps_op_convbn_train_supported(int64_t cin, int64_t cout) { return rc_ok(cin, cout) ? 1 : 0; }

extern "C" int ps_op_convbn_train_sums(ps_context* c, const float* x, int64_t ldx, const float* w, const float* b, int64_t R, int64_t cin, int64_t cout,
                                       double* sums)
{
    PS_CHECK(c && w && b && sums && rc_ok(cin, cout) && rc_rows_ok(x, ldx, cin), "ps_op_convbn_train_sums: unsupported (cin, cout) or unaligned rows");
    PS_HIP(hipSetDevice(c->device));
    if (R <= 0) {
        PS_HIP(hipMemsetAsync(sums, 0, sizeof(double) * 2 * cout, c->stream));
        return PS_OK;
    }
    Stage st(c, "train_convbn_fwd", 2);
    RcArgs a = {};
    a.x = x; a.ldx = (int)ldx; a.w = w; a.b = b; a.R = R;
    return rc_dispatch(c, cin, cout, a, 0, sums);
}

extern "C" int ps_op_convbn_train_apply(ps_context* c, const float* x, int64_t ldx, const float* w, const float* b, int64_t R, int64_t cin, int64_t cout,
                                        const float* mean, const float* scale, const float* beta, int leaky, float* out, int64_t ldo)
{
    PS_CHECK(c && w && b && mean && scale && beta && rc_ok(cin, cout) && rc_rows_ok(x, ldx, cin) && rc_rows_ok(out, ldo, cout),
             "ps_op_convbn_train_apply: unsupported (cin, cout) or unaligned rows");
    if (R <= 0) return PS_OK;
    PS_HIP(hipSetDevice(c->device));
    Stage st(c, "train_convbn_fwd", 1);
    RcArgs a = {};
    a.x = x; a.ldx = (int)ldx; a.w = w; a.b = b; a.R = R; a.mean = mean; a.scale = scale; a.beta = beta; a.leaky = leaky ? 1 : 0; a.out = out; a.ldo = (int)ldo;
    return rc_dispatch(c, cin, cout, a, 1, nullptr);
}

extern "C" int ps_op_convbn_train_apply_add(ps_context* c, const float* x, int64_t ldx, const float* w, const float* b, int64_t R, int64_t cin, int64_t cout,
                                            const float* mean, const float* scale, const float* beta, const float* addend, int64_t ld_add, float* out,
                                            int64_t ldo)
{
    PS_CHECK(c && w && b && mean && scale && beta && rc_ok(cin, cout) && rc_rows_ok(x, ldx, cin) && rc_rows_ok(out, ldo, cout) && rc_rows_ok(addend, ld_add, cout),
             "ps_op_convbn_train_apply_add: unsupported (cin, cout) or unaligned rows");
    if (R <= 0) return PS_OK;
    PS_HIP(hipSetDevice(c->device));
    Stage st(c, "train_convbn_fwd", 1);
    RcArgs a = {};
    a.x = x; a.ldx = (int)ldx; a.w = w; a.b = b; a.R = R; a.mean = mean; a.scale = scale; a.beta = beta; a.leaky = 0; a.out = out; a.ldo = (int)ldo;
    a.addend = addend; a.ld_add = (int)ld_add;
    return rc_dispatch(c, cin, cout, a, 1, nullptr);
}

extern "C" int ps_op_convbn_train_bwd_sums(ps_context* c, const float* x, int64_t ldx, const float* w, const float* b, int64_t R, int64_t cin, int64_t cout,
                                           const float* mean, const float* invstd, const float* scale, const float* beta, int leaky, const float* dz,
                                           int64_t lddz, float* s12)
{
    PS_CHECK(c && w && b && mean && invstd && scale && beta && s12 && rc_ok(cin, cout) && rc_rows_ok(x, ldx, cin) && rc_rows_ok(dz, lddz, cout),
             "ps_op_convbn_train_bwd_sums: unsupported (cin, cout) or unaligned rows");
    PS_HIP(hipSetDevice(c->device));
    if (R <= 0) {
        PS_HIP(hipMemsetAsync(s12, 0, sizeof(float) * 2 * cout, c->stream));
        return PS_OK;
    }
    Stage st(c, "train_convbn_bwd", 2);
    RcArgs a = {};
    a.x = x; a.ldx = (int)ldx; a.w = w; a.b = b; a.R = R; a.mean = mean; a.invstd = invstd; a.scale = scale; a.beta = beta; a.leaky = leaky ? 1 : 0;
    a.dz = dz; a.lddz = (int)lddz;
    return rc_dispatch(c, cin, cout, a, 2, s12);
}

extern "C" int ps_op_convbn_train_bwd_apply(ps_context* c, const float* x, int64_t ldx, const float* w, const float* b, int64_t R, int64_t cin, int64_t cout,
                                            const float* mean, const float* invstd, const float* scale, const float* beta, int leaky, const float* s12,
                                            float inv_rows, const float* dz, int64_t lddz, int accumulate, float* dx, int64_t lddx, float* dw, float* db)
{
    PS_CHECK(c && w && b && mean && invstd && scale && beta && s12 && dw && db && rc_ok(cin, cout) && rc_rows_ok(x, ldx, cin) && rc_rows_ok(dz, lddz, cout) &&
                 (!dx || rc_rows_ok(dx, lddx, cin)),
             "ps_op_convbn_train_bwd_apply: unsupported (cin, cout) or unaligned rows");
    PS_HIP(hipSetDevice(c->device));
    if (R <= 0) {
        PS_HIP(hipMemsetAsync(dw, 0, sizeof(float) * cin * cout, c->stream));
        PS_HIP(hipMemsetAsync(db, 0, sizeof(float) * cout, c->stream));
        return PS_OK;
    }
    Stage st(c, "train_convbn_bwd", 3);
    RcArgs a = {};
    a.x = x; a.ldx = (int)ldx; a.w = w; a.b = b; a.R = R; a.mean = mean; a.invstd = invstd; a.scale = scale; a.beta = beta; a.leaky = leaky ? 1 : 0;
    a.s12 = s12; a.inv_rows = inv_rows; a.dz = dz; a.lddz = (int)lddz; a.out = dx; a.ldo = (int)lddx; a.accum = accumulate ? 1 : 0;
    return rc_dispatch(c, cin, cout, a, 3, dw, db);
}
