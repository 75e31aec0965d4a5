// convbn_rows.hip -- conv2d(8 -> 8) + batch_normalization(training=True) + LeakyReLU on [N*K]-row tensors, one THREAD per row.
//
//   LFA mlp2 of building_block at encoder level 0 (PointSegment/RandLANet.py:331; helper_tf_util.conv2d :115-170): z = lrelu(BN(x . W + b)) on
//   the 23 M rows x 8 channels of a batch of 8 x 180 000 points (737 MB per tensor).  Same recompute scheme as smallconv_train.hip -- the
//   pre-BatchNorm product y and its gradient never reach memory -- but a row is 32 bytes: a thread loads it with two 16-byte loads, keeps it
//   in registers, takes the 8 x 8 weights as scalar operands and writes 16-byte stores, so every pass streams at the HBM rate (the 16-row MFMA
//   tiles of smallconv_train.hip pad 8 channels to 16 and issue ~250 instructions per tile: they lost to the op-by-op kernels).
//     forward   sums      : sum y, sum y^2, sum x per channel in float64                                   1 read
//               apply     : z = lrelu((y - mean) gamma invstd + beta)                                      1 read + 1 write
//     backward  sums      : g = dz lrelu'(.), xh = (y - mean) invstd:  S1 = sum g, S2 = sum g xh           2 reads
//               apply     : dy = gamma invstd (g - S1/M - xh S2/M);  dx (+)= dy . W^T;  dW = x^T dy,  db = sum dy (per-thread accumulators,
//                           merged per workgroup and then across workgroups in a fixed order)               2 reads + 1 write (+ 1 read)
//   6-7 passes over [rows, 8] tensors against 15 op by op (convolution 2, statistics 1, normalise 2, BatchNorm backward 2 + 3, input
//   gradient 3, weight gradient 2).  Deterministic: fixed grid, fixed reduction order, no atomics.
#include "common.h"
#include "reduce_partials.h"
#include "bf16_io.h"

namespace ps {

constexpr int kCbThreads = 256;
constexpr int kCbBlocks = 2048;

struct CbArgs {
    const float* x;   // [R, C] rows (ldx)
    const float* w;   // [C, C] row-major (in, out)
    const float* b;   // [C]
    const float* mean; const float* invstd; const float* scale; const float* beta;  // [C]; scale = gamma invstd
    const float* s12;  // [2 C] S1 | S2 summed over all rows of all ranks (backward apply)
    float inv_rows;    // 1 / rows of all ranks
    const float* dz;   // [R, C] (lddz)
    float* out;        // apply: z rows (ldo);  backward apply: dx rows (ldo)
    void* part;        // per-workgroup partial sums
    int64_t R;
    int ldx, lddz, ldo, accum;
    int x_bf16;        // x rows, apply's out rows AND the gradient rows dz / dx are bfloat16 (ps_set_train_act_bf16)
};

// the C x C weights and the bias: uniform addresses -> scalar loads, the products take them as SGPR operands (no LDS, no VGPRs)
template <int C>
struct CbWeights {
    float W[C * C];
    float Bv[C];
    __device__ __forceinline__ void stage(const CbArgs& a)
    {
#pragma unroll
        for (int i = 0; i < C * C; ++i) W[i] = a.w[i];
#pragma unroll
        for (int i = 0; i < C; ++i) Bv[i] = a.b[i];
    }
    // y = x . W + b
    __device__ __forceinline__ void product(const float (&x)[C], float (&y)[C]) const
    {
#pragma unroll
        for (int j = 0; j < C; ++j) y[j] = Bv[j];
#pragma unroll
        for (int k = 0; k < C; ++k)
#pragma unroll
            for (int j = 0; j < C; ++j) y[j] = __builtin_fmaf(x[k], W[k * C + j], y[j]);
    }
};

// a row of x: fp32 or (ps_set_train_act_bf16) bfloat16 -- 16 bytes for the eight channels
template <int C, bool XB>
__device__ __forceinline__ void cb_load_x(const CbArgs& a, int64_t r, float (&v)[C])
{
#pragma unroll
    for (int q = 0; q < C / 4; ++q) {
        const float4 t = load4_any(a.x, (size_t)r * a.ldx + 4 * q, XB);
        v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
    }
}
// a row of a gradient tensor (dz, dx): the format of the activation rows it belongs to (B16: bfloat16, ld in elements)
template <int C, bool B16>
__device__ __forceinline__ void cb_load_g(const float* __restrict__ p, int64_t ld, int64_t r, float (&v)[C])
{
#pragma unroll
    for (int q = 0; q < C / 4; ++q) {
        const float4 t = load4_any(p, (size_t)r * ld + 4 * q, B16);
        v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
    }
}
template <int C>
__device__ __forceinline__ void cb_load_row(const float* __restrict__ p, int64_t ld, int64_t r, float (&v)[C])
{
#pragma unroll
    for (int q = 0; q < C / 4; ++q) {
        const float4 t = *reinterpret_cast<const float4*>(p + r * ld + 4 * q);
        v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
    }
}

template <int C>
struct CbCols {
    float v[C];
    __device__ __forceinline__ CbCols(const float* p)
    {
#pragma unroll
        for (int j = 0; j < C; ++j) v[j] = p[j];
    }
};

// workgroup sum of NV per-thread values in a fixed order: xor butterfly inside the wave, the four waves in order through LDS
template <class T, int NV>
__device__ __forceinline__ void cb_block_reduce(T (&v)[NV], T* red /* [waves][NV] */, T* dst /* this workgroup's partial [NV] */)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        T s = v[i];
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) s += __shfl_xor(s, m);
        if (lane == 0) red[wave * NV + i] = s;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NV; i += kCbThreads) {
        T s = 0;
        for (int w = 0; w < kCbThreads / 64; ++w) s += red[w * NV + i];
        dst[i] = s;
    }
}

// ---- forward: statistics.  partial layout per workgroup (doubles): sy[CP] | sq[CP] | sx[CP], CP = 16 (the layout of smallconv_train.hip)
template <int C, bool XB>
__global__ __launch_bounds__(kCbThreads) void cb_sums_kernel(CbArgs a)
{
    constexpr int CP = 16;
    __shared__ double red[(kCbThreads / 64) * 3 * C];
    CbWeights<C> wt;
    wt.stage(a);
    double acc[3 * C];
#pragma unroll
    for (int i = 0; i < 3 * C; ++i) acc[i] = 0.;
    for (int64_t r = blockIdx.x * (int64_t)kCbThreads + threadIdx.x; r < a.R; r += (int64_t)gridDim.x * kCbThreads) {
        float x[C], y[C];
        cb_load_x<C, XB>(a, r, x);
        wt.product(x, y);
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const double yd = (double)y[j];
            acc[j] += yd;
            acc[C + j] = __builtin_fma(yd, yd, acc[C + j]);
            acc[2 * C + j] += (double)x[j];
        }
    }
    __shared__ double tot[3 * C];
    cb_block_reduce<double, 3 * C>(acc, red, tot);
    __syncthreads();
    double* dst = static_cast<double*>(a.part) + (size_t)blockIdx.x * 3 * CP;
    for (int i = threadIdx.x; i < 3 * CP; i += kCbThreads) {
        const int blk = i / CP, j = i - blk * CP;
        dst[i] = j < C ? tot[blk * C + j] : 0.;
    }
}

// ---- forward: normalise + LeakyReLU -> rows
template <int C, bool XB>
__global__ __launch_bounds__(kCbThreads) void cb_apply_kernel(CbArgs a)
{
    CbWeights<C> wt;
    wt.stage(a);
    const CbCols<C> mu(a.mean), sc(a.scale), be(a.beta);
    for (int64_t r = blockIdx.x * (int64_t)kCbThreads + threadIdx.x; r < a.R; r += (int64_t)gridDim.x * kCbThreads) {
        float x[C], y[C];
        cb_load_x<C, XB>(a, r, x);
        wt.product(x, y);
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const float z = __builtin_fmaf(y[j] - mu.v[j], sc.v[j], be.v[j]);
            y[j] = z < 0.f ? 0.2f * z : z;
        }
#pragma unroll
        for (int q = 0; q < C / 4; ++q)
            store4_any(a.out, (size_t)r * a.ldo + 4 * q, make_float4(y[4 * q], y[4 * q + 1], y[4 * q + 2], y[4 * q + 3]), XB);
    }
}

// ---- backward: S1 = sum g, S2 = sum g xh.  partial layout per workgroup (floats): S1[C] | S2[C]
template <int C, bool XB>
__global__ __launch_bounds__(kCbThreads) void cb_bwd_sums_kernel(CbArgs a)
{
    __shared__ float red[(kCbThreads / 64) * 2 * C];
    CbWeights<C> wt;
    wt.stage(a);
    const CbCols<C> mu(a.mean), is(a.invstd), sc(a.scale), be(a.beta);
    float acc[2 * C];
#pragma unroll
    for (int i = 0; i < 2 * C; ++i) acc[i] = 0.f;
    for (int64_t r = blockIdx.x * (int64_t)kCbThreads + threadIdx.x; r < a.R; r += (int64_t)gridDim.x * kCbThreads) {
        float x[C], y[C], g[C];
        cb_load_x<C, XB>(a, r, x);
        cb_load_g<C, XB>(a.dz, a.lddz, r, g);
        wt.product(x, y);
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const float yc = y[j] - mu.v[j];
            const float xh = yc * is.v[j];
            const float gv = __builtin_fmaf(yc, sc.v[j], be.v[j]) < 0.f ? 0.2f * g[j] : g[j];
            acc[j] += gv;
            acc[C + j] = __builtin_fmaf(gv, xh, acc[C + j]);
        }
    }
    cb_block_reduce<float, 2 * C>(acc, red, static_cast<float*>(a.part) + (size_t)blockIdx.x * 2 * C);
}

// ---- backward: input gradient rows + weight / bias gradient.  partial layout per workgroup (floats): dW[C][C] | db[C]
template <int C, bool XB>
__global__ __launch_bounds__(kCbThreads) void cb_bwd_apply_kernel(CbArgs a)
{
    constexpr int NV = C * C + C;
    __shared__ float red[(kCbThreads / 64) * NV];
    CbWeights<C> wt;
    wt.stage(a);
    const CbCols<C> mu(a.mean), is(a.invstd), sc(a.scale), be(a.beta);
    float m1[C], m2[C];
#pragma unroll
    for (int j = 0; j < C; ++j) {
        m1[j] = a.s12[j] * a.inv_rows;
        m2[j] = a.s12[C + j] * a.inv_rows;
    }
    float acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = 0.f;
    for (int64_t r = blockIdx.x * (int64_t)kCbThreads + threadIdx.x; r < a.R; r += (int64_t)gridDim.x * kCbThreads) {
        float x[C], y[C], g[C];
        cb_load_x<C, XB>(a, r, x);
        cb_load_g<C, XB>(a.dz, a.lddz, r, g);
        float old[C];
        if (a.accum) cb_load_g<C, XB>(a.out, a.ldo, r, old);
        wt.product(x, y);
        float dy[C];
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const float yc = y[j] - mu.v[j];
            const float xh = yc * is.v[j];
            const float gv = __builtin_fmaf(yc, sc.v[j], be.v[j]) < 0.f ? 0.2f * g[j] : g[j];
            dy[j] = sc.v[j] * (gv - m1[j] - xh * m2[j]);
            acc[C * C + j] += dy[j];
        }
        // dx[i] = sum_j dy[j] W[i][j]
        float dx[C];
#pragma unroll
        for (int i = 0; i < C; ++i) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < C; ++j) s = __builtin_fmaf(dy[j], wt.W[i * C + j], s);
            dx[i] = a.accum ? old[i] + s : s;
#pragma unroll
            for (int j = 0; j < C; ++j) acc[i * C + j] = __builtin_fmaf(x[i], dy[j], acc[i * C + j]);
        }
#pragma unroll
        for (int q = 0; q < C / 4; ++q)
            store4_any(a.out, (size_t)r * a.ldo + 4 * q, make_float4(dx[4 * q], dx[4 * q + 1], dx[4 * q + 2], dx[4 * q + 3]), XB);
    }
    cb_block_reduce<float, NV>(acc, red, static_cast<float*>(a.part) + (size_t)blockIdx.x * NV);
}

static int cb_blocks(int64_t R) { return (int)std::max<int64_t>(1, std::min<int64_t>((R + kCbThreads - 1) / kCbThreads, kCbBlocks)); }

// entry points for smallconv_train.hip's dispatch (C = 8 only)
int convbn_rows_sums(ps_context* c, const float* x, int64_t ldx, const float* w, const float* b, int64_t R, double* sums)
{
    CbArgs a = {};
    a.x_bf16 = c->train_act_bf16 && c->train_bf16 ? 1 : 0;  // (x, apply's out and the gradient rows dz / dx as bfloat16: ps_set_train_act_bf16)
    a.x = x; a.ldx = (int)ldx; a.w = w; a.b = b; a.R = R;
    const int blocks = cb_blocks(R);
    PS_TRY(c->red_ws.reserve(sizeof(double) * (size_t)blocks * 48 + 256));
    a.part = c->red_ws.as<void>();
    if (a.x_bf16) hipLaunchKernelGGL((cb_sums_kernel<8, true>), dim3(blocks), dim3(kCbThreads), 0, c->stream, a);
    else hipLaunchKernelGGL((cb_sums_kernel<8, false>), dim3(blocks), dim3(kCbThreads), 0, c->stream, a);
    hipLaunchKernelGGL(reduce_partials_kernel<double>, dim3(3), dim3(256), 0, c->stream, static_cast<const double*>(a.part), blocks, 48, sums);
    PS_HIP(hipGetLastError());
    return PS_OK;
}

int convbn_rows_apply(ps_context* c, const float* x, int64_t ldx, const float* w, const float* b, int64_t R, const float* mean, const float* scale,
                      const float* beta, float* out, int64_t ldo)
{
    CbArgs a = {};
    a.x_bf16 = c->train_act_bf16 && c->train_bf16 ? 1 : 0;  // (x, apply's out and the gradient rows dz / dx as bfloat16: ps_set_train_act_bf16)
    a.x = x; a.ldx = (int)ldx; a.w = w; a.b = b; a.R = R; a.mean = mean; a.scale = scale; a.beta = beta; a.out = out; a.ldo = (int)ldo;
    if (a.x_bf16) hipLaunchKernelGGL((cb_apply_kernel<8, true>), dim3(cb_blocks(R)), dim3(kCbThreads), 0, c->stream, a);
    else hipLaunchKernelGGL((cb_apply_kernel<8, false>), dim3(cb_blocks(R)), dim3(kCbThreads), 0, c->stream, a);
    PS_HIP(hipGetLastError());
    return PS_OK;
}

int convbn_rows_bwd_sums(ps_context* c, const float* x, int64_t ldx, const float* w, const float* b, int64_t R, const float* mean, const float* invstd,
                         const float* scale, const float* beta, const float* dz, int64_t lddz, float* s12)
{
    CbArgs a = {};
    a.x_bf16 = c->train_act_bf16 && c->train_bf16 ? 1 : 0;  // (x, apply's out and the gradient rows dz / dx as bfloat16: ps_set_train_act_bf16)
    a.x = x; a.ldx = (int)ldx; a.w = w; a.b = b; a.R = R; a.mean = mean; a.invstd = invstd; a.scale = scale; a.beta = beta; a.dz = dz; a.lddz = (int)lddz;
    const int blocks = cb_blocks(R);
    PS_TRY(c->red_ws.reserve(sizeof(float) * (size_t)blocks * 16 + 256));
    a.part = c->red_ws.as<void>();
    if (a.x_bf16) hipLaunchKernelGGL((cb_bwd_sums_kernel<8, true>), dim3(blocks), dim3(kCbThreads), 0, c->stream, a);
    else hipLaunchKernelGGL((cb_bwd_sums_kernel<8, false>), dim3(blocks), dim3(kCbThreads), 0, c->stream, a);
    hipLaunchKernelGGL(reduce_partials_kernel<float>, dim3(1), dim3(256), 0, c->stream, static_cast<const float*>(a.part), blocks, 16, s12);
    PS_HIP(hipGetLastError());
    return PS_OK;
}

int convbn_rows_bwd_apply(ps_context* c, const float* x, int64_t ldx, const float* w, const float* b, int64_t R, const float* mean, const float* invstd,
                          const float* scale, const float* beta, const float* s12, float inv_rows, const float* dz, int64_t lddz, int accumulate, float* dx,
                          int64_t lddx, float* dw, float* db)
{
    CbArgs a = {};
    a.x_bf16 = c->train_act_bf16 && c->train_bf16 ? 1 : 0;  // (x, apply's out and the gradient rows dz / dx as bfloat16: ps_set_train_act_bf16)
    a.x = x; a.ldx = (int)ldx; a.w = w; a.b = b; a.R = R; a.mean = mean; a.invstd = invstd; a.scale = scale; a.beta = beta; a.s12 = s12;
    a.inv_rows = inv_rows; a.dz = dz; a.lddz = (int)lddz; a.out = dx; a.ldo = (int)lddx; a.accum = accumulate ? 1 : 0;
    const int blocks = cb_blocks(R);
    PS_TRY(c->red_ws.reserve(sizeof(float) * (size_t)blocks * 72 + 256));
    a.part = c->red_ws.as<void>();
    if (a.x_bf16) hipLaunchKernelGGL((cb_bwd_apply_kernel<8, true>), dim3(blocks), dim3(kCbThreads), 0, c->stream, a);
    else hipLaunchKernelGGL((cb_bwd_apply_kernel<8, false>), dim3(blocks), dim3(kCbThreads), 0, c->stream, a);
    hipLaunchKernelGGL(reduce_partials2_kernel<float>, dim3(ceil_div(72, 16)), dim3(256), 0, c->stream, static_cast<const float*>(a.part), blocks, 72, 64, dw, db);
    PS_HIP(hipGetLastError());
    return PS_OK;
}

}  // namespace ps
