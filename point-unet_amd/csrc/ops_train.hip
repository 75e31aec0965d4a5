// ops_train.hip -- device ops of the TRAINING step (forward pieces that differ from inference + every backward):
// the reference trains with TF autodiff over the graph of PointSegment/RandLANet.py:110-152, 314-401 with
// tf.layers.batch_normalization(training=True) (helper_tf_util.py:167,246; RandLANet.py:115), the class-weighted
// softmax cross-entropy of RandLANet.py:267-274 and tf.train.AdamOptimizer (RandLANet.py:89).  The host graph
// (point-unet_amd/train.py) records a tape of these ops; the kernels here are the op-level forward/backward pairs.
//
// All tensors are dense row-major fp32 [rows, channels] on the device.  Per-channel reductions over rows (BatchNorm
// statistics, bias / gamma / beta gradients) are two-stage with a fixed merge order, so the forward pass is run-to-run
// bit-identical; the weight-gradient GEMM and the scatter-adds use float atomics (summation order not fixed, ~1e-6).
// Bound: every kernel here is HBM-bound (one or two passes over [rows, C]) except linear_wgrad, which is an
// MFMA GEMM with the row axis as K.
#include <type_traits>

#include "common.h"
#include "mfma_tile.h"
#include "wave_ops.h"

namespace ps {

// ---- per-channel sums over rows: out0[c] += sum_r f0(r,c), out1[c] += sum_r f1(r,c) -------------------------------
// Layout trick: a block covers a contiguous slab of rows; thread t handles elements t, t+T, ... of the slab; with
// T % C == 0 its channel never changes.  (C that does not divide T falls back to per-element modulo.)
template <class F>
__global__ __launch_bounds__(256) void colreduce2_kernel(F f, int64_t R, int C, int rows_per_block, int vec, float* __restrict__ part0,
                                                         float* __restrict__ part1)
{
    // part{0,1}[block][c]: per-block partial sums, merged in a fixed order by colreduce_finish_kernel (no float atomics:
    // the batch statistics, and with them which side of the leaky-ReLU kink an activation falls on, are run-to-run identical)
    __shared__ float s0[256], s1[256];
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < R ? r0 + rows_per_block : R;
    const int64_t e0 = r0 * C, e1 = r1 * C;
    const bool fixed = (256 % C) == 0;
    if (vec) {
        // float4 path: thread t owns elements 4t..4t+3 of every 1024-element stripe of the slab, i.e. four fixed channels
        __shared__ float v0s[1024], v1s[1024];
        const int c = (4 * threadIdx.x) % C;
        f.init4(c);
        float a[4] = {0.f, 0.f, 0.f, 0.f}, b[4] = {0.f, 0.f, 0.f, 0.f};
        int64_t e = e0 + 4 * threadIdx.x;
        for (; e + 1024 < e1; e += 2048) {
            float p0[4], p1[4], q0[4], q1[4];
            f.load4(e, c, p0, p1);
            f.load4(e + 1024, c, q0, q1);
#pragma unroll
            for (int j = 0; j < 4; ++j) { a[j] += p0[j]; b[j] += p1[j]; }
#pragma unroll
            for (int j = 0; j < 4; ++j) { a[j] += q0[j]; b[j] += q1[j]; }
        }
        if (e < e1) {
            float p0[4], p1[4];
            f.load4(e, c, p0, p1);
#pragma unroll
            for (int j = 0; j < 4; ++j) { a[j] += p0[j]; b[j] += p1[j]; }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { v0s[4 * threadIdx.x + j] = a[j]; v1s[4 * threadIdx.x + j] = b[j]; }
        __syncthreads();
        for (int ch = threadIdx.x; ch < C; ch += 256) {
            float t0 = 0.f, t1 = 0.f;
            for (int i = ch; i < 1024; i += C) { t0 += v0s[i]; t1 += v1s[i]; }
            part0[(size_t)blockIdx.x * C + ch] = t0;
            if (part1) part1[(size_t)blockIdx.x * C + ch] = t1;
        }
        return;
    }
    float a0 = 0.f, a1 = 0.f;
    if (fixed) {
        const int c = threadIdx.x % C;
        for (int64_t e = e0 + threadIdx.x; e < e1; e += 256) {
            float v0, v1;
            f(e, c, v0, v1);
            a0 += v0;
            a1 += v1;
        }
        s0[threadIdx.x] = a0;
        s1[threadIdx.x] = a1;
        __syncthreads();
        if ((int)threadIdx.x < C) {
            float t0 = 0.f, t1 = 0.f;
            for (int i = threadIdx.x; i < 256; i += C) { t0 += s0[i]; t1 += s1[i]; }
            part0[(size_t)blockIdx.x * C + threadIdx.x] = t0;
            if (part1) part1[(size_t)blockIdx.x * C + threadIdx.x] = t1;
        }
    } else {
        // generic: thread per channel group, rows strided
        for (int c = threadIdx.x; c < C; c += 256) {
            float t0 = 0.f, t1 = 0.f;
            for (int64_t r = r0; r < r1; ++r) {
                float v0, v1;
                f(r * C + c, c, v0, v1);
                t0 += v0;
                t1 += v1;
            }
            part0[(size_t)blockIdx.x * C + c] = t0;
            if (part1) part1[(size_t)blockIdx.x * C + c] = t1;
        }
    }
}

// out{0,1}[c] = sum_b part{0,1}[b][c]; one 64-lane wave per channel, lane-strided partials then a fixed shuffle tree
__global__ __launch_bounds__(64) void colreduce_finish_kernel(const float* __restrict__ part0, const float* __restrict__ part1, int blocks, int C,
                                                              float* __restrict__ out0, float* __restrict__ out1)
{
    const int c = blockIdx.x;
    // (four independent load chains per lane and output: up to 2 048 partials were 32 dependent L2 round trips per lane)
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
    int b = threadIdx.x;
    for (; b + 192 < blocks; b += 256) {
        a0 += part0[(size_t)b * C + c]; a1 += part0[(size_t)(b + 64) * C + c]; a2 += part0[(size_t)(b + 128) * C + c]; a3 += part0[(size_t)(b + 192) * C + c];
        if (part1) {
            b0 += part1[(size_t)b * C + c]; b1 += part1[(size_t)(b + 64) * C + c]; b2 += part1[(size_t)(b + 128) * C + c]; b3 += part1[(size_t)(b + 192) * C + c];
        }
    }
    for (; b < blocks; b += 64) {
        a0 += part0[(size_t)b * C + c];
        if (part1) b0 += part1[(size_t)b * C + c];
    }
    float t0 = (a0 + a1) + (a2 + a3), t1 = (b0 + b1) + (b2 + b3);
    for (int o = 32; o; o >>= 1) {
        t0 += __shfl_down(t0, o);
        t1 += __shfl_down(t1, o);
    }
    if (threadIdx.x == 0) {
        out0[c] = t0;
        if (out1) out1[c] = t1;
    }
}

// BatchNorm statistics finished in the reduction's own second stage: [sum | sum of squares] -> mean, population variance, invstd and
// (optionally) the moving-statistics update  moving = momentum * moving + (1 - momentum) * batch  (the reference's extra_update_ops,
// RandLANet.py:90,163) -- one launch instead of three
struct BnFinish {
    float* mean; float* invstd; float* var; float* mov_mean; float* mov_var;
    float rows, eps, momentum;
};
__global__ __launch_bounds__(64) void colreduce_finish_bn_kernel(const float* __restrict__ part0, const float* __restrict__ part1, int blocks, int C,
                                                                 float* __restrict__ out0, float* __restrict__ out1, BnFinish bn)
{
    const int c = blockIdx.x;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
    int b = threadIdx.x;
    for (; b + 192 < blocks; b += 256) {
        a0 += part0[(size_t)b * C + c]; a1 += part0[(size_t)(b + 64) * C + c]; a2 += part0[(size_t)(b + 128) * C + c]; a3 += part0[(size_t)(b + 192) * C + c];
        b0 += part1[(size_t)b * C + c]; b1 += part1[(size_t)(b + 64) * C + c]; b2 += part1[(size_t)(b + 128) * C + c]; b3 += part1[(size_t)(b + 192) * C + c];
    }
    for (; b < blocks; b += 64) {
        a0 += part0[(size_t)b * C + c];
        b0 += part1[(size_t)b * C + c];
    }
    float t0 = (a0 + a1) + (a2 + a3), t1 = (b0 + b1) + (b2 + b3);
    for (int o = 32; o; o >>= 1) {
        t0 += __shfl_down(t0, o);
        t1 += __shfl_down(t1, o);
    }
    if (threadIdx.x == 0) {
        out0[c] = t0;
        out1[c] = t1;
        const float m = t0 / bn.rows;
        float v = t1 / bn.rows - m * m;  // population variance (tf.nn.moments)
        v = v < 0.f ? 0.f : v;
        bn.mean[c] = m;
        bn.var[c] = v;
        bn.invstd[c] = rsqrtf(v + bn.eps);
        if (bn.mov_mean) {
            bn.mov_mean[c] = bn.mov_mean[c] * bn.momentum + m * (1.f - bn.momentum);
            bn.mov_var[c] = bn.mov_var[c] * bn.momentum + v * (1.f - bn.momentum);
        }
    }
}

template <class F>
static int colreduce2(ps_context* c, F f, int64_t R, int C, float* out0, float* out1, const BnFinish* bn = nullptr)
{
    if (R <= 0) {
        PS_HIP(hipMemsetAsync(out0, 0, sizeof(float) * C, c->stream));
        if (out1) PS_HIP(hipMemsetAsync(out1, 0, sizeof(float) * C, c->stream));
        return PS_OK;
    }
    // (4 096 elements per block while that keeps the grid within 2 048 blocks: a thread's walk is a chain of dependent round trips -- two
    //  16-byte loads per trip -- and the small layers of a one-cloud step spent 8 of them per launch: 10.4 us floor for the BatchNorm
    //  backward sums at 16 384 elements per block)
    int64_t blocks = (R * C + 256 * 16 - 1) / (256 * 16);
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
    if ((256 % C) != 0) blocks = blocks > 512 ? 512 : blocks;
    const int rpb = (int)((R + blocks - 1) / blocks);
    const int nb = (int)((R + rpb - 1) / rpb);
    PS_TRY(c->red_ws.reserve(sizeof(float) * 2 * (size_t)nb * C));
    float* p0 = c->red_ws.as<float>();
    float* p1 = out1 ? p0 + (size_t)nb * C : nullptr;
    const int vec = (C & 3) == 0 && (1024 % C) == 0 && f.aligned16();
    hipLaunchKernelGGL(colreduce2_kernel<F>, dim3((unsigned)nb), dim3(256), 0, c->stream, f, R, C, rpb, vec, p0, p1);
    if (bn)
        hipLaunchKernelGGL(colreduce_finish_bn_kernel, dim3((unsigned)C), dim3(64), 0, c->stream, p0, p1, nb, C, out0, out1, *bn);
    else
        hipLaunchKernelGGL(colreduce_finish_kernel, dim3((unsigned)C), dim3(64), 0, c->stream, p0, p1, nb, C, out0, out1);
    PS_HIP(hipGetLastError());
    return PS_OK;
}

struct SumSq {
    const float* x;
    __device__ void operator()(int64_t e, int, float& a, float& b) const { const float v = x[e]; a = v; b = v * v; }
    bool aligned16() const { return (reinterpret_cast<uintptr_t>(x) & 15) == 0; }
    __device__ void init4(int) {}
    __device__ void load4(int64_t e, int, float (&a)[4], float (&b)[4]) const
    {
        const float4 v = *reinterpret_cast<const float4*>(x + e);
        a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w;
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = a[j] * a[j];
    }
};
struct SumOnly {
    const float* x;
    __device__ void operator()(int64_t e, int, float& a, float& b) const { a = x[e]; b = 0.f; }
    bool aligned16() const { return (reinterpret_cast<uintptr_t>(x) & 15) == 0; }
    __device__ void init4(int) {}
    __device__ void load4(int64_t e, int, float (&a)[4], float (&b)[4]) const
    {
        const float4 v = *reinterpret_cast<const float4*>(x + e);
        a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w;
        b[0] = b[1] = b[2] = b[3] = 0.f;
    }
};
// BatchNorm backward sums: g = dy * act'(z), z = gamma*xhat + beta;  a = sum g, b = sum g*xhat
struct BnBwdSums {
    const float* dy; const float* x; const float* gamma; const float* beta; const float* mean; const float* invstd;
    int leaky;
    int C;          // channels: element e of x is (row e / C, channel e % C)
    int64_t lddy;   // row stride of dy in floats (>= C; dy may be a column block of a wider tensor)
    __device__ void operator()(int64_t e, int c, float& a, float& b) const
    {
        const float xh = (x[e] - mean[c]) * invstd[c];
        float g = dy[(e / C) * lddy + c];
        if (leaky && gamma[c] * xh + beta[c] < 0.f) g *= 0.2f;
        a = g;
        b = g * xh;
    }
    float ms[4], ss[4], gm[4], bt[4];  // the thread's four channels (float4 path)
    int c4, cshift;
    bool aligned16() const { return ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15) == 0 && (lddy & 3) == 0; }
    __device__ void init4(int c)
    {
        c4 = c;
        cshift = 31 - __clz(C);  // the float4 path runs only for power-of-two C
#pragma unroll
        for (int j = 0; j < 4; ++j) { ms[j] = mean[c + j]; ss[j] = invstd[c + j]; gm[j] = gamma[c + j]; bt[j] = beta[c + j]; }
    }
    __device__ void load4(int64_t e, int, float (&a)[4], float (&b)[4]) const
    {
        const float4 xv = *reinterpret_cast<const float4*>(x + e), gv = *reinterpret_cast<const float4*>(dy + (e >> cshift) * lddy + c4);
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, gs[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float xh = (xs[j] - ms[j]) * ss[j];
            float g = gs[j];
            if (leaky && gm[j] * xh + bt[j] < 0.f) g *= 0.2f;
            a[j] = g;
            b[j] = g * xh;
        }
    }
};

__global__ void bn_finish_stats_kernel(const float* __restrict__ sum, const float* __restrict__ sumsq, int64_t R, int C, float eps,
                                       float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ var)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float m = sum[c] / (float)R;
    float v = sumsq[c] / (float)R - m * m;  // population variance (tf.nn.moments)
    v = v < 0.f ? 0.f : v;
    mean[c] = m;
    var[c] = v;
    invstd[c] = rsqrtf(v + eps);
}

// ---- BatchNorm of a SMALL tensor in one launch -------------------------------------------------------------------------------------------
// The deep levels and the decoder of a one-cloud step see a few hundred to ~11 000 rows: statistics + finish + apply were three launches
// (backward: three more) of a few microseconds each, i.e. the launch floor six times per layer.  Here a workgroup owns FOUR channels --
// 16 bytes of every row -- and ALL rows: it sums them (fixed order: per-thread strided partials, then a tree over the 1 024 threads),
// finishes the statistics (and the moving-statistics update of the reference's extra_update_ops, RandLANet.py:90,163) and applies them in a
// second pass over rows that are still in L2 -- no hand-over between workgroups, hence no second launch.  R <= kBnSliceRows, C % 4 == 0.
constexpr int64_t kBnSliceRows = 4096;  // (11 250 rows measured SLOWER than three launches: 9.74 against 8.65 ms per one-cloud step)
constexpr int kBnSliceThreads = 1024;

// A workgroup owns 32 channels = ONE 128-byte line of every row: thread t reads channels 4 (t & 7) .. + 3 of rows t >> 3, + 128, ...  (the
// first form gave a workgroup four channels: 16 bytes of a line per row, every line fetched by eight workgroups -- 29 us per launch for the
// few-thousand-row layers it exists for).  Sums in a fixed order: a shuffle tree over the eight row slots of a wave that share a channel
// quad, then the sixteen wave totals in wave order -- one workgroup barrier.
constexpr int kBnSliceCh = 32;
__device__ __forceinline__ void bn_slice_reduce(float (&a)[4], float (&b)[4], float (*red)[8][8])
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, cq = lane & 7;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int o = 32; o >= 8; o >>= 1) {
            a[j] += __shfl_down(a[j], o);
            b[j] += __shfl_down(b[j], o);
        }
    }
    if (lane < 8) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { red[wave][cq][j] = a[j]; red[wave][cq][4 + j] = b[j]; }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int w = 0; w < kBnSliceThreads / 64; ++w) { s += red[w][cq][j]; q += red[w][cq][4 + j]; }
        a[j] = s;
        b[j] = q;
    }
}

__global__ __launch_bounds__(kBnSliceThreads) void bn_slice_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, int R,
                                                           int C, float eps, int leaky, float* __restrict__ y, int64_t ldy, float* __restrict__ mean,
                                                           float* __restrict__ invstd, float* __restrict__ var, float* __restrict__ sums,
                                                           float* __restrict__ mov_mean, float* __restrict__ mov_var, float momentum)
{
    __shared__ float red[kBnSliceThreads / 64][8][8];
    const int c0 = kBnSliceCh * blockIdx.x + 4 * (threadIdx.x & 7), r0 = threadIdx.x >> 3;
    constexpr int RS = kBnSliceThreads / 8;  // row slots
    float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
    // (eight rows per trip: a thread's walk is a chain of dependent round trips -- one row per trip is 32 of them for 4 096 rows)
    for (int r = r0; r < R; r += 8 * RS) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = r + u * RS < R ? *reinterpret_cast<const float4*>(x + (size_t)(r + u * RS) * C + c0) : float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            s[0] += v[u].x; s[1] += v[u].y; s[2] += v[u].z; s[3] += v[u].w;
            q[0] += v[u].x * v[u].x; q[1] += v[u].y * v[u].y; q[2] += v[u].z * v[u].z; q[3] += v[u].w * v[u].w;
        }
    }
    bn_slice_reduce(s, q, red);
    float m[4], is[4], ga[4], be[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        m[j] = s[j] / (float)R;
        float v = q[j] / (float)R - m[j] * m[j];  // population variance (tf.nn.moments)
        v = v < 0.f ? 0.f : v;
        is[j] = rsqrtf(v + eps);
        ga[j] = gamma[c0 + j];
        be[j] = beta[c0 + j];
        if (threadIdx.x < 8) {
            mean[c0 + j] = m[j];
            var[c0 + j] = v;
            invstd[c0 + j] = is[j];
            if (sums) { sums[c0 + j] = s[j]; sums[C + c0 + j] = q[j]; }
            if (mov_mean) {
                mov_mean[c0 + j] = mov_mean[c0 + j] * momentum + m[j] * (1.f - momentum);
                mov_var[c0 + j] = mov_var[c0 + j] * momentum + v * (1.f - momentum);
            }
        }
    }
    for (int r = r0; r < R; r += 8 * RS) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = r + u * RS < R ? *reinterpret_cast<const float4*>(x + (size_t)(r + u * RS) * C + c0) : float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            float z[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                z[j] = ga[j] * ((z[j] - m[j]) * is[j]) + be[j];
                if (leaky && z[j] < 0.f) z[j] *= 0.2f;
            }
            if (r + u * RS < R) *reinterpret_cast<float4*>(y + (size_t)(r + u * RS) * ldy + c0) = float4{z[0], z[1], z[2], z[3]};
        }
    }
}

__global__ __launch_bounds__(kBnSliceThreads) void bn_slice_bwd_kernel(const float* __restrict__ dy, int64_t lddy, const float* __restrict__ x, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           int R, int C, int leaky, float* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta)
{
    __shared__ float red[kBnSliceThreads / 64][8][8];
    const int c0 = kBnSliceCh * blockIdx.x + 4 * (threadIdx.x & 7), r0 = threadIdx.x >> 3;
    constexpr int RS = kBnSliceThreads / 8;
    float m[4], is[4], ga[4], be[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { m[j] = mean[c0 + j]; is[j] = invstd[c0 + j]; ga[j] = gamma[c0 + j]; be[j] = beta[c0 + j]; }
    float sg[4] = {0.f, 0.f, 0.f, 0.f}, sgx[4] = {0.f, 0.f, 0.f, 0.f};
    const float4 zero4 = {0.f, 0.f, 0.f, 0.f};
    for (int r = r0; r < R; r += 4 * RS) {
        float4 xv[4], gv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool in = r + u * RS < R;
            xv[u] = in ? *reinterpret_cast<const float4*>(x + (size_t)(r + u * RS) * C + c0) : zero4;
            gv[u] = in ? *reinterpret_cast<const float4*>(dy + (size_t)(r + u * RS) * lddy + c0) : zero4;  // (g = 0: adds nothing)
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float xs[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w};
            float g[4] = {gv[u].x, gv[u].y, gv[u].z, gv[u].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float xh = (xs[j] - m[j]) * is[j];
                if (leaky && ga[j] * xh + be[j] < 0.f) g[j] *= 0.2f;
                sg[j] += g[j];
                sgx[j] += g[j] * xh;
            }
        }
    }
    bn_slice_reduce(sg, sgx, red);
    if (threadIdx.x < 8) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { dbeta[c0 + j] = sg[j]; dgamma[c0 + j] = sgx[j]; }
    }
    const float invR = 1.0f / (float)R;
    for (int r = r0; r < R; r += 4 * RS) {
        float4 xv[4], gv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool in = r + u * RS < R;
            xv[u] = in ? *reinterpret_cast<const float4*>(x + (size_t)(r + u * RS) * C + c0) : zero4;
            gv[u] = in ? *reinterpret_cast<const float4*>(dy + (size_t)(r + u * RS) * lddy + c0) : zero4;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float xs[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w};
            float g[4] = {gv[u].x, gv[u].y, gv[u].z, gv[u].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float xh = (xs[j] - m[j]) * is[j];
                if (leaky && ga[j] * xh + be[j] < 0.f) g[j] *= 0.2f;
                g[j] = ga[j] * is[j] * (g[j] - sg[j] * invR - xh * sgx[j] * invR);
            }
            if (r + u * RS < R) *reinterpret_cast<float4*>(dx + (size_t)(r + u * RS) * C + c0) = float4{g[0], g[1], g[2], g[3]};
        }
    }
}

static bool bn_slice_ok(const Tuning& tn, int64_t R, int64_t C, const void* x, const void* y, int64_t ldy)
{
    // OFF by default: measured on MI355X the one-cloud step makes 39 launches fewer with it (644 -> 605) and takes the same time
    // (8.56 against 8.61 ms; with 11 250-row layers included 9.74) -- the step's small kernels already run back to back, a launch less is
    // not time less.  PS_BN_SLICE=1 switches it on for A/B.
    return tn.bn_slice && R <= kBnSliceRows && C % kBnSliceCh == 0 && ldy % 4 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
}

// Elementwise BatchNorm kernels.  VEC: float4 per thread with a grid stride that is a multiple of C (1024 % C == 0), so a
// thread's four channels never change and their parameters are loaded once.
template <bool VEC>
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ mean, const float* __restrict__ invstd, int64_t total, int C, int leaky,
                                                       float* __restrict__ y, int64_t ldy)
{
    if (VEC) {
        const int64_t first = 4 * (blockIdx.x * (int64_t)256 + threadIdx.x);
        const int c = (int)(first % C);
        const int cshift = 31 - __clz(C);  // power-of-two C on this path
        float sc[4], sh[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { sc[j] = invstd[c + j]; sh[j] = mean[c + j]; }
        float ga[4], be[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { ga[j] = gamma[c + j]; be[j] = beta[c + j]; }
        for (int64_t e = first; e < total; e += (int64_t)gridDim.x * 1024) {
            const float4 v = *reinterpret_cast<const float4*>(x + e);
            float z[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                z[j] = ga[j] * ((z[j] - sh[j]) * sc[j]) + be[j];
                if (leaky && z[j] < 0.f) z[j] *= 0.2f;
            }
            *reinterpret_cast<float4*>(y + (e >> cshift) * ldy + c) = float4{z[0], z[1], z[2], z[3]};
        }
        return;
    }
    for (int64_t e = blockIdx.x * (int64_t)256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % C);
        float z = gamma[c] * ((x[e] - mean[c]) * invstd[c]) + beta[c];
        if (leaky && z < 0.f) z *= 0.2f;
        y[(e / C) * ldy + c] = z;
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ sg, const float* __restrict__ sgx,
                                                           int64_t total, int C, float invR, int leaky, float* __restrict__ dx, int64_t lddy)
{
    // dx = gamma*invstd * (g - mean_r(g) - xhat*mean_r(g*xhat))
    if (VEC) {
        const int64_t first = 4 * (blockIdx.x * (int64_t)256 + threadIdx.x);
        const int c = (int)(first % C);
        const int cshift = 31 - __clz(C);  // power-of-two C on this path
        float sc[4], sh[4], ga[4], be[4], mg[4], mgx[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sc[j] = invstd[c + j]; sh[j] = mean[c + j]; ga[j] = gamma[c + j]; be[j] = beta[c + j];
            mg[j] = sg[c + j] * invR; mgx[j] = sgx[c + j] * invR;
        }
        for (int64_t e = first; e < total; e += (int64_t)gridDim.x * 1024) {
            const float4 xv = *reinterpret_cast<const float4*>(x + e), gv = *reinterpret_cast<const float4*>(dy + (e >> cshift) * lddy + c);
            const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
            float g[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float xh = (xs[j] - sh[j]) * sc[j];
                if (leaky && ga[j] * xh + be[j] < 0.f) g[j] *= 0.2f;
                g[j] = ga[j] * sc[j] * (g[j] - mg[j] - xh * mgx[j]);
            }
            *reinterpret_cast<float4*>(dx + e) = float4{g[0], g[1], g[2], g[3]};
        }
        return;
    }
    for (int64_t e = blockIdx.x * (int64_t)256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % C);
        const float xh = (x[e] - mean[c]) * invstd[c];
        float g = dy[(e / C) * lddy + c];
        if (leaky && gamma[c] * xh + beta[c] < 0.f) g *= 0.2f;
        dx[e] = gamma[c] * invstd[c] * (g - sg[c] * invR - xh * sgx[c] * invR);
    }
}

static inline bool bn_vec_ok(int64_t C, const void* a, const void* b, const void* c3, int64_t ld = 0)
{
    return (C & 3) == 0 && (1024 % C) == 0 && (ld & 3) == 0 &&
           ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c3)) & 15) == 0;
}

// ---- dW[cin,cout] += X^T . dY over a slab of rows; MFMA with the row axis as K -----------------------------------
// A workgroup of 4 waves owns a [(TI*WI*16) x (TJ*16)] block of dW (WI = 4/WK waves side by side along cin, each holding
// TI x TJ accumulator tiles) and a slab of rows.  The slab goes through LDS in chunks of wg_chunk() rows: both operands are
// read from HBM once per workgroup with full-row coalesced loads, and every A fragment feeds TJ MFMAs, every B fragment TI.
// When the block has fewer tiles than waves (narrow layers: 10->8, 16->16 ...) the WK wave groups split the k-steps of a
// chunk instead.  A[i][k] = X[r0+k][c0+i], B[k][j] = dY[r0+k][n0+j]; the LDS row strides are = 16 (mod 32) floats, so the
// four rows of a k-step land on disjoint banks.  The bias gradient (column sums of dY) rides along on the staged dY chunk.
__host__ __device__ constexpr int wg_chunk(int cols) { return cols <= 32 ? 256 : (cols <= 64 ? 128 : (cols <= 128 ? 64 : 32)); }  // ~8k staged floats
__host__ __device__ constexpr int wg_stride(int cols) { return ((cols + 15) / 32) * 32 + 16; }

// One operand's share of a chunk in flight between HBM and LDS: thread t holds items t, t+256, ... of the [CH x COLS] block
// (an item = one float4 with VEC, one float without); 256 % (items per row) == 0, so a thread's column never changes.
template <int COLS, int CH, bool VEC>
struct WgStage {
    static constexpr int Q = VEC ? COLS / 4 : COLS;
    static constexpr int N = (CH * Q + 255) / 256;
    using item = typename std::conditional<VEC, float4, float>::type;
    item v[N];
    __device__ __forceinline__ void load(const float* __restrict__ src, int ld, int c0, int w, int live, int64_t r)
    {
        const int q = threadIdx.x % Q;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int row = (threadIdx.x + 256 * i) / Q;
            const bool ok = row < live && (VEC ? 4 * q : q) < w;
            if constexpr (VEC) v[i] = ok ? *reinterpret_cast<const float4*>(src + (r + row) * ld + c0 + 4 * q) : float4{0.f, 0.f, 0.f, 0.f};
            else v[i] = ok ? src[(r + row) * ld + c0 + q] : 0.f;
        }
    }
    __device__ __forceinline__ void store(float* dst, int stride) const
    {
        const int q = threadIdx.x % Q;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int row = (threadIdx.x + 256 * i) / Q;
            if (row < CH) {
                if constexpr (VEC) *reinterpret_cast<float4*>(dst + row * stride + 4 * q) = v[i];
                else dst[row * stride + q] = v[i];
            }
        }
    }
};

// BF16: the k-step is 16 rows on v_mfma_f32_16x16x16_bf16 (lane (i, g) supplies rows 4g..4g+3 of the step), operands rounded to
// bf16 as they leave LDS; the LDS row strides are then = 4 (mod 16) floats so that the four row groups hit disjoint banks.
typedef short bf16x4s __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2t __attribute__((ext_vector_type(2)));
typedef float f32x2t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16x4s to_bf16x4(float a, float b, float c, float d)
{
    union { bf16x2t h[2]; bf16x4s s; } u;
    u.h[0] = __builtin_convertvector(f32x2t{a, b}, bf16x2t);
    u.h[1] = __builtin_convertvector(f32x2t{c, d}, bf16x2t);
    return u.s;
}

template <int TI, int TJ, int WK, bool VEC, bool BF16>
__global__ __launch_bounds__(256) void wgrad_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ dy, int lddy, int64_t R, int cin,
                                                    int cout, int64_t rows_per_block, float* __restrict__ dW, float* __restrict__ db)
{
    // dW / db point at this launch's PARTIALS: slab b (= blockIdx.x) owns dW[b][cin][cout] and db[b][cout] and every element of them is
    // written exactly once, by the one workgroup whose tile holds it -- plain stores, no atomics; wgrad_reduce_kernel adds the slabs
    // up in slab order afterwards (deterministic, and nothing has to be zeroed first)
    dW += (size_t)blockIdx.x * cin * cout;
    if (db != nullptr) db += (size_t)blockIdx.x * cout;
    constexpr int WI = 4 / WK;
    constexpr int CI = TI * WI * 16, CJ = TJ * 16;
    constexpr int SX = BF16 ? CI + 4 : wg_stride(CI), SD = BF16 ? CJ + 4 : wg_stride(CJ);
    constexpr int CH = BF16 ? (wg_chunk(CI + CJ) > 16 * WK ? wg_chunk(CI + CJ) : 16 * WK) : wg_chunk(CI + CJ);
    __shared__ float xs[CH * SX], ds[CH * SD];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i16 = lane & 15, k4 = lane >> 4;
    const int wi = wave % WI, wk = wave / WI;
    const int c0 = blockIdx.y * CI, n0 = blockIdx.z * CJ;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < R ? r0 + rows_per_block : R;
    const int wx = min(CI, cin - c0), wd = min(CJ, cout - n0);  // live columns of the two staged blocks
    f32x4 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    WgStage<CI, CH, VEC> gx;
    WgStage<CJ, CH, VEC> gd;
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};  // column sums of the thread's dY items (its column is fixed)
    gx.load(x, ldx, c0, wx, (int)min<int64_t>(CH, r1 - r0), r0);
    gd.load(dy, lddy, n0, wd, (int)min<int64_t>(CH, r1 - r0), r0);
    for (int64_t r = r0; r < r1; r += CH) {
        __syncthreads();  // the previous chunk's fragments have been read
        gx.store(xs, SX);
        gd.store(ds, SD);
#pragma unroll
        for (int i = 0; i < gd.N; ++i) {
            if constexpr (VEC) { bsum[0] += gd.v[i].x; bsum[1] += gd.v[i].y; bsum[2] += gd.v[i].z; bsum[3] += gd.v[i].w; }
            else bsum[0] += gd.v[i];
        }
        __syncthreads();
        if (r + CH < r1) {  // next chunk's loads fly while this one feeds the MFMAs
            const int live = (int)min<int64_t>(CH, r1 - r - CH);
            gx.load(x, ldx, c0, wx, live, r + CH);
            gd.load(dy, lddy, n0, wd, live, r + CH);
        }
        if constexpr (BF16) {
            // (the loop index starts at a compile-time 0: with the wave-dependent start `s = wk` the compiler could not unroll it --
            //  "-Wpass-failed: loop not unrolled", silenced by the Makefile -- and every k-step waited for its own LDS reads; same-box A/B of
            //  the training step: no measurable change, 38.0 / 7.91 ms either way: the fp32 kernel only takes the small products now)
#pragma unroll
            for (int s0 = 0; s0 < CH / 16; s0 += WK) {
                const int s = s0 + wk;
                if (s >= CH / 16) break;
                bf16x4s a[TI], b[TJ];
                const int rb = 16 * s + 4 * k4;
#pragma unroll
                for (int i = 0; i < TI; ++i) {
                    const float* q = xs + rb * SX + (wi * TI + i) * 16 + i16;
                    a[i] = to_bf16x4(q[0], q[SX], q[2 * SX], q[3 * SX]);
                }
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    const float* q = ds + rb * SD + j * 16 + i16;
                    b[j] = to_bf16x4(q[0], q[SD], q[2 * SD], q[3 * SD]);
                }
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        } else
#pragma unroll
        for (int s0 = 0; s0 < CH / 4; s0 += WK) {
            const int s = s0 + wk;
            if (s >= CH / 4) break;
            float a[TI], b[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) a[i] = xs[(4 * s + k4) * SX + (wi * TI + i) * 16 + i16];
#pragma unroll
            for (int j = 0; j < TJ; ++j) b[j] = ds[(4 * s + k4) * SD + j * 16 + i16];
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    if constexpr (WK > 1) {
        // the WK wave groups hold partial sums of the SAME tiles (they split the k-steps): groups 1 .. WK-1 hand theirs to group 0
        // through LDS, one after the other (fixed order)
        static_assert(CH * SD >= WI * TI * TJ * 256, "the dY staging buffer doubles as the wave-group reduction buffer");
        for (int k = 1; k < WK; ++k) {
            __syncthreads();
            if (wk == k) {
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q) ds[((wi * TI + i) * TJ + j) * 256 + q * 64 + lane] = acc[i][j][q];
            }
            __syncthreads();
            if (wk == 0) {
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc[i][j][q] += ds[((wi * TI + i) * TJ + j) * 256 + q * 64 + lane];
            }
        }
    }
    if (wk == 0) {
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    // C layout: row = (lane>>4)*4 + q, col = lane&15
                    const int ci = c0 + (wi * TI + i) * 16 + k4 * 4 + q, n = n0 + j * 16 + i16;
                    if (ci < cin && n < cout) dW[(size_t)ci * cout + n] = acc[i][j][q];
                }
    }
    if (db != nullptr && blockIdx.y == 0) {
        // threads sharing a column merge through LDS (the dY block is dead by now); one atomic per column and workgroup
        __syncthreads();
#pragma unroll
        for (int j = 0; j < (VEC ? 4 : 1); ++j) ds[threadIdx.x * (VEC ? 4 : 1) + j] = bsum[j];
        __syncthreads();
        if ((int)threadIdx.x < wd) {
            constexpr int Q = gd.Q;
            float t = 0.f;
            if constexpr (VEC) {
                for (int m = 0; m < 256 / Q; ++m) t += ds[((threadIdx.x >> 2) + Q * m) * 4 + (threadIdx.x & 3)];
            } else {
                for (int m = 0; m < 256 / Q; ++m) t += ds[threadIdx.x + Q * m];
            }
            db[n0 + threadIdx.x] = t;
        }
    }
}

// rows per slab and number of slabs of a weight-gradient launch (the partials are [slabs][cin][cout])
template <int TI, int TJ, int WK>
static void wgrad_slabs(const Tuning& tn, int64_t R, int cin, int cout, int64_t& rpb, int64_t& nb)
{
    constexpr int CI = TI * (4 / WK) * 16, CJ = TJ * 16;
    constexpr int kWgChunk = wg_chunk(CI + CJ) > 16 * WK ? wg_chunk(CI + CJ) : 16 * WK;  // >= either flavour's chunk
    const int ty = (cin + CI - 1) / CI, tz = (cout + CJ - 1) / CJ;
    // ~2 workgroups per CU; fewer, longer slabs when the dW block is large (every slab is a [cin, cout] partial the reduction reads back)
    const int64_t total = tn.wgrad_wgs;  // 512 (768 / 512 / 384 / 256 measured: one-cloud step 8.03 / 7.95 / 7.96 / 8.17 ms, batch 8: 39.0 / 38.1 / 38.8 / 39.3 -- every slab is a partial the finish reads back)
    int64_t slabs = total / ((int64_t)ty * tz);
    slabs = slabs < 1 ? 1 : slabs;
    rpb = (R + slabs - 1) / slabs;
    rpb = ((rpb + kWgChunk - 1) / kWgChunk) * kWgChunk;
    rpb = rpb < 4 * kWgChunk ? 4 * kWgChunk : rpb;
    nb = (R + rpb - 1) / rpb;
}

template <int TI, int TJ, int WK>
static void launch_wgrad(ps_context* c, const float* x, int ldx, const float* dy, int lddy, int64_t R, int cin, int cout, float* dW, float* db)
{
    constexpr int CI = TI * (4 / WK) * 16, CJ = TJ * 16;
    const int ty = (cin + CI - 1) / CI, tz = (cout + CJ - 1) / CJ;
    int64_t rpb, nb;
    wgrad_slabs<TI, TJ, WK>(c->tune, R, cin, cout, rpb, nb);
    // float4 staging needs every row start and block origin on a 16-byte boundary (CI, CJ are multiples of 16 already)
    const bool vec = ((cin | cout | ldx | lddy) & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15) == 0;
    const dim3 grid((unsigned)nb, ty, tz);
#define PS_WGK(V, B) hipLaunchKernelGGL((wgrad_kernel<TI, TJ, WK, V, B>), grid, dim3(256), 0, c->stream, x, ldx, dy, lddy, R, cin, cout, rpb, dW, db)
    if (c->train_bf16) {
        if (vec) PS_WGK(true, true); else PS_WGK(false, true);
    } else {
        if (vec) PS_WGK(true, false); else PS_WGK(false, false);
    }
#undef PS_WGK
}

// ---- scatter-add of gathered rows (backward of tf.batch_gather) ---------------------------------------------------
__global__ __launch_bounds__(256) void scatter_add_kernel(const float* __restrict__ drows, const int32_t* __restrict__ idx, float* __restrict__ dpc,
                                                          size_t rows, int rows_per_cloud, int n_cloud, int d, int64_t ldd)
{
    const size_t t = blockIdx.x * (size_t)256 + threadIdx.x;
    if (t >= rows * d) return;
    const size_t row = t / d;
    const int ch = (int)(t - row * d);
    const size_t b = row / rows_per_cloud;
    atomicAdd(&dpc[(b * n_cloud + idx[row]) * d + ch], drows[row * ldd + ch]);
}

// ---- softmax over K + weighted sum (att_pooling core, RandLANet.py:396-398) --------------------------------------
// KK > 0: K known at compile time, the K scores / features of a (row, channel) stay in registers (one pass over HBM)
template <int KK>
__global__ __launch_bounds__(256) void softpool_fwd_kernel(const float* __restrict__ fset, const float* __restrict__ scores, int64_t R, int K, int d,
                                                           float* __restrict__ probs, float* __restrict__ agg)
{
    const int64_t t = blockIdx.x * (int64_t)256 + threadIdx.x;  // (row, channel)
    if (t >= R * d) return;
    const int64_t r = t / d;
    const int c = (int)(t - r * d);
    const float* s = scores + r * K * d + c;
    const float* f = fset + r * K * d + c;
    if (KK > 0) {
        float sv[KK > 0 ? KK : 1], fv[KK > 0 ? KK : 1];
#pragma unroll
        for (int k = 0; k < KK; ++k) { sv[k] = s[(size_t)k * d]; fv[k] = f[(size_t)k * d]; }
        float m = sv[0];
#pragma unroll
        for (int k = 1; k < KK; ++k) m = fmaxf(m, sv[k]);
        float den = 0.f;
#pragma unroll
        for (int k = 0; k < KK; ++k) { sv[k] = expf(sv[k] - m); den += sv[k]; }
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < KK; ++k) {
            const float p = sv[k] / den;
            if (probs) probs[r * K * d + (size_t)k * d + c] = p;
            a += fv[k] * p;
        }
        agg[t] = a;
        return;
    }
    float m = s[0];
    for (int k = 1; k < K; ++k) m = fmaxf(m, s[(size_t)k * d]);
    float den = 0.f;
    for (int k = 0; k < K; ++k) den += expf(s[(size_t)k * d] - m);
    float a = 0.f;
    for (int k = 0; k < K; ++k) {
        const float p = expf(s[(size_t)k * d] - m) / den;
        if (probs) probs[r * K * d + (size_t)k * d + c] = p;
        a += f[(size_t)k * d] * p;
    }
    agg[t] = a;
}

// RECOMP: `probs` holds the SCORES (the forward kept no probabilities): the softmax is formed again, with the forward's arithmetic.  dscores
// may then alias the scores (a thread reads the K values of its (row, channel) before it writes them).
template <int KK, bool RECOMP>
__global__ __launch_bounds__(256) void softpool_bwd_kernel(const float* dagg, const float* fset, const float* probs, int64_t R, int K, int d, float* dfset,
                                                           float* dscores)
{
    const int64_t t = blockIdx.x * (int64_t)256 + threadIdx.x;
    if (t >= R * d) return;
    const int64_t r = t / d;
    const int c = (int)(t - r * d);
    const size_t base = (size_t)r * K * d + c;
    const float g = dagg[t];
    float dot = 0.f;  // sum_j p_j * dp_j,  dp_j = g * f_j
    if (KK > 0) {
        float pv[KK > 0 ? KK : 1], fv[KK > 0 ? KK : 1];
#pragma unroll
        for (int k = 0; k < KK; ++k) { pv[k] = probs[base + (size_t)k * d]; fv[k] = fset[base + (size_t)k * d]; }
        if (RECOMP) {
            float m = pv[0];
#pragma unroll
            for (int k = 1; k < KK; ++k) m = fmaxf(m, pv[k]);
            float den = 0.f;
#pragma unroll
            for (int k = 0; k < KK; ++k) { pv[k] = expf(pv[k] - m); den += pv[k]; }
#pragma unroll
            for (int k = 0; k < KK; ++k) pv[k] = pv[k] / den;
        }
#pragma unroll
        for (int k = 0; k < KK; ++k) dot += pv[k] * g * fv[k];
#pragma unroll
        for (int k = 0; k < KK; ++k) {
            dfset[base + (size_t)k * d] = g * pv[k];
            dscores[base + (size_t)k * d] = pv[k] * (g * fv[k] - dot);
        }
        return;
    }
    float m = 0.f, den = 1.f;
    if (RECOMP) {
        m = probs[base];
        for (int k = 1; k < K; ++k) m = fmaxf(m, probs[base + (size_t)k * d]);
        den = 0.f;
        for (int k = 0; k < K; ++k) den += expf(probs[base + (size_t)k * d] - m);
    }
    auto prob = [&](int k) { return RECOMP ? expf(probs[base + (size_t)k * d] - m) / den : probs[base + (size_t)k * d]; };
    for (int k = 0; k < K; ++k) dot += prob(k) * g * fset[base + (size_t)k * d];
    for (int k = 0; k < K; ++k) {
        const float p = prob(k), f = fset[base + (size_t)k * d];
        dfset[base + (size_t)k * d] = g * p;
        dscores[base + (size_t)k * d] = p * (g * f - dot);
    }
}

// ---- random_sample (max over K gathered rows) with tie bookkeeping; backward splits evenly among ties like
//      tf.reduce_max's gradient ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ out, const float* __restrict__ feat,
                                                          const int32_t* __restrict__ idx, float* __restrict__ dfeat, size_t rows, int m_cloud,
                                                          int n_cloud, int K, int d)
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
    const float g = dout[t] / (float)ties;
    for (int k = 0; k < K; ++k)
        if (feat[(base + ix[k]) * d + ch] == mx) atomicAdd(&dfeat[(base + ix[k]) * d + ch], g);
}

__global__ __launch_bounds__(256) void add_lrelu_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n, float* __restrict__ y)
{
    for (int64_t e = blockIdx.x * (int64_t)256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const float s = a[e] + b[e];
        y[e] = s >= 0.f ? s : 0.2f * s;
    }
}
__global__ __launch_bounds__(256) void add_lrelu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, int64_t n, float* __restrict__ ds)
{
    for (int64_t e = blockIdx.x * (int64_t)256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) ds[e] = y[e] >= 0.f ? dy[e] : 0.2f * dy[e];
}
__global__ __launch_bounds__(256) void axpy_kernel(float alpha, const float* __restrict__ x, int64_t n, float* __restrict__ y)
{
    for (int64_t e = blockIdx.x * (int64_t)256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) y[e] += alpha * x[e];
}

// ---- class-weighted softmax cross-entropy, mean over the VALID rows (RandLANet.py:62-84, 267-274) + its gradient -------
// A label outside [0, C) marks an ignored point (the reference drops the rows whose label is in cfg.ignored_label_inds before the
// loss and averages over the remaining ones; the host maps its ignored labels to -1 and the others to 0..C-1 like the reference's
// `reducing_list`): weight 0, zero gradient row, not counted in the mean.  Three small launches: count, per-workgroup partial
// sums, ordered final sum -- no float atomics, so the loss is bit-identical from run to run.
__global__ __launch_bounds__(256) void wce_count_kernel(const int32_t* __restrict__ labels, int64_t R, int C, int32_t* __restrict__ n_valid)
{
    int cnt = 0;
    for (int64_t r = blockIdx.x * (int64_t)256 + threadIdx.x; r < R; r += (int64_t)gridDim.x * 256) {
        const int y = labels[r];
        cnt += (y >= 0 && y < C) ? 1 : 0;
    }
    cnt = wave_sum(cnt);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(n_valid, cnt);
}

__global__ __launch_bounds__(256) void wce_kernel(const float* __restrict__ logits, const int32_t* __restrict__ labels, const float* __restrict__ cw,
                                                  int64_t R, int C, const int32_t* __restrict__ n_valid, float* __restrict__ partial,
                                                  float* __restrict__ dlogits)
{
    __shared__ float s[256];
    float acc = 0.f;
    const float inv = 1.f / (float)n_valid[0];
    for (int64_t r = blockIdx.x * (int64_t)256 + threadIdx.x; r < R; r += (int64_t)gridDim.x * 256) {
        const int y = labels[r];
        const bool valid = y >= 0 && y < C;
        if (!valid) {
            if (dlogits)
                for (int c = 0; c < C; ++c) dlogits[r * C + c] = 0.f;
            continue;
        }
        const float* z = logits + r * C;
        float m = z[0];
        for (int c = 1; c < C; ++c) m = fmaxf(m, z[c]);
        float den = 0.f;
        for (int c = 0; c < C; ++c) den += expf(z[c] - m);
        const float w = cw[y];
        acc += w * (logf(den) - (z[y] - m));
        if (dlogits)
            for (int c = 0; c < C; ++c) dlogits[r * C + c] = w * (expf(z[c] - m) / den - (c == y ? 1.f : 0.f)) * inv;
    }
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = s[0];
}

__global__ __launch_bounds__(256) void wce_finish_kernel(const float* __restrict__ partial, int n, const int32_t* __restrict__ n_valid, float* __restrict__ loss)
{
    __shared__ float s[256];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) acc += partial[i];
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = s[0] / (float)n_valid[0];  // no valid row: 0/0 = NaN, tf.reduce_mean of an empty tensor
}

// ---- Adam (tf.train.AdamOptimizer: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); p -= lr_t * m / (sqrt(v) + eps)) -------------
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                   int64_t n, float lr_t, float b1, float b2, float eps)
{
    for (int64_t e = blockIdx.x * (int64_t)256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const float gi = g[e];
        const float mi = b1 * m[e] + (1.f - b1) * gi;
        const float vi = b2 * v[e] + (1.f - b2) * gi * gi;
        m[e] = mi;
        v[e] = vi;
        p[e] -= lr_t * mi / (sqrtf(vi) + eps);
    }
}

// dropout with a counter-based hash (training only; tf.nn.dropout scales kept units by 1/keep_prob)
__device__ __forceinline__ unsigned hash32(unsigned x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, int64_t n, unsigned seed, float keep, float* __restrict__ y,
                                                      float* __restrict__ mask)
{
    for (int64_t e = blockIdx.x * (int64_t)256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const float u = (hash32((unsigned)e * 2654435761u ^ seed) >> 8) * (1.0f / 16777216.0f);
        const float mk = u < keep ? 1.0f / keep : 0.f;
        mask[e] = mk;
        y[e] = x[e] * mk;
    }
}
__global__ __launch_bounds__(256) void mul_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n, float* __restrict__ y)
{
    for (int64_t e = blockIdx.x * (int64_t)256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) y[e] = a[e] * b[e];
}

static inline unsigned ew_grid(int64_t n)
{
    int64_t b = (n + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace ps

using namespace ps;

extern "C" {

int ps_op_linear_wgrad(ps_context* c, const float* x, const float* dy, int64_t R, int64_t cin, int64_t cout, float* dW, float* db)
{
    return ps_op_linear_wgrad_ex(c, x, cin, dy, cout, R, cin, cout, dW, db);
}

}  // extern "C"

namespace ps {
// per-wave tile block TI x TJ and the number of wave groups splitting the k-steps (WK) for a [cin, cout] gradient; see wgrad_kernel
#define PS_WGRAD_DISPATCH(ti, tj, DO)     \
    do {                                   \
        if ((tj) >= 8) {                   \
            if ((ti) >= 8) DO(2, 8, 1);    \
            else if ((ti) >= 3) DO(1, 8, 1); \
            else if ((ti) == 2) DO(1, 8, 2); \
            else DO(1, 8, 4);              \
        } else if ((tj) >= 3) {            \
            if ((ti) >= 3) DO(1, 4, 1);    \
            else if ((ti) == 2) DO(1, 4, 2); \
            else DO(1, 4, 4);              \
        } else if ((tj) == 2) {            \
            if ((ti) >= 3) DO(1, 2, 1);    \
            else if ((ti) == 2) DO(1, 2, 2); \
            else DO(1, 2, 4);              \
        } else {                           \
            if ((ti) >= 3) DO(1, 1, 1);    \
            else if ((ti) == 2) DO(1, 1, 2); \
            else DO(1, 1, 4);              \
        }                                  \
    } while (0)

static bool wgrad_on_b3(ps_context* c, const float* x, int64_t ldx, const float* dy, int64_t lddy, int64_t R, int64_t cin, int64_t cout)
{
    return c->train_b3 && wgrad_b3_fits(c->tune, R, cin, cout, x, ldx, dy, lddy, c->train_bf16);  // (bf16-MLP mode: one plane of rounded operands)
}

int64_t wgrad_partial_slabs(ps_context* c, const float* x, int64_t ldx, const float* dy, int64_t lddy, int64_t R, int64_t cin, int64_t cout)
{
    if (R <= 0) return 0;
    if (wgrad_on_b3(c, x, ldx, dy, lddy, R, cin, cout)) return wgrad_b3_slabs(c->tune, R, cin, cout);
    const int ti = (int)((cin + 15) / 16), tj = (int)((cout + 15) / 16);
    int64_t rpb = 0, nb = 0;
#define PS_WG(TI, TJ, WK) wgrad_slabs<TI, TJ, WK>(c->tune, R, (int)cin, (int)cout, rpb, nb)
    PS_WGRAD_DISPATCH(ti, tj, PS_WG);
#undef PS_WG
    return nb;
}

// part [slabs][cin][cout], dbpart [slabs][cout] (may be null): written completely, nothing needs zeroing
int wgrad_partial(ps_context* c, const float* x, int64_t ldx, const float* dy, int64_t lddy, int64_t R, int64_t cin, int64_t cout, float* part,
                  float* dbpart)
{
    if (R <= 0) return PS_OK;
    if (wgrad_on_b3(c, x, ldx, dy, lddy, R, cin, cout)) return wgrad_b3_partial(c, x, ldx, dy, lddy, R, cin, cout, part, dbpart);
    const int ti = (int)((cin + 15) / 16), tj = (int)((cout + 15) / 16);
    const int ci = (int)cin, co = (int)cout;
#define PS_WG(TI, TJ, WK) launch_wgrad<TI, TJ, WK>(c, x, (int)ldx, dy, (int)lddy, R, ci, co, part, dbpart)
    PS_WGRAD_DISPATCH(ti, tj, PS_WG);
#undef PS_WG
    PS_HIP(hipGetLastError());
    return PS_OK;
}

// the same for X = [xl[xidx] | xr] (the neighbour set of att_pooling without its concat buffer): split-bf16 kernel only -- the caller
// checks wgrad_split_slabs() > 0 and takes the materialised form otherwise
int64_t wgrad_split_slabs(ps_context* c, const float* xl, int64_t ldxl, const int32_t* xidx, const float* xr, int64_t ldxr, const float* dy, int64_t lddy,
                          int64_t R, int64_t cin, int64_t cout)
{
    if (!c->train_b3 || !wgrad_b3_split_fits(c->tune, R, cin, cout, xl, ldxl, xidx, xr, ldxr, dy, lddy)) return 0;
    return wgrad_b3_slabs(c->tune, R, cin, cout);
}

// dst = sum over the slabs, in slab order (fixed order: deterministic).  blockIdx.y walks a table of jobs, so every weight and bias
// gradient of a training step is finished by ONE launch; transposed: dst is [cols][rows] (the conv2d_transpose kernels, stored [out, in])
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const WgradJob* __restrict__ jobs)
{
    __shared__ float red[4][64];
    const WgradJob j = jobs[blockIdx.y];
    const int64_t n = (int64_t)j.rows * j.cols;
    if (j.slabs <= 16 && (n & 3) == 0 && ((reinterpret_cast<uintptr_t>(j.part) | reinterpret_cast<uintptr_t>(j.dst)) & 15) == 0) {
        // few slabs (the large matrices: most of the bytes): a thread owns four consecutive elements and walks the slabs itself, slab b into
        // accumulator b % 4 -- 16-byte loads, four independent chains, no LDS round; one fixed summation order
        const float4* part = reinterpret_cast<const float4*>(j.part);
        const int64_t n4 = n >> 2;
        for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n4; e += (int64_t)gridDim.x * 256) {
            float4 a[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            int b = 0;
            for (; b + 4 <= j.slabs; b += 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float4 v = part[(size_t)(b + u) * n4 + e];
                    a[u].x += v.x; a[u].y += v.y; a[u].z += v.z; a[u].w += v.w;
                }
            }
            for (int u = 0; b + u < j.slabs; ++u) {
                const float4 v = part[(size_t)(b + u) * n4 + e];
                a[u].x += v.x; a[u].y += v.y; a[u].z += v.z; a[u].w += v.w;
            }
            const float4 sum = {(a[0].x + a[1].x) + (a[2].x + a[3].x), (a[0].y + a[1].y) + (a[2].y + a[3].y), (a[0].z + a[1].z) + (a[2].z + a[3].z),
                                (a[0].w + a[1].w) + (a[2].w + a[3].w)};
            if (j.transposed) {
                const int64_t e1 = e << 2;
                const int64_t r = e1 / j.cols, cc = e1 - r * j.cols;  // (cols % 4 == 0 is NOT implied by n % 4 == 0: per element)
                const float sv[4] = {sum.x, sum.y, sum.z, sum.w};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int64_t c1 = cc + u, rr = r + c1 / j.cols, c2 = c1 % j.cols;
                    j.dst[c2 * j.rows + rr] = sv[u];
                }
            } else {
                reinterpret_cast<float4*>(j.dst)[e] = sum;
            }
        }
        return;
    }
    // many slabs (small matrices under millions of rows: 768 slabs for the 8 x 8 layers of level 0): 16 elements per workgroup pass, the slabs
    // dealt over 16 thread groups x 4 accumulators (slab b goes to group b % 16, accumulator (b / 16) % 4): 64 independent load chains per
    // element -- 12 dependent round trips for 768 slabs, where 4 groups took 48 (the whole launch waited for these few workgroups: 112 us
    // of a one-cloud step) -- and still one fixed summation order
    const int el = threadIdx.x & 15, grp = threadIdx.x >> 4;
    float (*red16)[16] = reinterpret_cast<float (*)[16]>(&red[0][0]);  // [16 groups][16 elements]
    for (int64_t e0 = (int64_t)blockIdx.x * 16; e0 < n; e0 += (int64_t)gridDim.x * 16) {
        const int64_t e = e0 + el;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        if (e < n) {
            int b = grp;
            for (; b + 48 < j.slabs; b += 64) {
                a0 += j.part[(size_t)b * n + e];
                a1 += j.part[(size_t)(b + 16) * n + e];
                a2 += j.part[(size_t)(b + 32) * n + e];
                a3 += j.part[(size_t)(b + 48) * n + e];
            }
            if (b < j.slabs) a0 += j.part[(size_t)b * n + e];
            if (b + 16 < j.slabs) a1 += j.part[(size_t)(b + 16) * n + e];
            if (b + 32 < j.slabs) a2 += j.part[(size_t)(b + 32) * n + e];
        }
        __syncthreads();
        red16[grp][el] = (a0 + a1) + (a2 + a3);
        __syncthreads();
        if (grp == 0 && e < n) {
            float sum = 0.f;
#pragma unroll
            for (int g4 = 0; g4 < 16; g4 += 4) sum += (red16[g4][el] + red16[g4 + 1][el]) + (red16[g4 + 2][el] + red16[g4 + 3][el]);
            if (j.transposed) {
                const int64_t r = e / j.cols, cc = e - r * j.cols;
                j.dst[cc * j.rows + r] = sum;
            } else {
                j.dst[e] = sum;
            }
        }
    }
}

int wgrad_finish(ps_context* c, const WgradJob* d_jobs, int n_jobs, int64_t max_elems)
{
    if (n_jobs <= 0) return PS_OK;
    const unsigned gx = (unsigned)std::max<int64_t>(1, std::min<int64_t>((max_elems + 15) / 16, 256));
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(gx, (unsigned)n_jobs), dim3(256), 0, c->stream, d_jobs);
    PS_HIP(hipGetLastError());
    return PS_OK;
}
}  // namespace ps

extern "C" {

int ps_op_linear_wgrad_ex(ps_context* c, const float* x, int64_t ldx, const float* dy, int64_t lddy, int64_t R, int64_t cin, int64_t cout, float* dW,
                          float* db)
{
    PS_CHECK(c && x && dy && dW, "ps_op_linear_wgrad: NULL argument");
    PS_CHECK(ldx >= cin && lddy >= cout, "ps_op_linear_wgrad: row stride below the channel count");
    PS_HIP(hipSetDevice(c->device));
    Stage st(c, "train_wgrad", 2);
    if (R <= 0) {
        PS_HIP(hipMemsetAsync(dW, 0, sizeof(float) * cin * cout, c->stream));
        if (db) PS_HIP(hipMemsetAsync(db, 0, sizeof(float) * cout, c->stream));
        return PS_OK;
    }
    // per-slab partials (plain stores) + one reduction in slab order: deterministic, no float atomics, nothing zeroed first
    const int64_t nb = wgrad_partial_slabs(c, x, ldx, dy, lddy, R, cin, cout);
    const size_t wfl = (size_t)nb * cin * cout, bfl = db ? (size_t)nb * cout : 0;
    PS_TRY(c->wgrad_ws.reserve(sizeof(float) * (wfl + bfl) + 2 * sizeof(WgradJob) + 256));
    float* part = c->wgrad_ws.as<float>();
    float* dbpart = db ? part + wfl : nullptr;
    PS_TRY(wgrad_partial(c, x, ldx, dy, lddy, R, cin, cout, part, dbpart));
    WgradJob jobs[2];
    jobs[0] = WgradJob{part, dW, (int)nb, (int)cin, (int)cout, 0};
    jobs[1] = WgradJob{dbpart, db, (int)nb, 1, (int)cout, 0};
    WgradJob* dj = reinterpret_cast<WgradJob*>(c->wgrad_ws.as<char>() + ((sizeof(float) * (wfl + bfl) + 255) & ~size_t(255)));
    PS_TRY(c->upload_async(dj, jobs, sizeof(WgradJob) * (db ? 2 : 1)));
    return wgrad_finish(c, dj, db ? 2 : 1, cin * cout);
}

int ps_op_bn_train_fwd(ps_context* c, const float* x, const float* gamma, const float* beta, int64_t R, int64_t C, float eps, int leaky, float* y,
                       float* mean, float* invstd, float* var, float* scratch2C)
{
    return ps_op_bn_train_fwd_ex(c, x, gamma, beta, R, C, eps, leaky, y, C, mean, invstd, var, scratch2C);
}

int ps_op_bn_train_fwd_ex(ps_context* c, const float* x, const float* gamma, const float* beta, int64_t R, int64_t C, float eps, int leaky, float* y,
                          int64_t ldy, float* mean, float* invstd, float* var, float* scratch2C)
{
    PS_CHECK(c && x && gamma && beta && y && mean && invstd && var && scratch2C, "ps_op_bn_train_fwd: NULL argument");
    PS_CHECK(R >= 1 && C >= 1 && ldy >= C, "ps_op_bn_train_fwd: empty tensor");
    PS_HIP(hipSetDevice(c->device));
    if (bn_slice_ok(c->tune, R, C, x, y, ldy)) {
        Stage st1(c, "train_bn_fwd", 1);
        hipLaunchKernelGGL(bn_slice_fwd_kernel, dim3((unsigned)(C / kBnSliceCh)), dim3(kBnSliceThreads), 0, c->stream, x, gamma, beta, (int)R, (int)C, eps, leaky, y, ldy, mean, invstd,
                           var, scratch2C, static_cast<float*>(nullptr), static_cast<float*>(nullptr), 0.f);
        PS_HIP(hipGetLastError());
        return PS_OK;
    }
    Stage st(c, "train_bn_fwd", 3);
    PS_TRY(colreduce2(c, SumSq{x}, R, (int)C, scratch2C, scratch2C + C));
    hipLaunchKernelGGL(bn_finish_stats_kernel, dim3(ceil_div(C, 64)), dim3(64), 0, c->stream, scratch2C, scratch2C + C, R, (int)C, eps, mean, invstd, var);
    if (bn_vec_ok(C, x, y, x, ldy))
        hipLaunchKernelGGL(bn_apply_kernel<true>, dim3(ew_grid(R * C / 4)), dim3(256), 0, c->stream, x, gamma, beta, mean, invstd, R * C, (int)C, leaky, y,
                           ldy);
    else
        hipLaunchKernelGGL(bn_apply_kernel<false>, dim3(ew_grid(R * C)), dim3(256), 0, c->stream, x, gamma, beta, mean, invstd, R * C, (int)C, leaky, y,
                           ldy);
    PS_HIP(hipGetLastError());
    return PS_OK;
}

int ps_op_bn_train_fwd_mov(ps_context* c, const float* x, const float* gamma, const float* beta, int64_t R, int64_t C, float eps, int leaky, float* y,
                           int64_t ldy, float* mean, float* invstd, float* var, float* scratch2C, float* moving_mean, float* moving_var, float momentum)
{
    PS_CHECK(c && x && gamma && beta && y && mean && invstd && var && scratch2C && moving_mean && moving_var, "ps_op_bn_train_fwd_mov: NULL argument");
    PS_CHECK(R >= 1 && C >= 1 && ldy >= C, "ps_op_bn_train_fwd_mov: empty tensor");
    PS_HIP(hipSetDevice(c->device));
    if (bn_slice_ok(c->tune, R, C, x, y, ldy)) {
        Stage st1(c, "train_bn_fwd", 1);
        hipLaunchKernelGGL(bn_slice_fwd_kernel, dim3((unsigned)(C / kBnSliceCh)), dim3(kBnSliceThreads), 0, c->stream, x, gamma, beta, (int)R, (int)C, eps, leaky, y, ldy, mean, invstd,
                           var, scratch2C, moving_mean, moving_var, momentum);
        PS_HIP(hipGetLastError());
        return PS_OK;
    }
    Stage st(c, "train_bn_fwd", 3);
    const BnFinish bn = {mean, invstd, var, moving_mean, moving_var, (float)R, eps, momentum};
    PS_TRY(colreduce2(c, SumSq{x}, R, (int)C, scratch2C, scratch2C + C, &bn));
    if (bn_vec_ok(C, x, y, x, ldy))
        hipLaunchKernelGGL(bn_apply_kernel<true>, dim3(ew_grid(R * C / 4)), dim3(256), 0, c->stream, x, gamma, beta, mean, invstd, R * C, (int)C, leaky, y,
                           ldy);
    else
        hipLaunchKernelGGL(bn_apply_kernel<false>, dim3(ew_grid(R * C)), dim3(256), 0, c->stream, x, gamma, beta, mean, invstd, R * C, (int)C, leaky, y,
                           ldy);
    PS_HIP(hipGetLastError());
    return PS_OK;
}

int ps_op_bn_train_bwd(ps_context* c, const float* dy, const float* x, const float* gamma, const float* beta, const float* mean, const float* invstd,
                       int64_t R, int64_t C, int leaky, float* dx, float* dgamma, float* dbeta)
{
    return ps_op_bn_train_bwd_ex(c, dy, C, x, gamma, beta, mean, invstd, R, C, leaky, dx, dgamma, dbeta);
}

int ps_op_bn_train_bwd_ex(ps_context* c, const float* dy, int64_t lddy, const float* x, const float* gamma, const float* beta, const float* mean,
                          const float* invstd, int64_t R, int64_t C, int leaky, float* dx, float* dgamma, float* dbeta)
{
    PS_CHECK(c && dy && x && gamma && beta && mean && invstd && dx && dgamma && dbeta, "ps_op_bn_train_bwd: NULL argument");
    PS_CHECK(lddy >= C, "ps_op_bn_train_bwd: row stride of dy below the channel count");
    PS_HIP(hipSetDevice(c->device));
    if (R >= 1 && bn_slice_ok(c->tune, R, C, x, dy, lddy) && (reinterpret_cast<uintptr_t>(dx) & 15) == 0) {
        Stage st1(c, "train_bn_bwd", 1);
        hipLaunchKernelGGL(bn_slice_bwd_kernel, dim3((unsigned)(C / kBnSliceCh)), dim3(kBnSliceThreads), 0, c->stream, dy, lddy, x, gamma, beta, mean, invstd, (int)R, (int)C, leaky, dx,
                           dgamma, dbeta);
        PS_HIP(hipGetLastError());
        return PS_OK;
    }
    Stage st(c, "train_bn_bwd", 2);
    // dbeta = sum g, dgamma = sum g*xhat
    PS_TRY(colreduce2(c, BnBwdSums{dy, x, gamma, beta, mean, invstd, leaky, (int)C, lddy}, R, (int)C, dbeta, dgamma));
    if (bn_vec_ok(C, x, dy, dx, lddy))
        hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, dim3(ew_grid(R * C / 4)), dim3(256), 0, c->stream, dy, x, gamma, beta, mean, invstd, dbeta, dgamma,
                           R * C, (int)C, 1.0f / (float)R, leaky, dx, lddy);
    else
        hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, dim3(ew_grid(R * C)), dim3(256), 0, c->stream, dy, x, gamma, beta, mean, invstd, dbeta, dgamma,
                           R * C, (int)C, 1.0f / (float)R, leaky, dx, lddy);
    PS_HIP(hipGetLastError());
    return PS_OK;
}

// ---- the same two ops split at their reduction, for BatchNorm statistics shared by several GPUs (config 4): the caller
// all-reduces the 2*C sums between the halves and passes the global row count.
int ps_op_bn_train_sums(ps_context* c, const float* x, int64_t R, int64_t C, float* sums2C)
{
    PS_CHECK(c && x && sums2C, "ps_op_bn_train_sums: NULL argument");
    PS_CHECK(R >= 1 && C >= 1, "ps_op_bn_train_sums: empty tensor");
    PS_HIP(hipSetDevice(c->device));
    Stage st(c, "train_bn_fwd", 1);
    PS_TRY(colreduce2(c, SumSq{x}, R, (int)C, sums2C, sums2C + C));
    PS_HIP(hipGetLastError());
    return PS_OK;
}

int ps_op_bn_train_apply(ps_context* c, const float* x, const float* gamma, const float* beta, const float* sums2C, int64_t R, int64_t R_total,
                         int64_t C, float eps, int leaky, float* y, float* mean, float* invstd, float* var)
{
    return ps_op_bn_train_apply_ex(c, x, gamma, beta, sums2C, R, R_total, C, eps, leaky, y, C, mean, invstd, var);
}

int ps_op_bn_train_apply_ex(ps_context* c, const float* x, const float* gamma, const float* beta, const float* sums2C, int64_t R, int64_t R_total,
                            int64_t C, float eps, int leaky, float* y, int64_t ldy, float* mean, float* invstd, float* var)
{
    PS_CHECK(c && x && gamma && beta && sums2C && y && mean && invstd && var, "ps_op_bn_train_apply: NULL argument");
    PS_CHECK(R >= 1 && C >= 1 && R_total >= R && ldy >= C, "ps_op_bn_train_apply: bad row counts");
    PS_HIP(hipSetDevice(c->device));
    Stage st(c, "train_bn_fwd", 2);
    hipLaunchKernelGGL(bn_finish_stats_kernel, dim3(ceil_div(C, 64)), dim3(64), 0, c->stream, sums2C, sums2C + C, R_total, (int)C, eps, mean, invstd, var);
    if (bn_vec_ok(C, x, y, x, ldy))
        hipLaunchKernelGGL(bn_apply_kernel<true>, dim3(ew_grid(R * C / 4)), dim3(256), 0, c->stream, x, gamma, beta, mean, invstd, R * C, (int)C, leaky, y,
                           ldy);
    else
        hipLaunchKernelGGL(bn_apply_kernel<false>, dim3(ew_grid(R * C)), dim3(256), 0, c->stream, x, gamma, beta, mean, invstd, R * C, (int)C, leaky, y,
                           ldy);
    PS_HIP(hipGetLastError());
    return PS_OK;
}

int ps_op_bn_train_bwd_sums(ps_context* c, const float* dy, const float* x, const float* gamma, const float* beta, const float* mean,
                            const float* invstd, int64_t R, int64_t C, int leaky, float* dgamma, float* dbeta)
{
    return ps_op_bn_train_bwd_sums_ex(c, dy, C, x, gamma, beta, mean, invstd, R, C, leaky, dgamma, dbeta);
}

int ps_op_bn_train_bwd_sums_ex(ps_context* c, const float* dy, int64_t lddy, const float* x, const float* gamma, const float* beta, const float* mean,
                               const float* invstd, int64_t R, int64_t C, int leaky, float* dgamma, float* dbeta)
{
    PS_CHECK(c && dy && x && gamma && beta && mean && invstd && dgamma && dbeta, "ps_op_bn_train_bwd_sums: NULL argument");
    PS_CHECK(lddy >= C, "ps_op_bn_train_bwd_sums: row stride of dy below the channel count");
    PS_HIP(hipSetDevice(c->device));
    Stage st(c, "train_bn_bwd", 1);
    PS_TRY(colreduce2(c, BnBwdSums{dy, x, gamma, beta, mean, invstd, leaky, (int)C, lddy}, R, (int)C, dbeta, dgamma));
    PS_HIP(hipGetLastError());
    return PS_OK;
}

int ps_op_bn_train_bwd_apply(ps_context* c, const float* dy, const float* x, const float* gamma, const float* beta, const float* mean,
                             const float* invstd, const float* sum_g, const float* sum_gx, int64_t R, int64_t R_total, int64_t C, int leaky, float* dx)
{
    return ps_op_bn_train_bwd_apply_ex(c, dy, C, x, gamma, beta, mean, invstd, sum_g, sum_gx, R, R_total, C, leaky, dx);
}

int ps_op_bn_train_bwd_apply_ex(ps_context* c, const float* dy, int64_t lddy, const float* x, const float* gamma, const float* beta, const float* mean,
                                const float* invstd, const float* sum_g, const float* sum_gx, int64_t R, int64_t R_total, int64_t C, int leaky,
                                float* dx)
{
    PS_CHECK(c && dy && x && gamma && beta && mean && invstd && sum_g && sum_gx && dx, "ps_op_bn_train_bwd_apply: NULL argument");
    PS_CHECK(R >= 1 && C >= 1 && R_total >= R && lddy >= C, "ps_op_bn_train_bwd_apply: bad row counts");
    PS_HIP(hipSetDevice(c->device));
    Stage st(c, "train_bn_bwd", 1);
    if (bn_vec_ok(C, x, dy, dx, lddy))
        hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, dim3(ew_grid(R * C / 4)), dim3(256), 0, c->stream, dy, x, gamma, beta, mean, invstd, sum_g, sum_gx,
                           R * C, (int)C, 1.0f / (float)R_total, leaky, dx, lddy);
    else
        hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, dim3(ew_grid(R * C)), dim3(256), 0, c->stream, dy, x, gamma, beta, mean, invstd, sum_g, sum_gx,
                           R * C, (int)C, 1.0f / (float)R_total, leaky, dx, lddy);
    PS_HIP(hipGetLastError());
    return PS_OK;
}

int ps_op_scatter_add_rows(ps_context* c, const float* drows, const int32_t* idx, int64_t B, int64_t N, int64_t rows_per_cloud, int64_t d, float* dpc)
{
    return ps_op_scatter_add_rows_ex(c, drows, d, idx, B, N, rows_per_cloud, d, dpc);
}

int ps_op_scatter_add_rows_ex(ps_context* c, const float* drows, int64_t ldd, const int32_t* idx, int64_t B, int64_t N, int64_t rows_per_cloud, int64_t d,
                              float* dpc)
{
    PS_CHECK(c && drows && idx && dpc, "ps_op_scatter_add_rows: NULL argument");
    PS_CHECK(ldd >= d, "ps_op_scatter_add_rows: row stride below the channel count");
    const size_t rows = (size_t)B * rows_per_cloud;
    if (!rows) return PS_OK;
    PS_HIP(hipSetDevice(c->device));
    Stage st(c, "train_scatter_add", 1);
    hipLaunchKernelGGL(scatter_add_kernel, dim3(ceil_div(rows * d, 256)), dim3(256), 0, c->stream, drows, idx, dpc, rows, (int)rows_per_cloud, (int)N, (int)d,
                       ldd);
    PS_HIP(hipGetLastError());
    return PS_OK;
}

int ps_op_softmax_pool_fwd(ps_context* c, const float* fset, const float* scores, int64_t R, int64_t K, int64_t d, float* probs, float* agg)
{
    PS_CHECK(c && fset && scores && agg, "ps_op_softmax_pool_fwd: NULL argument");  // (probs may be NULL: not kept)
    if (!R) return PS_OK;
    PS_HIP(hipSetDevice(c->device));
    Stage st(c, "train_softpool_fwd", 1);
    const dim3 grid(ceil_div(R * d, 256));
    if (K == 16) hipLaunchKernelGGL(softpool_fwd_kernel<16>, grid, dim3(256), 0, c->stream, fset, scores, R, (int)K, (int)d, probs, agg);
    else if (K == 32) hipLaunchKernelGGL(softpool_fwd_kernel<32>, grid, dim3(256), 0, c->stream, fset, scores, R, (int)K, (int)d, probs, agg);
    else hipLaunchKernelGGL(softpool_fwd_kernel<0>, grid, dim3(256), 0, c->stream, fset, scores, R, (int)K, (int)d, probs, agg);
    PS_HIP(hipGetLastError());
    return PS_OK;
}

int ps_op_softmax_pool_bwd(ps_context* c, const float* dagg, const float* fset, const float* probs, int64_t R, int64_t K, int64_t d, float* dfset,
                           float* dscores)
{
    PS_CHECK(c && dagg && fset && probs && dfset && dscores, "ps_op_softmax_pool_bwd: NULL argument");
    if (!R) return PS_OK;
    PS_HIP(hipSetDevice(c->device));
    Stage st(c, "train_softpool_bwd", 1);
    const dim3 grid(ceil_div(R * d, 256));
    if (K == 16) hipLaunchKernelGGL((softpool_bwd_kernel<16, false>), grid, dim3(256), 0, c->stream, dagg, fset, probs, R, (int)K, (int)d, dfset, dscores);
    else if (K == 32) hipLaunchKernelGGL((softpool_bwd_kernel<32, false>), grid, dim3(256), 0, c->stream, dagg, fset, probs, R, (int)K, (int)d, dfset, dscores);
    else hipLaunchKernelGGL((softpool_bwd_kernel<0, false>), grid, dim3(256), 0, c->stream, dagg, fset, probs, R, (int)K, (int)d, dfset, dscores);
    PS_HIP(hipGetLastError());
    return PS_OK;
}

int ps_op_softmax_pool_bwd_scores(ps_context* c, const float* dagg, const float* fset, const float* scores, int64_t R, int64_t K, int64_t d, float* dfset,
                                  float* dscores)
{
    PS_CHECK(c && dagg && fset && scores && dfset && dscores, "ps_op_softmax_pool_bwd_scores: NULL argument");
    if (!R) return PS_OK;
    PS_HIP(hipSetDevice(c->device));
    Stage st(c, "train_softpool_bwd", 1);
    const dim3 grid(ceil_div(R * d, 256));
    if (K == 16) hipLaunchKernelGGL((softpool_bwd_kernel<16, true>), grid, dim3(256), 0, c->stream, dagg, fset, scores, R, (int)K, (int)d, dfset, dscores);
    else if (K == 32) hipLaunchKernelGGL((softpool_bwd_kernel<32, true>), grid, dim3(256), 0, c->stream, dagg, fset, scores, R, (int)K, (int)d, dfset, dscores);
    else hipLaunchKernelGGL((softpool_bwd_kernel<0, true>), grid, dim3(256), 0, c->stream, dagg, fset, scores, R, (int)K, (int)d, dfset, dscores);
    PS_HIP(hipGetLastError());
    return PS_OK;
}

int ps_op_random_sample_bwd(ps_context* c, const float* dout, const float* out, const float* feature, const int32_t* pool_idx, int64_t B, int64_t N,
                            int64_t M, int64_t K, int64_t d, float* dfeature)
{
    PS_CHECK(c && dout && out && feature && pool_idx && dfeature, "ps_op_random_sample_bwd: NULL argument");
    const size_t rows = (size_t)B * M;
    if (!rows) return PS_OK;
    PS_HIP(hipSetDevice(c->device));
    Stage st(c, "train_maxpool_bwd", 1);
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(ceil_div(rows * d, 256)), dim3(256), 0, c->stream, dout, out, feature, pool_idx, dfeature, rows, (int)M, (int)N,
                       (int)K, (int)d);
    PS_HIP(hipGetLastError());
    return PS_OK;
}

int ps_op_add_lrelu(ps_context* c, const float* a, const float* b, int64_t n, float* y)
{
    PS_CHECK(c && a && b && y, "ps_op_add_lrelu: NULL argument");
    PS_HIP(hipSetDevice(c->device));
    hipLaunchKernelGGL(add_lrelu_kernel, dim3(ew_grid(n)), dim3(256), 0, c->stream, a, b, n, y);
    PS_HIP(hipGetLastError());
    return PS_OK;
}

int ps_op_add_lrelu_bwd(ps_context* c, const float* dy, const float* y, int64_t n, float* ds)
{
    PS_CHECK(c && dy && y && ds, "ps_op_add_lrelu_bwd: NULL argument");
    PS_HIP(hipSetDevice(c->device));
    hipLaunchKernelGGL(add_lrelu_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, c->stream, dy, y, n, ds);
    PS_HIP(hipGetLastError());
    return PS_OK;
}

int ps_op_axpy(ps_context* c, float alpha, const float* x, int64_t n, float* y)
{
    PS_CHECK(c && x && y, "ps_op_axpy: NULL argument");
    PS_HIP(hipSetDevice(c->device));
    hipLaunchKernelGGL(axpy_kernel, dim3(ew_grid(n)), dim3(256), 0, c->stream, alpha, x, n, y);
    PS_HIP(hipGetLastError());
    return PS_OK;
}

int ps_op_weighted_ce(ps_context* c, const float* logits, const int32_t* labels, const float* class_weights, int64_t R, int64_t C, float* loss,
                      float* dlogits)
{
    PS_CHECK(c && logits && labels && class_weights && loss, "ps_op_weighted_ce: NULL argument");
    PS_CHECK(R >= 1 && C >= 1 && C <= 64, "ps_op_weighted_ce: bad shape");
    PS_HIP(hipSetDevice(c->device));
    Stage st(c, "train_loss", 3);
    const unsigned blocks = (unsigned)(ew_grid(R) > 1024 ? 1024 : ew_grid(R));
    PS_TRY(c->red_ws.reserve(sizeof(float) * (blocks + 64)));
    int32_t* n_valid = c->red_ws.as<int32_t>();
    float* partial = c->red_ws.as<float>() + 64;
    PS_HIP(hipMemsetAsync(n_valid, 0, sizeof(int32_t), c->stream));
    hipLaunchKernelGGL(wce_count_kernel, dim3(blocks), dim3(256), 0, c->stream, labels, R, (int)C, n_valid);
    hipLaunchKernelGGL(wce_kernel, dim3(blocks), dim3(256), 0, c->stream, logits, labels, class_weights, R, (int)C, n_valid, partial, dlogits);
    hipLaunchKernelGGL(wce_finish_kernel, dim3(1), dim3(256), 0, c->stream, partial, (int)blocks, n_valid, loss);
    PS_HIP(hipGetLastError());
    return PS_OK;
}

int ps_op_adam(ps_context* c, float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, int64_t step)
{
    PS_CHECK(c && p && g && m && v && step >= 1, "ps_op_adam: bad argument");
    PS_HIP(hipSetDevice(c->device));
    Stage st(c, "train_adam", 1);
    const double lr_t = (double)lr * std::sqrt(1.0 - std::pow((double)beta2, (double)step)) / (1.0 - std::pow((double)beta1, (double)step));
    hipLaunchKernelGGL(adam_kernel, dim3(ew_grid(n)), dim3(256), 0, c->stream, p, g, m, v, n, (float)lr_t, beta1, beta2, eps);
    PS_HIP(hipGetLastError());
    return PS_OK;
}

int ps_op_dropout(ps_context* c, const float* x, int64_t n, uint32_t seed, float keep_prob, float* y, float* mask)
{
    PS_CHECK(c && x && y && mask && keep_prob > 0.f && keep_prob <= 1.f, "ps_op_dropout: bad argument");
    PS_HIP(hipSetDevice(c->device));
    hipLaunchKernelGGL(dropout_kernel, dim3(ew_grid(n)), dim3(256), 0, c->stream, x, n, seed, keep_prob, y, mask);
    PS_HIP(hipGetLastError());
    return PS_OK;
}

int ps_op_mul(ps_context* c, const float* a, const float* b, int64_t n, float* y)
{
    PS_CHECK(c && a && b && y, "ps_op_mul: NULL argument");
    PS_HIP(hipSetDevice(c->device));
    hipLaunchKernelGGL(mul_kernel, dim3(ew_grid(n)), dim3(256), 0, c->stream, a, b, n, y);
    PS_HIP(hipGetLastError());
    return PS_OK;
}

}  // extern "C"

extern "C" int ps_op_linear_wgrad_split(ps_context* c, const float* xl, int64_t ldxl, const int32_t* xidx, int64_t B, int64_t n_src, int64_t n_q, int64_t K,
                                        const float* xr, int64_t ldxr, const float* dy, int64_t lddy, int64_t cin, int64_t cout, float* dW)
{
    PS_CHECK(c && xl && xidx && xr && dy && dW && B >= 0 && n_src > 0 && n_q > 0 && K > 0, "ps_op_linear_wgrad_split: NULL argument");
    const int64_t R = B * n_q * K;
    PS_HIP(hipSetDevice(c->device));
    const int64_t nb = ps::wgrad_split_slabs(c, xl, ldxl, xidx, xr, ldxr, dy, lddy, R, cin, cout);
    PS_CHECK(nb > 0, "ps_op_linear_wgrad_split: needs >= 16384 rows, cin and cout multiples of 128, 16-byte aligned rows (and ps_set_train_gemm_b3 on)");
    const size_t wfl = (size_t)nb * cin * cout;
    PS_TRY(c->wgrad_ws.reserve(sizeof(float) * wfl + sizeof(ps::WgradJob) + 256));
    Stage st(c, "train_wgrad", 2);
    PS_TRY(ps::wgrad_b3_partial_split(c, xl, ldxl, xidx, n_src, n_q * K, xr, ldxr, dy, lddy, R, cin, cout, c->wgrad_ws.as<float>()));
    // (one job, finished on the spot: the native step batches its jobs instead)
    const ps::WgradJob job = {c->wgrad_ws.as<float>(), dW, (int)nb, (int)cin, (int)cout, 0};
    ps::WgradJob* dj = reinterpret_cast<ps::WgradJob*>(c->wgrad_ws.as<char>() + ((sizeof(float) * wfl + 255) & ~size_t(255)));
    PS_TRY(c->upload_async(dj, &job, sizeof(job)));
    return ps::wgrad_finish(c, dj, 1, cin * cout);
}
