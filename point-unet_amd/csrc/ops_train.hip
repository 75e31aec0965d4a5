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
#include "common.h"
#include "mfma_tile.h"

namespace ps {

// ---- per-channel sums over rows: out0[c] += sum_r f0(r,c), out1[c] += sum_r f1(r,c) -------------------------------
// Layout trick: a block covers a contiguous slab of rows; thread t handles elements t, t+T, ... of the slab; with
// T % C == 0 its channel never changes.  (C that does not divide T falls back to per-element modulo.)
template <class F>
__global__ __launch_bounds__(256) void colreduce2_kernel(F f, int64_t R, int C, int rows_per_block, float* __restrict__ part0,
                                                         float* __restrict__ part1)
{
    // part{0,1}[block][c]: per-block partial sums, merged in a fixed order by colreduce_finish_kernel (no float atomics:
    // the batch statistics, and with them which side of the leaky-ReLU kink an activation falls on, are run-to-run identical)
    __shared__ float s0[256], s1[256];
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < R ? r0 + rows_per_block : R;
    const int64_t e0 = r0 * C, e1 = r1 * C;
    const bool fixed = (256 % C) == 0;
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
    float t0 = 0.f, t1 = 0.f;
    for (int b = threadIdx.x; b < blocks; b += 64) {
        t0 += part0[(size_t)b * C + c];
        if (part1) t1 += part1[(size_t)b * C + c];
    }
    for (int o = 32; o; o >>= 1) {
        t0 += __shfl_down(t0, o);
        t1 += __shfl_down(t1, o);
    }
    if (threadIdx.x == 0) {
        out0[c] = t0;
        if (out1) out1[c] = t1;
    }
}

template <class F>
static int colreduce2(ps_context* c, F f, int64_t R, int C, float* out0, float* out1)
{
    if (R <= 0) {
        PS_HIP(hipMemsetAsync(out0, 0, sizeof(float) * C, c->stream));
        if (out1) PS_HIP(hipMemsetAsync(out1, 0, sizeof(float) * C, c->stream));
        return PS_OK;
    }
    int64_t blocks = (R * C + 256 * 64 - 1) / (256 * 64);
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
    if ((256 % C) != 0) blocks = blocks > 512 ? 512 : blocks;
    const int rpb = (int)((R + blocks - 1) / blocks);
    const int nb = (int)((R + rpb - 1) / rpb);
    PS_TRY(c->red_ws.reserve(sizeof(float) * 2 * (size_t)nb * C));
    float* p0 = c->red_ws.as<float>();
    float* p1 = out1 ? p0 + (size_t)nb * C : nullptr;
    hipLaunchKernelGGL(colreduce2_kernel<F>, dim3((unsigned)nb), dim3(256), 0, c->stream, f, R, C, rpb, p0, p1);
    hipLaunchKernelGGL(colreduce_finish_kernel, dim3((unsigned)C), dim3(64), 0, c->stream, p0, p1, nb, C, out0, out1);
    PS_HIP(hipGetLastError());
    return PS_OK;
}

struct SumSq {
    const float* x;
    __device__ void operator()(int64_t e, int, float& a, float& b) const { const float v = x[e]; a = v; b = v * v; }
};
struct SumOnly {
    const float* x;
    __device__ void operator()(int64_t e, int, float& a, float& b) const { a = x[e]; b = 0.f; }
};
// BatchNorm backward sums: g = dy * act'(z), z = gamma*xhat + beta;  a = sum g, b = sum g*xhat
struct BnBwdSums {
    const float* dy; const float* x; const float* gamma; const float* beta; const float* mean; const float* invstd;
    int leaky;
    __device__ void operator()(int64_t e, int c, float& a, float& b) const
    {
        const float xh = (x[e] - mean[c]) * invstd[c];
        float g = dy[e];
        if (leaky && gamma[c] * xh + beta[c] < 0.f) g *= 0.2f;
        a = g;
        b = g * xh;
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

__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ mean, const float* __restrict__ invstd, int64_t total, int C, int leaky,
                                                       float* __restrict__ y)
{
    for (int64_t e = blockIdx.x * (int64_t)256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % C);
        float z = gamma[c] * ((x[e] - mean[c]) * invstd[c]) + beta[c];
        if (leaky && z < 0.f) z *= 0.2f;
        y[e] = z;
    }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ sg, const float* __restrict__ sgx,
                                                           int64_t total, int C, float invR, int leaky, float* __restrict__ dx)
{
    // dx = gamma*invstd * (g - mean_r(g) - xhat*mean_r(g*xhat))
    for (int64_t e = blockIdx.x * (int64_t)256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % C);
        const float xh = (x[e] - mean[c]) * invstd[c];
        float g = dy[e];
        if (leaky && gamma[c] * xh + beta[c] < 0.f) g *= 0.2f;
        dx[e] = gamma[c] * invstd[c] * (g - sg[c] * invR - xh * sgx[c] * invR);
    }
}

// ---- dW[cin,cout] += X^T . dY over a slab of rows; MFMA with the row axis as K -----------------------------------
// A[i][k] = X[r0+k][c0+i], B[k][j] = dY[r0+k][n0+j]: both read straight from global (16 consecutive channels per
// 16-lane group, 4 consecutive rows per k-step).
template <int NTB>
__global__ __launch_bounds__(256) void wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, int64_t R, int cin, int cout,
                                                    int64_t rows_per_wave, float* __restrict__ dW)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i16 = lane & 15, k4 = lane >> 4;
    const int c0 = blockIdx.y * 16, n0 = blockIdx.z * (16 * NTB);
    const int64_t w = (int64_t)blockIdx.x * 4 + wave;
    int64_t r = w * rows_per_wave;
    const int64_t rend = r + rows_per_wave < R ? r + rows_per_wave : R;
    if (r >= R) return;
    f32x4 acc[NTB];
#pragma unroll
    for (int j = 0; j < NTB; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool cok = c0 + i16 < cin;
    for (; r < rend; r += 4) {
        const int64_t row = r + k4;
        const bool rok = row < rend;
        const float a = (rok && cok) ? x[row * cin + c0 + i16] : 0.f;
#pragma unroll
        for (int j = 0; j < NTB; ++j) {
            const int n = n0 + j * 16 + i16;
            const float b = (rok && n < cout) ? dy[row * cout + n] : 0.f;
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int j = 0; j < NTB; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ci = c0 + k4 * 4 + q, n = n0 + j * 16 + i16;  // C layout: row = (lane>>4)*4 + q, col = lane&15
            if (ci < cin && n < cout) atomicAdd(&dW[(size_t)ci * cout + n], acc[j][q]);
        }
}

// ---- scatter-add of gathered rows (backward of tf.batch_gather) ---------------------------------------------------
__global__ __launch_bounds__(256) void scatter_add_kernel(const float* __restrict__ drows, const int32_t* __restrict__ idx, float* __restrict__ dpc,
                                                          size_t rows, int rows_per_cloud, int n_cloud, int d)
{
    const size_t t = blockIdx.x * (size_t)256 + threadIdx.x;
    if (t >= rows * d) return;
    const size_t row = t / d;
    const int ch = (int)(t - row * d);
    const size_t b = row / rows_per_cloud;
    atomicAdd(&dpc[(b * n_cloud + idx[row]) * d + ch], drows[t]);
}

// ---- softmax over K + weighted sum (att_pooling core, RandLANet.py:396-398) --------------------------------------
__global__ __launch_bounds__(256) void softpool_fwd_kernel(const float* __restrict__ fset, const float* __restrict__ scores, int64_t R, int K, int d,
                                                           float* __restrict__ probs, float* __restrict__ agg)
{
    const int64_t t = blockIdx.x * (int64_t)256 + threadIdx.x;  // (row, channel)
    if (t >= R * d) return;
    const int64_t r = t / d;
    const int c = (int)(t - r * d);
    const float* s = scores + r * K * d + c;
    const float* f = fset + r * K * d + c;
    float m = s[0];
    for (int k = 1; k < K; ++k) m = fmaxf(m, s[(size_t)k * d]);
    float den = 0.f;
    for (int k = 0; k < K; ++k) den += expf(s[(size_t)k * d] - m);
    float a = 0.f;
    for (int k = 0; k < K; ++k) {
        const float p = expf(s[(size_t)k * d] - m) / den;
        probs[r * K * d + (size_t)k * d + c] = p;
        a += f[(size_t)k * d] * p;
    }
    agg[t] = a;
}

__global__ __launch_bounds__(256) void softpool_bwd_kernel(const float* __restrict__ dagg, const float* __restrict__ fset, const float* __restrict__ probs,
                                                           int64_t R, int K, int d, float* __restrict__ dfset, float* __restrict__ dscores)
{
    const int64_t t = blockIdx.x * (int64_t)256 + threadIdx.x;
    if (t >= R * d) return;
    const int64_t r = t / d;
    const int c = (int)(t - r * d);
    const size_t base = (size_t)r * K * d + c;
    const float g = dagg[t];
    float dot = 0.f;  // sum_j p_j * dp_j,  dp_j = g * f_j
    for (int k = 0; k < K; ++k) dot += probs[base + (size_t)k * d] * g * fset[base + (size_t)k * d];
    for (int k = 0; k < K; ++k) {
        const float p = probs[base + (size_t)k * d], f = fset[base + (size_t)k * d];
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

// ---- class-weighted softmax cross-entropy, mean over rows (RandLANet.py:267-274) + its gradient ------------------
__global__ __launch_bounds__(256) void wce_kernel(const float* __restrict__ logits, const int32_t* __restrict__ labels, const float* __restrict__ cw,
                                                  int64_t R, int C, float* __restrict__ loss, float* __restrict__ dlogits)
{
    __shared__ float s[256];
    float acc = 0.f;
    for (int64_t r = blockIdx.x * (int64_t)256 + threadIdx.x; r < R; r += (int64_t)gridDim.x * 256) {
        const float* z = logits + r * C;
        float m = z[0];
        for (int c = 1; c < C; ++c) m = fmaxf(m, z[c]);
        float den = 0.f;
        for (int c = 0; c < C; ++c) den += expf(z[c] - m);
        const int y = labels[r];
        const float w = cw[y];
        acc += w * (logf(den) - (z[y] - m));
        if (dlogits)
            for (int c = 0; c < C; ++c) dlogits[r * C + c] = w * (expf(z[c] - m) / den - (c == y ? 1.f : 0.f)) / (float)R;
    }
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(loss, s[0] / (float)R);
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
    PS_CHECK(c && x && dy && dW, "ps_op_linear_wgrad: NULL argument");
    PS_HIP(hipSetDevice(c->device));
    Stage st(c, "train_wgrad", 2);
    PS_HIP(hipMemsetAsync(dW, 0, sizeof(float) * cin * cout, c->stream));
    if (R > 0) {
        const int ntb = cout >= 64 ? 4 : (cout >= 32 ? 2 : 1);
        const int ty = (int)((cin + 15) / 16), tz = (int)((cout + 16 * ntb - 1) / (16 * ntb));
        int64_t waves = 4096 / ((int64_t)ty * tz);
        waves = waves < 4 ? 4 : waves;
        int64_t rpw = (R + waves - 1) / waves;
        rpw = (rpw + 3) & ~int64_t(3);
        rpw = rpw < 64 ? 64 : rpw;
        const int64_t nw = (R + rpw - 1) / rpw;
        dim3 grid((unsigned)((nw + 3) / 4), ty, tz);
        if (ntb == 4) hipLaunchKernelGGL(wgrad_kernel<4>, grid, dim3(256), 0, c->stream, x, dy, R, (int)cin, (int)cout, rpw, dW);
        else if (ntb == 2) hipLaunchKernelGGL(wgrad_kernel<2>, grid, dim3(256), 0, c->stream, x, dy, R, (int)cin, (int)cout, rpw, dW);
        else hipLaunchKernelGGL(wgrad_kernel<1>, grid, dim3(256), 0, c->stream, x, dy, R, (int)cin, (int)cout, rpw, dW);
        PS_HIP(hipGetLastError());
    }
    if (db) PS_TRY(colreduce2(c, SumOnly{dy}, R, (int)cout, db, nullptr));
    return PS_OK;
}

int ps_op_bn_train_fwd(ps_context* c, const float* x, const float* gamma, const float* beta, int64_t R, int64_t C, float eps, int leaky, float* y,
                       float* mean, float* invstd, float* var, float* scratch2C)
{
    PS_CHECK(c && x && gamma && beta && y && mean && invstd && var && scratch2C, "ps_op_bn_train_fwd: NULL argument");
    PS_CHECK(R >= 1 && C >= 1, "ps_op_bn_train_fwd: empty tensor");
    PS_HIP(hipSetDevice(c->device));
    Stage st(c, "train_bn_fwd", 3);
    PS_TRY(colreduce2(c, SumSq{x}, R, (int)C, scratch2C, scratch2C + C));
    hipLaunchKernelGGL(bn_finish_stats_kernel, dim3(ceil_div(C, 64)), dim3(64), 0, c->stream, scratch2C, scratch2C + C, R, (int)C, eps, mean, invstd, var);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_grid(R * C)), dim3(256), 0, c->stream, x, gamma, beta, mean, invstd, R * C, (int)C, leaky, y);
    PS_HIP(hipGetLastError());
    return PS_OK;
}

int ps_op_bn_train_bwd(ps_context* c, const float* dy, const float* x, const float* gamma, const float* beta, const float* mean, const float* invstd,
                       int64_t R, int64_t C, int leaky, float* dx, float* dgamma, float* dbeta)
{
    PS_CHECK(c && dy && x && gamma && beta && mean && invstd && dx && dgamma && dbeta, "ps_op_bn_train_bwd: NULL argument");
    PS_HIP(hipSetDevice(c->device));
    Stage st(c, "train_bn_bwd", 2);
    // dbeta = sum g, dgamma = sum g*xhat
    PS_TRY(colreduce2(c, BnBwdSums{dy, x, gamma, beta, mean, invstd, leaky}, R, (int)C, dbeta, dgamma));
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_grid(R * C)), dim3(256), 0, c->stream, dy, x, gamma, beta, mean, invstd, dbeta, dgamma, R * C, (int)C,
                       1.0f / (float)R, leaky, dx);
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
    PS_CHECK(c && x && gamma && beta && sums2C && y && mean && invstd && var, "ps_op_bn_train_apply: NULL argument");
    PS_CHECK(R >= 1 && C >= 1 && R_total >= R, "ps_op_bn_train_apply: bad row counts");
    PS_HIP(hipSetDevice(c->device));
    Stage st(c, "train_bn_fwd", 2);
    hipLaunchKernelGGL(bn_finish_stats_kernel, dim3(ceil_div(C, 64)), dim3(64), 0, c->stream, sums2C, sums2C + C, R_total, (int)C, eps, mean, invstd, var);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_grid(R * C)), dim3(256), 0, c->stream, x, gamma, beta, mean, invstd, R * C, (int)C, leaky, y);
    PS_HIP(hipGetLastError());
    return PS_OK;
}

int ps_op_bn_train_bwd_sums(ps_context* c, const float* dy, const float* x, const float* gamma, const float* beta, const float* mean,
                            const float* invstd, int64_t R, int64_t C, int leaky, float* dgamma, float* dbeta)
{
    PS_CHECK(c && dy && x && gamma && beta && mean && invstd && dgamma && dbeta, "ps_op_bn_train_bwd_sums: NULL argument");
    PS_HIP(hipSetDevice(c->device));
    Stage st(c, "train_bn_bwd", 1);
    PS_TRY(colreduce2(c, BnBwdSums{dy, x, gamma, beta, mean, invstd, leaky}, R, (int)C, dbeta, dgamma));
    PS_HIP(hipGetLastError());
    return PS_OK;
}

int ps_op_bn_train_bwd_apply(ps_context* c, const float* dy, const float* x, const float* gamma, const float* beta, const float* mean,
                             const float* invstd, const float* sum_g, const float* sum_gx, int64_t R, int64_t R_total, int64_t C, int leaky, float* dx)
{
    PS_CHECK(c && dy && x && gamma && beta && mean && invstd && sum_g && sum_gx && dx, "ps_op_bn_train_bwd_apply: NULL argument");
    PS_CHECK(R >= 1 && C >= 1 && R_total >= R, "ps_op_bn_train_bwd_apply: bad row counts");
    PS_HIP(hipSetDevice(c->device));
    Stage st(c, "train_bn_bwd", 1);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_grid(R * C)), dim3(256), 0, c->stream, dy, x, gamma, beta, mean, invstd, sum_g, sum_gx, R * C, (int)C,
                       1.0f / (float)R_total, leaky, dx);
    PS_HIP(hipGetLastError());
    return PS_OK;
}

int ps_op_scatter_add_rows(ps_context* c, const float* drows, const int32_t* idx, int64_t B, int64_t N, int64_t rows_per_cloud, int64_t d, float* dpc)
{
    PS_CHECK(c && drows && idx && dpc, "ps_op_scatter_add_rows: NULL argument");
    const size_t rows = (size_t)B * rows_per_cloud;
    if (!rows) return PS_OK;
    PS_HIP(hipSetDevice(c->device));
    Stage st(c, "train_scatter_add", 1);
    hipLaunchKernelGGL(scatter_add_kernel, dim3(ceil_div(rows * d, 256)), dim3(256), 0, c->stream, drows, idx, dpc, rows, (int)rows_per_cloud, (int)N, (int)d);
    PS_HIP(hipGetLastError());
    return PS_OK;
}

int ps_op_softmax_pool_fwd(ps_context* c, const float* fset, const float* scores, int64_t R, int64_t K, int64_t d, float* probs, float* agg)
{
    PS_CHECK(c && fset && scores && probs && agg, "ps_op_softmax_pool_fwd: NULL argument");
    if (!R) return PS_OK;
    PS_HIP(hipSetDevice(c->device));
    Stage st(c, "train_softpool_fwd", 1);
    hipLaunchKernelGGL(softpool_fwd_kernel, dim3(ceil_div(R * d, 256)), dim3(256), 0, c->stream, fset, scores, R, (int)K, (int)d, probs, agg);
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
    hipLaunchKernelGGL(softpool_bwd_kernel, dim3(ceil_div(R * d, 256)), dim3(256), 0, c->stream, dagg, fset, probs, R, (int)K, (int)d, dfset, dscores);
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
    Stage st(c, "train_loss", 1);
    PS_HIP(hipMemsetAsync(loss, 0, sizeof(float), c->stream));
    hipLaunchKernelGGL(wce_kernel, dim3(ew_grid(R) > 1024 ? 1024 : ew_grid(R)), dim3(256), 0, c->stream, logits, labels, class_weights, R, (int)C, loss, dlogits);
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
