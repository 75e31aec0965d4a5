// locse_train.hip -- the LocSE branch of the TRAINING step without a single [N*K, .] tensor of its own.
//
//   relative_pos_encoding + conv2d(10 -> h) + batch_normalization(training=True) + LeakyReLU
//   (PointSegment/RandLANet.py:323-325, 377-386; helper_tf_util.conv2d :115-170): f_xyz = lrelu(BN(enc10 . W + b))
//
// Op by op that is: enc10 written and read, the product written, read for the statistics, read again and written normalised; and in the
// backward the statistics pass and the apply pass of BatchNorm over (dz, y), the written dy, and a weight-gradient GEMM over (enc10, dy)
// -- 0.74 GB per pass at levels 0 and 1 of a batch of 8 x 180 000 points.  The input of this branch, though, is 12 bytes of coordinates
// and a 4-byte index per row: everything is recomputed from those where it is needed.
//   forward   sums   : y = enc10 . W + b on the fly, per-channel sum / sum of squares            (reads idx + xyz, writes 2h floats)
//             apply  : y again, normalise, LeakyReLU -> f_xyz rows                                (writes the one tensor somebody else reads)
//   backward  ONE pass over dz: with xh = (y - mean) invstd, g = dz lrelu'(.), the BatchNorm backward dy = gamma invstd (g - mean(g) -
//             xh mean(g xh)) is linear in per-channel constants, so the weight gradient enc10^T . dy needs only sums of this pass:
//                 dW = gamma invstd (enc10^T g  -  (sum enc10) x mean(g)  -  (enc10^T xh) . mean(g xh))
//             together with dgamma = sum g xh, dbeta = sum g.  enc10 has no gradient (coordinates).  The finishing arithmetic on the
//             [10, h] sums is the caller's (train.py: it is also where the sums of all ranks meet under SyncBN).
// fp32 VALU work (10 h FMAs per row against 16 + 4 h bytes): HBM / issue bound, no MFMA.  All sums are per-block partials merged in a
// fixed order (deterministic).  h in {8, 16, 32, 64}.
#include "common.h"
#include "bf16_io.h"
#include "reduce_partials.h"
#include "wave_ops.h"

namespace ps {

struct LocseArgs {
    const float* xyz;    // [B*N, 3]
    const int32_t* idx;  // [B*N, K] cloud-local
    int64_t rows;        // B*N*K
    int n_cloud, K, kshift;  // kshift = log2 K when K is a power of two, else -1
    float inv_n;             // 1 / n_cloud
    const float* w;      // [10, H]
    const float* b;      // [H]
    const float* scale;  // [H] gamma * invstd            (apply / backward)
    const float* shift;  // [H] beta                      (apply / backward: z = (y - mean) scale + beta)
    const float* mean;   // [H]                           (backward)
    const float* invstd; // [H]
    const float* dz;     // [rows, H] (lddz)              (backward)
    float* out;          // apply: [rows, H] (ldo)
    float* part;         // per-block partial sums
    int ldo, lddz;
    int out_bf16;        // apply: out rows -- backward: dz rows -- are bfloat16 (ps_set_train_act_bf16; ldo / lddz in elements)
};

// (row counts are < 2^31 and point counts < 2^24 here -- checked on the host --, so the two divisions of the row index are a shift /
//  a float reciprocal with one correction instead of 64-bit integer divisions, which cost more than the rest of the row)
__device__ __forceinline__ void locse_enc(const LocseArgs& a, int64_t t64, float (&e)[10])
{
    const unsigned t = (unsigned)t64;
    const unsigned p = a.kshift >= 0 ? t >> a.kshift : t / (unsigned)a.K;
    unsigned cb = (unsigned)((float)p * a.inv_n);
    cb -= (cb * (unsigned)a.n_cloud > p);                    // the estimate is off by at most one
    cb += ((cb + 1u) * (unsigned)a.n_cloud <= p);
    const unsigned q = cb * (unsigned)a.n_cloud + (unsigned)a.idx[t];
    const float cx = a.xyz[3u * p], cy = a.xyz[3u * p + 1], cz = a.xyz[3u * p + 2];
    const float nx = a.xyz[3u * q], ny = a.xyz[3u * q + 1], nz = a.xyz[3u * q + 2];
    const float rx = cx - nx, ry = cy - ny, rz = cz - nz;
    e[0] = __fsqrt_rn(rx * rx + ry * ry + rz * rz);  // (same expression as ps_op_relative_pos_encoding)
    e[1] = rx; e[2] = ry; e[3] = rz;
    e[4] = cx; e[5] = cy; e[6] = cz;
    e[7] = nx; e[8] = ny; e[9] = nz;
}

// four consecutive channels of y = enc10 . W + b; Ws = [10][H] then [H] biases in LDS (every lane reads the same words: broadcast)
template <int H>
__device__ __forceinline__ void locse_y4(const float* Ws, const float (&e)[10], int c0, float (&y)[4])
{
    const float4 bb = *reinterpret_cast<const float4*>(Ws + 10 * H + c0);
    y[0] = bb.x; y[1] = bb.y; y[2] = bb.z; y[3] = bb.w;
#pragma unroll
    for (int j = 0; j < 10; ++j) {
        const float4 wv = *reinterpret_cast<const float4*>(Ws + j * H + c0);
        y[0] = __builtin_fmaf(e[j], wv.x, y[0]);
        y[1] = __builtin_fmaf(e[j], wv.y, y[1]);
        y[2] = __builtin_fmaf(e[j], wv.z, y[2]);
        y[3] = __builtin_fmaf(e[j], wv.w, y[3]);
    }
}

template <int H>
__device__ __forceinline__ void locse_stage_w(const LocseArgs& a, float* Ws)
{
    for (int i = threadIdx.x; i < 10 * H; i += blockDim.x) Ws[i] = a.w[i];
    for (int i = threadIdx.x; i < H; i += blockDim.x) Ws[10 * H + i] = a.b[i];
    __syncthreads();
}

// A row is shared by LPR = H/4 lanes, four channels each (their weight columns live in registers); a wave covers 64 / LPR rows per step.
template <int H>
struct LocseLanes {
    static constexpr int LPR = H / 4, RPW = 64 / LPR;
    int part, rw, c0;
    float w[10][4], bias[4];
    __device__ __forceinline__ LocseLanes(const LocseArgs& a)
    {
        const int lane = threadIdx.x & 63;
        part = lane % LPR; rw = lane / LPR; c0 = 4 * part;
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            const float4 v = *reinterpret_cast<const float4*>(a.w + j * H + c0);
            w[j][0] = v.x; w[j][1] = v.y; w[j][2] = v.z; w[j][3] = v.w;
        }
        const float4 v = *reinterpret_cast<const float4*>(a.b + c0);
        bias[0] = v.x; bias[1] = v.y; bias[2] = v.z; bias[3] = v.w;
    }
    __device__ __forceinline__ void y4(const float (&e)[10], float (&y)[4]) const
    {
#pragma unroll
        for (int q = 0; q < 4; ++q) y[q] = bias[q];
#pragma unroll
        for (int j = 0; j < 10; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) y[q] = __builtin_fmaf(e[j], w[j][q], y[q]);
    }
    // sum over the rows of a wave (lanes LPR apart share `part`)
    static __device__ __forceinline__ float rsum(float v)
    {
#pragma unroll
        for (int o = LPR; o < 64; o <<= 1) v += __shfl_xor(v, o);
        return v;
    }
};

// ---- forward: statistics -------------------------------------------------------------------------------------------------
// Accumulated in fp64 (eight accumulators per lane): the variance is the difference of two sums that agree in their leading digits
// whenever a channel's mean is large against its spread (coordinates are), and fp32 sums would leave it with 1e-7 (mean^2 / var) of
// relative error -- visible in the logits four levels later.
template <int H>
__global__ __launch_bounds__(256) void locse_sums_kernel(LocseArgs a)
{
    using LL = LocseLanes<H>;
    __shared__ double red[4 * 2 * H];
    const LL L(a);
    const int wave = threadIdx.x >> 6;
    double s[4] = {0., 0., 0., 0.}, q2[4] = {0., 0., 0., 0.};
    const int64_t wstride = (int64_t)gridDim.x * 4 * LL::RPW;
    for (int64_t t0 = ((int64_t)blockIdx.x * 4 + wave) * LL::RPW; t0 < a.rows; t0 += wstride) {
        const int64_t t = t0 + L.rw;
        if (t >= a.rows) continue;
        float e[10], y[4];
        locse_enc(a, t, e);
        L.y4(e, y);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double yd = (double)y[q];
            s[q] += yd;
            q2[q] = __builtin_fma(yd, yd, q2[q]);
        }
    }
    auto rsum = [&](double v) {
#pragma unroll
        for (int o = LL::LPR; o < 64; o <<= 1) v += __shfl_xor(v, o);
        return v;
    };
    double* r = red + wave * 2 * H;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const double v1 = rsum(s[q]), v2 = rsum(q2[q]);
        if (L.rw == 0) { r[L.c0 + q] = v1; r[H + L.c0 + q] = v2; }
    }
    __syncthreads();
    double* dst = reinterpret_cast<double*>(a.part) + (size_t)blockIdx.x * 2 * H;
    for (int i = threadIdx.x; i < 2 * H; i += 256) dst[i] = ((red[i] + red[2 * H + i]) + red[4 * H + i]) + red[6 * H + i];
}

// ---- forward: normalise + LeakyReLU -> rows ----------------------------------------------------------------------------------
template <int H>
__global__ __launch_bounds__(256) void locse_apply_kernel(LocseArgs a)
{
    using LL = LocseLanes<H>;
    const LL L(a);
    const int wave = threadIdx.x >> 6;
    const float4 sc = *reinterpret_cast<const float4*>(a.scale + L.c0), sh = *reinterpret_cast<const float4*>(a.shift + L.c0);
    const float4 mu = *reinterpret_cast<const float4*>(a.mean + L.c0);
    const int64_t wstride = (int64_t)gridDim.x * 4 * LL::RPW;
    for (int64_t t0 = ((int64_t)blockIdx.x * 4 + wave) * LL::RPW; t0 < a.rows; t0 += wstride) {
        const int64_t t = t0 + L.rw;
        if (t >= a.rows) continue;
        float e[10], y[4];
        locse_enc(a, t, e);
        L.y4(e, y);
        // (y - mean) first: the subtraction is (nearly) exact, y scale - mean scale would cancel
        float z[4] = {__builtin_fmaf(y[0] - mu.x, sc.x, sh.x), __builtin_fmaf(y[1] - mu.y, sc.y, sh.y), __builtin_fmaf(y[2] - mu.z, sc.z, sh.z),
                      __builtin_fmaf(y[3] - mu.w, sc.w, sh.w)};
#pragma unroll
        for (int j = 0; j < 4; ++j) z[j] = z[j] < 0.f ? 0.2f * z[j] : z[j];
        store4_any(a.out, (size_t)t * a.ldo + L.c0, make_float4(z[0], z[1], z[2], z[3]), a.out_bf16 != 0);  // the LPR lanes of a row write it whole
    }
}

// ---- backward: every sum of the branch in one pass over dz -------------------------------------------------------------------
// A row is shared by LPR = H/4 lanes (four channels each); a lane's accumulators: S1, S2, XS (4 each), A = enc10^T g and
// G = enc10^T xh (10 x 4 each), E = sum enc10 (10, counted by the lane of channel 0).  Layout of the NV = 23 H + 16 results:
// S1[H] | S2[H] | XS[H] | A[10][H] | G[10][H] | E[16].
template <int H>
__global__ __launch_bounds__(256) void locse_bwd_kernel(LocseArgs a)
{
    constexpr int LPR = H / 4, RPW = 64 / LPR, NV = 23 * H + 16;
    __shared__ __attribute__((aligned(16))) float Ws[11 * H];
    __shared__ __attribute__((aligned(16))) float Ss[4 * H];  // scale | shift | mean | invstd
    __shared__ float red[4 * NV];
    for (int i = threadIdx.x; i < H; i += 256) { Ss[i] = a.scale[i]; Ss[H + i] = a.shift[i]; Ss[2 * H + i] = a.mean[i]; Ss[3 * H + i] = a.invstd[i]; }
    locse_stage_w<H>(a, Ws);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int part = lane % LPR, rw = lane / LPR, c0 = 4 * part;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f}, xs[4] = {0.f, 0.f, 0.f, 0.f};
    float A[10][4], G[10][4], E[10];
#pragma unroll
    for (int j = 0; j < 10; ++j) {
        E[j] = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) { A[j][q] = 0.f; G[j][q] = 0.f; }
    }
    const float4 sc = *reinterpret_cast<const float4*>(Ss + c0), sh = *reinterpret_cast<const float4*>(Ss + H + c0);
    const float4 mu = *reinterpret_cast<const float4*>(Ss + 2 * H + c0), is = *reinterpret_cast<const float4*>(Ss + 3 * H + c0);
    const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w}, muv[4] = {mu.x, mu.y, mu.z, mu.w}, isv[4] = {is.x, is.y, is.z, is.w};
    const int64_t wstride = (int64_t)gridDim.x * 4 * RPW;
    for (int64_t t0 = ((int64_t)blockIdx.x * 4 + wave) * RPW; t0 < a.rows; t0 += wstride) {
        const int64_t t = t0 + rw;
        if (t >= a.rows) continue;
        float e[10];
        locse_enc(a, t, e);
        float y[4];
        locse_y4<H>(Ws, e, c0, y);
        const float4 dv = load4_any(a.dz, (size_t)t * a.lddz + c0, a.out_bf16 != 0);  // (the gradient rows have the format of the rows apply wrote)
        const float dzv[4] = {dv.x, dv.y, dv.z, dv.w};
        float g[4], xh[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            xh[q] = (y[q] - muv[q]) * isv[q];
            g[q] = __builtin_fmaf(y[q] - muv[q], scv[q], shv[q]) < 0.f ? 0.2f * dzv[q] : dzv[q];
            s1[q] += g[q];
            s2[q] = __builtin_fmaf(g[q], xh[q], s2[q]);
            xs[q] += xh[q];
        }
#pragma unroll
        for (int j = 0; j < 10; ++j) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                A[j][q] = __builtin_fmaf(e[j], g[q], A[j][q]);
                G[j][q] = __builtin_fmaf(e[j], xh[q], G[j][q]);
            }
            if (part == 0) E[j] += e[j];
        }
    }
    // rows of a wave that share `part` sit LPR lanes apart: butterfly over the row axis, then the four waves through LDS
    auto rsum = [&](float v) {
#pragma unroll
        for (int o = LPR; o < 64; o <<= 1) v += __shfl_xor(v, o);
        return v;
    };
    float* r = red + wave * NV;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float v1 = rsum(s1[q]), v2 = rsum(s2[q]), v3 = rsum(xs[q]);
        if (rw == 0) { r[c0 + q] = v1; r[H + c0 + q] = v2; r[2 * H + c0 + q] = v3; }
    }
#pragma unroll
    for (int j = 0; j < 10; ++j) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float va = rsum(A[j][q]), vg = rsum(G[j][q]);
            if (rw == 0) { r[3 * H + j * H + c0 + q] = va; r[13 * H + j * H + c0 + q] = vg; }
        }
        const float ve = rsum(E[j]);
        if (lane == 0) r[23 * H + j] = ve;
    }
    if (lane < 6) r[23 * H + 10 + lane] = 0.f;
    __syncthreads();
    float* dst = a.part + (size_t)blockIdx.x * NV;
    for (int i = threadIdx.x; i < NV; i += 256) dst[i] = ((red[i] + red[NV + i]) + red[2 * NV + i]) + red[3 * NV + i];
}

static bool locse_ok(int64_t K, int64_t h) { return K >= 1 && (h == 8 || h == 16 || h == 32 || h == 64); }

template <int H>
static int locse_launch(ps_context* c, LocseArgs a, int what, float* result)
{
    const int nv = what == 0 ? 2 * H : 23 * H + 16;
    const int64_t per_block = 4 * (64 / (H / 4));  // rows per workgroup and step
    if (what == 1) {
        // (a multiple of what the chip holds at once: all workgroups walk the rows with the same stride, a partial last round costs a full one)
        static const int occ = [] {
            int o = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, reinterpret_cast<const void*>(locse_apply_kernel<H>), 256, 0) != hipSuccess || o < 1) {
                (void)hipGetLastError();
                o = 4;
            }
            return o;
        }();
        const unsigned blocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>((a.rows + per_block - 1) / per_block, 256 * std::min(occ, 8)));
        hipLaunchKernelGGL(locse_apply_kernel<H>, dim3(blocks), dim3(256), 0, c->stream, a);
    } else {
        const unsigned blocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>((a.rows + per_block - 1) / per_block, 256 * 4));
        PS_TRY(c->red_ws.reserve(sizeof(double) * (size_t)blocks * nv + 256));
        a.part = c->red_ws.as<float>();
        if (what == 0) {
            hipLaunchKernelGGL(locse_sums_kernel<H>, dim3(blocks), dim3(256), 0, c->stream, a);
            hipLaunchKernelGGL(reduce_partials_kernel<double>, dim3(ceil_div(nv, 16)), dim3(256), 0, c->stream, reinterpret_cast<const double*>(a.part),
                               (int)blocks, nv, reinterpret_cast<double*>(result));
        } else {
            hipLaunchKernelGGL(locse_bwd_kernel<H>, dim3(blocks), dim3(256), 0, c->stream, a);
            hipLaunchKernelGGL(reduce_partials_kernel<float>, dim3(ceil_div(nv, 16)), dim3(256), 0, c->stream, static_cast<const float*>(a.part), (int)blocks, nv,
                               result);
        }
    }
    PS_HIP(hipGetLastError());
    return PS_OK;
}

static int locse_dispatch(ps_context* c, int64_t h, LocseArgs a, int what, float* result)
{
    PS_CHECK(a.rows < (1ll << 31) && a.rows / a.K < (1ll << 24), "ps_op_locse_train: %lld rows exceed the 32-bit row / 24-bit point index range",
             (long long)a.rows);
    a.kshift = -1;
    for (int sft = 0; sft < 31; ++sft)
        if ((1 << sft) == a.K) a.kshift = sft;
    a.inv_n = 1.0f / (float)a.n_cloud;
    switch (h) {
        case 8: return locse_launch<8>(c, a, what, result);
        case 16: return locse_launch<16>(c, a, what, result);
        case 32: return locse_launch<32>(c, a, what, result);
        default: return locse_launch<64>(c, a, what, result);
    }
}

}  // namespace ps

using namespace ps;

extern "C" int ps_op_locse_train_supported(int64_t K, int64_t h) { return locse_ok(K, h) ? 1 : 0; }

extern "C" int ps_op_locse_train_sums(ps_context* c, const float* xyz, const int32_t* idx, int64_t B, int64_t N, int64_t K, const float* w, const float* b,
                                      int64_t h, double* sums)
{
    PS_CHECK(c && xyz && idx && w && b && sums, "ps_op_locse_train_sums: NULL argument");
    PS_CHECK(locse_ok(K, h) && B >= 0 && N >= 0, "ps_op_locse_train_sums: h in {8, 16, 32, 64} (got %lld)", (long long)h);
    PS_HIP(hipSetDevice(c->device));
    if (B * N * K == 0) {
        PS_HIP(hipMemsetAsync(sums, 0, sizeof(double) * 2 * h, c->stream));
        return PS_OK;
    }
    Stage st(c, "train_locse_fwd", 2);
    LocseArgs a = {};
    a.xyz = xyz; a.idx = idx; a.rows = B * N * K; a.n_cloud = (int)N; a.K = (int)K; a.w = w; a.b = b;
    return locse_dispatch(c, h, a, 0, reinterpret_cast<float*>(sums));
}

extern "C" int ps_op_locse_train_apply(ps_context* c, const float* xyz, const int32_t* idx, int64_t B, int64_t N, int64_t K, const float* w, const float* b,
                                       int64_t h, const float* mean, const float* scale, const float* beta, float* out, int64_t ldo)
{
    const float* shift = beta;
    PS_CHECK(c && xyz && idx && w && b && mean && scale && beta && out, "ps_op_locse_train_apply: NULL argument");
    PS_CHECK(locse_ok(K, h) && ldo >= h && ldo % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0,
             "ps_op_locse_train_apply: h in {8, 16, 32, 64}, rows 16-byte aligned");
    if (B * N * K == 0) return PS_OK;
    PS_HIP(hipSetDevice(c->device));
    Stage st(c, "train_locse_fwd", 1);
    LocseArgs a = {};
    a.xyz = xyz; a.idx = idx; a.rows = B * N * K; a.n_cloud = (int)N; a.K = (int)K; a.w = w; a.b = b;
    a.scale = scale; a.shift = shift; a.mean = mean; a.out = out; a.ldo = (int)ldo;
    a.out_bf16 = c->train_act_bf16 && c->train_bf16 ? 1 : 0;  // (rows of bfloat16: ldo in elements)
    PS_CHECK(!a.out_bf16 || (ldo % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 7) == 0), "ps_op_locse_train_apply: bfloat16 rows need ldo % 4 == 0 and 8-byte alignment");
    return locse_dispatch(c, h, a, 1, nullptr);
}

extern "C" int ps_op_locse_train_bwd(ps_context* c, const float* xyz, const int32_t* idx, int64_t B, int64_t N, int64_t K, const float* w, const float* b,
                                     int64_t h, const float* scale, const float* beta, const float* mean, const float* invstd, const float* dz,
                                     int64_t lddz, float* sums)
{
    const float* shift = beta;
    PS_CHECK(c && xyz && idx && w && b && scale && shift && mean && invstd && dz && sums, "ps_op_locse_train_bwd: NULL argument");
    PS_CHECK(locse_ok(K, h) && lddz >= h && lddz % 4 == 0 && (reinterpret_cast<uintptr_t>(dz) & 15) == 0,
             "ps_op_locse_train_bwd: h in {8, 16, 32, 64}, rows 16-byte aligned");
    PS_HIP(hipSetDevice(c->device));
    if (B * N * K == 0) {
        PS_HIP(hipMemsetAsync(sums, 0, sizeof(float) * (23 * h + 16), c->stream));
        return PS_OK;
    }
    Stage st(c, "train_locse_bwd", 2);
    LocseArgs a = {};
    a.xyz = xyz; a.idx = idx; a.rows = B * N * K; a.n_cloud = (int)N; a.K = (int)K; a.w = w; a.b = b;
    a.scale = scale; a.shift = shift; a.mean = mean; a.invstd = invstd; a.dz = dz; a.lddz = (int)lddz;
    a.out_bf16 = c->train_act_bf16 && c->train_bf16 ? 1 : 0;  // (dz as bfloat16 rows)
    return locse_dispatch(c, h, a, 2, sums);
}
