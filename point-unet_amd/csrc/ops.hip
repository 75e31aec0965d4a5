// ops.hip -- the op-by-op surface: one entry per Network.* static method of the reference graph
// (PointSegment/RandLANet.py:337-401), for callers that keep the reference's unfused call sites.  These
// materialise the [B,N,K,C] tensors exactly like the reference does; the fused production path is randla.hip.
// All of them are gather / elementwise kernels: HBM-bound, coalesced along the channel axis.
#include "common.h"

#include <algorithm>

#include <hip/hip_fp16.h>
#include "rowgemm.h"

namespace ps {

// out[row, 0:d] = pc[batch(row)*N + idx[row], :]   rows = B*M*K; out rows are ldo floats apart (ldo > d: the rows land in the
// left columns of a wider tensor, the training step's concat buffer)
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ pc, const int32_t* __restrict__ idx, float* __restrict__ out,
                                                          size_t rows, int rows_per_cloud, int n_cloud, int d, int ldo)
{
    const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (t >= rows * d) return;
    const size_t row = t / d;
    const int ch = (int)(t - row * d);
    const size_t b = row / rows_per_cloud;
    out[row * ldo + ch] = pc[(b * n_cloud + idx[row]) * d + ch];
}
// the same with one float4 per thread (d, ldo multiples of 4, 16-byte aligned bases)
__global__ __launch_bounds__(256) void gather_rows4_kernel(const float* __restrict__ pc, const int32_t* __restrict__ idx, float* __restrict__ out,
                                                           size_t rows, int rows_per_cloud, int n_cloud, int d4, int ldo)
{
    const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (t >= rows * d4) return;
    const size_t row = t / d4;
    const int q = (int)(t - row * d4);
    const size_t b = row / rows_per_cloud;
    *reinterpret_cast<float4*>(out + row * ldo + 4 * q) = *reinterpret_cast<const float4*>(pc + (b * n_cloud + idx[row]) * (size_t)(4 * d4) + 4 * q);
}

static void launch_gather_rows(ps_context* c, const float* pc, const int32_t* idx, float* out, size_t rows, int rows_per_cloud, int n_cloud, int d, int ldo)
{
    if ((d & 3) == 0 && (ldo & 3) == 0 && ((reinterpret_cast<uintptr_t>(pc) | reinterpret_cast<uintptr_t>(out)) & 15) == 0)
        hipLaunchKernelGGL(gather_rows4_kernel, dim3(ceil_div(rows * (d / 4), 256)), dim3(256), 0, c->stream, pc, idx, out, rows, rows_per_cloud, n_cloud,
                           d / 4, ldo);
    else
        hipLaunchKernelGGL(gather_rows_kernel, dim3(ceil_div(rows * d, 256)), dim3(256), 0, c->stream, pc, idx, out, rows, rows_per_cloud, n_cloud, d, ldo);
}

// out[b,n,k,0:10] = [dis, rel(3), centre(3), nbr(3)]   (RandLANet.py:337-343)
__global__ __launch_bounds__(256) void relpos_kernel(const float* __restrict__ xyz, const int32_t* __restrict__ idx, float* __restrict__ out,
                                                     size_t total /* B*N*K */, int n_cloud, int K)
{
    const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (t >= total) return;
    const size_t p = t / K;  // global point row
    const size_t b = p / n_cloud;
    const size_t q = b * n_cloud + idx[t];
    const float cx = xyz[3 * p], cy = xyz[3 * p + 1], cz = xyz[3 * p + 2];
    const float nx = xyz[3 * q], ny = xyz[3 * q + 1], nz = xyz[3 * q + 2];
    const float rx = cx - nx, ry = cy - ny, rz = cz - nz;
    float* o = out + t * 10;
    o[0] = __fsqrt_rn(rx * rx + ry * ry + rz * rz);
    o[1] = rx; o[2] = ry; o[3] = rz;
    o[4] = cx; o[5] = cy; o[6] = cz;
    o[7] = nx; o[8] = ny; o[9] = nz;
}

// out[row, ch] = max_k feature[batch*N + pool_idx[row,k], ch]   (RandLANet.py:345-360)
__global__ __launch_bounds__(256) void pool_max_scalar_kernel(const float* __restrict__ feat, const int32_t* __restrict__ idx, float* __restrict__ out,
                                                              size_t rows, int m_cloud, int n_cloud, int K, int d)
{
    const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (t >= rows * d) return;
    const size_t row = t / d;
    const int ch = (int)(t - row * d);
    const size_t base = (row / m_cloud) * n_cloud;
    const int32_t* ix = idx + row * K;
    float m = feat[(base + ix[0]) * d + ch];
    for (int k = 1; k < K; ++k) m = fmaxf(m, feat[(base + ix[k]) * d + ch]);
    out[t] = m;
}

__global__ void pack_weights_kernel(const float* __restrict__ W, int64_t sk, int64_t sn, int cin, int cout, int ntb, int ks, int cblocks, float* __restrict__ out)
{
    const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t total = (size_t)cblocks * ks * 64 * ntb;
    if (t >= total) return;
    const int j = (int)(t % ntb);
    const int l = (int)((t / ntb) % 64);
    const int s = (int)((t / ntb / 64) % ks);
    const int cb = (int)(t / ntb / 64 / ks);
    const int k = s * 4 + (l >> 4), col = (cb * ntb + j) * 16 + (l & 15);
    out[t] = (k < cin && col < cout) ? W[k * sk + col * sn] : 0.f;
}

__global__ void pack_weights_kperm_kernel(const float* __restrict__ W, int64_t sk, int64_t sn, int cin, int cout, int ntb, int nc, int cblocks, float* __restrict__ out)
{
    // k-permuted image of rowgemm.h: out[((((cb*nc + c)*16 + s)*64 + l)*ntb) + j] = W[c*64 + 16*(l>>4) + s][(cb*ntb + j)*16 + (l&15)]
    const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t total = (size_t)cblocks * nc * 16 * 64 * ntb;
    if (t >= total) return;
    const int j = (int)(t % ntb);
    const int l = (int)((t / ntb) % 64);
    const int s = (int)((t / ntb / 64) % 16);
    const int c = (int)((t / ntb / 64 / 16) % nc);
    const int cb = (int)(t / ntb / 64 / 16 / nc);
    const int k = c * 64 + 16 * (l >> 4) + s, col = (cb * ntb + j) * 16 + (l & 15);
    out[t] = (k < cin && col < cout) ? W[k * sk + col * sn] : 0.f;
}

// bf16 image of rowgemm.h (PackedLinear::wb): one thread per bf16 element
__global__ void pack_weights_bf16_kernel(const float* __restrict__ W, int64_t sk, int64_t sn, int cin, int cout, int ntb, int nc, int cblocks, __bf16* __restrict__ out)
{
    const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t total = (size_t)cblocks * nc * 2 * 64 * ntb * 8;
    if (t >= total) return;
    const int e = (int)(t % 8);
    const int j = (int)((t / 8) % ntb);
    const int l = (int)((t / 8 / ntb) % 64);
    const int s = (int)((t / 8 / ntb / 64) % 2);
    const int c = (int)((t / 8 / ntb / 64 / 2) % nc);
    const int cb = (int)(t / 8 / ntb / 64 / 2 / nc);
    const int k = c * 64 + 16 * (l >> 4) + 8 * s + e, col = (cb * ntb + j) * 16 + (l & 15);
    out[t] = (__bf16)((k < cin && col < cout) ? W[k * sk + col * sn] : 0.f);
}

// ---- the weight images of a training step in one launch (PackCache, common.h): grid.y = job, grid.x strides over its elements ----------
// (IT: the index type of the element decomposition -- 32-bit for every image a network has; the 64-bit divisions of the general form cost
//  more than the copy itself: 92 -> 2x us per step measured for the one-cloud training step)
template <class IT>
__device__ __forceinline__ void pack_one(const PackJob& j, IT t)
{
    const float* __restrict__ W = j.w;
    const IT ntb = (IT)j.ntb;
    if (j.kind == 0) {  // pack_weights_kernel
        const IT ks = (IT)j.p0;
        const IT jj = t % ntb, l = (t / ntb) % 64, s = (t / ntb / 64) % ks, cb = t / ntb / 64 / ks;
        const int k = (int)(s * 4 + (l >> 4)), col = (int)((cb * ntb + jj) * 16 + (l & 15));
        static_cast<float*>(j.out)[t] = (k < j.cin && col < j.cout) ? W[k * j.sk + col * j.sn] : 0.f;
    } else if (j.kind == 1) {  // pack_weights_kperm_kernel
        const IT nc = (IT)j.p0;
        const IT jj = t % ntb, l = (t / ntb) % 64, s = (t / ntb / 64) % 16, c = (t / ntb / 64 / 16) % nc, cb = t / ntb / 64 / 16 / nc;
        const int k = (int)(c * 64 + 16 * (l >> 4) + s), col = (int)((cb * ntb + jj) * 16 + (l & 15));
        static_cast<float*>(j.out)[t] = (k < j.cin && col < j.cout) ? W[k * j.sk + col * j.sn] : 0.f;
    } else {  // pack_weights_bf16_kernel
        const IT nc = (IT)j.p0;
        const IT e = t % 8, jj = (t / 8) % ntb, l = (t / 8 / ntb) % 64, s = (t / 8 / ntb / 64) % 2, c = (t / 8 / ntb / 64 / 2) % nc,
                 cb = t / 8 / ntb / 64 / 2 / nc;
        const int k = (int)(c * 64 + 16 * (l >> 4) + 8 * s + e), col = (int)((cb * ntb + jj) * 16 + (l & 15));
        static_cast<__bf16*>(j.out)[t] = (__bf16)((k < j.cin && col < j.cout) ? W[k * j.sk + col * j.sn] : 0.f);
    }
}

__global__ __launch_bounds__(256) void pack_batch_ops_kernel(const PackJob* __restrict__ jobs)
{
    const PackJob j = jobs[blockIdx.y];
    if (j.total < (1ll << 31)) {
        const unsigned total = (unsigned)j.total;
        for (unsigned t = blockIdx.x * 256u + threadIdx.x; t < total; t += gridDim.x * 256u) pack_one<unsigned>(j, t);
    } else {
        for (int64_t t = blockIdx.x * (int64_t)256 + threadIdx.x; t < j.total; t += (int64_t)gridDim.x * 256) pack_one<int64_t>(j, t);
    }
}

int pack_batch_ops(ps_context* c, const PackJob* table, int n)
{
    if (n <= 0) return PS_OK;
    hipLaunchKernelGGL(pack_batch_ops_kernel, dim3(256, (unsigned)n), dim3(256), 0, c->stream, table);
    PS_HIP(hipGetLastError());
    return PS_OK;
}

void pack_cache_clear(PackCache& pc)
{
    for (auto& b : pc.bufs) b.release();
    pc.bufs.clear();
    pc.jobs.clear();
    pc.table.release();
    pc.mode = 0;
    pc.broken = false;
    pc.cursor = 0;
    pc.n_ops = pc.n_b3 = 0;
}

void* pack_slot(ps_context* c, const PackJob& key, size_t bytes, bool& launch)
{
    PackCache* pc = c->pack_cache;
    launch = true;
    if (pc && !pc->broken && pc->mode == 2) {
        if (pc->cursor < pc->jobs.size() && pc->jobs[pc->cursor].same_key(key)) {
            launch = false;  // packed at the start of the step
            return pc->jobs[pc->cursor++].out;
        }
        pc->broken = true;  // another sequence of products than the recorded one: this step packs call by call, the next one records again
    }
    if (pc && !pc->broken && pc->mode == 1) {
        DevBuf b;
        if (b.reserve(bytes) == PS_OK) {
            PackJob j = key;
            j.out = b.p;
            pc->jobs.push_back(j);
            pc->bufs.push_back(b);
            return b.p;
        }
        pc->broken = true;
    }
    ps::DevBuf& ws = c->ops_ring[c->ops_ring_pos];  // a small ring: consecutive calls on the stream must not overwrite weights still being read
    c->ops_ring_pos = (c->ops_ring_pos + 1) & 3;
    if (ws.reserve(bytes) != PS_OK) return nullptr;
    return ws.p;
}

int pack_cache_finish_recording(ps_context* c, PackCache& pc)
{
    // table order: ops.hip's kinds first, then gemm_b3.hip's (each batched kernel gets a contiguous slice); `jobs` keeps the call order
    std::vector<PackJob> tab;
    for (const auto& j : pc.jobs)
        if (j.kind <= 2) tab.push_back(j);
    pc.n_ops = (int)tab.size();
    for (const auto& j : pc.jobs)
        if (j.kind > 2) tab.push_back(j);
    pc.n_b3 = (int)tab.size() - pc.n_ops;
    if (tab.empty()) return PS_OK;
    PS_TRY(pc.table.reserve(sizeof(PackJob) * tab.size()));
    PS_HIP(hipMemcpyAsync(pc.table.p, tab.data(), sizeof(PackJob) * tab.size(), hipMemcpyHostToDevice, c->stream));
    PS_HIP(hipStreamSynchronize(c->stream));  // (`tab` is a local; once per recording)
    pc.mode = 2;
    return PS_OK;
}

int pack_cache_replay(ps_context* c, PackCache& pc)
{
    pc.cursor = 0;
    if (pc.mode != 2 || pc.broken) return PS_OK;
    Stage st(c, "op_conv1x1", 2);
    PS_TRY(pack_batch_ops(c, pc.table.as<PackJob>(), pc.n_ops));
    PS_TRY(pack_batch_b3(c, pc.table.as<PackJob>() + pc.n_ops, pc.n_b3));
    return PS_OK;
}

// agg[r, col] = sum_k fset[r,k,col] * softmax_k( (fset[r] . wfc)[k, col] )    (RandLANet.py:394-398)
template <int KMAX>
__global__ __launch_bounds__(256) void att_pool_op_kernel(const float* __restrict__ fset, const float* __restrict__ wfc, float* __restrict__ agg,
                                                          int R, int K, int d)
{
    extern __shared__ float tile[];  // [K][d]
    for (int r = blockIdx.x; r < R; r += gridDim.x) {
        __syncthreads();
        for (int e = threadIdx.x; e < K * d; e += blockDim.x) tile[e] = fset[(size_t)r * K * d + e];
        __syncthreads();
        for (int col = threadIdx.x; col < d; col += blockDim.x) {
            float s[KMAX];
#pragma unroll
            for (int k = 0; k < KMAX; ++k) s[k] = 0.f;
            for (int j = 0; j < d; ++j) {
                const float w = wfc[(size_t)j * d + col];
#pragma unroll
                for (int k = 0; k < KMAX; ++k)
                    if (k < K) s[k] = fmaf(tile[k * d + j], w, s[k]);
            }
            float m = s[0];
#pragma unroll
            for (int k = 1; k < KMAX; ++k)
                if (k < K) m = fmaxf(m, s[k]);
            float den = 0.f, num = 0.f;
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
                if (k < K) {
                    const float e = expf(s[k] - m);
                    den += e;
                    num += e * tile[k * d + col];
                }
            agg[(size_t)r * d + col] = num / den;
        }
    }
}

}  // namespace ps

using namespace ps;

extern "C" int ps_op_gather_neighbour_ex(ps_context* c, const float* pc, const int32_t* idx, int64_t B, int64_t N, int64_t M, int64_t K, int64_t d,
                                         float* out, int64_t ldo)
{
    PS_CHECK(c && pc && idx && out, "ps_op_gather_neighbour: NULL argument");
    PS_CHECK(B >= 0 && N >= 1 && M >= 0 && K >= 1 && d >= 1 && ldo >= d, "ps_op_gather_neighbour: bad shape");
    const size_t rows = (size_t)B * M * K;
    if (!rows) return PS_OK;
    PS_HIP(hipSetDevice(c->device));
    Stage st(c, "op_gather_neighbour", 1);
    launch_gather_rows(c, pc, idx, out, rows, (int)(M * K), (int)N, (int)d, (int)ldo);
    PS_HIP(hipGetLastError());
    return PS_OK;
}

extern "C" int ps_op_gather_neighbour(ps_context* c, const float* pc, const int32_t* idx, int64_t B, int64_t N, int64_t M, int64_t K, int64_t d,
                                      float* out)
{
    return ps_op_gather_neighbour_ex(c, pc, idx, B, N, M, K, d, out, d);
}

extern "C" int ps_op_relative_pos_encoding(ps_context* c, const float* xyz, const int32_t* idx, int64_t B, int64_t N, int64_t K, float* out)
{
    PS_CHECK(c && xyz && idx && out, "ps_op_relative_pos_encoding: NULL argument");
    const size_t total = (size_t)B * N * K;
    if (!total) return PS_OK;
    PS_HIP(hipSetDevice(c->device));
    Stage st(c, "op_relative_pos_encoding", 1);
    hipLaunchKernelGGL(relpos_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, c->stream, xyz, idx, out, total, (int)N, (int)K);
    PS_HIP(hipGetLastError());
    return PS_OK;
}

extern "C" int ps_op_random_sample(ps_context* c, const float* feature, const int32_t* pool_idx, int64_t B, int64_t N, int64_t M, int64_t K,
                                   int64_t d, float* out)
{
    PS_CHECK(c && feature && pool_idx && out, "ps_op_random_sample: NULL argument");
    const size_t rows = (size_t)B * M;
    if (!rows) return PS_OK;
    PS_HIP(hipSetDevice(c->device));
    Stage st(c, "op_random_sample", 1);
    // (rows of whole float4s: the vector kernel of the fused forward -- all K gathers of a thread in flight together)
    if (d % 4 == 0 && (reinterpret_cast<uintptr_t>(feature) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 && B * std::max(N, M) < (1ll << 31))
        return pool_max(c, feature, pool_idx, nullptr, out, B, N, M, (int)K, (int)d);
    hipLaunchKernelGGL(pool_max_scalar_kernel, dim3(ceil_div(rows * d, 256)), dim3(256), 0, c->stream, feature, pool_idx, out, rows, (int)M, (int)N,
                       (int)K, (int)d);
    PS_HIP(hipGetLastError());
    return PS_OK;
}

extern "C" int ps_op_random_sample_ties(ps_context* c, const float* feature, const int32_t* pool_idx, int64_t B, int64_t N, int64_t M, int64_t K, int64_t d,
                                        float* out, uint8_t* ties)
{
    PS_CHECK(c && feature && pool_idx && out && ties, "ps_op_random_sample_ties: NULL argument");
    PS_CHECK(d % 4 == 0 && K <= 255 && (reinterpret_cast<uintptr_t>(feature) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 &&
                 (reinterpret_cast<uintptr_t>(ties) & 3) == 0 && B * std::max(N, M) < (1ll << 31),
             "ps_op_random_sample_ties: d must be a multiple of 4, K <= 255, buffers 16-byte aligned");
    if (!(B * M)) return PS_OK;
    PS_HIP(hipSetDevice(c->device));
    Stage st(c, "op_random_sample", 1);
    return pool_max(c, feature, pool_idx, nullptr, out, B, N, M, (int)K, (int)d, ties);
}

extern "C" int ps_op_nearest_interpolation(ps_context* c, const float* feature, const int32_t* interp_idx, int64_t B, int64_t N, int64_t M, int64_t d,
                                           float* out)
{
    PS_CHECK(c && feature && interp_idx && out, "ps_op_nearest_interpolation: NULL argument");
    const size_t rows = (size_t)B * M;
    if (!rows) return PS_OK;
    PS_HIP(hipSetDevice(c->device));
    Stage st(c, "op_nearest_interpolation", 1);
    launch_gather_rows(c, feature, interp_idx, out, rows, (int)M, (int)N, (int)d, (int)d);
    PS_HIP(hipGetLastError());
    return PS_OK;
}

namespace ps {
// ---- tiny convolutions over millions of rows (level 0 of the training step: the h -> h LocSE convolution on [B*N*K, 8] rows and its input
// gradient): 64 multiply-adds per 64 bytes of row traffic.  The MFMA kernels stage such rows through LDS in 16-row tiles and ran at
// 2.2 TB/s; here a THREAD owns a row -- 16-byte loads, the CIN x COUT weights broadcast from LDS, 16-byte stores -- and the pass runs
// at the HBM rate.  bf16 mode: both operands rounded (RNE) before the fp32 multiply-add, like the MFMA flavours.
template <int CIN, int COUT>
__global__ __launch_bounds__(256) void tinyconv_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ w, int wt, const float* __restrict__ b,
                                                       int64_t R, int leaky, int accum, int bf16, float* __restrict__ y, int64_t ldy)
{
    __shared__ __attribute__((aligned(16))) float W[CIN * COUT];
    __shared__ float Bv[COUT];
    auto rb = [&](float v) {
        unsigned u = __float_as_uint(v);
        u += 0x7fffu + ((u >> 16) & 1u);
        return __uint_as_float(u & 0xffff0000u);
    };
    for (int i = threadIdx.x; i < CIN * COUT; i += 256) {  // W[k][n]; wt: the matrix is stored [n][k]
        const float v = wt ? w[(i % COUT) * CIN + i / COUT] : w[i];
        W[i] = bf16 ? rb(v) : v;
    }
    if ((int)threadIdx.x < COUT) Bv[threadIdx.x] = b ? b[threadIdx.x] : 0.f;
    __syncthreads();
    for (int64_t r = blockIdx.x * (int64_t)256 + threadIdx.x; r < R; r += (int64_t)gridDim.x * 256) {
        float xv[CIN];
#pragma unroll
        for (int q = 0; q < CIN / 4; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(x + r * ldx + 4 * q);
            xv[4 * q] = v.x; xv[4 * q + 1] = v.y; xv[4 * q + 2] = v.z; xv[4 * q + 3] = v.w;
        }
        if (bf16) {
#pragma unroll
            for (int k = 0; k < CIN; ++k) xv[k] = rb(xv[k]);
        }
        float acc[COUT];
#pragma unroll
        for (int n = 0; n < COUT; ++n) acc[n] = Bv[n];
#pragma unroll
        for (int k = 0; k < CIN; ++k)
#pragma unroll
            for (int q = 0; q < COUT / 4; ++q) {
                const float4 wv = *reinterpret_cast<const float4*>(&W[k * COUT + 4 * q]);  // (same address in every lane: an LDS broadcast)
                acc[4 * q] = __builtin_fmaf(xv[k], wv.x, acc[4 * q]);
                acc[4 * q + 1] = __builtin_fmaf(xv[k], wv.y, acc[4 * q + 1]);
                acc[4 * q + 2] = __builtin_fmaf(xv[k], wv.z, acc[4 * q + 2]);
                acc[4 * q + 3] = __builtin_fmaf(xv[k], wv.w, acc[4 * q + 3]);
            }
#pragma unroll
        for (int q = 0; q < COUT / 4; ++q) {
            float4 o = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
            if (leaky) { o.x = fmaxf(o.x, 0.2f * o.x); o.y = fmaxf(o.y, 0.2f * o.y); o.z = fmaxf(o.z, 0.2f * o.z); o.w = fmaxf(o.w, 0.2f * o.w); }
            float4* dst = reinterpret_cast<float4*>(y + r * ldy + 4 * q);
            if (accum) {
                const float4 old = *dst;
                o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
            }
            *dst = o;
        }
    }
}

static bool tinyconv_fits(int64_t R, int64_t cin, int64_t cout, const float* x, int64_t ldx, const float* y, int64_t ldy)
{
    return R >= (1 << 20) && (cin == 8 || cin == 16) && (cout == 8 || cout == 16) && ldx % 4 == 0 && ldy % 4 == 0 &&
           ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
}
}  // namespace ps

extern "C" int ps_op_conv1x1_ex(ps_context* c, const float* x, int64_t ldx, const float* w, const float* b, int64_t R, int64_t cin, int64_t cout, int leaky,
                                int accumulate, float* y, int64_t ldy)
{
    PS_CHECK(c && x && w && y, "ps_op_conv1x1: NULL argument");
    PS_CHECK(R >= 0 && cin >= 1 && cout >= 1 && ldx >= cin && ldy >= cout, "ps_op_conv1x1: bad shape");
    if (!R) return PS_OK;
    PS_HIP(hipSetDevice(c->device));
    // (internal switch of the native training step: `w` is stored [cout, cin] -- the transposed-convolution kernels of the decoder in the
    //  forward pass, every other layer's kernel in its input-gradient GEMM; the packing kernels read it through strides, no transposed copy)
    const bool wt = c->conv_w_transposed;
    const int64_t sk = wt ? 1 : cout, sn = wt ? cin : 1;
    if (tinyconv_fits(R, cin, cout, x, ldx, y, ldy)) {
        Stage st(c, "op_conv1x1", 1);
        const dim3 grid(2048);
        // (the MFMA bf16 flavour only covers cin % 16 == 0: the rounding follows the same rule here)
        const int bf = c->train_bf16 && cin % 16 == 0 ? 1 : 0;
#define PS_TINY(CI, CO) hipLaunchKernelGGL((tinyconv_kernel<CI, CO>), grid, dim3(256), 0, c->stream, x, ldx, w, wt ? 1 : 0, b, R, leaky, accumulate ? 1 : 0, bf, y, ldy)
        if (cin == 8 && cout == 8) PS_TINY(8, 8);
        else if (cin == 8) PS_TINY(8, 16);
        else if (cout == 8) PS_TINY(16, 8);
        else PS_TINY(16, 16);
#undef PS_TINY
        PS_HIP(hipGetLastError());
        return PS_OK;
    }
    if (c->train_b3 && gemm_b3_fits(c->tune, R, cin, cout, x, ldx, c->train_bf16)) {  // (bf16-MLP mode: the same tiling on ONE plane of rounded operands)
        // matrix-pipe bound shapes (att_pooling's score products at d >= 128): bf16 MFMA over exact splits, fp32-level error
        PackJob key = {};
        key.w = w; key.sk = sk; key.sn = sn; key.kind = c->train_bf16 ? 4 : 3; key.cin = (int)cin; key.cout = (int)cout;
        key.total = (int64_t)(cin / 16) * (cout / 32) * 64;
        bool launch = true;
        void* planes = pack_slot(c, key, gemm_b3_plane_bytes(cin, cout), launch);
        PS_CHECK(planes != nullptr, "ps_op_conv1x1: out of device memory for the weight planes");
        Stage st(c, "op_conv1x1", 2);
        return gemm_b3(c, x, ldx, w, sk, sn, b, R, cin, cout, leaky, accumulate, y, ldy, planes, launch ? 1 : 0);
    }
    PackedLinear L;
    L.cin = (int)cin; L.cout = (int)cout; L.leaky = leaky; L.accum = accumulate ? 1 : 0;
    L.ks = (L.cin + 3) / 4;
    L.ntb = choose_ntb(L.cout);
    L.cblocks = (L.cout + 16 * L.ntb - 1) / (16 * L.ntb);
    const bool kperm = (cin % 16) == 0 && (ldx % 4) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;  // the direct-load kernel's layout (rowgemm.h)
    const bool bf16 = kperm && c->train_bf16;
    const size_t need = bf16 ? L.bf16_bytes() : (kperm ? L.kperm_floats() : L.packed_floats()) * sizeof(float);
    PackJob key = {};
    key.w = w; key.sk = sk; key.sn = sn; key.kind = bf16 ? 2 : (kperm ? 1 : 0); key.cin = L.cin; key.cout = L.cout; key.ntb = L.ntb;
    key.p0 = (bf16 || kperm) ? L.nchunks() : L.ks; key.cblocks = L.cblocks;
    key.total = bf16 ? (int64_t)(L.bf16_bytes() / 2) : (int64_t)(kperm ? L.kperm_floats() : L.packed_floats());
    bool launch = true;
    void* wsp = pack_slot(c, key, need, launch);  // (the step's image when the native trainer replays its pack list, else a ring buffer)
    PS_CHECK(wsp != nullptr, "ps_op_conv1x1: out of device memory for the packed weights");
    Stage st(c, "op_conv1x1", 2);
    if (launch) {
        if (bf16)
            hipLaunchKernelGGL(pack_weights_bf16_kernel, dim3(ceil_div(L.bf16_bytes() / 2, 256)), dim3(256), 0, c->stream, w, sk, sn, L.cin, L.cout, L.ntb,
                               L.nchunks(), L.cblocks, static_cast<__bf16*>(wsp));
        else if (kperm)
            hipLaunchKernelGGL(pack_weights_kperm_kernel, dim3(ceil_div(L.kperm_floats(), 256)), dim3(256), 0, c->stream, w, sk, sn, L.cin, L.cout, L.ntb,
                               L.nchunks(), L.cblocks, static_cast<float*>(wsp));
        else
            hipLaunchKernelGGL(pack_weights_kernel, dim3(ceil_div(L.packed_floats(), 256)), dim3(256), 0, c->stream, w, sk, sn, L.cin, L.cout, L.ntb, L.ks,
                               L.cblocks, static_cast<float*>(wsp));
        PS_HIP(hipGetLastError());
    }
    L.wp = kperm ? nullptr : static_cast<float*>(wsp);
    L.wq = (kperm && !bf16) ? static_cast<float*>(wsp) : nullptr;
    L.wb = bf16 ? wsp : nullptr;
    L.bias = b;
    RowSrc s1, none;
    s1.x = x; s1.ld = (int)ldx; s1.c = L.cin;
    return rowgemm(c, L, s1, none, R, y, (int)ldy);
}

extern "C" int ps_op_conv1x1(ps_context* c, const float* x, const float* w, const float* b, int64_t R, int64_t cin, int64_t cout, int leaky, float* y)
{
    return ps_op_conv1x1_ex(c, x, cin, w, b, R, cin, cout, leaky, 0, y, cout);
}

extern "C" int ps_op_att_pool(ps_context* c, const float* fset, const float* wfc, int64_t R, int64_t K, int64_t d, float* agg)
{
    PS_CHECK(c && fset && wfc && agg, "ps_op_att_pool: NULL argument");
    PS_CHECK(K >= 1 && K <= 32 && d >= 1, "ps_op_att_pool: K must be in 1..32");
    if (!R) return PS_OK;
    PS_HIP(hipSetDevice(c->device));
    const size_t smem = (size_t)K * d * sizeof(float);
    PS_CHECK(smem <= 160 * 1024, "ps_op_att_pool: K*d too large for the LDS");
    Stage st(c, "op_att_pool", 1);
    const int blocks = (int)std::min<int64_t>(R, 256 * 8);
    if (K <= 16) {
        auto kern = att_pool_op_kernel<16>;
        if (smem > 48 * 1024) PS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), smem, c->stream, fset, wfc, agg, (int)R, (int)K, (int)d);
    } else {
        auto kern = att_pool_op_kernel<32>;
        if (smem > 48 * 1024) PS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), smem, c->stream, fset, wfc, agg, (int)R, (int)K, (int)d);
    }
    PS_HIP(hipGetLastError());
    return PS_OK;
}

// --------------------------------------------------------------------------------------------------------
// point -> volume scatter of the class probabilities (the post-processing loop of the reference's tester:
// PointSegment/testBraTS.py:83-101 point2prod + :226-231; testPancreas.py:71-85):
//     test_probs = zeros[total, C];  test_probs[p_idx] = softmax(logits)          (last duplicate wins)
//     volume[z][x][y] = test_probs[i] for i in 0..total (last duplicate wins);  np.moveaxis(volume, 1, 2)
// i.e. out[z, y, x, :] = softmax(logits[j]) where i is the LAST point sitting on voxel (x,y,z) and j the LAST sampled row
// with p_idx[j] == i (zeros if the voxel's last point was not sampled).  Deterministic: winners by atomicMax.
// --------------------------------------------------------------------------------------------------------
namespace ps {

__global__ __launch_bounds__(256) void winner_rows_kernel(const int32_t* __restrict__ p_idx, int n, int total, int32_t* __restrict__ winner_pt)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const int i = p_idx ? p_idx[j] : j;
    if (i >= 0 && i < total) atomicMax(&winner_pt[i], j);
}

__global__ __launch_bounds__(256) void winner_vox_kernel(const int32_t* __restrict__ xyz, int total, int Z, int X, int Y, int32_t* __restrict__ winner_vox)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int x = xyz[3 * (size_t)i], y = xyz[3 * (size_t)i + 1], z = xyz[3 * (size_t)i + 2];
    if (x < 0 || x >= X || y < 0 || y >= Y || z < 0 || z >= Z) return;
    atomicMax(&winner_vox[((size_t)z * Y + y) * X + x], i);
}

__global__ __launch_bounds__(256) void fill_volume_kernel(const float* __restrict__ logits, const int32_t* __restrict__ winner_pt,
                                                          const int32_t* __restrict__ winner_vox, size_t nvox, int C, float* __restrict__ out)
{
    const size_t v = blockIdx.x * (size_t)256 + threadIdx.x;
    if (v >= nvox) return;
    const int i = winner_vox[v];
    const int j = i >= 0 ? winner_pt[i] : -1;
    float* o = out + v * C;
    if (j < 0) {
        for (int c = 0; c < C; ++c) o[c] = 0.f;
        return;
    }
    const float* z = logits + (size_t)j * C;
    float m = z[0];
    for (int c = 1; c < C; ++c) m = fmaxf(m, z[c]);
    float den = 0.f;
    for (int c = 0; c < C; ++c) den += expf(z[c] - m);
    for (int c = 0; c < C; ++c) o[c] = expf(z[c] - m) / den;
}

}  // namespace ps

namespace ps {
__global__ void half_to_float_kernel(const __half* __restrict__ in, float* __restrict__ out, int64_t n)
{
    for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = __half2float(in[i]);
}
}  // namespace ps

extern "C" int ps_op_half_to_float(ps_context* c, const uint16_t* in, int64_t n, float* out)
{
    PS_CHECK(c && in && out && n >= 0, "ps_op_half_to_float: bad argument");
    if (!n) return PS_OK;
    PS_HIP(hipSetDevice(c->device));
    Stage st(c, "op_half_to_float", 1);
    hipLaunchKernelGGL(half_to_float_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 4096)), dim3(256), 0, c->stream,
                       reinterpret_cast<const __half*>(in), out, n);
    PS_HIP(hipGetLastError());
    return PS_OK;
}

extern "C" int ps_op_probs_to_volume(ps_context* c, const float* logits, int64_t n, int64_t C, const int32_t* p_idx, const int32_t* xyz_origin,
                                     int64_t total, int64_t Z, int64_t X, int64_t Y, float* volume, int32_t* scratch)
{
    PS_CHECK(c && logits && xyz_origin && volume && scratch, "ps_op_probs_to_volume: NULL argument");
    PS_CHECK(n >= 0 && total >= 1 && C >= 1 && Z >= 1 && X >= 1 && Y >= 1, "ps_op_probs_to_volume: bad shape");
    PS_HIP(hipSetDevice(c->device));
    const size_t nvox = (size_t)Z * X * Y;
    int32_t* winner_pt = scratch;
    int32_t* winner_vox = scratch + total;
    Stage st(c, "op_probs_to_volume", 3);
    PS_HIP(hipMemsetAsync(scratch, 0xff, sizeof(int32_t) * (total + nvox), c->stream));  // -1
    if (n) hipLaunchKernelGGL(winner_rows_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, c->stream, p_idx, (int)n, (int)total, winner_pt);
    hipLaunchKernelGGL(winner_vox_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, c->stream, xyz_origin, (int)total, (int)Z, (int)X, (int)Y, winner_vox);
    hipLaunchKernelGGL(fill_volume_kernel, dim3(ceil_div(nvox, 256)), dim3(256), 0, c->stream, logits, winner_pt, winner_vox, nvox, (int)C, volume);
    PS_HIP(hipGetLastError());
    return PS_OK;
}
