// grid.hip -- voxel-grid barycentre subsampling on the device (ps_grid_subsample).
//
// Replaces grid_subsampling() (PointSegment/utils/cpp_wrappers/cpp_subsampling/grid_subsampling/
// grid_subsampling.cpp:5-106; accumulators grid_subsampling.h:10-80; bounds cpp_utils/cloud/cloud.cpp:27-67).
// The reference accumulates into an unordered_map keyed by the cell index while walking the points in input
// order; per cell that is an in-order fp32 running sum.  The device form keeps those sums bit-exact:
//   1. min/max reduction -> origin = floor(min * (1/dl)) * dl, NX, NY            (grid_subsampling.cpp:23-30)
//   2. cell key per point = iX + NX*iY + NX*NY*iZ                                (:51-54)
//   3. stable LSD radix sort of (key, point index)  -> points of a cell are contiguous AND in input order
//   4. head flags + exclusive scan -> cell number of every sorted element, M = number of cells
//   5. one thread per cell walks its run in order: point sum * float(1.0/count), feature sum / float(count),
//      majority label (ties -> smallest label; the reference's tie order is unspecified, :100-101)
// Rows come out in ascending key order.  HBM-bound integer/byte work; sort and scan are sortscan.hip's (eight key bits per
// pass, only the passes the largest cell key needs).
#include "common.h"
#include "sortscan.h"

namespace ps {

__device__ __forceinline__ unsigned f2ord(float f)
{
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ __forceinline__ float ord2f(unsigned u)
{
    u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}

__global__ __launch_bounds__(256) void minmax_kernel(const float* __restrict__ pts, size_t n, unsigned* __restrict__ mm /* min[3], max[3] */)
{
    unsigned lo[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, hi[3] = {0u, 0u, 0u};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        for (int a = 0; a < 3; ++a) {
            const unsigned v = f2ord(pts[3 * i + a]);
            lo[a] = min(lo[a], v);
            hi[a] = max(hi[a], v);
        }
    for (int a = 0; a < 3; ++a) {
        for (int o = 32; o > 0; o >>= 1) {
            lo[a] = min(lo[a], (unsigned)__shfl_xor((int)lo[a], o));
            hi[a] = max(hi[a], (unsigned)__shfl_xor((int)hi[a], o));
        }
        if ((threadIdx.x & 63) == 0) {
            atomicMin(&mm[a], lo[a]);
            atomicMax(&mm[3 + a], hi[a]);
        }
    }
}

struct GridGeom {
    float org[3];
    float dl;
    unsigned long long NX, NY;
};

__global__ __launch_bounds__(256) void cell_key_kernel(const float* __restrict__ pts, size_t n, GridGeom g, unsigned long long* __restrict__ keys,
                                                       unsigned* __restrict__ order)
{
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long iX = (unsigned long long)floorf(__fdiv_rn(__fsub_rn(pts[3 * i], g.org[0]), g.dl));
    const unsigned long long iY = (unsigned long long)floorf(__fdiv_rn(__fsub_rn(pts[3 * i + 1], g.org[1]), g.dl));
    const unsigned long long iZ = (unsigned long long)floorf(__fdiv_rn(__fsub_rn(pts[3 * i + 2], g.org[2]), g.dl));
    keys[i] = iX + g.NX * iY + g.NX * g.NY * iZ;
    order[i] = (unsigned)i;
}

__global__ __launch_bounds__(256) void head_flag_kernel(const unsigned long long* __restrict__ keys, size_t n, unsigned* __restrict__ flag)
{
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    flag[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}

__global__ __launch_bounds__(256) void seg_start_kernel(const unsigned* __restrict__ flag, const unsigned* __restrict__ cell, size_t n,
                                                        unsigned* __restrict__ start)
{
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (flag[i]) start[cell[i]] = (unsigned)i;
}

__global__ __launch_bounds__(128) void cell_reduce_kernel(const float* __restrict__ pts, const float* __restrict__ feats, const int32_t* __restrict__ cls,
                                                          const unsigned* __restrict__ order, const unsigned* __restrict__ start, unsigned M, size_t n,
                                                          int fdim, int ldim, float* __restrict__ out_pts, float* __restrict__ out_feats,
                                                          int32_t* __restrict__ out_cls)
{
    const unsigned m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const size_t s = start[m], e = (m + 1 < M) ? start[m + 1] : n;
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (size_t j = s; j < e; ++j) {
        const size_t p = order[j];
        sx = __fadd_rn(sx, pts[3 * p]);
        sy = __fadd_rn(sy, pts[3 * p + 1]);
        sz = __fadd_rn(sz, pts[3 * p + 2]);
    }
    const int count = (int)(e - s);
    const float w = (float)(1.0 / (double)count);  // point * (1.0 / count): the double quotient converts to const float&
    out_pts[3 * (size_t)m] = __fmul_rn(sx, w);
    out_pts[3 * (size_t)m + 1] = __fmul_rn(sy, w);
    out_pts[3 * (size_t)m + 2] = __fmul_rn(sz, w);
    if (out_feats) {
        const float fc = (float)count;
        for (int c = 0; c < fdim; ++c) {
            float acc = 0.f;
            for (size_t j = s; j < e; ++j) acc = __fadd_rn(acc, feats[(size_t)order[j] * fdim + c]);
            out_feats[(size_t)m * fdim + c] = __fdiv_rn(acc, fc);
        }
    }
    if (out_cls) {
        for (int l = 0; l < ldim; ++l) {
            int best = 0, best_c = 0;
            for (size_t j = s; j < e; ++j) {
                const int v = cls[(size_t)order[j] * ldim + l];
                int cnt = 0;
                for (size_t k = s; k < e; ++k) cnt += (cls[(size_t)order[k] * ldim + l] == v);
                if (cnt > best_c || (cnt == best_c && v < best)) {
                    best_c = cnt;
                    best = v;
                }
            }
            out_cls[(size_t)m * ldim + l] = best;
        }
    }
}

}  // namespace ps

using namespace ps;

// dev: every pointer but M_out is DEVICE memory (inputs read in place, outputs written by the reduction kernel itself: no staging copy);
// capacity: rows the output buffers hold (dev only; the host form sizes its buffers from the count call)
static int grid_subsample_impl(ps_context* c, const float* points, int64_t n, const float* features, int64_t fdim, const int32_t* classes,
                               int64_t ldim, float sampleDl, int64_t* M_out, float* out_points, float* out_features, int32_t* out_classes,
                               bool dev, int64_t capacity)
{
    PS_CHECK(c && points && M_out, "ps_grid_subsample: NULL argument");
    PS_CHECK(n >= 1 && n < (1ll << 32), "ps_grid_subsample: n out of range");
    PS_CHECK(sampleDl > 0.f, "ps_grid_subsample: sampleDl must be positive");
    PS_CHECK(fdim >= 0 && ldim >= 0, "ps_grid_subsample: negative dims");
    if (!features) fdim = 0;
    if (!classes) ldim = 0;
    PS_HIP(hipSetDevice(c->device));
    hipStream_t st = c->stream;

    // workspace layout
    const size_t sort_words = sort_workspace_words((size_t)n), scan_words = scan_workspace_words((size_t)n);
    Arena& A = c->knn_arena;  // shares the KNN workspace (the two ops never overlap on one context)
    float *d_pts = nullptr, *d_feat = nullptr, *o_pts = nullptr, *o_feat = nullptr;
    int32_t *d_cls = nullptr, *o_cls = nullptr;
    unsigned long long *k0 = nullptr, *k1 = nullptr;
    unsigned *v0 = nullptr, *v1 = nullptr, *flag = nullptr, *cell = nullptr, *start = nullptr, *mm = nullptr;
    unsigned *tmp1 = nullptr, *tmp2 = nullptr;
    for (int pass = 0; pass < 2; ++pass) {
        A.begin(pass == 0);
        d_pts = A.take<float>(dev ? 1 : 3 * (size_t)n);
        d_feat = A.take<float>(dev ? 1 : (size_t)n * fdim + 1);
        d_cls = A.take<int32_t>(dev ? 1 : (size_t)n * ldim + 1);
        k0 = A.take<unsigned long long>(n);
        k1 = A.take<unsigned long long>(n);
        v0 = A.take<unsigned>(n);
        v1 = A.take<unsigned>(n);
        flag = A.take<unsigned>(n);
        cell = A.take<unsigned>(n);
        start = A.take<unsigned>(n + 1);
        mm = A.take<unsigned>(8);
        tmp1 = A.take<unsigned>(sort_words);
        tmp2 = A.take<unsigned>(scan_words);
        o_pts = A.take<float>(dev ? 1 : 3 * (size_t)n);
        o_feat = A.take<float>(dev ? 1 : (size_t)n * fdim + 1);
        o_cls = A.take<int32_t>(dev ? 1 : (size_t)n * ldim + 1);
        if (pass == 0) PS_TRY(A.buf.reserve(A.off));
    }
    Stage stg(c, "grid_subsample", 8);
    if (dev) {
        d_pts = const_cast<float*>(points);
        d_feat = const_cast<float*>(features);
        d_cls = const_cast<int32_t*>(classes);
        o_pts = out_points;
        o_feat = out_features;
        o_cls = out_classes;
    } else {
        PS_HIP(hipMemcpyAsync(d_pts, points, sizeof(float) * 3 * (size_t)n, hipMemcpyHostToDevice, st));
        if (fdim) PS_HIP(hipMemcpyAsync(d_feat, features, sizeof(float) * (size_t)n * fdim, hipMemcpyHostToDevice, st));
        if (ldim) PS_HIP(hipMemcpyAsync(d_cls, classes, sizeof(int32_t) * (size_t)n * ldim, hipMemcpyHostToDevice, st));
    }
    const unsigned init[8] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0u, 0u};
    PS_HIP(hipMemcpyAsync(mm, init, sizeof init, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(minmax_kernel, dim3(std::min(ceil_div(n, 256), 1024)), dim3(256), 0, st, d_pts, (size_t)n, mm);
    unsigned h_mm[8];
    PS_HIP(hipMemcpyAsync(h_mm, mm, sizeof h_mm, hipMemcpyDeviceToHost, st));
    PS_HIP(hipStreamSynchronize(st));
    GridGeom g;
    g.dl = sampleDl;
    float mx[3];
    {
        // scalar geometry exactly as grid_subsampling.cpp:23-30 (fp32, one rounding per operation)
        volatile float inv = 1 / sampleDl;
        for (int a = 0; a < 3; ++a) {
            volatile float t = ord2f(h_mm[a]) * inv;
            volatile float fl = floorf(t);
            volatile float o = fl * sampleDl;
            g.org[a] = o;
            mx[a] = ord2f(h_mm[3 + a]);
        }
        volatile float ex = (mx[0] - g.org[0]) / sampleDl, ey = (mx[1] - g.org[1]) / sampleDl;
        g.NX = (unsigned long long)floorf(ex) + 1;
        g.NY = (unsigned long long)floorf(ey) + 1;
    }
    int key_bits = 64;  // the largest key any point can get: (NX - 1) + NX (NY - 1) + NX NY iZmax
    {
        volatile float ez = (mx[2] - g.org[2]) / sampleDl;
        const long double cells = (long double)g.NX * (long double)g.NY * ((long double)floorf(ez) + 1.0L);
        if (cells < 18446744073709551615.0L) {
            unsigned long long top = (unsigned long long)cells;  // keys are < cells
            key_bits = 1;
            while (key_bits < 64 && (top >> key_bits) != 0) ++key_bits;
        }
    }
    const dim3 grid(ceil_div(n, 256)), blk(256);
    hipLaunchKernelGGL(cell_key_kernel, grid, blk, 0, st, d_pts, (size_t)n, g, k0, v0);
    if (radix_sort_pairs_u64(st, k0, k1, v0, v1, (size_t)n, key_bits, tmp1) == 0) {
        std::swap(k0, k1);
        std::swap(v0, v1);
    }
    hipLaunchKernelGGL(head_flag_kernel, grid, blk, 0, st, k1, (size_t)n, flag);
    exclusive_scan_u32(st, flag, cell, (size_t)n, tmp2);
    PS_HIP(hipGetLastError());
    hipLaunchKernelGGL(seg_start_kernel, grid, blk, 0, st, flag, cell, (size_t)n, start);
    unsigned last_cell = 0, last_flag = 0;
    PS_HIP(hipMemcpyAsync(&last_cell, cell + (n - 1), 4, hipMemcpyDeviceToHost, st));
    PS_HIP(hipMemcpyAsync(&last_flag, flag + (n - 1), 4, hipMemcpyDeviceToHost, st));
    PS_HIP(hipStreamSynchronize(st));
    // cell[] is the EXCLUSIVE scan of the head flags: the exact cell number at head positions (the only place it
    // is read); the number of cells is the inclusive total
    const unsigned M = last_cell + last_flag;
    *M_out = M;
    if (!out_points) return PS_OK;
    if (dev) PS_CHECK((int64_t)M <= capacity, "ps_grid_subsample_dev: the output buffers hold %lld rows, the sub-cloud has %u", (long long)capacity, M);
    hipLaunchKernelGGL(cell_reduce_kernel, dim3(ceil_div(M, 128)), dim3(128), 0, st, d_pts, d_feat, d_cls, v1, start, M, (size_t)n, (int)fdim,
                       (int)ldim, o_pts, (fdim && out_features) ? o_feat : nullptr, (ldim && out_classes) ? o_cls : nullptr);
    PS_HIP(hipGetLastError());
    if (dev) return PS_OK;  // (stream-ordered: the caller's next kernel on this stream reads the rows)
    PS_HIP(hipMemcpyAsync(out_points, o_pts, sizeof(float) * 3 * (size_t)M, hipMemcpyDeviceToHost, st));
    if (fdim && out_features) PS_HIP(hipMemcpyAsync(out_features, o_feat, sizeof(float) * (size_t)M * fdim, hipMemcpyDeviceToHost, st));
    if (ldim && out_classes) PS_HIP(hipMemcpyAsync(out_classes, o_cls, sizeof(int32_t) * (size_t)M * ldim, hipMemcpyDeviceToHost, st));
    PS_HIP(hipStreamSynchronize(st));
    return PS_OK;
}

extern "C" int ps_grid_subsample(ps_context* c, const float* points, int64_t n, const float* features, int64_t fdim, const int32_t* classes,
                                 int64_t ldim, float sampleDl, int64_t* M_out, float* out_points, float* out_features, int32_t* out_classes)
{
    return grid_subsample_impl(c, points, n, features, fdim, classes, ldim, sampleDl, M_out, out_points, out_features, out_classes, false, 0);
}

extern "C" int ps_grid_subsample_dev(ps_context* c, const float* points, int64_t n, const float* features, int64_t fdim, const int32_t* classes,
                                     int64_t ldim, float sampleDl, int64_t capacity, int64_t* M_out, float* out_points, float* out_features,
                                     int32_t* out_classes)
{
    PS_CHECK(out_points && capacity >= 1, "ps_grid_subsample_dev: out_points is NULL or capacity < 1 (one call: size the buffers for n rows, or for a known bound)");
    return grid_subsample_impl(c, points, n, features, fdim, classes, ldim, sampleDl, M_out, out_points, out_features, out_classes, true, capacity);
}
