// volume.hip -- MR volume -> point cloud (ps_volume_to_cloud): the first half of the reference's dataset preparation,
// PointSegment/utils/dataPrepareBraTS.py:
//   itensity_normalize_one_volume (:33-49)  per modality: z-score with the mean / population std of the voxels > 0, computed
//                                           in float64 like NumPy does; zero voxels stay 0
//   convert_pc2ply (:75-89)                 every voxel where any normalised modality is non-zero becomes a point, in x-major
//                                           (x, y, z) order: xyz = index / shape (float64 division, then float32), the four
//                                           normalised modalities as float32, the label (4 -> 3 is the caller's business),
//                                           and the integer voxel index (`xyz_origin`)
// Device form: two reduction passes per modality (NumPy's std is two-pass: mean first, then mean |x - mean|^2), a flag pass,
// an exclusive scan (sortscan.hip) for the compaction offsets, one scatter.  Summation order differs from NumPy's pairwise sum, so
// the normalised values agree to float64 rounding (and are identical after the cast to float32 except at rounding ties);
// the point set, its order and the coordinates are exact.  HBM-bound streaming work; offline in the reference.
#include "common.h"

#include "sortscan.h"

namespace ps {

struct VolStats {
    double sum[4], dev2[4];
    unsigned long long count[4];
};

__device__ __forceinline__ double wave_sum(double v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// pass 1: count and sum of the voxels > 0, per modality
__global__ __launch_bounds__(256) void vol_sum_kernel(const float* __restrict__ vol, size_t nvox, VolStats* __restrict__ st)
{
    const int m = blockIdx.y;
    const float* v = vol + (size_t)m * nvox;
    double s = 0.0;
    unsigned long long c = 0;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < nvox; i += (size_t)gridDim.x * 256)
        if (v[i] > 0.f) { s += (double)v[i]; ++c; }
    s = wave_sum(s);
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor((long long)c, o);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&st->sum[m], s);
        atomicAdd(&st->count[m], c);
    }
}

// pass 2: sum of squared deviations from the mean
__global__ __launch_bounds__(256) void vol_dev_kernel(const float* __restrict__ vol, size_t nvox, VolStats* __restrict__ st)
{
    const int m = blockIdx.y;
    const float* v = vol + (size_t)m * nvox;
    const double mean = st->sum[m] / (double)st->count[m];
    double s = 0.0;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < nvox; i += (size_t)gridDim.x * 256)
        if (v[i] > 0.f) { const double d = (double)v[i] - mean; s += d * d; }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) atomicAdd(&st->dev2[m], s);
}

__device__ __forceinline__ double vol_norm(float v, double mean, double sd) { return v == 0.f ? 0.0 : ((double)v - mean) / sd; }

__global__ __launch_bounds__(256) void vol_flag_kernel(const float* __restrict__ vol, size_t nvox, const VolStats* __restrict__ st,
                                                       unsigned* __restrict__ flag)
{
    double mean[4], sd[4];
    for (int m = 0; m < 4; ++m) { mean[m] = st->sum[m] / (double)st->count[m]; sd[m] = sqrt(st->dev2[m] / (double)st->count[m]); }
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < nvox; i += (size_t)gridDim.x * 256) {
        bool any = false;
        for (int m = 0; m < 4; ++m) any |= vol_norm(vol[(size_t)m * nvox + i], mean[m], sd[m]) != 0.0;
        flag[i] = any ? 1u : 0u;
    }
}

__global__ __launch_bounds__(256) void vol_scatter_kernel(const float* __restrict__ vol, const int32_t* __restrict__ seg, size_t nvox, int X, int Y, int Z,
                                                          const VolStats* __restrict__ st, const unsigned* __restrict__ flag,
                                                          const unsigned* __restrict__ pos, float* __restrict__ xyz, float* __restrict__ colors,
                                                          int32_t* __restrict__ labels, int32_t* __restrict__ origin)
{
    double mean[4], sd[4];
    for (int m = 0; m < 4; ++m) { mean[m] = st->sum[m] / (double)st->count[m]; sd[m] = sqrt(st->dev2[m] / (double)st->count[m]); }
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < nvox; i += (size_t)gridDim.x * 256) {
        if (!flag[i]) continue;
        const size_t o = pos[i];
        const int z = (int)(i % Z), y = (int)((i / Z) % Y), x = (int)(i / ((size_t)Z * Y));
        xyz[3 * o] = (float)((double)x / (double)X);
        xyz[3 * o + 1] = (float)((double)y / (double)Y);
        xyz[3 * o + 2] = (float)((double)z / (double)Z);
        for (int m = 0; m < 4; ++m) colors[4 * o + m] = (float)vol_norm(vol[(size_t)m * nvox + i], mean[m], sd[m]);
        if (labels) labels[o] = seg ? seg[i] : 0;
        if (origin) { origin[3 * o] = x; origin[3 * o + 1] = y; origin[3 * o + 2] = z; }
    }
}

}  // namespace ps

using namespace ps;

// dev: every pointer but n_out is DEVICE memory -- the volumes are read in place and the compaction kernel writes the caller's buffers
static int volume_to_cloud_impl(ps_context* c, const float* volumes, const int32_t* seg, int64_t X, int64_t Y, int64_t Z, int64_t* n_out,
                                float* xyz, float* colors, int32_t* labels, int32_t* xyz_origin, bool dev)
{
    PS_CHECK(c && volumes && n_out, "ps_volume_to_cloud: NULL argument");
    PS_CHECK(X >= 1 && Y >= 1 && Z >= 1 && X * Y * Z < (1ll << 31), "ps_volume_to_cloud: bad volume shape");
    PS_CHECK((xyz == nullptr) == (colors == nullptr), "ps_volume_to_cloud: pass both xyz and colors, or neither (count-only call)");
    PS_HIP(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const size_t nvox = (size_t)X * Y * Z;
    const size_t scan_words = scan_workspace_words(nvox + 1);
    Arena& A = c->knn_arena;  // shares the KNN / grid workspace (these ops never overlap on one context)
    float *d_vol = nullptr, *d_xyz = nullptr, *d_col = nullptr;
    int32_t *d_seg = nullptr, *d_lab = nullptr, *d_org = nullptr;
    unsigned *flag = nullptr, *pos = nullptr;
    VolStats* stats = nullptr;
    unsigned* tmp = nullptr;
    const bool fill = xyz != nullptr;
    for (int pass = 0; pass < 2; ++pass) {
        A.begin(pass == 0);
        d_vol = A.take<float>(dev ? 1 : 4 * nvox);
        d_seg = A.take<int32_t>(seg && !dev ? nvox : 1);
        flag = A.take<unsigned>(nvox + 1);
        pos = A.take<unsigned>(nvox + 1);
        stats = A.take<VolStats>(1);
        tmp = A.take<unsigned>(scan_words);
        d_xyz = A.take<float>(fill && !dev ? 3 * nvox : 1);
        d_col = A.take<float>(fill && !dev ? 4 * nvox : 1);
        d_lab = A.take<int32_t>(fill && !dev ? nvox : 1);
        d_org = A.take<int32_t>(fill && !dev ? 3 * nvox : 1);
        if (pass == 0) PS_TRY(A.buf.reserve(A.off));
    }
    Stage stg(c, "volume_to_cloud", 6);
    if (dev) {
        d_vol = const_cast<float*>(volumes);
        d_seg = const_cast<int32_t*>(seg);
        d_xyz = xyz; d_col = colors; d_lab = labels; d_org = xyz_origin;
    } else {
        PS_HIP(hipMemcpyAsync(d_vol, volumes, sizeof(float) * 4 * nvox, hipMemcpyHostToDevice, st));
        if (seg) PS_HIP(hipMemcpyAsync(d_seg, seg, sizeof(int32_t) * nvox, hipMemcpyHostToDevice, st));
    }
    PS_HIP(hipMemsetAsync(stats, 0, sizeof(VolStats), st));
    const dim3 grid((unsigned)std::min<size_t>(ceil_div(nvox, 256), 1024), 4);
    hipLaunchKernelGGL(vol_sum_kernel, grid, dim3(256), 0, st, d_vol, nvox, stats);
    hipLaunchKernelGGL(vol_dev_kernel, grid, dim3(256), 0, st, d_vol, nvox, stats);
    hipLaunchKernelGGL(vol_flag_kernel, dim3(grid.x), dim3(256), 0, st, d_vol, nvox, stats, flag);
    PS_HIP(hipMemsetAsync(flag + nvox, 0, sizeof(unsigned), st));
    exclusive_scan_u32(st, flag, pos, nvox + 1, tmp);
    PS_HIP(hipGetLastError());
    unsigned n_pts = 0;
    VolStats h_stats;
    PS_HIP(hipMemcpyAsync(&n_pts, pos + nvox, sizeof(unsigned), hipMemcpyDeviceToHost, st));
    PS_HIP(hipMemcpyAsync(&h_stats, stats, sizeof(VolStats), hipMemcpyDeviceToHost, st));
    PS_HIP(hipStreamSynchronize(st));
    for (int m = 0; m < 4; ++m)
        PS_CHECK(h_stats.count[m] > 0, "ps_volume_to_cloud: modality %d has no voxel above zero (mean / std undefined)", m);
    if (!fill) {
        *n_out = (int64_t)n_pts;
        return PS_OK;
    }
    PS_CHECK(*n_out >= (int64_t)n_pts, "ps_volume_to_cloud: output buffers hold %lld rows, the cloud has %u", (long long)*n_out, n_pts);
    *n_out = (int64_t)n_pts;
    if (!n_pts) return PS_OK;
    hipLaunchKernelGGL(vol_scatter_kernel, dim3(grid.x), dim3(256), 0, st, d_vol, seg ? d_seg : nullptr, nvox, (int)X, (int)Y, (int)Z, stats, flag, pos,
                       d_xyz, d_col, labels ? d_lab : nullptr, xyz_origin ? d_org : nullptr);
    PS_HIP(hipGetLastError());
    if (dev) return PS_OK;  // (stream-ordered: the caller's next kernel on this stream reads the rows)
    PS_HIP(hipMemcpyAsync(xyz, d_xyz, sizeof(float) * 3 * n_pts, hipMemcpyDeviceToHost, st));
    PS_HIP(hipMemcpyAsync(colors, d_col, sizeof(float) * 4 * n_pts, hipMemcpyDeviceToHost, st));
    if (labels) PS_HIP(hipMemcpyAsync(labels, d_lab, sizeof(int32_t) * n_pts, hipMemcpyDeviceToHost, st));
    if (xyz_origin) PS_HIP(hipMemcpyAsync(xyz_origin, d_org, sizeof(int32_t) * 3 * n_pts, hipMemcpyDeviceToHost, st));
    PS_HIP(hipStreamSynchronize(st));
    return PS_OK;
}

extern "C" int ps_volume_to_cloud(ps_context* c, const float* volumes, const int32_t* seg, int64_t X, int64_t Y, int64_t Z, int64_t* n_out,
                                  float* xyz, float* colors, int32_t* labels, int32_t* xyz_origin)
{
    return volume_to_cloud_impl(c, volumes, seg, X, Y, Z, n_out, xyz, colors, labels, xyz_origin, false);
}

extern "C" int ps_volume_to_cloud_dev(ps_context* c, const float* volumes, const int32_t* seg, int64_t X, int64_t Y, int64_t Z, int64_t* n_out,
                                      float* xyz, float* colors, int32_t* labels, int32_t* xyz_origin)
{
    PS_CHECK(xyz && colors, "ps_volume_to_cloud_dev: xyz and colors are required (one call: *n holds the row capacity of the buffers, X*Y*Z at most)");
    return volume_to_cloud_impl(c, volumes, seg, X, Y, Z, n_out, xyz, colors, labels, xyz_origin, true);
}
