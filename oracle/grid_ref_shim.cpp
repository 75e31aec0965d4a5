// oracle/grid_ref_shim.cpp -- TEST INFRASTRUCTURE ONLY.
//
// A C-ABI doorway into the REAL reference grid_subsampling() so that Python (ctypes) can drive it.
// The reference's own CPython wrapper (PointSegment/utils/cpp_wrappers/cpp_subsampling/wrapper.cpp:58-286)
// does not compile against numpy 2 headers, so this shim does what wrapper.cpp:202-265 does -- copy the
// inputs into std::vector, call grid_subsampling(), copy the outputs out -- and nothing else.
// The algorithm itself is compiled from the reference sources where they lie (see oracle/Makefile `ref`).
#include "grid_subsampling/grid_subsampling.h"

#include <cstring>
#include <vector>

extern "C" {

// Two-call protocol: run once, keep the result in a static, copy out on the second call.
static std::vector<PointXYZ> g_pts;
static std::vector<float> g_feat;
static std::vector<int> g_cls;

// returns M (number of occupied cells)
long ref_grid_subsample_run(const float* points, long n, const float* features, long fdim,
                            const int* classes, long ldim, float sampleDl)
{
    std::vector<PointXYZ> original_points(n);
    for (long i = 0; i < n; ++i)
        original_points[i] = PointXYZ(points[3 * i], points[3 * i + 1], points[3 * i + 2]);
    std::vector<float> original_features;
    if (features && fdim > 0) original_features.assign(features, features + n * fdim);
    std::vector<int> original_classes;
    if (classes && ldim > 0) original_classes.assign(classes, classes + n * ldim);
    g_pts.clear();
    g_feat.clear();
    g_cls.clear();
    grid_subsampling(original_points, g_pts, original_features, g_feat, original_classes, g_cls,
                     sampleDl, 0);
    return (long)g_pts.size();
}

void ref_grid_subsample_fetch(float* out_points, float* out_features, int* out_classes)
{
    for (size_t i = 0; i < g_pts.size(); ++i) {
        out_points[3 * i] = g_pts[i].x;
        out_points[3 * i + 1] = g_pts[i].y;
        out_points[3 * i + 2] = g_pts[i].z;
    }
    if (out_features && !g_feat.empty())
        std::memcpy(out_features, g_feat.data(), g_feat.size() * sizeof(float));
    if (out_classes && !g_cls.empty())
        std::memcpy(out_classes, g_cls.data(), g_cls.size() * sizeof(int));
}

}  // extern "C"
