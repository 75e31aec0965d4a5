/* oracle/grid_oracle.c -- TEST INFRASTRUCTURE ONLY. See oracle/oracle.h for the rules of use.
 *
 * CPU restatement of grid_subsampling()
 *   PointSegment/utils/cpp_wrappers/cpp_subsampling/grid_subsampling/grid_subsampling.cpp:5-106
 *   SampledData accumulators: .../grid_subsampling/grid_subsampling.h:10-80
 *   min_point / max_point:    PointSegment/utils/cpp_wrappers/cpp_utils/cloud/cloud.cpp:27-67
 *
 * Per occupied voxel: barycentre of the points (fp32 running sum in input order, times float(1.0/count)),
 * mean of the features (fp32 running sum in input order, divided by float(count)), majority label.
 * Rows come out in ascending cell-key order (the reference emits unordered_map iteration order,
 * grid_subsampling.cpp:85; parity is checked after a canonical row sort).
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    uint64_t key;
    int64_t i;
} kv_t;

static int kv_cmp(const void* a, const void* b)
{
    const kv_t* x = (const kv_t*)a;
    const kv_t* y = (const kv_t*)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->i < y->i ? -1 : (x->i > y->i); /* stable: input order inside a cell */
}

static int int_cmp(const void* a, const void* b)
{
    int32_t x = *(const int32_t*)a, y = *(const int32_t*)b;
    return x < y ? -1 : (x > y);
}

int64_t oracle_grid_subsample(const float* points, int64_t n, const float* features, int64_t fdim,
                              const int32_t* classes, int64_t ldim, float sampleDl, float* out_points,
                              float* out_features, int32_t* out_classes)
{
    if (n <= 0) return 0;
    float mn[3], mx[3], org[3];
    for (int a = 0; a < 3; ++a) mn[a] = mx[a] = points[a];
    for (int64_t i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a) {
            float v = points[3 * i + a];
            if (v < mn[a]) mn[a] = v;
            if (v > mx[a]) mx[a] = v;
        }
    const float inv = 1 / sampleDl; /* grid_subsampling.cpp:26 */
    for (int a = 0; a < 3; ++a) org[a] = floorf(mn[a] * inv) * sampleDl;
    const uint64_t NX = (uint64_t)floorf((mx[0] - org[0]) / sampleDl) + 1;
    const uint64_t NY = (uint64_t)floorf((mx[1] - org[1]) / sampleDl) + 1;

    kv_t* kv = (kv_t*)malloc((size_t)n * sizeof(kv_t));
    for (int64_t i = 0; i < n; ++i) {
        uint64_t iX = (uint64_t)floorf((points[3 * i + 0] - org[0]) / sampleDl);
        uint64_t iY = (uint64_t)floorf((points[3 * i + 1] - org[1]) / sampleDl);
        uint64_t iZ = (uint64_t)floorf((points[3 * i + 2] - org[2]) / sampleDl);
        kv[i].key = iX + NX * iY + NX * NY * iZ;
        kv[i].i = i;
    }
    qsort(kv, (size_t)n, sizeof(kv_t), kv_cmp);

    int64_t M = 0;
    int32_t* lab = (int32_t*)malloc((size_t)n * sizeof(int32_t));
    float* fsum = (float*)malloc((size_t)(fdim > 0 ? fdim : 1) * sizeof(float));
    for (int64_t s = 0; s < n;) {
        int64_t e = s;
        while (e < n && kv[e].key == kv[s].key) ++e;
        if (out_points) {
            float sx = 0.f, sy = 0.f, sz = 0.f;
            for (int64_t j = s; j < e; ++j) {
                const float* p = points + 3 * kv[j].i;
                sx += p[0];
                sy += p[1];
                sz += p[2];
            }
            int count = (int)(e - s);
            float w = (float)(1.0 / count); /* point * (1.0 / count): double -> const float& */
            out_points[3 * M + 0] = sx * w;
            out_points[3 * M + 1] = sy * w;
            out_points[3 * M + 2] = sz * w;
            if (out_features && fdim > 0 && features) {
                for (int64_t c = 0; c < fdim; ++c) fsum[c] = 0.f;
                for (int64_t j = s; j < e; ++j)
                    for (int64_t c = 0; c < fdim; ++c) fsum[c] += features[kv[j].i * fdim + c];
                float fc = (float)count;
                for (int64_t c = 0; c < fdim; ++c) out_features[M * fdim + c] = fsum[c] / fc;
            }
            if (out_classes && ldim > 0 && classes) {
                for (int64_t l = 0; l < ldim; ++l) {
                    int64_t m = e - s;
                    for (int64_t j = 0; j < m; ++j) lab[j] = classes[kv[s + j].i * ldim + l];
                    qsort(lab, (size_t)m, sizeof(int32_t), int_cmp);
                    int32_t best = lab[0];
                    int64_t best_c = 0;
                    for (int64_t j = 0; j < m;) {
                        int64_t k = j;
                        while (k < m && lab[k] == lab[j]) ++k;
                        if (k - j > best_c) {
                            best_c = k - j;
                            best = lab[j];
                        }
                        j = k;
                    }
                    out_classes[M * ldim + l] = best;
                }
            }
        }
        ++M;
        s = e;
    }
    free(kv);
    free(lab);
    free(fsum);
    return M;
}
