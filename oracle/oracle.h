/* oracle/oracle.h -- TEST INFRASTRUCTURE ONLY (CPU restatement of the reference algorithms).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load liboracle.so.
 * The product (libpointseg_hip.so) never links, loads or calls anything in this directory.
 *
 * Parity status: PINNED for KNN and grid subsampling -- the restatement is checked index-for-index
 * against the real reference C++ compiled into oracle/_ref (see oracle/Makefile `ref`,
 * tests/test_oracle_vs_ref.py) and against the .npz fixtures under tests/golden generated from it.
 * The network forward restatement (oracle/randla_oracle.py) is UNPINNED: TensorFlow 1.11 is not
 * installable here and the reference holds no golden vectors for it.
 */
#ifndef PS_ORACLE_H
#define PS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Batched exact KNN, restating cpp_knn_batch_omp (PointSegment/utils/nearest_neighbors/knn_.cxx:104-135)
 * + nanoflann 1.2.3 kd-tree (nanoflann.hpp:79-145, 313-355, 916-1061, 1216-1258, 1321-1408).
 * support  f32 [B, n_support, 3], queries f32 [B, n_queries, 3] row-major; out int64 [B, n_queries, K]
 * (slots beyond n_support keep the caller's contents, as the reference's np.zeros + partial fill does,
 * knn.pyx:93). threads<=1: serial over the batch (cpp_knn_batch); >1: OpenMP over the batch only. */
void oracle_knn_batch(const float* support, const float* queries, int64_t B, int64_t n_support,
                      int64_t n_queries, int64_t K, int64_t* out_idx, int threads);

/* Same search but OpenMP over the QUERIES of each cloud (like cpp_knn_omp, knn_.cxx:46-69);
 * used for the "all cores" CPU baseline leg. */
void oracle_knn_batch_qpar(const float* support, const float* queries, int64_t B, int64_t n_support,
                           int64_t n_queries, int64_t K, int64_t* out_idx, int threads);

/* Tree export for white-box tests of the device tree builder.
 * Builds the tree of one cloud and writes it in flat arrays (pre-order node numbering):
 *   vind      int32 [n]         permuted point indices (nanoflann `vind`)
 *   node_a    int32 [max_nodes] leaf: left  (vind range start); inner: index of child1
 *   node_b    int32 [max_nodes] leaf: right (vind range end);   inner: index of child2
 *   node_axis int32 [max_nodes] -1 for a leaf, else divfeat
 *   node_lo / node_hi  f32 [max_nodes]   divlow / divhigh (inner nodes)
 *   root_bbox f32 [6]           lo[3], hi[3]
 * returns the number of nodes, or -1 if max_nodes is too small. */
int64_t oracle_kdtree_export(const float* support, int64_t n, int32_t* vind, int32_t* node_a,
                             int32_t* node_b, int32_t* node_axis, float* node_lo, float* node_hi,
                             float* root_bbox, int64_t max_nodes);

/* Grid subsampling restating grid_subsampling() (PointSegment/utils/cpp_wrappers/cpp_subsampling/
 * grid_subsampling/grid_subsampling.cpp:5-106). Output rows are emitted in ascending cell-key order
 * (the reference's order is unordered_map iteration order; parity is after a canonical row sort).
 * Two-call protocol: call with out_* == NULL to get M; then with buffers.
 * classes int32 [n, ldim]; label ties resolve to the smallest label (fixtures avoid ties). */
int64_t oracle_grid_subsample(const float* points, int64_t n, const float* features, int64_t fdim,
                              const int32_t* classes, int64_t ldim, float sampleDl,
                              float* out_points, float* out_features, int32_t* out_classes);

#ifdef __cplusplus
}
#endif
#endif
