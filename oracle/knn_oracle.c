/* oracle/knn_oracle.c -- TEST INFRASTRUCTURE ONLY. See oracle/oracle.h for the rules of use.
 *
 * A scalar CPU restatement, written from the observed behaviour, of the reference's batched exact KNN:
 *   cpp_knn_batch_omp           PointSegment/utils/nearest_neighbors/knn_.cxx:104-135
 *   kd-tree build               PointSegment/utils/nearest_neighbors/nanoflann.hpp:916-1043, 1216-1226, 1321-1343
 *   kd-tree search              nanoflann.hpp:1244-1258, 1045-1061, 1351-1408
 *   result set (stable insert)  nanoflann.hpp:79-145
 *   metric                      nanoflann.hpp:313-355 (dim 3: only the tail loop runs)
 *
 * Everything that decides WHICH index wins an equal-distance tie is reproduced: tree shape, the
 * permutation the two-pass partition leaves in `vind`, child visiting order, fp32 pruning arithmetic,
 * first-visited-wins insertion. All arithmetic is fp32 with separately rounded products and sums
 * (build with -ffp-contract=off).
 */
#include "oracle.h"

#include <float.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define LEAF_MAX 10 /* KDTreeTableAdaptor(npts, dim, points, 10), knn_.cxx:116 */

typedef struct {
    int32_t a;    /* leaf: vind range start; inner: child1 */
    int32_t b;    /* leaf: vind range end;   inner: child2 */
    int32_t axis; /* -1 leaf */
    float lo;     /* divlow  */
    float hi;     /* divhigh */
} node_t;

typedef struct {
    const float* pts; /* [n,3] */
    int64_t n;
    int32_t* vind;
    node_t* nodes;
    int64_t n_nodes, cap_nodes;
    float root_lo[3], root_hi[3];
} tree_t;

static int32_t new_node(tree_t* t)
{
    if (t->n_nodes == t->cap_nodes) {
        t->cap_nodes = t->cap_nodes ? 2 * t->cap_nodes : 1024;
        t->nodes = (node_t*)realloc(t->nodes, (size_t)t->cap_nodes * sizeof(node_t));
    }
    return (int32_t)t->n_nodes++;
}

static inline float coord(const tree_t* t, int32_t p, int ax) { return t->pts[3 * (int64_t)p + ax]; }

/* nanoflann.hpp:897-907 */
static void min_max(const tree_t* t, const int32_t* ind, int64_t count, int ax, float* mn, float* mx)
{
    float lo = coord(t, ind[0], ax), hi = lo;
    for (int64_t i = 1; i < count; ++i) {
        float v = coord(t, ind[i], ax);
        if (v < lo) lo = v;
        if (v > hi) hi = v;
    }
    *mn = lo;
    *mx = hi;
}

/* nanoflann.hpp:1016-1043: two Hoare passes, `< cut | == cut | > cut`. */
static void plane_split(const tree_t* t, int32_t* ind, int64_t count, int ax, float cut, int64_t* lim1,
                        int64_t* lim2)
{
    int64_t left = 0, right = count - 1;
    for (;;) {
        while (left <= right && coord(t, ind[left], ax) < cut) ++left;
        while (right && left <= right && coord(t, ind[right], ax) >= cut) --right;
        if (left > right || !right) break;
        int32_t tmp = ind[left];
        ind[left] = ind[right];
        ind[right] = tmp;
        ++left;
        --right;
    }
    *lim1 = left;
    right = count - 1;
    for (;;) {
        while (left <= right && coord(t, ind[left], ax) <= cut) ++left;
        while (right && left <= right && coord(t, ind[right], ax) > cut) --right;
        if (left > right || !right) break;
        int32_t tmp = ind[left];
        ind[left] = ind[right];
        ind[right] = tmp;
        ++left;
        --right;
    }
    *lim2 = left;
}

/* nanoflann.hpp:916-1005. bbox_lo/hi are in/out (incoming box in, tight union out). */
static int32_t divide(tree_t* t, int64_t left, int64_t right, float* lo, float* hi)
{
    int32_t id = new_node(t);
    if (right - left <= LEAF_MAX) {
        node_t nd = {(int32_t)left, (int32_t)right, -1, 0.f, 0.f};
        t->nodes[id] = nd;
        for (int ax = 0; ax < 3; ++ax) lo[ax] = hi[ax] = coord(t, t->vind[left], ax);
        for (int64_t k = left + 1; k < right; ++k)
            for (int ax = 0; ax < 3; ++ax) {
                float v = coord(t, t->vind[k], ax);
                if (lo[ax] > v) lo[ax] = v;
                if (hi[ax] < v) hi[ax] = v;
            }
        return id;
    }
    int32_t* ind = t->vind + left;
    int64_t count = right - left;
    /* middleSplit_ */
    const float EPS = 0.00001f;
    float max_span = hi[0] - lo[0];
    for (int ax = 1; ax < 3; ++ax) {
        float span = hi[ax] - lo[ax];
        if (span > max_span) max_span = span;
    }
    float max_spread = -1.f;
    int cutfeat = 0;
    for (int ax = 0; ax < 3; ++ax) {
        float span = hi[ax] - lo[ax];
        if (span > (1 - EPS) * max_span) {
            float mn, mx;
            min_max(t, ind, count, ax, &mn, &mx);
            float spread = mx - mn;
            if (spread > max_spread) {
                cutfeat = ax;
                max_spread = spread;
            }
        }
    }
    float split_val = (lo[cutfeat] + hi[cutfeat]) / 2;
    float mn, mx, cutval;
    min_max(t, ind, count, cutfeat, &mn, &mx);
    if (split_val < mn) cutval = mn;
    else if (split_val > mx) cutval = mx;
    else cutval = split_val;
    int64_t lim1, lim2, idx;
    plane_split(t, ind, count, cutfeat, cutval, &lim1, &lim2);
    if (lim1 > count / 2) idx = lim1;
    else if (lim2 < count / 2) idx = lim2;
    else idx = count / 2;

    float llo[3], lhi[3], rlo[3], rhi[3];
    memcpy(llo, lo, sizeof llo);
    memcpy(lhi, hi, sizeof lhi);
    memcpy(rlo, lo, sizeof rlo);
    memcpy(rhi, hi, sizeof rhi);
    lhi[cutfeat] = cutval;
    int32_t c1 = divide(t, left, left + idx, llo, lhi);
    rlo[cutfeat] = cutval;
    int32_t c2 = divide(t, left + idx, right, rlo, rhi);
    node_t nd = {c1, c2, cutfeat, lhi[cutfeat], rlo[cutfeat]};
    t->nodes[id] = nd; /* (t->nodes may have been reallocated by the recursion: index, not pointer) */
    for (int ax = 0; ax < 3; ++ax) {
        lo[ax] = llo[ax] < rlo[ax] ? llo[ax] : rlo[ax];
        hi[ax] = lhi[ax] > rhi[ax] ? lhi[ax] : rhi[ax];
    }
    return id;
}

static void tree_build(tree_t* t, const float* pts, int64_t n)
{
    memset(t, 0, sizeof *t);
    t->pts = pts;
    t->n = n;
    t->vind = (int32_t*)malloc((size_t)(n > 0 ? n : 1) * sizeof(int32_t));
    for (int64_t i = 0; i < n; ++i) t->vind[i] = (int32_t)i;
    if (n == 0) return;
    /* computeBoundingBox, nanoflann.hpp:1321-1343 */
    for (int ax = 0; ax < 3; ++ax) t->root_lo[ax] = t->root_hi[ax] = pts[ax];
    for (int64_t k = 1; k < n; ++k)
        for (int ax = 0; ax < 3; ++ax) {
            float v = pts[3 * k + ax];
            if (v < t->root_lo[ax]) t->root_lo[ax] = v;
            if (v > t->root_hi[ax]) t->root_hi[ax] = v;
        }
    /* divideTree mutates the box it is given: root_bbox ends as the (identical) union. */
    divide(t, 0, n, t->root_lo, t->root_hi);
}

static void tree_free(tree_t* t)
{
    free(t->vind);
    free(t->nodes);
}

/* --- search ---------------------------------------------------------------------------------- */

typedef struct {
    float* dist;  /* [K] ascending */
    int64_t* idx; /* [K] */
    int64_t cap, count;
} rset_t;

/* nanoflann.hpp:115-139: shift while stored > d (strict), so the first-visited of equals stays first. */
static inline void rset_add(rset_t* r, float d, int64_t index)
{
    int64_t i;
    for (i = r->count; i > 0; --i) {
        if (r->dist[i - 1] > d) {
            if (i < r->cap) {
                r->dist[i] = r->dist[i - 1];
                r->idx[i] = r->idx[i - 1];
            }
        } else
            break;
    }
    if (i < r->cap) {
        r->dist[i] = d;
        r->idx[i] = index;
    }
    if (r->count < r->cap) r->count++;
}

static void search(const tree_t* t, const float* q, int32_t node, float mindistsq, float* dists, rset_t* r)
{
    const node_t* nd = &t->nodes[node];
    if (nd->axis < 0) {
        float worst = r->dist[r->cap - 1]; /* sampled once per leaf, nanoflann.hpp:1357 */
        for (int32_t i = nd->a; i < nd->b; ++i) {
            int32_t p = t->vind[i];
            float d = 0.f; /* nanoflann.hpp:343-346: result += diff*diff, three times */
            for (int ax = 0; ax < 3; ++ax) {
                float diff = q[ax] - coord(t, p, ax);
                d += diff * diff;
            }
            if (d < worst) rset_add(r, d, p);
        }
        return;
    }
    int ax = nd->axis;
    float val = q[ax];
    float diff1 = val - nd->lo, diff2 = val - nd->hi;
    int32_t best, other;
    float cut;
    if (diff1 + diff2 < 0) {
        best = nd->a;
        other = nd->b;
        cut = (val - nd->hi) * (val - nd->hi);
    } else {
        best = nd->b;
        other = nd->a;
        cut = (val - nd->lo) * (val - nd->lo);
    }
    search(t, q, best, mindistsq, dists, r);
    float dst = dists[ax];
    mindistsq = mindistsq + cut - dst;
    dists[ax] = cut;
    if (mindistsq * 1.0f /* epsError = 1 + eps(0) */ <= r->dist[r->cap - 1]) search(t, q, other, mindistsq, dists, r);
    dists[ax] = dst;
}

static void query_one(const tree_t* t, const float* q, int64_t K, float* dist_buf, int64_t* idx_buf,
                      int64_t* out)
{
    /* the reference reuses one out_ids vector per cloud (zero-initialised, knn_.cxx:120-121) and copies
     * all K slots after every query: slots never filled (n < K) read 0. */
    rset_t r = {dist_buf, idx_buf, K, 0};
    if (K) dist_buf[K - 1] = FLT_MAX;
    if (t->n > 0) {
        float dists[3] = {0.f, 0.f, 0.f};
        float distsq = 0.f; /* computeInitialDistances, nanoflann.hpp:1045-1061 */
        for (int ax = 0; ax < 3; ++ax) {
            if (q[ax] < t->root_lo[ax]) {
                dists[ax] = (q[ax] - t->root_lo[ax]) * (q[ax] - t->root_lo[ax]);
                distsq += dists[ax];
            }
            if (q[ax] > t->root_hi[ax]) {
                dists[ax] = (q[ax] - t->root_hi[ax]) * (q[ax] - t->root_hi[ax]);
                distsq += dists[ax];
            }
        }
        search(t, q, 0, distsq, dists, &r);
    }
    for (int64_t j = 0; j < K; ++j) out[j] = idx_buf[j];
}

static void knn_cloud(const float* pts, const float* qs, int64_t n, int64_t nq, int64_t K, int64_t* out,
                      int qthreads)
{
    tree_t t;
    tree_build(&t, pts, n);
    if (qthreads <= 1) {
        float* dist_buf = (float*)calloc((size_t)(K > 0 ? K : 1), sizeof(float));
        int64_t* idx_buf = (int64_t*)calloc((size_t)(K > 0 ? K : 1), sizeof(int64_t));
        for (int64_t i = 0; i < nq; ++i) query_one(&t, qs + 3 * i, K, dist_buf, idx_buf, out + i * K);
        free(dist_buf);
        free(idx_buf);
    } else {
#pragma omp parallel num_threads(qthreads)
        {
            float* dist_buf = (float*)calloc((size_t)(K > 0 ? K : 1), sizeof(float));
            int64_t* idx_buf = (int64_t*)calloc((size_t)(K > 0 ? K : 1), sizeof(int64_t));
#pragma omp for schedule(dynamic, 256)
            for (int64_t i = 0; i < nq; ++i) query_one(&t, qs + 3 * i, K, dist_buf, idx_buf, out + i * K);
            free(dist_buf);
            free(idx_buf);
        }
    }
    tree_free(&t);
}

void oracle_knn_batch(const float* support, const float* queries, int64_t B, int64_t n_support,
                      int64_t n_queries, int64_t K, int64_t* out_idx, int threads)
{
    if (threads < 1) threads = 1;
#pragma omp parallel for num_threads(threads) schedule(static) if (threads > 1)
    for (int64_t b = 0; b < B; ++b)
        knn_cloud(support + b * n_support * 3, queries + b * n_queries * 3, n_support, n_queries, K,
                  out_idx + b * n_queries * K, 1);
}

void oracle_knn_batch_qpar(const float* support, const float* queries, int64_t B, int64_t n_support,
                           int64_t n_queries, int64_t K, int64_t* out_idx, int threads)
{
    for (int64_t b = 0; b < B; ++b)
        knn_cloud(support + b * n_support * 3, queries + b * n_queries * 3, n_support, n_queries, K,
                  out_idx + b * n_queries * K, threads);
}

int64_t oracle_kdtree_export(const float* support, int64_t n, int32_t* vind, int32_t* node_a,
                             int32_t* node_b, int32_t* node_axis, float* node_lo, float* node_hi,
                             float* root_bbox, int64_t max_nodes)
{
    tree_t t;
    tree_build(&t, support, n);
    int64_t nn = t.n_nodes;
    if (nn > max_nodes) {
        tree_free(&t);
        return -1;
    }
    memcpy(vind, t.vind, (size_t)n * sizeof(int32_t));
    for (int64_t i = 0; i < nn; ++i) {
        node_a[i] = t.nodes[i].a;
        node_b[i] = t.nodes[i].b;
        node_axis[i] = t.nodes[i].axis;
        node_lo[i] = t.nodes[i].lo;
        node_hi[i] = t.nodes[i].hi;
    }
    for (int ax = 0; ax < 3; ++ax) {
        root_bbox[ax] = t.root_lo[ax];
        root_bbox[3 + ax] = t.root_hi[ax];
    }
    tree_free(&t);
    return nn;
}
