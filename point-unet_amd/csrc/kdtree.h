// kdtree.h -- device layout of the kd-tree and the per-query search, shared by the tree builders and the
// search kernels.  The search routine is __host__ __device__ so the exact same code is exercised by the CPU
// unit tests (tests/test_host_logic.py via ps_debug_*) and by the HIP kernel.
//
// Behaviour being reproduced (so that equal-distance ties resolve to the same index as the reference):
//   nanoflann 1.2.3 as used by PointSegment/utils/nearest_neighbors/knn_.cxx:104-135
//     tree shape / vind permutation   nanoflann.hpp:916-1043
//     query descent, pruning          nanoflann.hpp:1244-1258, 1045-1061, 1351-1408
//     stable result insertion         nanoflann.hpp:115-139
//     metric (dim 3 tail loop)        nanoflann.hpp:343-346
//
// Layout (one tree = one cloud at one pyramid level, n points):
//   pts[n]    float4  points in `vind` order: (x, y, z, bit-cast original index)  -> a leaf is a contiguous,
//                     coalescable run and needs no second indirection
//   nodes[2n] int4    indexed by a deterministic node id:
//                       leaf  over vind range [l, r)            id = 2*l      (even)  {l, r, 0, 0}
//                       inner whose children meet at position m id = 2*m - 1  (odd)   {ref1 | axis<<30,
//                                                                                      ref2, divlow, divhigh}
//                     every split position belongs to exactly one inner node and every range start to exactly
//                     one leaf, so ids never collide and do not depend on scheduling order.
//                     A child REFERENCE (and TreeMeta::root) is the id plus, for a leaf, its point count in bits
//                     26..29: the search gets a leaf's range [l, l+count) from the reference it already holds
//                     and never loads the leaf record (one dependent memory round trip less per leaf visit).
#pragma once

#include <cfloat>
#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PS_HD __host__ __device__ __forceinline__
#else
#define PS_HD inline
struct int4 { int x, y, z, w; };
struct float4 { float x, y, z, w; };
#endif

#include "gmem.h"

namespace ps {

constexpr int kLeafMax = 10;   // KDTreeTableAdaptor(npts, dim, points, 10)  knn_.cxx:116
constexpr int kStackMax = 64;  // deferred far-children per query; builders report the tree depth
constexpr int kRefIdBits = 26;  // node ids < 2^26  =>  n < 2^25 points per tree
constexpr int kRefIdMask = (1 << kRefIdBits) - 1;
constexpr int kMaxTreePoints = 1 << 25;

// Produced by the builder (on the device in production): root id, depth, root bounding box.
struct TreeMeta {
    int32_t root;
    int32_t depth;
    float lo[3], hi[3];
};

struct TreeView {
    const int4* nodes;
    const float4* pts;
    const TreeMeta* meta;
    int32_t n;
    // optional "fat" image of the inner nodes, three records per node id: fat[3*id] = nodes[id], fat[3*id + 1] / [3*id + 2] = the records
    // of its left / right child where that child is an inner node.  One 48-byte fetch then decides TWO levels of a descent: the
    // search is a chain of dependent loads, and the chain gets a third shorter (55 -> 37 round trips per K = 16 self query of a
    // 180 000-point cloud).  Written by fatten_kernel (kdtree_build.hip) behind the builders; nullptr = walk `nodes`.
    const int4* fat = nullptr;
};

// fp32 arithmetic with one rounding per operation, never contracted into FMA (the reference is built without
// -march / -ffast-math: plain SSE2 mulss/addss).
#if defined(__HIP_DEVICE_COMPILE__)
PS_HD float f_mul(float a, float b) { return __fmul_rn(a, b); }
PS_HD float f_add(float a, float b) { return __fadd_rn(a, b); }
PS_HD float f_sub(float a, float b) { return __fsub_rn(a, b); }
PS_HD float as_f(int v) { return __int_as_float(v); }
PS_HD int as_i(float v) { return __float_as_int(v); }
#else
PS_HD float f_mul(float a, float b) { volatile float r = a * b; return r; }
PS_HD float f_add(float a, float b) { volatile float r = a + b; return r; }
PS_HD float f_sub(float a, float b) { volatile float r = a - b; return r; }
PS_HD float as_f(int v) { float f; __builtin_memcpy(&f, &v, 4); return f; }
PS_HD int as_i(float v) { int i; __builtin_memcpy(&i, &v, 4); return i; }
#endif

PS_HD int leaf_ref(int l, int count) { return 2 * l | (count << kRefIdBits); }

PS_HD float sq_dist(float qx, float qy, float qz, float px, float py, float pz)
{
    // ((dx*dx) + dy*dy) + dz*dz with diff = query - point   (nanoflann.hpp:343-346)
    float dx = f_sub(qx, px), dy = f_sub(qy, py), dz = f_sub(qz, pz);
    return f_add(f_add(f_mul(dx, dx), f_mul(dy, dy)), f_mul(dz, dz));
}

#if defined(__HIP_DEVICE_COMPILE__)
// Device insertion, slot J downwards, with a WAVE-UNIFORM exit: the lists are ascending, so once no lane of the wave has
// dist[J-1] > d nothing at or below slot J-1 changes for anybody -- and the predicate of slot J-1 is a compare the insertion
// needs anyway, so the test is its mask against zero and a scalar branch.  A candidate's rank is uniform over the list, and late in
// a search only a few lanes of a wave have one at all: the deepest slot any of them reaches is usually well above slot 0.  Template
// recursion instead of a loop with a break: every slot index stays a compile-time constant, the lists stay in registers.
template <int K, int J>
struct TopkInsertFrom {
    static __device__ __forceinline__ void run(float (&dist)[K], int (&idx)[K], float d, int p, bool mine)
    {
        if constexpr (J == 0) {
            idx[0] = mine ? p : idx[0];
            dist[0] = fminf(dist[0], d);
        } else {
            const bool prev = dist[J - 1] > d;
            const int ni = prev ? idx[J - 1] : p;
            idx[J] = mine ? ni : idx[J];
            dist[J] = __builtin_amdgcn_fmed3f(dist[J - 1], dist[J], d);
            if (__builtin_amdgcn_ballot_w64(prev) == 0ull) return;
            TopkInsertFrom<K, J - 1>::run(dist, idx, d, p, prev);
        }
    }
};
#endif

// Ascending list of the K best (distance, index); first-visited wins among equals (shift while stored > d).
// Written with compile-time indices only so that dist/idx stay in registers on the device.
template <int K>
PS_HD void topk_insert(float (&dist)[K], int (&idx)[K], float d, int p)
{
#if defined(__HIP_DEVICE_COMPILE__) && defined(PS_KNN_EXP_DISTONLY)
    // EXPERIMENT (wrong indices, timing only): what the kernel costs when an insertion moves distances alone
#pragma unroll
    for (int j = K - 1; j >= 1; --j) dist[j] = __builtin_amdgcn_fmed3f(dist[j - 1], dist[j], d);
    dist[0] = fminf(dist[0], d);
    idx[0] = p;
#elif defined(__HIP_DEVICE_COMPILE__) && !defined(PS_KNN_FULL_INSERT)
    TopkInsertFrom<K, K - 1>::run(dist, idx, d, p, dist[K - 1] > d);
#elif defined(__HIP_DEVICE_COMPILE__)
    // the list is ascending, so the new value of slot j is the MEDIAN of (dist[j-1], dist[j], d): dist[j] when d is not below it,
    // d when it falls between the two, dist[j-1] when both move up -- one v_med3_f32 instead of a compare and two selects; the
    // indices follow the same two predicates
#pragma unroll
    for (int j = K - 1; j >= 1; --j) {
        const bool mine = dist[j] > d, prev = dist[j - 1] > d;
        const int ni = prev ? idx[j - 1] : p;
        idx[j] = mine ? ni : idx[j];
        dist[j] = __builtin_amdgcn_fmed3f(dist[j - 1], dist[j], d);
    }
    idx[0] = dist[0] > d ? p : idx[0];
    dist[0] = fminf(dist[0], d);
#else
#pragma unroll
    for (int j = K - 1; j >= 0; --j) {
        const bool mine = dist[j] > d;                             // slot j moves or takes the newcomer
        const bool prev = (j > 0) ? (dist[j > 0 ? j - 1 : 0] > d) : false;  // the element below also moves up
        const float nd = prev ? dist[j > 0 ? j - 1 : 0] : d;
        const int ni = prev ? idx[j > 0 ? j - 1 : 0] : p;
        dist[j] = mine ? nd : dist[j];
        idx[j] = mine ? ni : idx[j];
    }
#endif
}

// Deferred far children of one query: (node id, lower bound m, per-axis offsets d0..d2).  PrivateStack keeps them in
// per-lane arrays (host build: plain locals; device: scratch); knn.hip supplies an LDS-windowed variant with the same
// interface for the search kernels.
struct PrivateStack {
    int id[kStackMax];
    float m[kStackMax], d0[kStackMax], d1[kStackMax], d2[kStackMax];
    int sp = 0;
    PS_HD bool push(int node, float mm, float a, float b, float c)
    {
        if (sp >= kStackMax) return false;
        id[sp] = node; m[sp] = mm; d0[sp] = a; d1[sp] = b; d2[sp] = c;
        ++sp;
        return true;
    }
    // most recent entry that still passes the prune test `m <= worst`; false when none is left
    PS_HD bool pop(float worst, int& node, float& mm, float& a, float& b, float& c)
    {
        while (sp > 0) {
            --sp;
            if (m[sp] <= worst) {
                node = id[sp]; mm = m[sp]; a = d0[sp]; b = d1[sp]; c = d2[sp];
                return true;
            }
        }
        return false;
    }
};

// One query against one tree.  dist/idx must be initialised by the caller (FLT_MAX / 0).
// Returns false if the deferred-node stack overflowed (tree deeper than kStackMax).
template <int K, class Stack>
PS_HD bool knn_search_one(const TreeView& t, float qx, float qy, float qz, float (&dist)[K], int (&idx)[K], Stack& st)
{
    if (t.n <= 0) return true;
    const TreeMeta mt = *t.meta;
    // computeInitialDistances, nanoflann.hpp:1045-1061
    float d0 = 0.f, d1 = 0.f, d2 = 0.f, m = 0.f;
    if (qx < mt.lo[0]) { d0 = f_mul(f_sub(qx, mt.lo[0]), f_sub(qx, mt.lo[0])); m = f_add(m, d0); }
    if (qx > mt.hi[0]) { d0 = f_mul(f_sub(qx, mt.hi[0]), f_sub(qx, mt.hi[0])); m = f_add(m, d0); }
    if (qy < mt.lo[1]) { d1 = f_mul(f_sub(qy, mt.lo[1]), f_sub(qy, mt.lo[1])); m = f_add(m, d1); }
    if (qy > mt.hi[1]) { d1 = f_mul(f_sub(qy, mt.hi[1]), f_sub(qy, mt.hi[1])); m = f_add(m, d1); }
    if (qz < mt.lo[2]) { d2 = f_mul(f_sub(qz, mt.lo[2]), f_sub(qz, mt.lo[2])); m = f_add(m, d2); }
    if (qz > mt.hi[2]) { d2 = f_mul(f_sub(qz, mt.hi[2]), f_sub(qz, mt.hi[2])); m = f_add(m, d2); }

    bool ok = true;
    int cur = mt.root;
    for (;;) {
        // ---- descend to a leaf, deferring the far children (searchLevel, nanoflann.hpp:1372-1406) ----
        // one inner node: decide the near child, defer the far one (a macro, not a lambda: a closure over the stack object and the
        // lists made the compiler keep them in memory)
#define PS_KD_VISIT(nd, left_first)                                                                                          \
    {                                                                                                                        \
        const int ax = (int)((unsigned)(nd).x >> 30);                                                                        \
        const int c1 = (nd).x & 0x3fffffff, c2 = (nd).y;                                                                     \
        const float divlow = as_f((nd).z), divhigh = as_f((nd).w);                                                           \
        const float val = ax == 0 ? qx : (ax == 1 ? qy : qz);                                                                \
        const float diff1 = f_sub(val, divlow), diff2 = f_sub(val, divhigh);                                                 \
        left_first = f_add(diff1, diff2) < 0.f;                                                                              \
        const float e = left_first ? diff2 : diff1; /* accum_dist(val, divhigh) or (val, divlow) */                          \
        const float cut = f_mul(e, e);                                                                                       \
        const float dax = ax == 0 ? d0 : (ax == 1 ? d1 : d2);                                                                \
        const float m2 = f_sub(f_add(m, cut), dax);                                                                          \
        /* The reference tests `m2 <= worstDist()` AFTER the near subtree returns; worstDist() only ever decreases, so a far \
           child that already fails now can never pass later: do not defer it. */                                            \
        if (m2 <= dist[K - 1])                                                                                               \
            ok &= st.push(left_first ? c2 : c1, m2, ax == 0 ? cut : d0, ax == 1 ? cut : d1, ax == 2 ? cut : d2);             \
        cur = left_first ? c1 : c2;                                                                                          \
    }
        if (t.fat) {
            while (cur & 1) {
                const int4* f = t.fat + 3 * (size_t)cur;
                const int4 nd = gload(f), kl = gload(f + 1), kr = gload(f + 2);  // the node and both children's records: one round trip
                bool left;
                PS_KD_VISIT(nd, left)
                if (cur & 1) {  // the near child is an inner node: its record is already here
                    int4 kid;
                    kid.x = left ? kl.x : kr.x; kid.y = left ? kl.y : kr.y; kid.z = left ? kl.z : kr.z; kid.w = left ? kl.w : kr.w;
                    bool left2;
                    PS_KD_VISIT(kid, left2)
                }
            }
        } else {
            while (cur & 1) {
                const int4 nd = gload(t.nodes + cur);
                bool left;
                PS_KD_VISIT(nd, left)
            }
        }
#undef PS_KD_VISIT
        // ---- leaf: scan its points in vind order (nanoflann.hpp:1355-1369) ----
        {
            const int lf_x = (cur & kRefIdMask) >> 1, lf_y = lf_x + (cur >> kRefIdBits);  // from the reference, no node load
#if defined(__HIP_DEVICE_COMPILE__)
            // all (<= kLeafMax) point loads are issued before the first distance is needed: one memory round trip per
            // leaf instead of one per point
            float4 pv[kLeafMax];
#ifdef PS_KNN_EXP_ONELEAFLOAD
            // EXPERIMENT (wrong results, timing only): one record load per leaf visit instead of ten -- what the vector-memory address path costs
            pv[0] = gload(t.pts + lf_x);
#pragma unroll
            for (int j = 1; j < kLeafMax; ++j) { pv[j] = pv[0]; pv[j].x += 1e-3f * j; }
#elif defined(PS_KNN_EXP_MASKED_LEAF)
            // EXPERIMENT: only the slots the leaf has -- a lane's record loads are one vector-memory lookup each, and a leaf holds ~7 of 10
#pragma unroll
            for (int j = 0; j < kLeafMax; ++j) pv[j] = lf_x + j < lf_y ? gload(t.pts + lf_x + j) : make_float4(0.f, 0.f, 0.f, 0.f);
#else
#pragma unroll
            for (int j = 0; j < kLeafMax; ++j) pv[j] = gload(t.pts + lf_x + j);  // one address, ten immediate offsets: the record
            // array is padded by kLeafMax entries (TreeSetPlan::carve), slots past the leaf's end are read and ignored below
#endif
#pragma unroll
            for (int j = 0; j < kLeafMax; ++j) {
                // A slot past the leaf's end gets d = FLT_MAX, which never beats the current worst.  The insertion then runs
                // UNPREDICATED whenever any lane of the wave has a candidate: a lane whose d is not below its worst distance has
                // no list slot with dist[s] > d, so the selects leave its list untouched -- one wave-uniform branch instead of a
                // second layer of per-lane selects around every slot of the list.
                const float d = lf_x + j < lf_y ? sq_dist(qx, qy, qz, pv[j].x, pv[j].y, pv[j].z) : FLT_MAX;
                if (__ballot(d < dist[K - 1]) != 0ull) topk_insert<K>(dist, idx, d, as_i(pv[j].w));
            }
#else
            for (int i = lf_x; i < lf_y; ++i) {
                const float4 p = t.pts[i];
                const float d = sq_dist(qx, qy, qz, p.x, p.y, p.z);
                // the reference filters on a worst distance sampled once per leaf and lets addPoint drop the
                // late-comers; both together accept exactly the points with d < current worst.
                if (d < dist[K - 1]) topk_insert<K>(dist, idx, d, as_i(p.w));
            }
#endif
        }
        // ---- resume at the most recent deferred child that still passes the prune test ----
        if (!st.pop(dist[K - 1], cur, m, d0, d1, d2)) break;
    }
    return ok;
}

template <int K>
PS_HD bool knn_search_one(const TreeView& t, float qx, float qy, float qz, float (&dist)[K], int (&idx)[K])
{
    PrivateStack st;
    return knn_search_one<K, PrivateStack>(t, qx, qy, qz, dist, idx, st);
}

}  // namespace ps
