// kdtree_host.hip -- host-side construction of the kd-tree in the device layout of kdtree.h.
//
// TEST-ONLY: linked into libpointseg_debug.so (and the host sanitizer binary), not into the product library.  The C-ABI debug
// doors ps_debug_* use it to let the CPU test-suite drive the very search routine the HIP kernel runs, and the GPU tests compare
// the device builder (kdtree_build.hip, the only builder of the product) against it array for array (tests/test_gpu_knn.py).
//
// The construction rules are nanoflann 1.2.3's (PointSegment/utils/nearest_neighbors/nanoflann.hpp:916-1043,
// 1321-1343) re-stated for an explicit work stack and the id scheme of kdtree.h.
#include "kdtree_host.h"

#include <algorithm>
#include <cstring>

namespace ps {

namespace {

struct Box {
    float lo[3], hi[3];
};

struct Builder {
    const float* pts;
    int32_t n;
    std::vector<int32_t>& vind;
    std::vector<int4>& nodes;
    int depth = 0;

    float at(int32_t p, int ax) const { return pts[3 * (size_t)p + ax]; }

    void span(const int32_t* ind, int32_t count, int ax, float& mn, float& mx) const
    {
        mn = mx = at(ind[0], ax);
        for (int32_t i = 1; i < count; ++i) {
            float v = at(ind[i], ax);
            if (v < mn) mn = v;
            if (v > mx) mx = v;
        }
    }

    // One Hoare sweep: elements satisfying `keep_left` end in front.  Returns the boundary.
    template <class Pred>
    int32_t sweep(int32_t* ind, int32_t from, int32_t count, Pred keep_left) const
    {
        int64_t left = from, right = (int64_t)count - 1;
        for (;;) {
            while (left <= right && keep_left(ind[left])) ++left;
            while (right && left <= right && !keep_left(ind[right])) --right;
            if (left > right || !right) break;
            std::swap(ind[left], ind[right]);
            ++left;
            --right;
        }
        return (int32_t)left;
    }

    // Returns the reference (kdtree.h) of the node built over vind[l, r); box is the incoming box and leaves as the tight box.
    int32_t build(int32_t l, int32_t r, Box& box, int level)
    {
        depth = std::max(depth, level);
        if (r - l <= kLeafMax) {
            for (int ax = 0; ax < 3; ++ax) box.lo[ax] = box.hi[ax] = at(vind[l], ax);
            for (int32_t k = l + 1; k < r; ++k)
                for (int ax = 0; ax < 3; ++ax) {
                    float v = at(vind[k], ax);
                    if (box.lo[ax] > v) box.lo[ax] = v;
                    if (box.hi[ax] < v) box.hi[ax] = v;
                }
            nodes[2 * (size_t)l] = make_int4(l, r, 0, 0);
            return leaf_ref(l, r - l);
        }
        int32_t* ind = vind.data() + l;
        const int32_t count = r - l;
        float max_span = box.hi[0] - box.lo[0];
        for (int ax = 1; ax < 3; ++ax) max_span = std::max(max_span, box.hi[ax] - box.lo[ax]);
        int cutfeat = 0;
        float max_spread = -1.f;
        const float thresh = (1 - 0.00001f) * max_span;
        for (int ax = 0; ax < 3; ++ax) {
            if (box.hi[ax] - box.lo[ax] > thresh) {
                float mn, mx;
                span(ind, count, ax, mn, mx);
                if (mx - mn > max_spread) {
                    cutfeat = ax;
                    max_spread = mx - mn;
                }
            }
        }
        float mn, mx;
        span(ind, count, cutfeat, mn, mx);
        float cut = (box.lo[cutfeat] + box.hi[cutfeat]) / 2;
        cut = cut < mn ? mn : (cut > mx ? mx : cut);
        const int ax = cutfeat;
        const int32_t lim1 = sweep(ind, 0, count, [&](int32_t p) { return at(p, ax) < cut; });
        const int32_t lim2 = sweep(ind, lim1, count, [&](int32_t p) { return at(p, ax) <= cut; });
        const int32_t half = count / 2;
        const int32_t idx = lim1 > half ? lim1 : (lim2 < half ? lim2 : half);

        Box lb = box, rb = box;
        lb.hi[ax] = cut;
        rb.lo[ax] = cut;
        const int32_t m = l + idx;
        const int32_t c1 = build(l, m, lb, level + 1);
        const int32_t c2 = build(m, r, rb, level + 1);
        nodes[2 * (size_t)m - 1] = make_int4((int)((unsigned)c1 | ((unsigned)ax << 30)), c2, as_i(lb.hi[ax]), as_i(rb.lo[ax]));
        for (int a = 0; a < 3; ++a) {
            box.lo[a] = std::min(lb.lo[a], rb.lo[a]);
            box.hi[a] = std::max(lb.hi[a], rb.hi[a]);
        }
        return 2 * m - 1;
    }
};

}  // namespace

void build_tree_host(const float* pts, int32_t n, HostTree& t)
{
    t.n = n;
    t.vind.resize(n);
    t.nodes.assign(2 * (size_t)std::max(n, 1), make_int4(0, 0, 0, 0));
    t.pts.resize(n);
    for (int32_t i = 0; i < n; ++i) t.vind[i] = i;
    t.meta.root = 0;
    t.meta.depth = 0;
    for (int ax = 0; ax < 3; ++ax) t.meta.lo[ax] = t.meta.hi[ax] = 0.f;
    if (n == 0) return;
    Box box;
    for (int ax = 0; ax < 3; ++ax) box.lo[ax] = box.hi[ax] = pts[ax];
    for (int32_t k = 1; k < n; ++k)
        for (int ax = 0; ax < 3; ++ax) {
            float v = pts[3 * (size_t)k + ax];
            if (v < box.lo[ax]) box.lo[ax] = v;
            if (v > box.hi[ax]) box.hi[ax] = v;
        }
    Builder b{pts, n, t.vind, t.nodes};
    t.meta.root = b.build(0, n, box, 0);
    t.meta.depth = b.depth;
    for (int ax = 0; ax < 3; ++ax) {
        t.meta.lo[ax] = box.lo[ax];
        t.meta.hi[ax] = box.hi[ax];
    }
    for (int32_t i = 0; i < n; ++i) {
        const int32_t p = t.vind[i];
        t.pts[i] = make_float4(pts[3 * (size_t)p], pts[3 * (size_t)p + 1], pts[3 * (size_t)p + 2], as_f(p));
    }
}

}  // namespace ps
