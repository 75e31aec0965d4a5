// kdtree_build.h -- a set of kd-trees carved from the context's workspace and built in one go.
#pragma once

#include <vector>

#include "common.h"
#include "kdtree.h"

namespace ps {

struct TreeSetPlan {
    // inputs
    std::vector<int32_t> n;          // points per tree
    std::vector<const float*> src;   // device pointer to that tree's [n,3] fp32 rows
    std::vector<float*> copy_dst;    // optional, per tree: device [n,3] buffer that also receives the tree's source rows (may be short / empty)
    int extra_jobs = 0;              // additional KnnJob slots wanted in d_jobs (beyond one per tree)
    // carved device storage
    std::vector<int4*> d_nodes;      // [2n] per tree
    std::vector<int4*> d_fat;        // [3 * 2n] per tree: every inner node with its children's records (TreeView::fat)
    std::vector<float4*> d_pts;      // [n]  per tree (vind order, w = original index)
    TreeMeta* d_meta = nullptr;      // [trees]
    // ONE contiguous "head" block that ONE host-to-device copy initialises (round 6: it was an upload of the tree table, a kernel zeroing
    // the counters and priming the bounding-box accumulators, and a second upload of the job table -- three dependent stream operations):
    //   [ KnnJob table | 16 status words (zero) | BuildTree table | bounding-box accumulators (primed) | queue counters (zero) | level counters (zero) ]
    char* d_head = nullptr;
    size_t head_bytes = 0, off_flags = 0, off_trees = 0, off_bbox = 0, off_cnt = 0, off_hcnt = 0;
    void* d_jobs = nullptr;          // = d_head: KnnJob table (trees + extra_jobs entries of 128 B)
    int32_t* d_flags = nullptr;      // [16] error flags, zero after the head copy
    void* d_scratch = nullptr;       // builder scratch (everything that needs no initial contents)
    size_t scratch_bytes = 0;
    int launches = 0;                // kernels launched by build_trees (for the stage timer)
    std::vector<char> host_blob;     // host image of the head block; the caller writes its job table at offset 0 BEFORE build_trees
    char* host_jobs() { return host_blob.data(); }

    void add(int32_t count)
    {
        n.push_back(count);
        src.push_back(nullptr);
        d_nodes.push_back(nullptr);
        d_fat.push_back(nullptr);
        d_pts.push_back(nullptr);
    }
    size_t total_points() const
    {
        size_t t = 0;
        for (int32_t v : n) t += (size_t)v;
        return t;
    }
    void carve(Arena& a);
    TreeView view(int i) const
    {
        TreeView v;
        v.nodes = d_nodes[i];
        v.fat = d_fat[i];
        v.pts = d_pts[i];
        v.meta = d_meta + i;
        v.n = n[i];
        return v;
    }
};

// Builds every tree of the plan on the context's stream WITHOUT synchronising.  The build always completes on the device
// (very unbalanced clouds included: kdtree_build.hip, straggler kernel); d_flags[1] / d_flags[0] report the two degenerate
// cases that remain (builder queue overflow, tree deeper than the traversal stack) -- even then every search writes valid
// point indices, so whatever is enqueued behind the build never reads out of bounds.
int build_trees(ps_context* c, TreeSetPlan& plan);

}  // namespace ps
