// kdtree_build.hip -- builds the kd-trees of a TreeSetPlan.
//
// Bring-up version: the points are copied to the host, the tree is built there by kdtree_host.hip and the
// arrays are uploaded.  (The device builder replaces this function; the layout and the contract are the same.)
#include "kdtree_build.h"

#include "kdtree_host.h"

namespace ps {

void TreeSetPlan::carve(Arena& a)
{
    const size_t T = n.size();
    for (size_t i = 0; i < T; ++i) {
        d_nodes[i] = a.take<int4>(2 * (size_t)(n[i] > 0 ? n[i] : 1));
        d_pts[i] = a.take<float4>((size_t)(n[i] > 0 ? n[i] : 1));
    }
    d_meta = a.take<TreeMeta>(T);
    d_jobs = a.take<char>(128 * (T + (size_t)extra_jobs));
    d_flags = a.take<int32_t>(16);
    scratch_bytes = 0;
    d_scratch = nullptr;
}

int build_trees(ps_context* c, TreeSetPlan& plan)
{
    const size_t T = plan.n.size();
    PS_HIP(hipMemsetAsync(plan.d_flags, 0, 16 * sizeof(int32_t), c->stream));
    std::vector<TreeMeta> metas(T);
    std::vector<float> host_pts;
    HostTree ht;
    for (size_t i = 0; i < T; ++i) {
        const int32_t n = plan.n[i];
        host_pts.resize(3 * (size_t)(n > 0 ? n : 1));
        if (n > 0) {
            PS_HIP(hipMemcpyAsync(host_pts.data(), plan.src[i], sizeof(float) * 3 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
            PS_HIP(hipStreamSynchronize(c->stream));
        }
        build_tree_host(host_pts.data(), n, ht);
        metas[i] = ht.meta;
        if (n > 0) {
            PS_HIP(hipMemcpyAsync(plan.d_nodes[i], ht.nodes.data(), sizeof(int4) * 2 * (size_t)n, hipMemcpyHostToDevice, c->stream));
            PS_HIP(hipMemcpyAsync(plan.d_pts[i], ht.pts.data(), sizeof(float4) * (size_t)n, hipMemcpyHostToDevice, c->stream));
            PS_HIP(hipStreamSynchronize(c->stream));  // ht is reused by the next tree
        }
    }
    PS_HIP(hipMemcpyAsync(plan.d_meta, metas.data(), sizeof(TreeMeta) * T, hipMemcpyHostToDevice, c->stream));
    PS_HIP(hipStreamSynchronize(c->stream));
    return PS_OK;
}

}  // namespace ps
