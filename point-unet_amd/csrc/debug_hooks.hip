// debug_hooks.hip -- the TEST-ONLY library libpointseg_debug.so (declarations: debug_hooks.h): doors used by the test-suite to
// exercise the product's OWN host logic without a GPU -- the kd-tree construction rules (kdtree_host.hip) and the per-query
// search routine that the HIP kernel instantiates (kdtree.h, __host__ __device__) -- and to read a device-built tree back.
// Built next to the product library and linked against it; nothing in the product library or the Python package refers to it
// (tests/conftest.py binds it) -- it is not a CPU fallback.
#include "common.h"
#include "kdtree_build.h"
#include "kdtree_host.h"
#include "rowgemm.h"
#include "attpool.h"

#include <cfloat>

using namespace ps;

template <int K>
static void search_all(const TreeView& t, const float* q, int64_t nq, int32_t* out)
{
    for (int64_t i = 0; i < nq; ++i) {
        float dist[K];
        int idx[K];
        for (int j = 0; j < K; ++j) {
            dist[j] = FLT_MAX;
            idx[j] = 0;
        }
        knn_search_one<K>(t, q[3 * i], q[3 * i + 1], q[3 * i + 2], dist, idx);
        for (int j = 0; j < K; ++j) out[i * K + j] = idx[j];
    }
}

extern "C" int ps_debug_knn_host(const float* support, const float* queries, int64_t B, int64_t n1, int64_t n2, int64_t K, int32_t* out)
{
    PS_CHECK(support && queries && out, "ps_debug_knn_host: NULL argument");
    HostTree ht;
    for (int64_t b = 0; b < B; ++b) {
        build_tree_host(support + b * n1 * 3, (int32_t)n1, ht);
        const TreeView v = ht.view();
        const float* q = queries + b * n2 * 3;
        int32_t* o = out + b * n2 * K;
        switch (K) {
            case 1: search_all<1>(v, q, n2, o); break;
            case 5: search_all<5>(v, q, n2, o); break;
            case 7: search_all<7>(v, q, n2, o); break;
            case 16: search_all<16>(v, q, n2, o); break;
            case 32: search_all<32>(v, q, n2, o); break;
            default: set_error("ps_debug_knn_host: K=%lld not instantiated (1,5,7,16,32)", (long long)K); return PS_EINVAL;
        }
    }
    return PS_OK;
}

extern "C" int ps_debug_kdtree_host(const float* support, int64_t n, int32_t* vind, int32_t* nodes /* [2n,4] */, float* pts /* [n,4] */,
                                    int32_t* root_depth /* [2] */, float* bbox /* [6] */)
{
    PS_CHECK(support && vind && nodes && pts && root_depth && bbox, "ps_debug_kdtree_host: NULL argument");
    HostTree ht;
    build_tree_host(support, (int32_t)n, ht);
    std::memcpy(vind, ht.vind.data(), sizeof(int32_t) * (size_t)n);
    std::memcpy(nodes, ht.nodes.data(), sizeof(int4) * 2 * (size_t)n);
    std::memcpy(pts, ht.pts.data(), sizeof(float4) * (size_t)n);
    root_depth[0] = ht.meta.root;
    root_depth[1] = ht.meta.depth;
    for (int a = 0; a < 3; ++a) {
        bbox[a] = ht.meta.lo[a];
        bbox[3 + a] = ht.meta.hi[a];
    }
    return PS_OK;
}

extern "C" int ps_debug_pack_weights(const float* W, int cin, int cout, int ntb, float* out)
{
    PS_CHECK(W && out && (ntb == 1 || ntb == 2 || ntb == 4), "ps_debug_pack_weights: bad argument");
    pack_weights(W, cin, cout, ntb, out);
    return PS_OK;
}

extern "C" int ps_debug_pack_b3(const float* W, int cin, int cout, uint16_t* out)
{
    PS_CHECK(W && out && cin > 0 && cin % 32 == 0 && cout > 0 && cout % 32 == 0, "ps_debug_pack_b3: bad argument");
    pack_b3(W, cin, cout, out);
    return PS_OK;
}

// --------------------------------------------------------------------------------------------------------
// white-box door for the GPU test-suite: build one tree on the DEVICE and copy its arrays back (same layout as
// ps_debug_kdtree_host) so the two builders can be compared array for array.
// --------------------------------------------------------------------------------------------------------
extern "C" int ps_debug_kdtree_device(ps_context* c, const float* support, int64_t n, int32_t* vind, int32_t* nodes, float* pts,
                                      int32_t* root_depth, float* bbox)
{
    PS_CHECK(c && support && vind && nodes && pts && root_depth && bbox && n >= 1, "ps_debug_kdtree_device: bad argument");
    PS_HIP(hipSetDevice(c->device));
    PS_TRY(c->stage_in.reserve(sizeof(float) * 3 * (size_t)n));
    PS_HIP(hipMemcpyAsync(c->stage_in.p, support, sizeof(float) * 3 * (size_t)n, hipMemcpyHostToDevice, c->stream));
    TreeSetPlan plan;
    plan.add((int32_t)n);
    for (int pass = 0; pass < 2; ++pass) {
        c->knn_arena.begin(pass == 0);
        plan.carve(c->knn_arena);
        if (pass == 0) PS_TRY(c->knn_arena.buf.reserve(c->knn_arena.off));
    }
    plan.src[0] = c->stage_in.as<float>();
    PS_TRY(build_trees(c, plan));
    int32_t flag[3] = {0, 0, 0};
    PS_HIP(hipMemcpyAsync(flag, plan.d_flags, 12, hipMemcpyDeviceToHost, c->stream));
    PS_HIP(hipStreamSynchronize(c->stream));
    PS_CHECK(flag[1] == 0, "ps_debug_kdtree_device: builder queue overflow");
    TreeMeta m;
    PS_HIP(hipMemcpyAsync(nodes, plan.d_nodes[0], sizeof(int4) * 2 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    PS_HIP(hipMemcpyAsync(pts, plan.d_pts[0], sizeof(float4) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    PS_HIP(hipMemcpyAsync(&m, plan.d_meta, sizeof(TreeMeta), hipMemcpyDeviceToHost, c->stream));
    PS_HIP(hipStreamSynchronize(c->stream));
    for (int64_t i = 0; i < n; ++i) std::memcpy(&vind[i], &pts[4 * i + 3], 4);
    root_depth[0] = m.root;
    root_depth[1] = m.depth;
    for (int a = 0; a < 3; ++a) {
        bbox[a] = m.lo[a];
        bbox[3 + a] = m.hi[a];
    }
    return PS_OK;
}
