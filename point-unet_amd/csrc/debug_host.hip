// debug_host.hip -- the pure-host doors of the TEST-ONLY library libpointseg_debug.so (declarations: debug_hooks.h): the product's
// own kd-tree construction rules (kdtree_host.hip) and the per-query search routine that the HIP kernel instantiates (kdtree.h,
// __host__ __device__), run on the CPU.  No HIP runtime call in this file or in kdtree_host.hip: `make asan-host` compiles exactly
// these two files host-only under -fsanitize=address,undefined together with asan_host_main.hip (tests/test_sanitizers.py).
#include "common.h"
#include "kdtree_host.h"

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
