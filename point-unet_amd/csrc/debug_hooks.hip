// debug_hooks.hip -- the TEST-ONLY library libpointseg_debug.so (declarations: debug_hooks.h): doors used by the test-suite to
// exercise the product's OWN host logic without a GPU -- the kd-tree construction rules (kdtree_host.hip) and the per-query
// search routine that the HIP kernel instantiates (kdtree.h, __host__ __device__) -- and to read a device-built tree back.
// Built next to the product library and linked against it; nothing in the product library or the Python package refers to it
// (tests/conftest.py binds it) -- it is not a CPU fallback.  The two pure-host doors (ps_debug_knn_host, ps_debug_kdtree_host) live
// in debug_host.hip, which is also built on its own under the host sanitizers (`make asan-host`).
#include "common.h"

#include <cstring>
#include <vector>
#include "kdtree_build.h"
#include "kdtree_host.h"
#include "rowgemm.h"
#include "attpool.h"

using namespace ps;

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

// one dense layer through gemm32b.hip / gemm32.hip with freshly packed weight images (the product packs them once per network)
extern "C" int ps_debug_gemm32(ps_context* c, int split_bf16, const float* x1, int ld1, int c1, const int32_t* g1, const float* x2, int ld2, int c2,
                               const int32_t* g2, int gm, int gn, const float* W, const float* bias, int64_t R, int cout, int leaky, float* y, int ldy)
{
    PS_CHECK(c && x1 && W && bias && y && R >= 0 && c1 > 0 && c2 >= 0 && cout > 0, "ps_debug_gemm32: bad argument");
    PS_HIP(hipSetDevice(c->device));
    const int cin = c1 + c2;
    PS_CHECK(cin % 16 == 0 && cout % 32 == 0, "ps_debug_gemm32: cin %% 16 and cout %% 32 must be 0");
    std::vector<float> img((size_t)cin * cout * (split_bf16 ? 3 : 2) / 2);
    if (split_bf16) pack_p32b(W, cin, cout, reinterpret_cast<uint16_t*>(img.data()));
    else pack_p32(W, cin, cout, img.data());
    float *d_img = nullptr, *d_bias = nullptr;
    PS_HIP(hipMalloc(reinterpret_cast<void**>(&d_img), img.size() * sizeof(float)));
    PS_HIP(hipMalloc(reinterpret_cast<void**>(&d_bias), sizeof(float) * (size_t)cout));
    PS_HIP(hipMemcpyAsync(d_img, img.data(), img.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
    PS_HIP(hipMemcpyAsync(d_bias, bias, sizeof(float) * (size_t)cout, hipMemcpyHostToDevice, c->stream));
    PackedLinear L;
    L.cin = cin; L.cout = cout; L.leaky = leaky; L.bias = d_bias;
    if (split_bf16) L.w32b = d_img; else L.w32 = d_img;
    RowSrc s1, s2;
    s1.x = x1; s1.ld = ld1; s1.c = c1; s1.gather = g1; s1.gm = g1 ? gm : 0; s1.gn = g1 ? gn : 0;
    s2.x = x2; s2.ld = ld2; s2.c = c2; s2.gather = g2; s2.gm = g2 ? gm : 0; s2.gn = g2 ? gn : 0;
    int rc = PS_EINVAL;
    if (split_bf16 ? gemm32b_fits(L, s1, s2, R, ldy) : gemm32_fits(L, s1, s2, R, ldy))
        rc = split_bf16 ? gemm32b(c, L, s1, s2, R, y, ldy) : gemm32(c, L, s1, s2, R, y, ldy);
    else
        set_error("ps_debug_gemm32: the shape does not fit the kernel");
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(d_img);
    (void)hipFree(d_bias);
    return rc;
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
