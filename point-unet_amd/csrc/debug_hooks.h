/* debug_hooks.h -- C declarations of the TEST-ONLY library point-unet_amd/libpointseg_debug.so (csrc/debug_hooks.hip).
 * Not part of the product ABI (include/pointseg.h) and not in the product library: the doors below let the test-suite run the
 * product's own host logic (kd-tree construction rules, the per-query search routine the HIP kernel instantiates, the MFMA weight
 * packing) without a GPU, and read a device-built tree back array for array.  The library links against libpointseg_hip.so. */
#ifndef POINTSEG_DEBUG_H
#define POINTSEG_DEBUG_H
#include "../../include/pointseg.h"
#ifdef __cplusplus
extern "C" {
#endif
/* The product's own kd-tree construction + the per-query search routine the HIP kernel instantiates, run on the
 * host.  K in {1,5,7,16,32}. */
int ps_debug_knn_host(const float* support, const float* queries, int64_t B, int64_t n_support,
                      int64_t n_queries, int64_t K, int32_t* out_idx);
/* vind i32[n], nodes i32[2n,4], pts f32[n,4], root_depth i32[2], bbox f32[6] (layout: csrc/kdtree.h). */
int ps_debug_kdtree_host(const float* support, int64_t n, int32_t* vind, int32_t* nodes, float* pts,
                         int32_t* root_depth, float* bbox);
/* The same arrays from the DEVICE builder (csrc/kdtree_build.hip); needs a GPU.  Unreached node slots are
 * unspecified: compare by walking from the root. */
int ps_debug_kdtree_device(ps_context* ctx, const float* support, int64_t n, int32_t* vind, int32_t* nodes,
                           float* pts, int32_t* root_depth, float* bbox);
/* MFMA B-fragment packing of a row-major W[cin,cout] (csrc/rowgemm.h). */
int ps_debug_pack_weights(const float* W, int cin, int cout, int ntb, float* out);
/* Three-plane bfloat16 image of a row-major W[cin,cout] for the split-bf16 attention kernels (csrc/attpool32b.hip: pack_b3):
 * out = uint16 [cout/32][cin/16][3 planes][64 lanes][8]. */
int ps_debug_pack_b3(const float* W, int cin, int cout, uint16_t* out);
/* One dense layer of the deep levels, Y = act([X1[g1] | X2[g2]] . W + b), through gemm32b.hip (split_bf16 != 0: bf16 MFMA over exact
 * three-way splits) or gemm32.hip (fp32 MFMA); needs a GPU.  x1 / x2 / g1 / g2 / y: DEVICE pointers (g* may be NULL: plain rows; gm / gn:
 * batched gather, row r reads x[(r / gm) * gn + g[r]] when gm != 0), W [c1 + c2, cout] and bias [cout]: HOST.  Returns PS_EINVAL when the
 * shape does not fit the kernel. */
int ps_debug_gemm32(ps_context* ctx, int split_bf16, const float* x1, int ld1, int c1, const int32_t* g1, const float* x2, int ld2, int c2,
                    const int32_t* g2, int gm, int gn, const float* W, const float* bias, int64_t R, int cout, int leaky, float* y, int ldy);

#ifdef __cplusplus
}
#endif
#endif /* POINTSEG_DEBUG_H */
