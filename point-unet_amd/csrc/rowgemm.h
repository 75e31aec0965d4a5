// rowgemm.h -- per-point dense layer  Y = act([X1[g1] | X2[g2]] . W + b)  on fp32 MFMA (v_mfma_f32_16x16x4_f32).
//
// This is the device form of every 1x1 conv / dense layer of the reference graph applied to [B,N,1,C] tensors:
// helper_tf_util.conv2d (PointSegment/helper_tf_util.py:115-170), conv2d_transpose (:173-250) and
// tf.layers.dense (PointSegment/RandLANet.py:114), with inference-mode BatchNorm folded into W and b on the
// host.  Two row sources with optional gather indices cover
//   * tf.concat([skip, nearest_interpolation(feature, interp_idx)]) -> conv2d_transpose (RandLANet.py:137-141)
//   * mlp2(f_pc) + shortcut(feature) as one GEMM over the concatenated K axis       (RandLANet.py:317-321)
// without materialising the concatenation.
#pragma once

#include <vector>

#include "common.h"

namespace ps {

// Weights packed on the host into MFMA B-fragment order (see pack_weights() in rowgemm.hip):
//   for column block cb (NTB*16 output channels), k-step s (4 input channels), lane l, tile j:
//     wp[((cb*KS + s)*64 + l)*NTB + j] = W[s*4 + (l>>4)][(cb*NTB + j)*16 + (l&15)]      (0 outside W)
// A second packing ("k-permuted", pack_weights_kperm) serves the direct-load kernel, whose lanes read 16 CONTIGUOUS
// input channels of their row straight from global memory: within a 64-channel chunk c, lane group g = lane>>4 owns
// channels [c*64 + 16g, c*64 + 16g + 16) and k-step s multiplies channel c*64 + 16g + s:
//     wq[((((cb*NC + c)*16 + s)*64 + l)*NTB) + j] = W[c*64 + 16*(l>>4) + s][(cb*NTB + j)*16 + (l&15)]
// (the sum over k is the same set of products in a different order).
struct PackedLinear {
    const float* wp = nullptr;  // device
    const float* wq = nullptr;  // device, k-permuted image (nullptr when cin % 16 != 0)
    // device, bf16 image for the training step's "bf16 MLP" mode (BASELINE configs[2]); when set, the direct-load kernel rounds the
    // activations to bf16 on the fly and runs v_mfma_f32_16x16x32_bf16 (fp32 accumulate).  8 bf16 per (column block cb, chunk c,
    // half s, lane l, tile j):  wb[(((cb*NC + c)*2 + s)*64 + l)*NTB + j][t] = bf16(W[c*64 + 16*(l>>4) + 8s + t][(cb*NTB + j)*16 + (l&15)])
    const void* wb = nullptr;
    // device, pack_p32 image (attpool.h) for the 32x32x2 kernel of the deep levels (gemm32.hip); nullptr when not built
    const float* w32 = nullptr;
    // device, pack_p32b image (three bfloat16 planes, 6 bytes per weight) for the split-bf16 form of that kernel (gemm32b.hip); nullptr when not built
    const void* w32b = nullptr;
    const float* bias = nullptr;  // device [cout_pad] (zeros when the layer has no bias)
    int cin = 0, cout = 0;
    int ks = 0;       // k-steps = ceil(cin/4)
    int ntb = 0;      // 1, 2 or 4
    int cblocks = 0;  // ceil(cout / (16*ntb))
    int leaky = 0;
    int accum = 0;    // add the result to what y already holds (training: gradient accumulation in the GEMM epilogue)
    size_t packed_floats() const { return (size_t)cblocks * ks * 64 * ntb; }
    int nchunks() const { return (cin + 63) / 64; }
    size_t kperm_floats() const { return (size_t)cblocks * nchunks() * 16 * 64 * ntb; }
    size_t bf16_bytes() const { return (size_t)cblocks * nchunks() * 2 * 64 * ntb * 16; }
    int cout_pad() const { return cblocks * ntb * 16; }
};

inline int choose_ntb(int cout) { return cout >= 64 ? 4 : (cout >= 32 ? 2 : 1); }

// Host-side packing: W is row-major [cin, cout]; out must hold packed_floats() floats.
void pack_weights(const float* W, int cin, int cout, int ntb, float* out);
void pack_weights_kperm(const float* W, int cin, int cout, int ntb, float* out);

struct RowSrc {
    const float* x = nullptr;
    const int32_t* gather = nullptr;  // row r reads x[gather[r]] when non-null
    int ld = 0;                       // row stride in floats
    int c = 0;                        // channels taken from this source
    int gm = 0, gn = 0;               // batched gather: row r reads x[(r / gm) * gn + gather[r]] when gm != 0
};

// the same layer on 32x32x2 tiles (gemm32.hip): few rows x wide channels (encoder levels 2-4, decoder); rowgemm() takes this
// route by itself when the layer carries a w32 image and the shape qualifies
bool gemm32_fits(const PackedLinear& L, const RowSrc& s1, const RowSrc& s2, int64_t R, int ldy);
int gemm32(ps_context* c, const PackedLinear& L, const RowSrc& s1, const RowSrc& s2, int64_t R, float* y, int ldy);
// ... and on v_mfma_f32_32x32x16_bf16 over exact three-way bfloat16 splits of both operands (gemm32b.hip): the products large enough
// to be bound by the matrix pipe
void pack_p32b(const float* W, int cin, int cout, uint16_t* out);  // [cin, cout], cin % 16 == 0, cout % 32 == 0 -> cin*cout*3 uint16
bool gemm32b_fits(const PackedLinear& L, const RowSrc& s1, const RowSrc& s2, int64_t R, int ldy);
int gemm32b(ps_context* c, const PackedLinear& L, const RowSrc& s1, const RowSrc& s2, int64_t R, float* y, int ldy);

// y[r, 0:cout] (row stride ldy) for r in [0, R)
int rowgemm(ps_context* c, const PackedLinear& L, const RowSrc& s1, const RowSrc& s2, int64_t R, float* y, int ldy);

// A chain of up to kChainMaxSteps dense layers evaluated per 16-row tile, the activations staying in LDS between the
// layers: for the wide-N / narrow-C end of the network (level 0, last decoder step, head) every separate layer is a full
// HBM round trip of [N, C] and a ~5 us dependent launch; chained, only the rows that somebody else reads are written.
//   step k: act_k = act([act_{k-1} | extra_k] . W_k + b_k)      act_{-1} = [s1 | s2] (gather allowed, as in rowgemm)
// All channel counts (inputs incl. extras, outputs) must be <= kChainMaxC.
constexpr int kChainMaxC = 96;
constexpr int kChainMaxSteps = 4;
struct ChainStep {
    const PackedLinear* L = nullptr;
    float* y = nullptr;  // also store this step's output rows here (nullptr: the output only feeds the next step)
    int ldy = 0;
    RowSrc extra;        // extra.x != nullptr: `extra.c` channels of plain rows appended after the previous activations
};
bool rowchain_fits(const ChainStep* steps, int n_steps, const RowSrc& s1, const RowSrc& s2);
// register-resident evaluation of the network's own chain shapes (regchain.hip); rowchain() prefers it when the shape matches
bool regchain_fits(const ChainStep* steps, int n_steps, const RowSrc& s1, const RowSrc& s2);
// per-network cache of the chains' re-ordered weight images (built on first use on the caller's stream; clear() when the
// weights change).  Entries are keyed by the first / last layer's packed weights and the split of the first K axis.
struct ChainCache {
    struct Entry { const float* wp0 = nullptr; const float* wp_last = nullptr; int n = 0, ca = 0, cb = 0; DevBuf img; };
    std::vector<Entry> entries;
    void clear()
    {
        for (auto& e : entries) e.img.release();
        entries.clear();
    }
};
int regchain(ps_context* c, const ChainStep* steps, int n_steps, const RowSrc& s1, const RowSrc& s2, int64_t R, ChainCache* cache);
int rowchain(ps_context* c, const ChainStep* steps, int n_steps, const RowSrc& s1, const RowSrc& s2, int64_t R, ChainCache* cache = nullptr);

// gemm_b3.hip: large fp32 products of the training step on bf16 MFMA over exact three-way splits
bool gemm_b3_fits(const Tuning& tn, int64_t R, int64_t K, int64_t N, const float* x, int64_t ldx, bool one_plane);
size_t gemm_b3_plane_bytes(int64_t K, int64_t N);
int gemm_b3(ps_context* c, const float* x, int64_t ldx, const float* w, int64_t sk, int64_t sn, const float* bias, int64_t R, int64_t K, int64_t N, int leaky,
            int accumulate, float* y, int64_t ldy, void* planes, int pack = 1);  // pack = 0: `planes` already hold this matrix (PackCache)

}  // namespace ps
