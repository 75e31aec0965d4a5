// attpool.h -- fused LocSE + gather + attentive pooling (see attpool.hip).
#pragma once

#include "common.h"
#include "rowgemm.h"

namespace ps {

// Weight images of one encoder level for the 32x32x2 kernels (attpool32.hip: pack_p32 / pack_p32_locse), device pointers.
struct Att32Weights {
    const float* w1 = nullptr;   // LocSE mlp1 (10 -> h)
    const float* w2 = nullptr;   // LFA mlp2 (h -> h)
    const float* wb1 = nullptr;  // att_pooling_1 Wfc[h:, :] (h -> d)
    const float* wb2 = nullptr;  // att_pooling_2 Wfc[h:, :]
    // the same four matrices as three-plane bfloat16 images (attpool32b.hip: pack_b3 / pack_b3_locse); null when not packed
    const float* w1b = nullptr;
    const float* w2b = nullptr;
    const float* wb1b = nullptr;
    const float* wb2b = nullptr;
};
void pack_b3(const float* W, int cin, int cout, uint16_t* out);   // [cin, cout], cin % 32 == 0, cout % 32 == 0 -> cin*cout*3 uint16
void pack_b3_locse(const float* W1, int cout, uint16_t* out);     // [10, cout] -> (cout/32)*3*64*8 uint16
void pack_p32(const float* W, int cin, int cout, float* out);   // [cin, cout] row-major, cin % 8 == 0, cout % 32 == 0 -> cin*cout floats
void pack_p32_locse(const float* W1, int cout, float* out);     // [10, cout] -> (cout/32)*5*64 floats

struct AttStage {
    const float* xyz = nullptr;      // [n_total, 3]
    const int32_t* idx = nullptr;    // [n_total, k] cloud-local neighbour indices
    const int32_t* order = nullptr;  // optional [n_total]: cloud-local row of the t-th point in a spatially coherent order (ps_pyramid.order)
    const float* fg = nullptr;       // [n_total, d/2 + d]: features f | G = f . Wfc[:d/2, :]
    const PackedLinear* lfa1 = nullptr;  // 10 -> d/2 (bias + folded BN, LeakyReLU)
    const PackedLinear* lfa2 = nullptr;  // d/2 -> d/2, stage 2 only (nullptr = stage 1)
    const PackedLinear* wbot = nullptr;  // Wfc[d/2:, :]  (d/2 -> d, no bias): pre-product formulation, fg = [f | G]
    const PackedLinear* wfull = nullptr; // Wfc (d -> d): direct formulation, fg = f only (row stride ldf)
    int ldf = 0;
    float* agg = nullptr;            // [n_total, d]
    int64_t n_total = 0, n_cloud = 0;
    int d = 0, k = 16;
    const Att32Weights* p32 = nullptr;  // when set (d >= 64, pre-product formulation): the 32x32x2 kernels (attpool32.hip)
};

bool att_pool32_fits(const AttStage& s);
int att_pool32_stage(ps_context* c, const AttStage& s);
bool att_pool32b_fits(const AttStage& s);
int att_pool32b_stage(ps_context* c, const AttStage& s);

int att_pool_stage(ps_context* c, const AttStage& s);

}  // namespace ps
