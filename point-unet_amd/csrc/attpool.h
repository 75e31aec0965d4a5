// attpool.h -- fused LocSE + gather + attentive pooling (see attpool.hip).
#pragma once

#include "common.h"
#include "rowgemm.h"

namespace ps {

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
};

int att_pool_stage(ps_context* c, const AttStage& s);

}  // namespace ps
