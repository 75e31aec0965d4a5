// attpool_train.h -- the argument block shared by the per-point fused attentive-pooling kernels of the training step
// (attpool_train.hip) and their d = 64 form on the split-bf16 matrix pipe (attpool_gemm.hip).
#pragma once

#include "common.h"

namespace ps {

struct AttTrainArgs {
    const float* f;     // [R*K, ld]
    const float* w;     // [D, D] row-major
    const float* dagg;  // [R, D] (backward)
    float* agg;         // [R, D] (forward)
    float* df;          // [R*K, lddf] (backward)
    float* dw_part;     // [gridDim.x, D*D] (backward)
    int64_t R;
    int ld, lddf, bf16;
    // split-source form (gather_neighbour + concat folded in, RandLANet.py:326-333): F = [fl[idx] | f]; `f` / `df` then hold only the
    // right half ([R*K, D/2] rows); the left half's gradient is added into dfl with float atomics (a scatter-add like
    // ps_op_scatter_add_rows)
    const float* fl;     // [B*n_src, D/2] rows (ldl), nullptr = F is materialised in f
    const int32_t* idx;  // [R, K] cloud-local source rows
    float* dfl;          // [B*n_src, D/2] rows (lddl), accumulated into (backward)
    int64_t n_src, n_q;  // rows per cloud of fl / points per cloud
    int ldl, lddl;
    float* dfl_rows;     // non-null: the gathered half's gradient goes HERE as plain rows [R*K, D/2] (ld_rows) instead of being scatter-added into
    int ld_rows;         // dfl with float atomics; ps_op_gather_reduce_rows then adds the rows up in a fixed order (deterministic step)
    int df_accum;        // split form: df (the f_xyz half's gradient) is ADDED to what the rows already hold (a second gradient of the same tensor)
    int fr_bf16;         // split form: the rows of `f` (the f_xyz half) are STORED as bfloat16 (ps_set_train_act_bf16; ld in elements)
    int vec_store;       // backward: the row outputs (df, dfl_rows) are 16-byte aligned with pitches % 4 == 0 -> staged through LDS, float4 stores
};

// d = 64 (encoder level 1) on v_mfma_f32_32x32x16_bf16 with both weight images resident in LDS (attpool_gemm.hip): forward, or backward with
// the weight gradient accumulated in registers (dW: [64, 64], overwritten).  Handles the plain form and the split-source form with row
// outputs (a.dfl_rows); the float-atomic scatter form stays with attpool_train.hip.
bool att64_gemm_fits(const Tuning& tn, const AttTrainArgs& a, bool backward);
int att64_gemm(ps_context* c, AttTrainArgs a, bool backward, float* dW);

}  // namespace ps
