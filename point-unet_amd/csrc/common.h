// common.h -- context, workspace arena, error plumbing and per-stage timing shared by every .hip file of
// libpointseg_hip.so.  Nothing here is visible through the C ABI (include/pointseg.h).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/pointseg.h"
#include "../../include/pointseg_train_ops.h"

namespace ps {

void set_error(const char* fmt, ...);

#define PS_HIP(expr)                                                                                  \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess) {                                                                       \
            ps::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return PS_EHIP;                                                                           \
        }                                                                                             \
    } while (0)

#define PS_CHECK(cond, ...)             \
    do {                                \
        if (!(cond)) {                  \
            ps::set_error(__VA_ARGS__); \
            return PS_EINVAL;           \
        }                               \
    } while (0)

#define PS_TRY(expr)                \
    do {                            \
        int _rc = (expr);           \
        if (_rc != PS_OK) return _rc; \
    } while (0)

// A grow-only device buffer: sized to the high-water mark, then reused (no hipMalloc on the steady-state path).
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes);
    void release();
    template <class T>
    T* as() const { return static_cast<T*>(p); }
};

// Bump allocator over one DevBuf.  Usage per API call: plan sizes with `dry` = true, reserve, then carve.
struct Arena {
    DevBuf buf;
    size_t off = 0;
    bool dry = false;
    void begin(bool dry_run) { off = 0; dry = dry_run; }
    template <class T>
    T* take(size_t count)
    {
        size_t bytes = (count * sizeof(T) + 255) & ~size_t(255);
        T* r = dry ? nullptr : reinterpret_cast<T*>(static_cast<char*>(buf.p) + off);
        off += bytes;
        return r;
    }
};

struct StageTimer {
    std::string name;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> spans;
    int64_t launches = 0;
};

}  // namespace ps

namespace ps {
// One weight image of ps_op_conv1x1_ex: which matrix (pointer + strides), which kernel's layout, where it goes.
// kind 0: pack_weights (rowgemm LDS form), 1: k-permuted (direct-load form), 2: bfloat16 k-permuted, 3: split-bf16 planes (gemm_b3), 4: one
// RNE bfloat16 plane (gemm_b3 in the bf16-MLP mode), 5 / 6: kinds 3 / 4 with the K axis in accumulator order (attpool_gemm.hip)
struct PackJob {
    const float* w;
    int64_t sk, sn;
    int kind, cin, cout, ntb, p0, cblocks;
    void* out;
    int64_t total;  // threads of the stand-alone packing kernel
    bool same_key(const PackJob& o) const
    {
        return w == o.w && sk == o.sk && sn == o.sn && kind == o.kind && cin == o.cin && cout == o.cout && ntb == o.ntb && p0 == o.p0 && cblocks == o.cblocks;
    }
};
// The weight images of a training step, recorded during one step and from then on produced by one launch per source file at the start of
// every step (the weights only change in Adam, at the end): a step packs ~75 matrices, 5 us of launch each.  mode 0: off (every call packs
// for itself into the ring), 1: recording, 2: replaying -- calls are matched to the recorded sequence in order; a call that does not match
// (another batch size, another option set) drops the cache for the rest of the step and the next step records again.
struct PackCache {
    int mode = 0;
    bool broken = false;
    size_t cursor = 0;
    std::vector<PackJob> jobs;
    std::vector<DevBuf> bufs;
    DevBuf table;       // device copy of `jobs`
    int n_ops = 0, n_b3 = 0;  // jobs of ops.hip's kinds / gemm_b3.hip's kinds (tables are split: [ops jobs | b3 jobs])
};
// returns the image's buffer; launch = the caller still has to run its packing kernel into it
void* pack_slot(ps_context* c, const PackJob& key, size_t bytes, bool& launch);
int pack_cache_finish_recording(ps_context* c, PackCache& pc);
int pack_cache_replay(ps_context* c, PackCache& pc);
void pack_cache_clear(PackCache& pc);  // (releases the images)
int pack_batch_ops(ps_context* c, const PackJob* table, int n);  // ops.hip
int pack_batch_b3(ps_context* c, const PackJob* table, int n);   // gemm_b3.hip
}  // namespace ps

namespace ps {
// Experiment knobs of profiles/tools/*.  The DEFAULT build never reads the environment: the values below are what ships, every context holds
// the same ones, and nothing else in the library calls getenv.  A library built with -DPS_TUNING_ENV (profiles/tools/build_variant.sh) fills
// the struct ONCE, in ps_create (tuning_from_env, context.hip), from the PS_* variables named here, clamped to what the kernels are compiled
// for; the trainer reads its knobs from its context when it is created.  No function-static caches, no per-call getenv.
struct Tuning {
    // gemm32b.hip / gemm32.hip (forward dense layers of the deep levels)
    double gemm32b_min_flops = 3e8;  // PS_GEMM32B_MIN_FLOPS: products at least this large run on split-bf16 MFMA (0 = all that fit, 1e30 = none)
    int gemm32b_rw = 0, gemm32b_cw = 0;  // PS_GEMM32B_RW / _CW: tile shape override, 1 or 2 (0 = the heuristic)
    bool gemm32_no_sk8 = false;      // PS_GEMM32_NO_SK8
    // attpool_gemm.hip / attpool_train.hip
    bool att64_gemm = true;          // PS_ATT64_GEMM=0: level 1's training pooling back on attpool_train.hip's per-point kernels
    int att64_occ = 0;               // PS_ATT64_OCC: 1 | 2 = the other occupancy of the d = 64 backward (bf16 / fp32)
    bool att_no_split = false;       // PS_ATT_NO_SPLIT: the one-wave-per-point bf16 backward at d = 128
    // gemm_b3.hip / ops_train.hip
    int64_t wgrad_b3_min_rows = 0;   // PS_WGRAD_B3_MIN_ROWS (0 = 4 096 three planes / 16 384 one plane)
    int64_t gemm_b3_min_rows = 0;    // PS_GEMM_B3_MIN_ROWS (0 = 4 096 / 8 192)
    int64_t wgrad_wgs = 512;         // PS_WGRAD_WGS, clamped to [64, 4096]: workgroups (= partial slabs x blocks) of a weight-gradient launch
    bool bn_slice = false;           // PS_BN_SLICE=1: the one-launch BatchNorm for small tensors (measured no faster: DESIGN.md 4.3)
    // invidx.hip
    bool inv_bucket = true;          // PS_INV_BUCKET=0: the radix-sort form of the inverse index
    int inv_tile = 4096;             // PS_INV_TILE in {4096, 8192}
    bool gather_reduce_ordered = true;  // PS_GATHER_REDUCE_ORDERED=0
    int maxpool_bwd_ordered = 1;     // PS_MAXPOOL_BWD_ORDERED=0: the one-entry cloud-order walk
    // trainer.hip (read at ps_trainer_create)
    bool train_act_bf16 = true;      // PS_TRAIN_ACT_BF16=0 (next to ps_train_options.act_bf16)
    int train_att_gemm_split = -1;   // PS_TRAIN_ATT_GEMM_SPLIT: -1 = bf16-MLP mode only
    bool train_att_gemm = true;      // PS_TRAIN_ATT_GEMM=0
    bool train_att128_fwd_gemm = true;  // PS_TRAIN_ATT128_FWD_GEMM=0
    bool train_fuse_residual = true; // PS_TRAIN_FUSE_RESIDUAL=0
    bool train_merge_syncbn = true;  // PS_TRAIN_MERGE_SYNCBN=0: every BatchNorm layer its own all-reduces (89 instead of 74 calls per step)
    int convbn_max_c = 64;           // PS_CONVBN_MAX_C
    int convbn_rect_max = 1 << 30;   // PS_CONVBN_RECT_MAX (cin * cout)
    bool wgrad_debug = false;        // PS_WGRAD_DEBUG: print the first step's weight-gradient shapes
};
void tuning_from_env(Tuning& t);  // context.hip: a no-op unless the library is built with -DPS_TUNING_ENV
}  // namespace ps

struct ps_context {
    int device = 0;
    ps::Tuning tune;         // (see above: constants in the default build)
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    ps::Arena knn_arena;     // kd-trees + query scratch
    ps::Arena net_arena;     // activations of the forward pass
    ps::DevBuf stage_in;     // host<->device staging for device_ptrs == 0 calls
    ps::DevBuf stage_out;
    ps::DevBuf red_ws;       // per-block partial sums of the per-channel reductions (ops_train.hip)
    ps::DevBuf wgrad_ws;     // per-slab partials of ps_op_linear_wgrad[_ex] (ops_train.hip)
    ps::PackCache* pack_cache = nullptr;  // (set by the native training step around its work: the step's weight images, packed by ONE launch)
    ps::DevBuf ops_ring[4];  // packed weights of ps_op_conv1x1 (a ring: consecutive calls never repack into the buffer the previous GEMM is still reading)
    int ops_ring_pos = 0;
    // (dense layers of the deep levels / the decoder with at least tune.gemm32b_min_flops FLOPs run on bf16 MFMA over exact three-way splits,
    //  gemm32b.hip, instead of the fp32 MFMA, gemm32.hip; switched with the attention's form by ps_set_att_bf16x3)
    bool att_bf16x3 = true;   // ps_set_att_bf16x3: attentive pooling at d = 64 / 128 on bf16 MFMA over three-way splits (attpool32b.hip)
    bool train_b3 = true;     // ps_set_train_gemm_b3: large fp32 op-level GEMMs on bf16 MFMA over exact three-way splits (gemm_b3.hip)
    bool conv_w_transposed = false;  // (internal, set around a call by the native trainer) ps_op_conv1x1_ex: w is stored [cout, cin]
    bool att_df_accum = false;  // (internal, set around a call by the native trainer) ps_op_att_pool_train_bwd_split*: dfr += instead of dfr =
    // a hint of the trainer for the op it enqueues next: i32[B, walk_order_n], the spatially coherent processing order (ps_pyramid.order) of the
    // walk_order_n points per cloud the op's destinations are; NULL = none (ops called through the C ABI on their own never see one)
    const int32_t* walk_order = nullptr;
    int64_t walk_order_n = 0;
    bool pool_bwd_overwrite = false;  // ps_op_random_sample_bwd_inv (tie-count form) STORES the gradient instead of adding into a zeroed buffer (trainer.hip)
    bool train_bf16 = false;  // ps_set_train_gemm_bf16: the op-level GEMMs round their operands to bf16 (fp32 accumulate)
    bool train_act_bf16 = false;  // ps_set_train_act_bf16: the [N*K, h] activation rows of the LFA branch are STORED as bfloat16 (include/pointseg_train_ops.h)
    // per-device kernel attributes already raised by this context (dynamic LDS above the default limit)
    bool mid_lds_attr = false;
    size_t chain_lds_attr = 48 * 1024;
    std::map<const void*, size_t> regchain_lds_attr;  // per chain kernel (regchain.hip): dynamic-LDS limit already raised on this device
    // deferred status checks (ps_set_deferred_checks): device flags land in pinned slots, validated at ps_synchronize
    bool deferred = false;
    int32_t* h_flags = nullptr;   // pinned [8][4]
    unsigned pending_mask = 0;
    uint64_t builds = 0;          // ps_pyramid_build calls on this context so far (error messages name the failing one)
    uint64_t flag_serial[8] = {}; // which build each pending slot belongs to
    int flag_slot = 0;
    int sticky_rc = 0;            // a failed deferred check found while reusing its slot: reported by the next ps_synchronize
    std::string sticky_msg;
    hipEvent_t flag_ev[8] = {};   // recorded behind the copy into each slot: reusing a slot waits for THAT copy, not for the stream
    std::vector<char> host_ring[8];  // host staging kept alive behind asynchronous uploads
    int ring_pos = 0;
    std::vector<char>& ring_next() { ring_pos = (ring_pos + 1) & 7; return host_ring[ring_pos]; }
    int check_deferred();
    int check_flag_slot(int s);  // after a stream sync: PS_OK or PS_ESTATE with the message set
    // small host->device uploads that never stall the host: the data is copied into a pinned ring slot first
    struct PinSlot { void* p = nullptr; size_t cap = 0; hipEvent_t ev = nullptr; bool busy = false; };
    PinSlot pin[16];
    int pin_pos = 0;
    int upload_async(void* dst, const void* src, size_t bytes);
    // timing
    bool timing = false;
    std::string timing_only;  // when non-empty only this stage records events (keeps the timed region undisturbed)
    std::vector<ps::StageTimer> stages;
    std::vector<hipEvent_t> event_pool;
    size_t event_next = 0;
    int cur_stage = -1;
    hipEvent_t cur_start = nullptr;

    hipEvent_t get_event();
    void stage_begin(const char* name);
    void stage_end(int launches);
};

namespace ps {
// RAII span: times everything enqueued on ctx->stream between construction and destruction when timing is armed.
struct Stage {
    ps_context* c;
    int n;
    bool on = false;
    Stage(ps_context* ctx, const char* name, int launches = 1) : c(ctx), n(launches)
    {
        on = c->timing && (c->timing_only.empty() || c->timing_only == name);
        if (on) c->stage_begin(name);
    }
    ~Stage()
    {
        if (on) c->stage_end(n);
    }
};

inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ps_pyramid.built: what ps_pyramid_build leaves behind and the trainer recomputes -- any change of a table pointer or shape breaks it.
// It vouches for STRUCTURE (sub_idx is the prefix of neigh_idx, order[] is a permutation), which holds for every build, also one whose
// status words later report a degenerate cloud (the searches then write index 0 rows and order[t] = t: kdtree.h / knn.hip); the CONTENTS of
// the tables are the caller's to leave alone -- point-unet_amd/pyramid.py: Pyramid.invalidate() drops the stamp after an in-place edit.
inline uint64_t pyramid_stamp(const ps_pyramid* p)
{
    uint64_t h = 0x50535059524D4944ull ^ (uint64_t)p->B ^ ((uint64_t)p->K << 32) ^ ((uint64_t)p->num_layers << 48);
    for (int i = 0; i < p->num_layers && i < PS_MAX_LAYERS; ++i) {
        h = (h ^ (uint64_t)reinterpret_cast<uintptr_t>(p->sub_idx[i])) * 0x9E3779B97F4A7C15ull;
        h = (h ^ (uint64_t)reinterpret_cast<uintptr_t>(p->neigh_idx[i])) * 0x9E3779B97F4A7C15ull;
        h = (h ^ (uint64_t)reinterpret_cast<uintptr_t>(p->interp_idx[i])) * 0x9E3779B97F4A7C15ull;
        h = (h ^ (uint64_t)reinterpret_cast<uintptr_t>(p->order[i])) * 0x9E3779B97F4A7C15ull;  // (the trainer scatters through order[]: a swapped table breaks the stamp)
        h = (h ^ (uint64_t)p->n[i] ^ ((uint64_t)p->n[i + 1] << 32)) * 0x9E3779B97F4A7C15ull;
    }
    return h | 1ull;
}

// random_sample forward, vector form (randla.hip): out[b, m, :] = max over k of feat[b, idx[b, m, k], :], ch % 4 == 0; order: optional row walk
int pool_max(ps_context* c, const float* feat, const int32_t* idx, const int32_t* order, float* out, int64_t B, int64_t n, int64_t m, int K, int ch,
             unsigned char* ties = nullptr);

// weight-gradient partials (ops_train.hip): the GEMM writes one [rows, cols] partial per row slab, wgrad_finish adds them up in slab
// order for a whole table of jobs in one launch (the native training step finishes every gradient of a step with it)
struct WgradJob {
    const float* part;  // [slabs][rows][cols]
    float* dst;         // [rows][cols], or [cols][rows] when transposed
    int slabs, rows, cols, transposed;
};
// (both take the operands: which kernel runs -- fp32 MFMA or, for 128-multiple shapes of many rows, split-bf16 MFMA (gemm_b3.hip) --
//  depends on their alignment, and the two must agree on the number of slabs)
int64_t wgrad_partial_slabs(ps_context* c, const float* x, int64_t ldx, const float* dy, int64_t lddy, int64_t R, int64_t cin, int64_t cout);
int wgrad_partial(ps_context* c, const float* x, int64_t ldx, const float* dy, int64_t lddy, int64_t R, int64_t cin, int64_t cout, float* part,
                  float* dbpart);
bool wgrad_b3_fits(const ps::Tuning& tn, int64_t R, int64_t cin, int64_t cout, const float* x, int64_t ldx, const float* dy, int64_t lddy, bool one_plane);
int64_t wgrad_b3_slabs(const ps::Tuning& tn, int64_t R, int64_t cin, int64_t cout);
int wgrad_b3_partial(ps_context* c, const float* x, int64_t ldx, const float* dy, int64_t lddy, int64_t R, int64_t cin, int64_t cout, float* part,
                     float* dbpart);
int64_t wgrad_split_slabs(ps_context* c, const float* xl, int64_t ldxl, const int32_t* xidx, const float* xr, int64_t ldxr, const float* dy, int64_t lddy,
                          int64_t R, int64_t cin, int64_t cout);  // 0: the split-source weight gradient does not apply to these operands
bool wgrad_b3_split_fits(const ps::Tuning& tn, int64_t R, int64_t cin, int64_t cout, const float* xl, int64_t ldxl, const int32_t* xidx, const float* xr, int64_t ldxr, const float* dy,
                         int64_t lddy);
int wgrad_b3_partial_split(ps_context* c, const float* xl, int64_t ldxl, const int32_t* xidx, int64_t n_src, int64_t rows_per_cloud, const float* xr, int64_t ldxr,
                           const float* dy, int64_t lddy, int64_t R, int64_t cin, int64_t cout, float* part);
int wgrad_finish(ps_context* c, const WgradJob* d_jobs, int n_jobs, int64_t max_elems);
}  // namespace ps
