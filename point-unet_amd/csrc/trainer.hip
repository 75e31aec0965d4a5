// trainer.hip -- the training step of the PointSegment RandLA-Net behind ONE C-ABI call (ps_randla_train_step / ps_randla_backward).
//
// What it replaces in the reference (TensorFlow autodiff over the graph of Network.inference):
//     loss + optimizer                         PointSegment/RandLANet.py:62-90, 267-274
//     sess.run([train_op, extra_update_ops..]) PointSegment/RandLANet.py:162-169
//     tf.layers.batch_normalization(training)  PointSegment/helper_tf_util.py:167,246; RandLANet.py:115
//     the graph itself                         PointSegment/RandLANet.py:110-152, 314-401
//
// Structure: a tape in C++.  The forward pass (Trainer::forward, a line-by-line counterpart of Network.inference in training mode)
// calls the op-level kernels of this library (ops.hip, ops_train.hip, attpool_train.hip, locse_train.hip, gemm_b3.hip ...) and records,
// per op, a closure that computes the input gradients from the output gradient; backward() replays the closures in reverse.  No
// Python, no torch: device memory for activations and gradients comes from a pool owned by the trainer (blocks are reference counted
// and go back to the pool the moment the last user drops them, so the step's footprint is that of the live tensors), parameters /
// gradients / Adam moments / BatchNorm moving statistics are CALLER-OWNED flat device buffers (ps_trainer_bind) in the layout
// ps_trainer_layout reports, BatchNorm moving statistics are updated by the kernels that finish the batch statistics, and collectives
// (gradient mean, BatchNorm statistics shared by the ranks) go through a callback the host supplies (ps_trainer_set_collective: the
// library does not link a communication library; RCCL sits behind torch.distributed in bench.py, behind ncclAllReduce in a C++ host).
#include "common.h"

#include <cmath>
#include <functional>
#include <array>
#include <chrono>
#include <map>
#include <memory>
#include <unordered_map>

using namespace ps;

namespace ps {

static constexpr float kBnEps = 1e-6f;      // tf.layers.batch_normalization(x, -1, 0.99, 1e-6)  RandLANet.py:115, helper_tf_util.py:167
static constexpr float kBnMomentum = 0.99f;

struct TrainError {
    int rc;
};
#define TK(expr)                                \
    do {                                        \
        int rc_ = (expr);                       \
        if (rc_ != PS_OK) throw TrainError{rc_}; \
    } while (0)
#define TK_HIP(expr)                                                                                  \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) {                                                                       \
            ps::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            throw TrainError{PS_EHIP};                                                                \
        }                                                                                             \
    } while (0)

// ---- device memory pool -----------------------------------------------------------------------------------------------------------
// First-fit over address-ordered free blocks with coalescing, in chunks obtained from hipMalloc.  Everything runs on ONE stream, so a
// block can be handed out again the moment the host drops it: the kernels that still read it were enqueued before the kernels of its
// next user.  Chunks are never returned; the first step grows the pool to its high-water mark (a handful of chunks), every later
// step of the same shape repeats the same allocation sequence and finds the same places (steady state: no hipMalloc).
struct Pool {
    struct Chunk {
        char* base;
        size_t size;
    };
    std::vector<Chunk> chunks;
    std::map<char*, size_t> free_;
    std::unordered_map<char*, size_t> used;
    size_t in_use = 0, peak = 0, total = 0;

    void add_chunk(size_t bytes)
    {
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) {
            ps::set_error("training pool: hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
            throw TrainError{PS_ENOMEM};
        }
        chunks.push_back({static_cast<char*>(p), bytes});
        free_[static_cast<char*>(p)] = bytes;
        total += bytes;
    }
    void* alloc(size_t bytes)
    {
        bytes = (bytes + 255) & ~size_t(255);
        if (!bytes) bytes = 256;
        for (int attempt = 0; attempt < 2; ++attempt) {
            for (auto it = free_.begin(); it != free_.end(); ++it) {
                if (it->second < bytes) continue;
                char* p = it->first;
                const size_t rest = it->second - bytes;
                free_.erase(it);
                if (rest) free_[p + bytes] = rest;
                used[p] = bytes;
                in_use += bytes;
                if (in_use > peak) peak = in_use;
                return p;
            }
            // grow: at least the request, at least 1 GiB, at least half of what there is (few chunks even for a first, unsized step)
            size_t want = std::max(bytes, std::max<size_t>(size_t(1) << 30, total / 2));
            add_chunk(want);
        }
        ps::set_error("training pool: allocation of %zu bytes failed", bytes);
        throw TrainError{PS_ENOMEM};
    }
    void release(void* q)
    {
        char* p = static_cast<char*>(q);
        auto u = used.find(p);
        if (u == used.end()) return;
        size_t bytes = u->second;
        used.erase(u);
        in_use -= bytes;
        auto nx = free_.lower_bound(p);
        // merge with the following block (same chunk only: chunks are separate allocations and never adjacent by contract)
        if (nx != free_.end() && p + bytes == nx->first && same_chunk(p, nx->first)) {
            bytes += nx->second;
            nx = free_.erase(nx);
        }
        if (nx != free_.begin()) {
            auto pv = std::prev(nx);
            if (pv->first + pv->second == p && same_chunk(pv->first, p)) {
                pv->second += bytes;
                return;
            }
        }
        free_[p] = bytes;
    }
    bool same_chunk(const char* a, const char* b) const
    {
        for (const Chunk& c : chunks)
            if (a >= c.base && a < c.base + c.size) return b >= c.base && b < c.base + c.size;
        return false;
    }
    // in front of a step: everything is free again
    void begin_step()
    {
        if (!used.empty()) {  // a step that failed half-way
            used.clear();
            in_use = 0;
            free_.clear();
            for (const Chunk& c : chunks) free_[c.base] = c.size;
        }
        // (several chunks stay several chunks: the allocation sequence of a step is deterministic, so first-fit places every tensor of
        //  the next step exactly where it was -- merging them into one allocation cost a 20 GB hipFree + hipMalloc, ~1.4 s, in step 2)
        peak = 0;
    }
    void destroy()
    {
        if (!chunks.empty()) (void)hipDeviceSynchronize();
        for (const Chunk& c : chunks) (void)hipFree(c.base);
        chunks.clear();
        free_.clear();
        used.clear();
        in_use = total = 0;
    }
};

struct Block {
    Pool* pool;
    void* p;
    Block(Pool* pl, void* q) : pool(pl), p(q) {}
    ~Block() { pool->release(p); }
    Block(const Block&) = delete;
    Block& operator=(const Block&) = delete;
};

// a 2-D fp32 device tensor (rows contiguous, row stride ld >= C); `own` keeps the pool block alive (null: caller-owned memory)
struct Tn {
    float* p = nullptr;
    int64_t R = 0, C = 0, ld = 0;
    int id = -1;
    bool req = false;  // takes part in the backward pass
    bool b16 = false;  // rows STORED as bfloat16 (ps_train_options.act_bf16): only ever handed to the ops that read them that way
    std::shared_ptr<Block> own;
    bool contiguous() const { return ld == C; }
    int64_t numel() const { return R * C; }
    explicit operator bool() const { return p != nullptr; }
};

// ---- small kernels of the tape itself ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tr_transpose_kernel(const float* __restrict__ src, int rows, int cols, float* __restrict__ dst)
{
    // dst[c, r] = src[r, c]; 32x32 tiles through LDS (weights only: at most 1.5 M elements)
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8)
        if (by + j < rows && bx + tx < cols) tile[j][tx] = src[(size_t)(by + j) * cols + bx + tx];
    __syncthreads();
    for (int j = ty; j < 32; j += 8)
        if (bx + j < cols && by + tx < rows) dst[(size_t)(bx + j) * rows + by + tx] = tile[tx][j];
}

template <bool ADD>
__global__ __launch_bounds__(256) void tr_copy2d_kernel(const float* __restrict__ src, int64_t lds, float* __restrict__ dst, int64_t ldd, int64_t n, int C)
{
    for (int64_t e = blockIdx.x * (int64_t)256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / C;
        const int c = (int)(e - r * C);
        const float v = src[r * lds + c];
        if (ADD)
            dst[r * ldd + c] += v;
        else
            dst[r * ldd + c] = v;
    }
}

// is pool_idx [B, M, K] the first M rows per cloud of neigh [B, N, K]?  (the deterministic max-pool backward walks the neighbour
// table's inverse index and relies on it: ps_pyramid_build's tables are, a caller-filled pyramid need not be)
__global__ __launch_bounds__(256) void tr_prefix_check_kernel(const int32_t* __restrict__ pool_idx, const int32_t* __restrict__ neigh, int64_t B, int64_t N, int64_t M,
                                                              int64_t K, int32_t* __restrict__ mismatch)
{
    const int64_t per = M * K, total = B * per;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / per, r = i - b * per;
        if (pool_idx[i] != neigh[b * N * K + r]) *mismatch = 1;
    }
}

__global__ __launch_bounds__(256) void tr_scale_kernel(float* __restrict__ x, int64_t n, float s)
{
    for (int64_t e = blockIdx.x * (int64_t)256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) x[e] *= s;
}

// moving statistics (the reference's extra_update_ops, RandLANet.py:90,163): moving = momentum * moving + (1 - momentum) * batch
__global__ void tr_ema2_kernel(float* __restrict__ mov_mean, float* __restrict__ mov_var, const float* __restrict__ mean, const float* __restrict__ var, int C,
                               float momentum)
{
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= C) return;
    mov_mean[c] = mov_mean[c] * momentum + mean[c] * (1.f - momentum);
    mov_var[c] = mov_var[c] * momentum + var[c] * (1.f - momentum);
}

// labels -> training labels (RandLANet.py:77-81: a 0 inserted at every ignored index of range(C); ignored entries become -1 here)
__global__ __launch_bounds__(256) void tr_label_map_kernel(const int32_t* __restrict__ lab, const int32_t* __restrict__ map, int nmap, int64_t n,
                                                           int32_t* __restrict__ out)
{
    for (int64_t e = blockIdx.x * (int64_t)256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const int v = lab[e];
        out[e] = (v >= 0 && v < nmap) ? map[v] : -1;
    }
}

// LocSE branch, forward finish: float64 sums of y and y^2 over all rows (of all ranks) -> mean, variance, invstd, scale = gamma invstd,
// and the moving-statistics update.  out = mean[h] | var[h] | invstd[h] | scale[h]
__global__ void tr_locse_stats_kernel(const double* __restrict__ sums, double rows, const float* __restrict__ gamma, int h, float eps, float* __restrict__ out,
                                      float* __restrict__ mov_mean, float* __restrict__ mov_var, float momentum, int sq_off)
{
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= h) return;
    const double m = sums[c] / rows;
    double v = sums[sq_off + c] / rows - m * m;  // population variance (tf.nn.moments)
    if (v < 0.0) v = 0.0;
    const float mean = (float)m, var = (float)v;
    const float invstd = rsqrtf(var + eps);
    out[c] = mean;
    out[h + c] = var;
    out[2 * h + c] = invstd;
    out[3 * h + c] = gamma[c] * invstd;
    mov_mean[c] = mov_mean[c] * momentum + mean * (1.f - momentum);
    mov_var[c] = mov_var[c] * momentum + var * (1.f - momentum);
}

// LocSE branch, backward finish.  acc = S1[h] | S2[h] | XS[h] | A[10,h] | G[10,h] | E[16] (ps_op_locse_train_bwd); tot = S1 | S2 summed over
// the ranks.  dgamma = S2, dbeta = S1 (this rank's); dw = k (A - E x m1 - G . m2), db = k (S1 - R m1 - XS m2), k = gamma invstd,
// m = tot / rows of all ranks
__global__ void tr_locse_local_kernel(const float* __restrict__ acc, int h, float* __restrict__ ggamma, float* __restrict__ gbeta, float* __restrict__ tot)
{
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= h) return;
    gbeta[c] = acc[c];
    ggamma[c] = acc[h + c];
    tot[c] = acc[c];
    tot[h + c] = acc[h + c];
}
__global__ void tr_locse_wgrad_kernel(const float* __restrict__ acc, const float* __restrict__ tot, const float* __restrict__ gamma,
                                      const float* __restrict__ invstd, int h, float rows_local, float inv_rows_total, float* __restrict__ gW,
                                      float* __restrict__ gb)
{
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= h) return;
    const float m1 = tot[c] * inv_rows_total, m2 = tot[h + c] * inv_rows_total;
    const float k = gamma[c] * invstd[c];
    const float* A = acc + 3 * h;
    const float* G = acc + 13 * h;
    const float* E = acc + 23 * h;
    for (int j = 0; j < 10; ++j) gW[j * h + c] = k * (A[j * h + c] - E[j] * m1 - G[j * h + c] * m2);
    gb[c] = k * (acc[c] - rows_local * m1 - acc[2 * h + c] * m2);
}

__global__ void tr_pair_copy_kernel(const float* __restrict__ a, const float* __restrict__ b, int C, float* __restrict__ out)
{
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= C) return;
    out[c] = a[c];
    out[C + c] = b[c];
}

// float <-> double copies of a handful of per-channel sums (the merged SyncBN all-reduces carry float sums in a double buffer)
__global__ void tr_f2d_kernel(const float* __restrict__ in, int n, double* __restrict__ out)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i < n) out[i] = (double)in[i];
}
__global__ void tr_d2f_kernel(const double* __restrict__ in, int n, float* __restrict__ out)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i < n) out[i] = (float)in[i];
}

static inline unsigned tr_grid(int64_t n)
{
    const int64_t b = (n + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

// ---- parameter layout (point-unet_amd/weights.py: layer_dims) -------------------------------------------------------------------------
enum LayerKind { kDense, kDenseNoBias, kConv, kConvNoBn, kDeconv };

struct LayerP {
    std::string scope;
    LayerKind kind;
    int cin, cout;
    int64_t w = -1, b = -1, gamma = -1, beta = -1;  // offsets into the flat parameter / gradient buffers
    int64_t mov_mean = -1, mov_var = -1;            // offsets into the flat BatchNorm-statistics buffer
};

struct LayoutRow {
    std::string name;
    int64_t offset, rows, cols;
    int is_buffer;
};

}  // namespace ps

struct ps_trainer {
    ps_context* c = nullptr;
    ps_randla_config cfg{};
    ps_train_options opt{};
    std::vector<LayerP> layers;
    std::unordered_map<std::string, int> by_scope;
    std::vector<LayoutRow> rows;
    int64_t n_params = 0, n_buffers = 0;
    float *params = nullptr, *grads = nullptr, *adam_m = nullptr, *adam_v = nullptr, *buffers = nullptr;
    ps_allreduce_fn coll = nullptr;
    void* coll_user = nullptr;
    int world = 1, rank = 0;
    bool sync_bn = false;
    bool coll_at_one = false;  // PS_COLLECTIVE_AT_WORLD_ONE: a one-rank world still goes through the callback
    // collectives of the last step (ps_trainer_collective_stats): calls into the host's callback, bytes handed over, host time inside
    // the callback, and -- on profiled steps -- the device time between an event pair around every call
    int64_t coll_calls = 0, coll_bytes = 0;
    double coll_host_ms = 0.0, coll_device_ms = 0.0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> coll_marks;
    int64_t step = 0;
    // ---- second stream (ps_train_options.overlap_wgrad): the weight-gradient products of the backward pass feed nothing before the step's
    // one reduction launch, so they run on `side` behind an event of the main stream (their operands exist by then) while the main
    // stream goes on with the input-gradient chain; the reduction waits for the side stream.  The operands stay referenced until then
    // (wkeep): the pool hands a released block to the NEXT main-stream kernel, which would overwrite it under a product still reading.
    hipStream_t side = nullptr;
    std::vector<hipEvent_t> fork_events;
    size_t forks_used = 0;
    hipEvent_t join_event = nullptr;
    bool side_busy = false;
    bool use_side() const { return opt.overlap_wgrad != 0; }
    void fork_to_side()
    {
        if (!side) {
            TK_HIP(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
            TK_HIP(hipEventCreateWithFlags(&join_event, hipEventDisableTiming));
        }
        if (forks_used == fork_events.size()) {
            hipEvent_t e;
            TK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            fork_events.push_back(e);
        }
        hipEvent_t e = fork_events[forks_used++];
        TK_HIP(hipEventRecord(e, stream()));
        TK_HIP(hipStreamWaitEvent(side, e, 0));
        side_busy = true;
    }
    void join_side()
    {
        if (!side_busy) return;
        TK_HIP(hipEventRecord(join_event, side));
        TK_HIP(hipStreamWaitEvent(stream(), join_event, 0));
        side_busy = false;
    }
    void destroy_side()
    {
        if (side) {
            (void)hipStreamSynchronize(side);
            (void)hipStreamDestroy(side);
            side = nullptr;
        }
        for (hipEvent_t e : fork_events) (void)hipEventDestroy(e);
        fork_events.clear();
        if (join_event) (void)hipEventDestroy(join_event);
        join_event = nullptr;
    }
    Pool pool;
    DevBuf label_map;  // int32 [num_classes + ignored]
    int n_label_map = 0;
    int next_id = 0;

    // ---- per-section device time of a step (ps_trainer_profile): hipEvents at the section boundaries of the forward pass, and in the
    // backward pass wherever the section of the op in hand changes
    bool profile = false;
    int section = 0;
    std::vector<std::string> section_names;
    std::vector<std::pair<std::string, hipEvent_t>> marks;
    std::vector<std::pair<std::string, double>> profile_rows;
    int begin_section(const std::string& name)
    {
        section_names.push_back(name);
        section = (int)section_names.size() - 1;
        mark("fwd " + name);
        return section;
    }
    void mark(const std::string& name)
    {
        if (!profile) return;
        hipEvent_t e;
        TK_HIP(hipEventCreate(&e));
        TK_HIP(hipEventRecord(e, stream()));
        marks.emplace_back(name, e);
    }
    void finish_profile()
    {
        if (!profile) return;
        mark("end");
        (void)hipStreamSynchronize(stream());
        std::map<std::string, double> acc;
        std::vector<std::string> order;
        for (size_t i = 0; i + 1 < marks.size(); ++i) {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, marks[i].second, marks[i + 1].second);
            if (!acc.count(marks[i].first)) order.push_back(marks[i].first);
            acc[marks[i].first] += ms;
        }
        for (auto& m : marks) (void)hipEventDestroy(m.second);
        marks.clear();
        coll_device_ms = 0.0;
        for (auto& m : coll_marks) {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, m.first, m.second);
            coll_device_ms += ms;
            (void)hipEventDestroy(m.first);
            (void)hipEventDestroy(m.second);
        }
        coll_marks.clear();
        profile_rows.clear();
        for (const std::string& n : order) profile_rows.emplace_back(n, acc[n]);
    }

    // ---- tape state of the step in flight
    struct Op {
        int out_id;
        int section;
        std::function<void(const Tn&)> bw;
    };
    std::vector<Op> ops;
    std::unordered_map<int, Tn> grad_of;
    std::unordered_map<int, std::vector<std::function<void()>>> deferred;  // input-gradient GEMMs waiting for another consumer's plain store
    // the wide levels without their concat buffers (attpool_gemm.hip, split-source forms).  Measured on MI355X, batch 8: the bf16-MLP step
    // 35.20 -> 34.97 ms with it, the fp32 step 39.71 -> 39.97 ms (the gathering loader of the weight-gradient kernel: 0.71 against 0.40 ms
    // per level-2 pooling, and the forward kernel drops from three to two waves per SIMD) -- on by default in the bf16-MLP mode only;
    // PS_TRAIN_ATT_GEMM_SPLIT = 0 | 1 overrides (-1: by mode)
    // (copies of the context's experiment knobs, taken at ps_trainer_create: common.h, struct Tuning)
    bool act_bf16_on = true;  // next to ps_train_options.act_bf16
    int att_split_env = -1;
    bool att_gemm_on = true;  // attpool_gemm.hip
    bool merge_syncbn = true;  // shared BatchNorm statistics of INDEPENDENT layers in one all-reduce (mlp2 || shortcut, mlp1 || LocSE-mlp1)
    ps::PackCache pack;  // the step's weight images (recorded during the first step, then packed by one launch per step: common.h)
    // inverse indices of the step's gather tables (deterministic mode): built at their first use in the backward pass, kept to its end
    struct Inv {
        Tn offsets, src;
        int64_t n_dst;
        const int32_t* order = nullptr;  // the destinations' spatially coherent processing order (ps_pyramid.order of their level), or NULL
        int64_t n_cloud = 0;
    };
    std::map<const int32_t*, Inv> inv_cache;
    std::map<const int32_t*, const int32_t*> order_of;  // gather table -> leaf order of the level it gathers FROM (set per step from the pyramid)
    void reduce_rows(const Inv& iv, const float* rows, int64_t ldr, int64_t d, float* dst, int64_t ldd, int accumulate)
    {
        TK(ps_op_gather_reduce_rows_ordered(c, rows, ldr, reinterpret_cast<const int32_t*>(iv.offsets.p), reinterpret_cast<const int32_t*>(iv.src.p), iv.n_dst, d, dst,
                                            ldd, accumulate, iv.order, iv.n_cloud));
    }
    const Inv& inverse(const int32_t* idx, int64_t B, int64_t N, int64_t rows_per_cloud)
    {
        auto it = inv_cache.find(idx);
        if (it != inv_cache.end()) return it->second;
        Inv v;
        v.n_dst = B * N;
        v.n_cloud = N;
        auto oo = order_of.find(idx);
        v.order = oo != order_of.end() ? oo->second : nullptr;
        v.offsets = alloc(1, v.n_dst + 1, false);
        v.src = alloc(1, std::max<int64_t>(B * rows_per_cloud, 1), false);
        Tn ws = alloc(1, ps_op_inverse_index_workspace(v.n_dst, B * rows_per_cloud), false);
        TK(ps_op_inverse_index(c, idx, B, N, rows_per_cloud, reinterpret_cast<int32_t*>(v.offsets.p), reinterpret_cast<int32_t*>(v.src.p),
                               reinterpret_cast<int32_t*>(ws.p)));
        return inv_cache.emplace(idx, v).first->second;
    }
    // Is sub_idx the prefix of neigh_idx (the precondition of the fixed-order max-pool backward)?  A pyramid ps_pyramid_build wrote says
    // so itself (ps_pyramid.built).  Any other pyramid is compared on EVERY step -- its tables may have been rewritten in place, or a
    // freed table's address handed to another pyramid of the same shape by the caller's allocator -- and only a NEGATIVE answer is
    // remembered (key: both pointers and the shape): it selects the float-atomic form, which is correct for every table.
    std::map<std::array<int64_t, 6>, bool> prefix_checked;
    bool pyramid_vouched = false;  // (set per step: pyr->built == pyramid_stamp(pyr))
    bool pool_is_prefix(const int32_t* pool_idx, const int32_t* neigh, int64_t B, int64_t N, int64_t M, int64_t K)
    {
        if (pool_idx == neigh && B == 1) return true;
        if (pyramid_vouched) return true;
        const std::array<int64_t, 6> key = {(int64_t)reinterpret_cast<uintptr_t>(pool_idx), (int64_t)reinterpret_cast<uintptr_t>(neigh), B, N, M, K};
        auto it = prefix_checked.find(key);
        if (it != prefix_checked.end() && !it->second) return false;
        Tn flag = alloc(1, 1, false);
        TK_HIP(hipMemsetAsync(flag.p, 0, sizeof(int32_t), stream()));
        hipLaunchKernelGGL(tr_prefix_check_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(B * M * K, 256), 2048)), dim3(256), 0, stream(), pool_idx, neigh, B, N,
                           M, K, reinterpret_cast<int32_t*>(flag.p));
        TK_HIP(hipGetLastError());
        int32_t h = 0;
        TK_HIP(hipMemcpyAsync(&h, flag.p, sizeof(int32_t), hipMemcpyDeviceToHost, stream()));
        TK_HIP(hipStreamSynchronize(stream()));
        if (h != 0) prefix_checked[key] = false;
        return h == 0;
    }
    std::vector<WgradJob> wjobs;  // weight / bias gradient partials waiting for the step's one reduction launch
    std::vector<Tn> wkeep;
    void finish_wgrads()
    {
        join_side();  // (the partial products may still be running on the second stream)
        if (wjobs.empty()) return;
        Stage st(c, "train_wgrad", 1);
        Tn table = alloc(1, (int64_t)((sizeof(WgradJob) * wjobs.size() + 3) / 4), false);
        TK(c->upload_async(table.p, wjobs.data(), sizeof(WgradJob) * wjobs.size()));
        int64_t most = 1;
        for (const WgradJob& j : wjobs) most = std::max<int64_t>(most, (int64_t)j.rows * j.cols);
        if (c->tune.wgrad_debug && step < 1)
            for (const WgradJob& j : wjobs)
                fprintf(stderr, "wgrad job: slabs %d rows %d cols %d transposed %d part%%16 %d dst%%16 %d\n", j.slabs, j.rows, j.cols, j.transposed,
                        (int)(reinterpret_cast<uintptr_t>(j.part) & 15), (int)(reinterpret_cast<uintptr_t>(j.dst) & 15));
        TK(wgrad_finish(c, reinterpret_cast<const WgradJob*>(table.p), (int)wjobs.size(), most));
        wjobs.clear();
        wkeep.clear();
    }

    hipStream_t stream() const { return c->stream; }
    // tells the storage-aware ops (ps_set_train_act_bf16) that the [N*K, h] rows of this call are bfloat16, for the lifetime of the object
    struct ActScope {
        ps_context* c;
        bool was;
        ActScope(ps_context* ctx, bool on) : c(ctx), was(ctx->train_act_bf16) { c->train_act_bf16 = on; }
        ~ActScope() { c->train_act_bf16 = was; }
    };

    // ---- tensors
    Tn alloc(int64_t R, int64_t C, bool req = true)
    {
        Tn t;
        t.R = R; t.C = C; t.ld = C;
        t.id = next_id++;
        t.req = req;
        void* p = pool.alloc(sizeof(float) * (size_t)std::max<int64_t>(R * C, 1));
        t.own = std::make_shared<Block>(&pool, p);
        t.p = static_cast<float*>(p);
        return t;
    }
    Tn zeros(int64_t R, int64_t C)
    {
        Tn t = alloc(R, C);
        TK_HIP(hipMemsetAsync(t.p, 0, sizeof(float) * (size_t)(R * C), stream()));
        return t;
    }
    Tn cols(const Tn& t, int64_t c0, int64_t nc, bool fresh_id = true)
    {
        Tn v = t;
        v.p = t.p + c0;
        v.C = nc;
        if (fresh_id) v.id = next_id++;
        return v;
    }
    Tn rows_of(const Tn& t, int64_t r0, int64_t nr)  // (row blocks of a parameter matrix: contiguous)
    {
        Tn v = t;
        v.p = t.p + r0 * t.ld;
        v.R = nr;
        v.id = next_id++;
        return v;
    }
    Tn external(float* p, int64_t R, int64_t C, bool req)
    {
        Tn t;
        t.p = p; t.R = R; t.C = C; t.ld = C; t.id = next_id++; t.req = req;
        return t;
    }
    void copy2d(const Tn& src, const Tn& dst, bool add)
    {
        const int64_t n = src.R * src.C;
        if (!n) return;
        Stage st(c, "train_copies", 1);
        if (add)
            hipLaunchKernelGGL(tr_copy2d_kernel<true>, dim3(tr_grid(n)), dim3(256), 0, stream(), src.p, src.ld, dst.p, dst.ld, n, (int)src.C);
        else
            hipLaunchKernelGGL(tr_copy2d_kernel<false>, dim3(tr_grid(n)), dim3(256), 0, stream(), src.p, src.ld, dst.p, dst.ld, n, (int)src.C);
        TK_HIP(hipGetLastError());
    }
    Tn contig(const Tn& t)
    {
        if (t.contiguous()) return t;
        Tn o = alloc(t.R, t.C, t.req);
        copy2d(t, o, false);
        return o;
    }
    Tn transpose(const Tn& w)  // [r, c] contiguous -> fresh [c, r]
    {
        Tn o = alloc(w.C, w.R, false);
        Stage st(c, "train_copies", 1);
        hipLaunchKernelGGL(tr_transpose_kernel, dim3(ceil_div(w.C, 32), ceil_div(w.R, 32)), dim3(256), 0, stream(), w.p, (int)w.R, (int)w.C, o.p);
        TK_HIP(hipGetLastError());
        return o;
    }
    void transpose_into(const Tn& src, float* dst)
    {
        Stage st(c, "train_copies", 1);
        hipLaunchKernelGGL(tr_transpose_kernel, dim3(ceil_div(src.C, 32), ceil_div(src.R, 32)), dim3(256), 0, stream(), src.p, (int)src.R, (int)src.C, dst);
        TK_HIP(hipGetLastError());
    }
    // a world of ONE rank is the identity: the callback is skipped (and the BatchNorm layers keep their cheaper one-rank form) unless
    // ps_trainer_set_collective asked for it with PS_COLLECTIVE_AT_WORLD_ONE -- bench.py's measurement of the host collective's floor
    bool coll_active() const { return coll != nullptr && (world > 1 || coll_at_one); }
    void allreduce(void* buf, int64_t count, int dtype)
    {
        if (!coll_active()) return;
        struct EventPair {  // (owned until handed to coll_marks: nothing leaks when a HIP call in between throws)
            hipEvent_t e0 = nullptr, e1 = nullptr;
            ~EventPair()
            {
                if (e0) (void)hipEventDestroy(e0);
                if (e1) (void)hipEventDestroy(e1);
            }
        } ev;
        if (profile) {
            TK_HIP(hipEventCreate(&ev.e0));
            TK_HIP(hipEventCreate(&ev.e1));
            TK_HIP(hipEventRecord(ev.e0, stream()));
        }
        const auto h0 = std::chrono::steady_clock::now();
        const int rc = coll(coll_user, buf, count, dtype, (void*)stream());
        coll_host_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - h0).count();
        coll_calls += 1;
        coll_bytes += count * (dtype == 1 ? 8 : 4);
        if (profile) {
            (void)hipEventRecord(ev.e1, stream());
            coll_marks.emplace_back(ev.e0, ev.e1);
            ev.e0 = ev.e1 = nullptr;
        }
        if (rc != 0) {
            ps::set_error("the host's all-reduce callback failed with code %d", rc);
            throw TrainError{PS_ESTATE};
        }
    }

    // ---- gradient bookkeeping (keys = tensor ids, like the identity keys of a Python tape)
    void accum(const Tn& t, const Tn& g)
    {
        auto it = grad_of.find(t.id);
        if (it == grad_of.end()) {
            grad_of[t.id] = g;
            return;
        }
        Tn& have = it->second;
        if (have.b16 || g.b16) {  // (bfloat16-stored gradient rows are only ever accumulated by the ops that read them that way)
            ps::set_error("trainer: a second gradient of a bfloat16-stored tensor reached the generic accumulation");
            throw TrainError{PS_ESTATE};
        }
        if (have.contiguous() && g.contiguous())
            TK(ps_op_axpy(c, 1.0f, g.p, g.numel(), have.p));
        else
            copy2d(g, have, true);
    }
    // the gradient buffer of t for ops that ADD into their output (scatter-add, max-pool backward): the one already recorded, or a
    // fresh zero tensor that becomes it
    Tn accum_buffer(const Tn& t)
    {
        auto it = grad_of.find(t.id);
        if (it != grad_of.end()) return it->second;
        Tn z = zeros(t.R, t.C);
        grad_of[t.id] = z;
        return z;
    }
    void record(const Tn& out, std::function<void(const Tn&)> bw) { ops.push_back({out.id, section, std::move(bw)}); }
    // runs the input-gradient products that were parked on tensor `id` (linear(..., defer_dgrad)): called once the consumer they waited
    // for has stored its own gradient, and in any case before the producer of `id` runs its backward
    void run_deferred(int id)
    {
        auto it = deferred.find(id);
        if (it == deferred.end()) return;
        std::vector<std::function<void()>> fs = std::move(it->second);
        deferred.erase(it);
        for (auto& f : fs) f();
    }
    void backward(const Tn& out, const Tn& dout)
    {
        grad_of[out.id] = dout;
        int cur = -1;
        for (size_t i = ops.size(); i-- > 0;) {
            if (profile && ops[i].section != cur) {
                cur = ops[i].section;
                mark("bwd " + section_names[cur]);
            }
            run_deferred(ops[i].out_id);
            auto it = grad_of.find(ops[i].out_id);
            if (it == grad_of.end()) {
                ops[i].bw = nullptr;
                continue;
            }
            Tn g = it->second;
            grad_of.erase(it);
            ops[i].bw(g);
            ops[i].bw = nullptr;  // drops the activations only this op held
        }
        ops.clear();
        grad_of.clear();
        deferred.clear();
        inv_cache.clear();
        finish_wgrads();
    }

    // ---- parameters
    const LayerP& layer(const std::string& scope) const
    {
        auto it = by_scope.find(scope);
        if (it == by_scope.end()) {
            ps::set_error("trainer: no layer named %s", scope.c_str());
            throw TrainError{PS_EINVAL};
        }
        return layers[it->second];
    }
    Tn P(int64_t off, int64_t R, int64_t C) { return external(params + off, R, C, false); }
    Tn G(int64_t off, int64_t R, int64_t C) { return external(grads + off, R, C, false); }

    // ---- ops (each: forward kernels now, a closure for the backward pass) -----------------------------------------------------------
    // y = x . W (+ b).  W is [cin, cout], or [cout, cin] when transposed (conv2d_transpose kernels, helper_tf_util.py:208-212).
    // into: an existing [R, cout] tensor of the tape that the product is ADDED to (the GEMM's accumulate epilogue); the result is that
    // same tensor and its gradient is handed on unchanged to the op that produced it.
    // fp32_only: this layer is not one of the "bf16 MLPs" (the LocSE convolution 10 -> h: part of the position encoding, K = 10 is no
    // matrix-pipe shape, and its fused form computes in fp32): its three GEMMs keep fp32 operands in the bf16 mode too.
    // defer_dgrad: x has a later-running consumer whose backward STORES its gradient of x (the fused attention kernels); this layer's
    // input-gradient GEMM then waits for that store and adds into it in its epilogue (a streaming read-modify-write) instead of making the
    // attention kernel read-modify-write a tensor this GEMM wrote first
    Tn linear(const Tn& x, const Tn& W, const float* b, const Tn& gW, float* gb, bool transposed = false, const Tn* into = nullptr, bool fp32_only = false,
              bool defer_dgrad = false)
    {
        struct Fp32Scope {  // switches the context's bf16-GEMM mode off for the lifetime of the object
            ps_context* c;
            bool was;
            Fp32Scope(ps_context* ctx, bool on) : c(ctx), was(ctx->train_bf16) { if (on) c->train_bf16 = false; }
            ~Fp32Scope() { c->train_bf16 = was; }
        };
        struct WtScope {  // tells ps_op_conv1x1_ex that its weight matrix is stored [cout, cin] (read through strides while packing)
            ps_context* c;
            WtScope(ps_context* ctx, bool on) : c(ctx) { c->conv_w_transposed = on; }
            ~WtScope() { c->conv_w_transposed = false; }
        };
        Fp32Scope fwd_scope(c, fp32_only);
        const int64_t R = x.R, cin = x.C, cout = transposed ? W.R : W.C;
        Tn y;
        {
            WtScope wt(c, transposed);  // conv2d_transpose kernels are stored [out, in]
            if (!into) {
                y = alloc(R, cout);
                TK(ps_op_conv1x1_ex(c, x.p, x.ld, W.p, b, R, cin, cout, 0, 0, y.p, y.ld));
            } else {
                y = *into;
                TK(ps_op_conv1x1_ex(c, x.p, x.ld, W.p, b, R, cin, cout, 0, 1, y.p, y.ld));
            }
        }
        const bool had_into = into != nullptr;
        const Tn into_t = had_into ? *into : Tn();
        record(y, [=](const Tn& dy) {
            Fp32Scope bwd_scope(c, fp32_only);
            if (had_into) grad_of[into_t.id] = dy;  // d(into + x.W)/d(into) = 1: the producer of `into` (earlier on the tape) gets the same gradient
            {
                // weight / bias gradient: per-slab partials now (plain stores), summed in slab order by the ONE wgrad_finish launch at the
                // end of the backward pass -- deterministic, no memsets, and the [out, in] layout of the transposed kernels is just a flag
                const int64_t nb = wgrad_partial_slabs(c, x.p, x.ld, dy.p, dy.ld, R, cin, cout);
                Tn part = alloc(nb, cin * cout, false);
                Tn dbp = gb ? alloc(nb, cout, false) : Tn();
                {
                    struct StreamScope {  // the product is enqueued on the second stream: the context's stream is what every launch below uses
                        ps_context* c;
                        hipStream_t was;
                        StreamScope(ps_context* ctx, hipStream_t s) : c(ctx), was(ctx->stream) { if (s) c->stream = s; }
                        ~StreamScope() { c->stream = was; }
                    };
                    const bool on_side = use_side();
                    if (on_side) fork_to_side();
                    StreamScope ss(c, on_side ? side : nullptr);
                    Stage st(c, "train_wgrad", 1);
                    TK(wgrad_partial(c, x.p, x.ld, dy.p, dy.ld, R, cin, cout, part.p, gb ? dbp.p : nullptr));
                    if (on_side) {
                        wkeep.push_back(x);
                        wkeep.push_back(dy);
                    }
                }
                wjobs.push_back(WgradJob{part.p, gW.p, (int)nb, (int)cin, (int)cout, transposed ? 1 : 0});
                wkeep.push_back(part);
                if (gb) {
                    wjobs.push_back(WgradJob{dbp.p, gb, (int)nb, 1, (int)cout, 0});
                    wkeep.push_back(dbp);
                }
            }
            if (x.req) {
                // dx = dy . W^T: the GEMM's [cout, cin] matrix IS the stored kernel for the transposed layers and its transpose for all others
                auto dgrad = [=]() {
                    Fp32Scope scope(c, fp32_only);
                    WtScope wt(c, !transposed);
                    auto it = grad_of.find(x.id);
                    if (it != grad_of.end()) {
                        // x already has a gradient from another consumer: add this one in the GEMM epilogue
                        Tn& have = it->second;
                        TK(ps_op_conv1x1_ex(c, dy.p, dy.ld, W.p, nullptr, R, cout, cin, 0, 1, have.p, have.ld));
                    } else {
                        Tn dx = alloc(R, cin);
                        TK(ps_op_conv1x1_ex(c, dy.p, dy.ld, W.p, nullptr, R, cout, cin, 0, 0, dx.p, dx.ld));
                        accum(x, dx);
                    }
                };
                if (defer_dgrad && !grad_of.count(x.id))
                    deferred[x.id].push_back(dgrad);
                else
                    dgrad();
            }
        });
        return y;
    }

    // y = act(BN_train(x)); out: optional [R, C] column block of a wider tensor (rows contiguous) that receives y
    Tn bn_act(const Tn& x_in, const LayerP& lp, bool leaky, const Tn* out = nullptr)
    {
        const Tn x = contig(x_in);
        const int64_t R = x.R, C = x.C;
        Tn y = out ? *out : alloc(R, C);
        y.req = true;
        Tn stats = alloc(5, C, false);  // mean, invstd, var, [sum x | sum x^2]
        float *mean = stats.p, *invstd = stats.p + C, *var = stats.p + 2 * C, *sums = stats.p + 3 * C;
        const float *gamma = params + lp.gamma, *beta = params + lp.beta;
        const bool sync = sync_bn && coll_active();
        const int64_t R_total = sync ? R * world : R;
        if (!sync) {
            // statistics, moving-statistics update and the apply pass: three launches
            TK(ps_op_bn_train_fwd_mov(c, x.p, gamma, beta, R, C, kBnEps, leaky ? 1 : 0, y.p, y.ld, mean, invstd, var, sums, buffers + lp.mov_mean,
                                      buffers + lp.mov_var, kBnMomentum));
        } else {
            // statistics over the rows of ALL ranks: two small all-reduces per layer (2*C floats forward, 2*C backward)
            TK(ps_op_bn_train_sums(c, x.p, R, C, sums));
            allreduce(sums, 2 * C, 0);
            TK(ps_op_bn_train_apply_ex(c, x.p, gamma, beta, sums, R, R_total, C, kBnEps, leaky ? 1 : 0, y.p, y.ld, mean, invstd, var));
        }
        if (sync) {
            Stage st(c, "train_bn_fwd", 1);
            hipLaunchKernelGGL(tr_ema2_kernel, dim3(ceil_div(C, 64)), dim3(64), 0, stream(), buffers + lp.mov_mean, buffers + lp.mov_var, mean, var, (int)C,
                               kBnMomentum);
            TK_HIP(hipGetLastError());
        }
        float *ggamma = grads + lp.gamma, *gbeta = grads + lp.beta;
        const Tn xin = x_in;
        record(y, [=](const Tn& dy) {
            Tn dx = alloc(R, C);
            if (!sync) {
                TK(ps_op_bn_train_bwd_ex(c, dy.p, dy.ld, x.p, gamma, beta, mean, invstd, R, C, leaky ? 1 : 0, dx.p, ggamma, gbeta));
            } else {
                // local sums are this rank's dgamma / dbeta (averaged with every other gradient later); dx needs the global ones
                TK(ps_op_bn_train_bwd_sums_ex(c, dy.p, dy.ld, x.p, gamma, beta, mean, invstd, R, C, leaky ? 1 : 0, ggamma, gbeta));
                Tn tot = alloc(2, C, false);
                hipLaunchKernelGGL(tr_pair_copy_kernel, dim3(ceil_div(C, 64)), dim3(64), 0, stream(), gbeta, ggamma, (int)C, tot.p);
                TK_HIP(hipGetLastError());
                allreduce(tot.p, 2 * C, 0);
                TK(ps_op_bn_train_bwd_apply_ex(c, dy.p, dy.ld, x.p, gamma, beta, mean, invstd, tot.p, tot.p + C, R, R_total, C, leaky ? 1 : 0, dx.p));
            }
            (void)stats;  // (mean / invstd live in it)
            accum(xin, dx);
        });
        return y;
    }

    // ---- shared statistics of INDEPENDENT BatchNorm layers in one all-reduce (SyncBN, world > 1) ---------------------------------------
    // bn_act's shared-statistics form cut in two: bn_begin leaves this rank's [sum | sum x^2] in p.sums(); whoever all-reduces them (a
    // companion layer's call) is followed by bn_end: apply pass, moving statistics, backward closure (its own all-reduce of the 2 C
    // backward sums -- the backward of the two layers does not run at the same time).
    struct PendingBn {
        Tn x, x_in, y, stats, shared;  // shared: a [2 Ca + 2 Cb] buffer two pending layers put their sums into (kept alive by both)
        const LayerP* lp = nullptr;
        float* ext = nullptr;          // this layer's 2 C sums inside `shared`
        bool leaky = false, open = false;
        float* sums() const { return ext ? ext : stats.p + 3 * x.C; }
    };
    PendingBn bn_begin(const Tn& x_in, const LayerP& lp, bool leaky, const Tn* out = nullptr, const Tn* shared = nullptr, int64_t shared_off = 0)
    {
        PendingBn p;
        p.x_in = x_in;
        p.x = contig(x_in);
        p.y = out ? *out : alloc(p.x.R, p.x.C);
        p.y.req = true;
        p.stats = alloc(5, p.x.C, false);
        if (shared) {
            p.shared = *shared;
            p.ext = shared->p + shared_off;
        }
        p.lp = &lp;
        p.leaky = leaky;
        p.open = true;
        TK(ps_op_bn_train_sums(c, p.x.p, p.x.R, p.x.C, p.sums()));
        return p;
    }
    Tn bn_end(PendingBn& p)
    {
        const LayerP& lp = *p.lp;
        const Tn x = p.x, xin = p.x_in, y = p.y, stats = p.stats;
        const int64_t R = x.R, C = x.C, R_total = R * world;
        const bool leaky = p.leaky;
        float *mean = stats.p, *invstd = stats.p + C, *var = stats.p + 2 * C;
        const float *gamma = params + lp.gamma, *beta = params + lp.beta;
        TK(ps_op_bn_train_apply_ex(c, x.p, gamma, beta, p.sums(), R, R_total, C, kBnEps, leaky ? 1 : 0, y.p, y.ld, mean, invstd, var));
        {
            Stage st(c, "train_bn_fwd", 1);
            hipLaunchKernelGGL(tr_ema2_kernel, dim3(ceil_div(C, 64)), dim3(64), 0, stream(), buffers + lp.mov_mean, buffers + lp.mov_var, mean, var, (int)C,
                               kBnMomentum);
            TK_HIP(hipGetLastError());
        }
        float *ggamma = grads + lp.gamma, *gbeta = grads + lp.beta;
        record(y, [=](const Tn& dy) {
            Tn dx = alloc(R, C);
            TK(ps_op_bn_train_bwd_sums_ex(c, dy.p, dy.ld, x.p, gamma, beta, mean, invstd, R, C, leaky ? 1 : 0, ggamma, gbeta));
            Tn tot = alloc(2, C, false);
            hipLaunchKernelGGL(tr_pair_copy_kernel, dim3(ceil_div(C, 64)), dim3(64), 0, stream(), gbeta, ggamma, (int)C, tot.p);
            TK_HIP(hipGetLastError());
            allreduce(tot.p, 2 * C, 0);
            TK(ps_op_bn_train_bwd_apply_ex(c, dy.p, dy.ld, x.p, gamma, beta, mean, invstd, tot.p, tot.p + C, R, R_total, C, leaky ? 1 : 0, dx.p));
            (void)stats;
            accum(xin, dx);
        });
        p.open = false;
        p.shared = Tn();
        return y;
    }

    // LeakyReLU(BN(xa) + BN(xb)) of dilated_res_block (RandLANet.py:303-307: mlp2 and the shortcut) with shared statistics: the two layers
    // are independent and their output gradients are the SAME tensor, so both directions need ONE all-reduce for the pair -- [2 Ca + 2 Cb]
    // floats forward, the same backward -- instead of two each.
    Tn res_pair_sync(const Tn& xa_in, const LayerP& la, const Tn& xb_in, const LayerP& lb)
    {
        const Tn xa = contig(xa_in), xb = contig(xb_in);
        const int64_t R = xa.R, C = xa.C, R_total = R * world;
        if (xb.R != R || xb.C != C) {
            ps::set_error("trainer: res_pair_sync: the two branches differ in shape");
            throw TrainError{PS_ESTATE};
        }
        Tn st = alloc(10, C, false);  // per branch: mean | invstd | var, then the four sums [sum a | sum a^2 | sum b | sum b^2] contiguously
        float *mean_a = st.p, *invstd_a = st.p + C, *var_a = st.p + 2 * C, *mean_b = st.p + 3 * C, *invstd_b = st.p + 4 * C, *var_b = st.p + 5 * C;
        float* sums = st.p + 6 * C;
        const float *ga = params + la.gamma, *ba = params + la.beta, *gb = params + lb.gamma, *bb = params + lb.beta;
        TK(ps_op_bn_train_sums(c, xa.p, R, C, sums));
        TK(ps_op_bn_train_sums(c, xb.p, R, C, sums + 2 * C));
        allreduce(sums, 4 * C, 0);
        Tn ya = alloc(R, C), yb = alloc(R, C);
        TK(ps_op_bn_train_apply_ex(c, xa.p, ga, ba, sums, R, R_total, C, kBnEps, 0, ya.p, ya.ld, mean_a, invstd_a, var_a));
        TK(ps_op_bn_train_apply_ex(c, xb.p, gb, bb, sums + 2 * C, R, R_total, C, kBnEps, 0, yb.p, yb.ld, mean_b, invstd_b, var_b));
        {
            Stage stg(c, "train_bn_fwd", 2);
            hipLaunchKernelGGL(tr_ema2_kernel, dim3(ceil_div(C, 64)), dim3(64), 0, stream(), buffers + la.mov_mean, buffers + la.mov_var, mean_a, var_a, (int)C, kBnMomentum);
            hipLaunchKernelGGL(tr_ema2_kernel, dim3(ceil_div(C, 64)), dim3(64), 0, stream(), buffers + lb.mov_mean, buffers + lb.mov_var, mean_b, var_b, (int)C, kBnMomentum);
            TK_HIP(hipGetLastError());
        }
        Tn y = alloc(R, C);
        y.req = true;
        TK(ps_op_add_lrelu(c, ya.p, yb.p, y.numel(), y.p));
        float *gga = grads + la.gamma, *gba = grads + la.beta, *ggb = grads + lb.gamma, *gbb = grads + lb.beta;
        record(y, [=](const Tn& dy_in) {
            const Tn dy = contig(dy_in);
            Tn ds = alloc(R, C);
            TK(ps_op_add_lrelu_bwd(c, dy.p, y.p, y.numel(), ds.p));
            // this rank's dgamma / dbeta of both layers (averaged with every other gradient later); dx needs the global sums
            TK(ps_op_bn_train_bwd_sums_ex(c, ds.p, ds.ld, xa.p, ga, ba, mean_a, invstd_a, R, C, 0, gga, gba));
            TK(ps_op_bn_train_bwd_sums_ex(c, ds.p, ds.ld, xb.p, gb, bb, mean_b, invstd_b, R, C, 0, ggb, gbb));
            Tn tot = alloc(4, C, false);
            hipLaunchKernelGGL(tr_pair_copy_kernel, dim3(ceil_div(C, 64)), dim3(64), 0, stream(), gba, gga, (int)C, tot.p);
            hipLaunchKernelGGL(tr_pair_copy_kernel, dim3(ceil_div(C, 64)), dim3(64), 0, stream(), gbb, ggb, (int)C, tot.p + 2 * C);
            TK_HIP(hipGetLastError());
            allreduce(tot.p, 4 * C, 0);
            Tn dxa = alloc(R, C), dxb = alloc(R, C);
            TK(ps_op_bn_train_bwd_apply_ex(c, ds.p, ds.ld, xa.p, ga, ba, mean_a, invstd_a, tot.p, tot.p + C, R, R_total, C, 0, dxa.p));
            TK(ps_op_bn_train_bwd_apply_ex(c, ds.p, ds.ld, xb.p, gb, bb, mean_b, invstd_b, tot.p + 2 * C, tot.p + 3 * C, R, R_total, C, 0, dxb.p));
            (void)st; (void)ya; (void)yb;
            accum(xa_in, dxa);
            accum(xb_in, dxb);
        });
        return y;
    }

    // f_xyz = LeakyReLU(BN_train(relative_pos_encoding(xyz, idx) . W + b)) -> [B*N*K, h] (out: optional column block), with nothing but
    // that output in memory: statistics, output and every gradient are recomputed from xyz [B*N,3] and idx [B,N,K] (locse_train.hip)
    // companion: an open PendingBn of an independent layer whose 2 C float sums ride in this layer's all-reduce (as doubles, behind the 2 h)
    Tn locse_bn_act(const float* xyz, const int32_t* idx, int64_t B, int64_t N, int64_t K, const LayerP& lp, const Tn* out = nullptr, bool act16 = false,
                    PendingBn* companion = nullptr)
    {
        const int64_t h = lp.cout, R = B * N * K;
        const bool sync = sync_bn && coll_active();
        const int64_t R_total = sync ? R * world : R;
        const float *W = params + lp.w, *b = params + lp.b, *gamma = params + lp.gamma, *beta = params + lp.beta;
        const int64_t n_comp = (sync && companion && companion->open) ? 2 * companion->x.C : 0;
        Tn sums = alloc(1, 4 * h + 2 * n_comp, false);  // 2h doubles: the variance is a difference of nearly equal sums (+ the companion's sums)
        double* s64 = reinterpret_cast<double*>(sums.p);
        TK(ps_op_locse_train_sums(c, xyz, idx, B, N, K, W, b, h, s64));
        if (n_comp) {
            Stage st(c, "train_bn_fwd", 1);
            hipLaunchKernelGGL(tr_f2d_kernel, dim3(ceil_div(n_comp, 64)), dim3(64), 0, stream(), companion->sums(), (int)n_comp, s64 + 2 * h);
            TK_HIP(hipGetLastError());
        }
        if (sync) allreduce(s64, 2 * h + n_comp, 1);
        if (n_comp) {
            Stage st(c, "train_bn_fwd", 1);
            hipLaunchKernelGGL(tr_d2f_kernel, dim3(ceil_div(n_comp, 64)), dim3(64), 0, stream(), s64 + 2 * h, (int)n_comp, companion->sums());
            TK_HIP(hipGetLastError());
        }
        Tn st4 = alloc(4, h, false);  // mean | var | invstd | scale
        float *mean = st4.p, *invstd = st4.p + 2 * h, *scale = st4.p + 3 * h;
        {
            Stage st(c, "train_locse_fwd", 1);
            hipLaunchKernelGGL(tr_locse_stats_kernel, dim3(ceil_div(h, 64)), dim3(64), 0, stream(), s64, (double)R_total, gamma, (int)h, kBnEps, st4.p,
                               buffers + lp.mov_mean, buffers + lp.mov_var, kBnMomentum, (int)h);
            TK_HIP(hipGetLastError());
        }
        Tn y = out ? *out : alloc(R, h);
        y.req = true;
        y.b16 = act16;
        {
            ActScope as(c, act16);
            TK(ps_op_locse_train_apply(c, xyz, idx, B, N, K, W, b, h, mean, scale, beta, y.p, y.ld));
        }
        float *gW = grads + lp.w, *gb = grads + lp.b, *ggamma = grads + lp.gamma, *gbeta = grads + lp.beta;
        record(y, [=](const Tn& dz) {
            Tn acc = alloc(1, 23 * h + 16, false);
            if (dz.b16 != act16) {  // (the gradient rows of a bfloat16-stored tensor are bfloat16 themselves: attpool_split, conv_bn_fused)
                ps::set_error("trainer: locse_bn_act: gradient rows in the wrong storage format");
                throw TrainError{PS_ESTATE};
            }
            {
                ActScope as(c, act16);
                TK(ps_op_locse_train_bwd(c, xyz, idx, B, N, K, W, b, h, scale, beta, mean, invstd, dz.p, dz.ld, acc.p));
            }
            Tn tot = alloc(2, h, false);
            Stage st(c, "train_locse_bwd", 2);
            hipLaunchKernelGGL(tr_locse_local_kernel, dim3(ceil_div(h, 64)), dim3(64), 0, stream(), acc.p, (int)h, ggamma, gbeta, tot.p);
            TK_HIP(hipGetLastError());
            if (sync) allreduce(tot.p, 2 * h, 0);
            hipLaunchKernelGGL(tr_locse_wgrad_kernel, dim3(ceil_div(h, 64)), dim3(64), 0, stream(), acc.p, tot.p, gamma, invstd, (int)h, (float)R,
                               1.0f / (float)R_total, gW, gb);
            TK_HIP(hipGetLastError());
            (void)st4;
        });
        return y;
    }

    // LFA mlp2 on the [N*K, h] rows (conv h -> h + BatchNorm + LeakyReLU, RandLANet.py:331) with the pre-BatchNorm product recomputed from x
    // wherever it is needed instead of stored (smallconv_train.hip; h = 8: convbn_rows.hip): 3 + 6 passes over [rows, h] tensors instead of
    // 5 + 10.  defer_dgrad as in linear(): the input-gradient pass waits for the other consumer's plain store and adds into it.
    bool convbn_fused_ok(const Tn& x, const LayerP& lp) const
    {
        if (!opt.fused_convbn || lp.cin != lp.cout || !ps_op_conv_bn_train_supported(lp.cout) || lp.kind == kDeconv || lp.b < 0) return false;
        // (bf16-MLP mode: the tile kernels round the operands of their three products like the GEMMs they replace -- cin % 16 == 0 --, the
        //  8-channel layer stays fp32 in both forms)
        const int max_c = c->tune.convbn_max_c;  // 64 (A/B knob of the experiments in DESIGN.md)
        if (lp.cout > max_c) return false;
        return x.ld % 4 == 0 && (reinterpret_cast<uintptr_t>(x.p) & 15) == 0;  // (a column block of a concat buffer is fine)
    }
    Tn conv_bn_fused(const Tn& x, const LayerP& lp, bool defer_dgrad, const Tn* out = nullptr)
    {
        const int64_t R = x.R, h = lp.cout, CP = h < 16 ? 16 : h;
        const bool sync = sync_bn && coll_active();
        const int64_t R_total = sync ? R * world : R;
        const float *W = params + lp.w, *b = params + lp.b, *gamma = params + lp.gamma, *beta = params + lp.beta;
        Tn sums = alloc(1, 6 * CP, false);  // 3 CP doubles: sum y | sum y^2 | sum x
        double* s64 = reinterpret_cast<double*>(sums.p);
        const bool act16 = x.b16;  // (bfloat16 rows in, bfloat16 rows out)
        {
            ActScope as(c, act16);
            TK(ps_op_conv_bn_train_sums(c, x.p, x.ld, W, b, R, h, s64));
        }
        if (sync) allreduce(s64, 2 * CP, 1);
        Tn st4 = alloc(4, h, false);  // mean | var | invstd | scale
        float *mean = st4.p, *invstd = st4.p + 2 * h, *scale = st4.p + 3 * h;
        {
            Stage st(c, "train_convbn_fwd", 1);
            hipLaunchKernelGGL(tr_locse_stats_kernel, dim3(ceil_div(h, 64)), dim3(64), 0, stream(), s64, (double)R_total, gamma, (int)h, kBnEps, st4.p,
                               buffers + lp.mov_mean, buffers + lp.mov_var, kBnMomentum, (int)CP);
            TK_HIP(hipGetLastError());
        }
        Tn z = out ? *out : alloc(R, h);
        z.req = true;
        z.b16 = act16;
        {
            ActScope as(c, act16);
            TK(ps_op_conv_bn_train_apply(c, x.p, x.ld, W, b, R, h, mean, scale, beta, z.p, z.ld));
        }
        float *gW = grads + lp.w, *gb = grads + lp.b, *ggamma = grads + lp.gamma, *gbeta = grads + lp.beta;
        record(z, [=](const Tn& dz_in) {
            const bool dz_ok = dz_in.ld % 4 == 0 && (reinterpret_cast<uintptr_t>(dz_in.p) & 15) == 0;
            if (dz_in.b16 != act16 || (act16 && !dz_ok)) {  // (bfloat16 rows in and out: so are the gradient rows dz and dx)
                ps::set_error("trainer: conv_bn_fused: gradient rows in the wrong storage format");
                throw TrainError{PS_ESTATE};
            }
            const Tn dz = dz_ok ? dz_in : contig(dz_in);
            Tn acc = alloc(1, 3 * h, false);
            {
                ActScope as(c, act16);
                TK(ps_op_conv_bn_train_bwd_sums2(c, x.p, x.ld, W, b, R, h, mean, invstd, scale, beta, dz.p, dz.ld, acc.p));
            }
            Tn tot = alloc(2, h, false);
            {
                Stage st(c, "train_convbn_bwd", 1);
                hipLaunchKernelGGL(tr_locse_local_kernel, dim3(ceil_div(h, 64)), dim3(64), 0, stream(), acc.p, (int)h, ggamma, gbeta, tot.p);
                TK_HIP(hipGetLastError());
            }
            if (sync) allreduce(tot.p, 2 * h, 0);
            auto apply = [=]() {
                auto it = grad_of.find(x.id);
                const bool add = it != grad_of.end();
                Tn dx = add ? it->second : alloc(R, h);
                if (!add) dx.b16 = act16;
                if (add && !(dx.ld % 4 == 0 && dx.R == R && dx.C == h && dx.b16 == act16)) {
                    ps::set_error("trainer: conv_bn_fused: unexpected gradient layout");
                    throw TrainError{PS_ESTATE};
                }
                {
                    ActScope as(c, act16);
                    TK(ps_op_conv_bn_train_bwd_apply_w(c, x.p, x.ld, W, b, R, h, mean, invstd, scale, beta, tot.p, 1.0f / (float)R_total, dz.p, dz.ld, add ? 1 : 0,
                                                       dx.p, dx.ld, gW, gb));
                }
                if (!add) accum(x, dx);
                (void)st4;
            };
            if (!x.req) {
                ps::set_error("trainer: conv_bn_fused on an input without gradient");
                throw TrainError{PS_ESTATE};
            }
            if (defer_dgrad && !grad_of.count(x.id))
                deferred[x.id].push_back(apply);
            else
                apply();
        });
        z.req = true;
        return z;
    }

    // The widening shared MLPs (Encoder mlp2 / shortcut, fc1) in the same recompute form (rectconv_train.hip): cin -> cout with cout > cin,
    // 5 cin + 3 cout row passes instead of 3 cin + 11 cout.
    bool convbn_rect_ok(const Tn& x, const LayerP& lp) const
    {
        if (!opt.fused_convbn || lp.kind == kDeconv || lp.b < 0 || lp.gamma < 0 || !ps_op_convbn_train_supported(lp.cin, lp.cout)) return false;
        const int max_w = c->tune.convbn_rect_max;  // (A/B knob: cin * cout)
        if (lp.cin * lp.cout > max_w) return false;
        return x.ld % 4 == 0 && (reinterpret_cast<uintptr_t>(x.p) & 15) == 0;
    }
    // addend (optional): z = LeakyReLU(BN(x . W + b) + addend) -- the residual sum of dilated_res_block in the pass that writes the second
    // summand (ps_op_convbn_train_apply_add); `leaky` must be false then.  Its backward first forms ds = dz lrelu'(z), which is the
    // gradient of the addend AND of this layer's BatchNorm output.
    Tn conv_bn_rect(const Tn& x, const LayerP& lp, bool leaky, const Tn* addend_in = nullptr)
    {
        const bool fused_add = addend_in != nullptr;
        const Tn addend = fused_add ? *addend_in : Tn();
        const int64_t R = x.R, ci = lp.cin, co = lp.cout;
        const bool sync = sync_bn && coll_active();
        const int64_t R_total = sync ? R * world : R;
        const float *W = params + lp.w, *b = params + lp.b, *gamma = params + lp.gamma, *beta = params + lp.beta;
        Tn sums = alloc(1, 4 * co, false);  // 2 cout doubles: sum y | sum y^2
        double* s64 = reinterpret_cast<double*>(sums.p);
        TK(ps_op_convbn_train_sums(c, x.p, x.ld, W, b, R, ci, co, s64));
        if (sync) allreduce(s64, 2 * co, 1);
        Tn st4 = alloc(4, co, false);  // mean | var | invstd | scale
        float *mean = st4.p, *invstd = st4.p + 2 * co, *scale = st4.p + 3 * co;
        {
            Stage st(c, "train_convbn_fwd", 1);
            hipLaunchKernelGGL(tr_locse_stats_kernel, dim3(ceil_div(co, 64)), dim3(64), 0, stream(), s64, (double)R_total, gamma, (int)co, kBnEps, st4.p,
                               buffers + lp.mov_mean, buffers + lp.mov_var, kBnMomentum, (int)co);
            TK_HIP(hipGetLastError());
        }
        Tn z = alloc(R, co);
        if (fused_add)
            TK(ps_op_convbn_train_apply_add(c, x.p, x.ld, W, b, R, ci, co, mean, scale, beta, addend.p, addend.ld, z.p, z.ld));
        else
            TK(ps_op_convbn_train_apply(c, x.p, x.ld, W, b, R, ci, co, mean, scale, beta, leaky ? 1 : 0, z.p, z.ld));
        float *gW = grads + lp.w, *gb = grads + lp.b, *ggamma = grads + lp.gamma, *gbeta = grads + lp.beta;
        record(z, [=](const Tn& dz_raw) {
            Tn dz_in = dz_raw;
            if (fused_add) {
                // ds = dz lrelu'(z): the gradient of both summands (add_lrelu's backward); the addend has no other consumer in this graph,
                // so it takes the tensor itself when it holds no gradient yet
                const Tn dy = contig(dz_raw);
                Tn ds = alloc(R, co);
                TK(ps_op_add_lrelu_bwd(c, dy.p, z.p, z.numel(), ds.p));
                if (grad_of.count(addend.id)) {
                    Tn ds2 = alloc(R, co);
                    TK_HIP(hipMemcpyAsync(ds2.p, ds.p, sizeof(float) * (size_t)z.numel(), hipMemcpyDeviceToDevice, stream()));
                    accum(addend, ds2);
                } else {
                    grad_of[addend.id] = ds;
                }
                dz_in = ds;
            }
            const bool dz_ok = dz_in.ld % 4 == 0 && (reinterpret_cast<uintptr_t>(dz_in.p) & 15) == 0;
            const Tn dz = dz_ok ? dz_in : contig(dz_in);
            Tn acc = alloc(1, 2 * co, false);
            TK(ps_op_convbn_train_bwd_sums(c, x.p, x.ld, W, b, R, ci, co, mean, invstd, scale, beta, leaky ? 1 : 0, dz.p, dz.ld, acc.p));
            Tn tot = alloc(2, co, false);
            {
                Stage st(c, "train_convbn_bwd", 1);
                hipLaunchKernelGGL(tr_locse_local_kernel, dim3(ceil_div(co, 64)), dim3(64), 0, stream(), acc.p, (int)co, ggamma, gbeta, tot.p);
                TK_HIP(hipGetLastError());
            }
            if (sync) allreduce(tot.p, 2 * co, 0);
            Tn dx;
            int add = 0;
            if (x.req) {
                auto it = grad_of.find(x.id);
                add = it != grad_of.end() && it->second.ld % 4 == 0 && (reinterpret_cast<uintptr_t>(it->second.p) & 15) == 0 ? 1 : 0;
                dx = add ? it->second : alloc(R, ci);
            }
            TK(ps_op_convbn_train_bwd_apply(c, x.p, x.ld, W, b, R, ci, co, mean, invstd, scale, beta, leaky ? 1 : 0, tot.p, 1.0f / (float)R_total, dz.p, dz.ld, add,
                                            x.req ? dx.p : nullptr, x.req ? dx.ld : 0, gW, gb));
            if (x.req && !add) accum(x, dx);
            (void)st4;
        });
        return z;
    }

    // x [B*N, d], idx [B, M, K] -> [B*M*K, d]; out: optional column block of a wider tensor that receives the rows
    Tn gather(const Tn& x_in, const int32_t* idx, int64_t B, int64_t M, int64_t K, const Tn* out = nullptr)
    {
        const Tn x = contig(x_in);
        const int64_t N = x.R / B, d = x.C;
        Tn o = out ? *out : alloc(B * M * K, d);
        o.req = true;
        TK(ps_op_gather_neighbour_ex(c, x.p, idx, B, N, M, K, d, o.p, o.ld));
        const Tn xin = x_in;
        record(o, [=](const Tn& dy) {
            if (opt.deterministic && !grad_of.count(xin.id)) {
                // first gradient of x: the gather-reduction writes every row (empty segments as zeros): no zero-filled buffer to add into
                Tn fresh = alloc(xin.R, xin.C);
                const Inv& iv = inverse(idx, B, N, M * K);
                reduce_rows(iv, dy.p, dy.ld, d, fresh.p, fresh.ld, 0);
                grad_of[xin.id] = fresh;
                return;
            }
            Tn buf = accum_buffer(xin);
            if (!buf.contiguous()) {
                ps::set_error("trainer: scatter-add into a strided gradient");
                throw TrainError{PS_ESTATE};
            }
            if (opt.deterministic) {
                const Inv& iv = inverse(idx, B, N, M * K);
                reduce_rows(iv, dy.p, dy.ld, d, buf.p, buf.ld, 1);
            } else {
                TK(ps_op_scatter_add_rows_ex(c, dy.p, dy.ld, idx, B, N, M * K, d, buf.p));
            }
        });
        return o;
    }

    // buf [R, ca+cb] whose left / right column blocks a and b were written in place by their producers: the concat costs nothing, and
    // its backward hands the column blocks of the gradient on as views
    Tn concat_views(const Tn& buf, const Tn& a, const Tn& b)
    {
        Tn o = buf;
        o.req = true;
        const int64_t ca = a.C, cb = b.C;
        record(o, [=](const Tn& dy) {
            if (a.req) accum(a, cols(dy, 0, ca));
            if (b.req) accum(b, cols(dy, ca, cb));
        });
        return o;
    }

    Tn softpool(const Tn& fset, const Tn& scores, int64_t K)
    {
        const int64_t RK = fset.R, d = fset.C, R = RK / K;
        const Tn fs = contig(fset), sc = contig(scores);
        // (the probabilities are not kept: the backward forms the softmax again from the scores, one [R*K, d] write + one live tensor less)
        Tn agg = alloc(R, d);
        TK(ps_op_softmax_pool_fwd(c, fs.p, sc.p, R, K, d, nullptr, agg.p));
        record(agg, [=](const Tn& dy_in) {
            const Tn dy = contig(dy_in);
            Tn dfset = alloc(RK, d), dscores = alloc(RK, d);
            TK(ps_op_softmax_pool_bwd_scores(c, dy.p, fs.p, sc.p, R, K, d, dfset.p, dscores.p));
            accum(fset, dfset);
            accum(scores, dscores);
        });
        return agg;
    }

    // att_pooling's score product + softmax over K + weighted sum as ONE kernel per direction (attpool_train.hip)
    Tn attpool(const Tn& fset, const Tn& W, const Tn& gW, int64_t K)
    {
        const int64_t RK = fset.R, d = fset.C, R = RK / K;
        Tn agg = alloc(R, d);
        TK(ps_op_att_pool_train_fwd(c, fset.p, fset.ld, W.p, R, K, d, agg.p));
        record(agg, [=](const Tn& dy_in) {
            const Tn dy = contig(dy_in);
            Tn dfset = alloc(RK, d);
            TK(ps_op_att_pool_train_bwd(c, fset.p, fset.ld, W.p, dy.p, R, K, d, dfset.p, d, gW.p));
            accum(fset, dfset);
        });
        return agg;
    }

    // att_pooling's core at the wide levels (d = 128 / 256; attpool_gemm.hip): scores and probabilities only ever live in registers; the
    // backward hands dS on to the step's ordinary weight-gradient product (partials now, summed by the one wgrad_finish launch)
    Tn attpool_gemm(const Tn& fset, const Tn& W, const Tn& gW, int64_t K)
    {
        const int64_t RK = fset.R, d = fset.C, R = RK / K;
        Tn agg = alloc(R, d);
        TK(ps_op_att_pool_gemm_fwd(c, fset.p, fset.ld, W.p, R, K, d, agg.p));
        record(agg, [=](const Tn& dy_in) {
            const Tn dy = contig(dy_in);
            Tn ds = alloc(RK, d, false);
            // fset's gradient so far (none in this graph: the pooling is the concat buffer's only consumer) is added to in the epilogue
            auto have = grad_of.find(fset.id);
            const bool add = have != grad_of.end() && have->second.R == RK && have->second.C == d && have->second.ld % 4 == 0;
            Tn dfset = add ? have->second : alloc(RK, d);
            TK(ps_op_att_pool_gemm_bwd(c, fset.p, fset.ld, W.p, dy.p, R, K, d, dfset.p, dfset.ld, add ? 1 : 0, ds.p, ds.ld));
            if (!add) accum(fset, dfset);
            const int64_t nb = wgrad_partial_slabs(c, fset.p, fset.ld, ds.p, ds.ld, RK, d, d);
            Tn part = alloc(nb, d * d, false);
            {
                Stage st(c, "train_wgrad", 1);
                TK(wgrad_partial(c, fset.p, fset.ld, ds.p, ds.ld, RK, d, d, part.p, nullptr));
            }
            wjobs.push_back(WgradJob{part.p, gW.p, (int)nb, (int)d, (int)d, 0});
            wkeep.push_back(part);
        });
        return agg;
    }

    // d = 512 (level 4) exists on the frame too (round 6: its backward as two launches over halves of the dF columns, 64 + 128 accumulators
    // each) and is NOT used by the step: measured on MI355X, 8 x 703 points (profiles/tools/exp_attg.py), the two launches take 1.13 ms
    // against 0.46 ms for the softmax-pool backward + input-gradient GEMM they would replace (bf16-MLP: 0.63 against 0.36) -- every launch
    // recomputes all four score panels at one wave per SIMD --, which the forward's gain (0.37 -> 0.30) does not buy back: batch-8 step
    // 38.2 -> 39.8 ms with it.  Level 4 stays op by op.
    static bool att_gemm_pays(int64_t K, int64_t d) { return ps_op_att_pool_gemm_supported(K, d) != 0 && d <= 256; }

    // the wide-level pooling over fset = [gather(f_src, idx) | f_xyz] without the gather and the concat buffer (attpool_gemm.hip, split-source
    // forms): the gathered half's gradient leaves as rows for the fixed-order gather-reduction, the f_xyz half is added in place, dS feeds the
    // split-bf16 weight-gradient kernel, whose loader gathers the rows of X it needs through idx
    bool att_gemm_split_ok(const Tn& f_src, const int32_t* idx, const Tn& f_xyz, int64_t B, int64_t M, int64_t K) const
    {
        const int64_t d = 2 * f_src.C, rows = B * M * K;
        auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
        return opt.fused_att && att_gemm_on && c->train_b3 && ps_op_att_pool_gemm_supported(K, d) && d <= 256 /* (the split-source kernels) */ && rows >= 16384 && rows < (1ll << 31) && d % 128 == 0 &&
               f_src.ld % 4 == 0 && f_xyz.ld % 4 == 0 && f_xyz.C == f_src.C && al(f_src.p) && al(f_xyz.p) && al(idx) && f_src.R / B * f_src.ld < (1ll << 31);
    }
    Tn attpool_gemm_split(const Tn& f_src_in, const int32_t* idx, int64_t B, int64_t M, int64_t K, const Tn& f_xyz, const Tn& W, const Tn& gW)
    {
        const Tn f_src = f_src_in;
        const int64_t N = f_src.R / B, h = f_src.C, d = 2 * h, RK = B * M * K;
        Tn agg = alloc(B * M, d);
        TK(ps_op_att_pool_gemm_fwd_split(c, f_src.p, f_src.ld, idx, B, N, M, f_xyz.p, f_xyz.ld, W.p, K, d, agg.p));
        record(agg, [=](const Tn& dy_in) {
            const Tn dy = contig(dy_in);
            auto have = grad_of.find(f_xyz.id);
            const bool add_in_place = have != grad_of.end() && have->second.C == h && have->second.R == RK && have->second.ld % 4 == 0;
            Tn dfx = add_in_place ? have->second : alloc(RK, h);
            Tn rows = alloc(RK, h, false), ds = alloc(RK, d, false);
            TK(ps_op_att_pool_gemm_bwd_split(c, f_src.p, f_src.ld, idx, B, N, M, f_xyz.p, f_xyz.ld, W.p, dy.p, K, d, rows.p, h, dfx.p, dfx.ld, add_in_place ? 1 : 0,
                                             ds.p, ds.ld));
            if (opt.deterministic) {
                const bool fresh_src = !grad_of.count(f_src.id);  // (first gradient of f_src: written, not added into zeros)
                Tn dsrc = fresh_src ? alloc(f_src.R, f_src.C) : accum_buffer(f_src);
                const Inv& iv = inverse(idx, B, N, M * K);
                reduce_rows(iv, rows.p, h, h, dsrc.p, dsrc.ld, fresh_src ? 0 : 1);
                if (fresh_src) grad_of[f_src.id] = dsrc;
            } else {
                Tn dsrc = accum_buffer(f_src);
                TK(ps_op_scatter_add_rows_ex(c, rows.p, h, idx, B, N, M * K, h, dsrc.p));
            }
            const int64_t nb = wgrad_split_slabs(c, f_src.p, f_src.ld, idx, f_xyz.p, f_xyz.ld, ds.p, ds.ld, RK, d, d);
            if (nb <= 0) {
                ps::set_error("trainer: the split-source weight gradient does not apply (rows %lld, d %lld)", (long long)RK, (long long)d);
                throw TrainError{PS_ESTATE};
            }
            Tn part = alloc(nb, d * d, false);
            {
                Stage st(c, "train_wgrad", 1);
                TK(wgrad_b3_partial_split(c, f_src.p, f_src.ld, idx, N, M * K, f_xyz.p, f_xyz.ld, ds.p, ds.ld, RK, d, d, part.p));
            }
            wjobs.push_back(WgradJob{part.p, gW.p, (int)nb, (int)d, (int)d, 0});
            wkeep.push_back(part);
            if (!add_in_place) accum(f_xyz, dfx);
            run_deferred(f_xyz.id);
        });
        return agg;
    }

    // attpool over fset = [gather(f_src, idx) | f_xyz] without the gather, the concat buffer or the scatter-add of its gradient
    Tn attpool_split(const Tn& f_src_in, const int32_t* idx, int64_t B, int64_t M, int64_t K, const Tn& f_xyz, const Tn& W, const Tn& gW)
    {
        const Tn f_src = f_src_in;
        const int64_t N = f_src.R / B, h = f_src.C, d = 2 * h;
        Tn agg = alloc(B * M, d);
        const bool act16 = f_xyz.b16;  // (the f_xyz half as bfloat16 rows)
        {
            ActScope as(c, act16);
            // d = 128 in the bf16-MLP mode: the forward on the frame of the large GEMMs (attpool_gemm.hip: scores in accumulator tiles,
            // the bfloat16 rows of f_xyz ARE operand fragments) -- 0.37 -> 0.2x ms per pooling; the backward stays with the per-point
            // kernel, which owns the weight gradient
            const bool fwd_gemm = c->tune.train_att128_fwd_gemm;
            const bool al = ((reinterpret_cast<uintptr_t>(f_src.p) | reinterpret_cast<uintptr_t>(f_xyz.p) | reinterpret_cast<uintptr_t>(idx)) & 15) == 0;
            if (fwd_gemm && d == 128 && opt.mlp_bf16 && act16 && al && f_src.ld % 4 == 0 && f_xyz.ld % 8 == 0 && B * M * K < (1ll << 31) && N * f_src.ld < (1ll << 31))
                TK(ps_op_att_pool_gemm_fwd_split(c, f_src.p, f_src.ld, idx, B, N, M, f_xyz.p, f_xyz.ld, W.p, K, d, agg.p));
            else
                TK(ps_op_att_pool_train_fwd_split(c, f_src.p, f_src.ld, idx, B, N, M, f_xyz.p, f_xyz.ld, W.p, K, d, agg.p));
        }
        record(agg, [=](const Tn& dy_in) {
            ActScope as(c, act16);
            const Tn dy = contig(dy_in);
            // f_xyz usually has a gradient already (the h -> h convolution's input gradient ran first): the kernel adds into it instead of
            // writing a second tensor that an axpy pass then folds in (4 passes over [N*K, h] -> 2)
            auto have = grad_of.find(f_xyz.id);
            const bool add_in_place = have != grad_of.end() && have->second.C == h && have->second.R == B * M * K && have->second.b16 == act16;
            Tn dfx = add_in_place ? have->second : alloc(B * M * K, h);
            if (!add_in_place) dfx.b16 = act16;  // (the gradient rows of the bfloat16-stored half are bfloat16 as well)
            struct Flag {
                ps_context* c;
                Flag(ps_context* ctx, bool on) : c(ctx) { c->att_df_accum = on; }
                ~Flag() { c->att_df_accum = false; }
            } flag(c, add_in_place);
            const bool fresh_src = opt.deterministic && !grad_of.count(f_src.id);  // (first gradient of f_src: written, not added into zeros)
            Tn dsrc = fresh_src ? alloc(f_src.R, f_src.C) : accum_buffer(f_src);  // the gathered half's gradient is added in place
            if (opt.deterministic) {
                // ... as plain rows first, then summed per source row in ascending row order (no float atomics)
                Tn rows = alloc(B * M * K, h, false);
                TK(ps_op_att_pool_train_bwd_split_rows(c, f_src.p, f_src.ld, idx, B, N, M, f_xyz.p, f_xyz.ld, W.p, dy.p, K, d, rows.p, h, dfx.p, dfx.ld, gW.p));
                const Inv& iv = inverse(idx, B, N, M * K);
                reduce_rows(iv, rows.p, h, h, dsrc.p, dsrc.ld, fresh_src ? 0 : 1);
                if (fresh_src) grad_of[f_src.id] = dsrc;
            } else {
                TK(ps_op_att_pool_train_bwd_split(c, f_src.p, f_src.ld, idx, B, N, M, f_xyz.p, f_xyz.ld, W.p, dy.p, K, d, dsrc.p, dsrc.ld, dfx.p, dfx.ld, gW.p));
            }
            if (!add_in_place) accum(f_xyz, dfx);
            run_deferred(f_xyz.id);
        });
        return agg;
    }

    // neigh: the level's neighbour table [B, N, K] whose first M rows per cloud ARE pool_idx (ps_pyramid: sub_idx = neigh_idx[:, :M])
    Tn maxpool(const Tn& x_in, const int32_t* pool_idx, const int32_t* neigh, int64_t B, int64_t M, int64_t K)
    {
        const Tn x = contig(x_in);
        const int64_t N = x.R / B, d = x.C;
        Tn out = alloc(B * M, d);
        Tn ties;  // deterministic mode: the forward leaves the tie counts for the backward (one byte per output)
        // the fixed-order backward walks the PREFIX of the neighbour table's inverse index: only for pooling tables that are that prefix
        // (a caller-filled pyramid whose pooling table is something else takes the float-atomic form: still correct)
        const bool by_inverse = opt.deterministic && pool_is_prefix(pool_idx, neigh, B, N, M, K);
        if (by_inverse && d % 4 == 0 && K <= 255) {
            ties = alloc(1, (B * M * d + 3) / 4, false);
            TK(ps_op_random_sample_ties(c, x.p, pool_idx, B, N, M, K, d, out.p, reinterpret_cast<uint8_t*>(ties.p)));
        } else {
            TK(ps_op_random_sample(c, x.p, pool_idx, B, N, M, K, d, out.p));
        }
        const Tn xin = x_in;
        record(out, [=](const Tn& dy_in) {
            const Tn dy = contig(dy_in);
            // (first gradient of x, tie-count form: the kernel stores -- zeros where nothing was pooled from a row -- instead of adding into a
            //  zero-filled buffer)
            // (the overwriting kernel's own conditions, csrc/invidx.hip maxpool_bwd_inv4_kernel: tie counts, 16-byte rows, 32-bit element and
            //  table offsets -- a batch beyond them takes the additive form instead of failing)
            const bool fresh = by_inverse && ties && !grad_of.count(xin.id) && xin.contiguous() && d % 4 == 0 && B * N * d < (1ll << 32) &&
                               B * N * K < (1ll << 31) &&
                               ((reinterpret_cast<uintptr_t>(dy.p) | reinterpret_cast<uintptr_t>(out.p) | reinterpret_cast<uintptr_t>(x.p)) & 15) == 0;
            Tn buf = fresh ? alloc(xin.R, xin.C) : accum_buffer(xin);
            if (fresh) grad_of[xin.id] = buf;
            struct Flag {
                ps_context* c;
                Flag(ps_context* ctx, bool on) : c(ctx) { c->pool_bwd_overwrite = on; }
                ~Flag()
                {
                    c->pool_bwd_overwrite = false;
                    c->walk_order = nullptr;
                }
            } flag(c, fresh);
            if (by_inverse && buf.contiguous()) {
                const Inv& iv = inverse(neigh, B, N, N * K);  // (shared with the level's gathers: the pooling rows are a prefix of every segment)
                c->walk_order = iv.order;  // (the level's leaf order: the kernel walks its destinations in it, XCD by XCD)
                c->walk_order_n = iv.n_cloud;
                Tn share = ties ? Tn() : alloc(B * M, d, false);
                TK(ps_op_random_sample_bwd_inv(c, dy.p, out.p, x.p, pool_idx, reinterpret_cast<const int32_t*>(iv.offsets.p),
                                               reinterpret_cast<const int32_t*>(iv.src.p), B, N, M, K, d,
                                               ties ? reinterpret_cast<const uint8_t*>(ties.p) : nullptr, ties ? nullptr : share.p, buf.p));
            } else {
                TK(ps_op_random_sample_bwd(c, dy.p, out.p, x.p, pool_idx, B, N, M, K, d, buf.p));
            }
        });
        return out;
    }

    Tn add_lrelu(const Tn& a, const Tn& b)
    {
        Tn y = alloc(a.R, a.C);
        TK(ps_op_add_lrelu(c, a.p, b.p, a.numel(), y.p));
        record(y, [=](const Tn& dy_in) {
            const Tn dy = contig(dy_in);
            Tn ds = alloc(a.R, a.C);
            TK(ps_op_add_lrelu_bwd(c, dy.p, y.p, a.numel(), ds.p));
            // the same values are the gradient of both summands.  In this graph both (mlp2's and the shortcut's BatchNorm outputs) have no
            // other consumer, so nothing is ever added into either gradient: they share ONE read-only tensor.  Only where a summand already
            // holds a gradient (another graph) does the second one get its own copy.
            if (grad_of.count(a.id) || grad_of.count(b.id)) {
                Tn ds2 = alloc(a.R, a.C);
                TK_HIP(hipMemcpyAsync(ds2.p, ds.p, sizeof(float) * (size_t)a.numel(), hipMemcpyDeviceToDevice, stream()));
                accum(a, ds);
                accum(b, ds2);
            } else {
                grad_of[a.id] = ds;
                Tn alias = ds;
                grad_of[b.id] = alias;
            }
        });
        return y;
    }

    Tn dropout(const Tn& x, float keep_prob, uint32_t seed)
    {
        if (keep_prob >= 1.0f) return x;
        Tn y = alloc(x.R, x.C), mask = alloc(x.R, x.C, false);
        TK(ps_op_dropout(c, x.p, x.numel(), seed, keep_prob, y.p, mask.p));
        record(y, [=](const Tn& dy_in) {
            const Tn dy = contig(dy_in);
            Tn dx = alloc(x.R, x.C);
            TK(ps_op_mul(c, dy.p, mask.p, x.numel(), dx.p));
            accum(x, dx);
        });
        return y;
    }

    // ---- graph pieces (RandLANet.py call sites) -----------------------------------------------------------------------------------------
    Tn Wt(const LayerP& lp) { return lp.kind == kDeconv ? P(lp.w, lp.cout, lp.cin) : P(lp.w, lp.cin, lp.cout); }
    Tn gWt(const LayerP& lp) { return lp.kind == kDeconv ? G(lp.w, lp.cout, lp.cin) : G(lp.w, lp.cin, lp.cout); }

    // helper_tf_util.conv2d / conv2d_transpose (:115-250): 1x1 conv + bias [+ BatchNorm(training) [+ LeakyReLU(0.2)]]
    Tn conv(const Tn& x, const std::string& scope, bool bn = true, bool act = true, const Tn* out = nullptr, bool fp32_only = false, bool defer_dgrad = false)
    {
        const LayerP& lp = layer(scope);
        if (bn && !out && !fp32_only && !defer_dgrad && convbn_rect_ok(x, lp)) return conv_bn_rect(x, lp, act);
        // (the square layers on [N] rows -- mlp1, att_pooling's mlp: 16 -> 16, 32 -> 32, 64 -> 64 -- measured no gain in the recompute form:
        //  43.9 / 43.8 ms without, 44.5 / 43.9 ms with; they keep the GEMM + BatchNorm kernels)
        Tn y = linear(x, Wt(lp), lp.b >= 0 ? params + lp.b : nullptr, gWt(lp), lp.b >= 0 ? grads + lp.b : nullptr, lp.kind == kDeconv, nullptr, fp32_only,
                      defer_dgrad);
        if (bn) y = bn_act(y, lp, act, out);
        return y;
    }
    Tn att_split(const Tn& f_src, const int32_t* idx, int64_t B, int64_t M, int64_t K, const Tn& f_xyz, const std::string& name, bool on_gemm_frame = false)
    {
        const LayerP& fc = layer(name + "fc");
        Tn agg = on_gemm_frame ? attpool_gemm_split(f_src, idx, B, M, K, f_xyz, P(fc.w, fc.cin, fc.cout), G(fc.w, fc.cin, fc.cout))
                               : attpool_split(f_src, idx, B, M, K, f_xyz, P(fc.w, fc.cin, fc.cout), G(fc.w, fc.cin, fc.cout));
        return conv(agg, name + "mlp");
    }
    // att_pooling with the score product in the pre-product form of the inference kernels: fset . Wfc = (f . Wfc[:h])[idx] + f_xyz . Wfc[h:]
    Tn att_pre(const Tn& f_src, const int32_t* idx, int64_t B, int64_t M, int64_t K, const Tn& fcat, const Tn& f_xyz, const std::string& name)
    {
        const LayerP& fc = layer(name + "fc");
        const Tn W = P(fc.w, fc.cin, fc.cout), gW = G(fc.w, fc.cin, fc.cout);
        const int64_t h = f_src.C;
        Tn s = gather(linear(f_src, rows_of(W, 0, h), nullptr, rows_of(gW, 0, h), nullptr), idx, B, M, K);
        // (defer_dgrad: f_xyz's other gradient is a column block of this pooling's dF, handed on by the concat's backward AFTER this product's;
        //  parked, the input-gradient GEMM adds into that view in its epilogue instead of a strided add pass folding the view into its output)
        s = linear(f_xyz, rows_of(W, h, W.R - h), nullptr, rows_of(gW, h, W.R - h), nullptr, false, &s, false, true);
        return conv(softpool(fcat, s, K), name + "mlp");
    }
    Tn att(const Tn& fcat, const std::string& name, int64_t K)
    {
        const LayerP& fc = layer(name + "fc");
        const Tn W = P(fc.w, fc.cin, fc.cout), gW = G(fc.w, fc.cin, fc.cout);
        Tn agg;
        if (opt.fused_att && ps_op_att_pool_train_supported(K, fcat.C)) {
            agg = attpool(fcat, W, gW, K);  // levels whose [N*K, d] tensors are large: one kernel per direction
        } else if (opt.fused_att && att_gemm_on && att_gemm_pays(K, fcat.C) && fcat.ld % 4 == 0 &&
                   (reinterpret_cast<uintptr_t>(fcat.p) & 15) == 0) {
            agg = attpool_gemm(fcat, W, gW, K);  // the wide levels: scores in registers on the frame of the large GEMMs
        } else {
            Tn s = linear(fcat, W, nullptr, gW, nullptr);
            agg = softpool(fcat, s, K);
        }
        return conv(agg, name + "mlp");
    }

    // Network.inference in training mode (RandLANet.py:110-152); features [B*N0, Cin] -> logits [B*N0, classes]
    Tn forward(const ps_pyramid* pyr, const float* features)
    {
        const int L = cfg.num_layers;
        const int64_t B = pyr->B, K = cfg.k_n;
        Tn x = external(const_cast<float*>(features), B * pyr->n[0], cfg.in_channels, false);
        begin_section("fc0");
        const LayerP& fc0 = layer("fc0");
        Tn f = linear(x, P(fc0.w, fc0.cin, fc0.cout), params + fc0.b, G(fc0.w, fc0.cin, fc0.cout), grads + fc0.b);
        f = bn_act(f, fc0, true);
        std::vector<Tn> enc;
        for (int i = 0; i < L; ++i) {
            const std::string n = "Encoder_layer_" + std::to_string(i);
            begin_section("enc" + std::to_string(i));
            const int32_t* idx = pyr->neigh_idx[i];
            const int64_t N = pyr->n[i];
            const Tn feature = f;
            const LayerP& lfa1 = layer(n + "LFAmlp1");
            const int64_t hloc = lfa1.cout;
            const bool locse_fused = opt.fused_locse && ps_op_locse_train_supported(K, hloc);
            // shared statistics (world > 1): mlp1 and the LocSE convolution are independent -- mlp1's sums ride in the LocSE layer's all-reduce
            // (bn_begin here, bn_end right behind the LocSE layer); mlp2 || shortcut further down
            const bool sync_merge = sync_bn && coll_active() && merge_syncbn;
            const LayerP& l_mlp1 = layer(n + "mlp1");
            PendingBn pend1;
            Tn f_pc, pair_sums;
            if (sync_merge && l_mlp1.gamma >= 0 && lfa1.gamma >= 0 && !convbn_rect_ok(feature, l_mlp1)) {
                Tn x1 = linear(feature, Wt(l_mlp1), l_mlp1.b >= 0 ? params + l_mlp1.b : nullptr, gWt(l_mlp1), l_mlp1.b >= 0 ? grads + l_mlp1.b : nullptr,
                               l_mlp1.kind == kDeconv);
                if (!locse_fused) pair_sums = alloc(1, 2 * l_mlp1.cout + 2 * hloc, false);  // (op-by-op LocSE: both layers' float sums side by side)
                pend1 = bn_begin(x1, l_mlp1, true, nullptr, locse_fused ? nullptr : &pair_sums, 0);
                f_pc = pend1.y;  // (shape, pitch and address are final; the values arrive with bn_end, before the first reader is enqueued)
            } else {
                f_pc = conv(feature, n + "mlp1");
            }
            Tn rel;
            if (!locse_fused) {
                rel = alloc(B * N * K, 10, false);
                TK(ps_op_relative_pos_encoding(c, pyr->xyz[i], idx, B, N, K, rel.p));
            }
            bool act16 = false;  // (set below for the levels that store their [N*K, h] rows as bfloat16)
            auto locse = [&](const Tn* out) -> Tn {
                if (!locse_fused && pend1.open) {
                    // the op-by-op form of the pair: both layers' sums in ONE float all-reduce
                    Tn xr = linear(rel, Wt(lfa1), lfa1.b >= 0 ? params + lfa1.b : nullptr, gWt(lfa1), lfa1.b >= 0 ? grads + lfa1.b : nullptr, lfa1.kind == kDeconv,
                                   nullptr, /*fp32_only*/ true);
                    PendingBn p2 = bn_begin(xr, lfa1, true, out, &pair_sums, 2 * l_mlp1.cout);
                    allreduce(pair_sums.p, 2 * l_mlp1.cout + 2 * hloc, 0);
                    bn_end(pend1);
                    return bn_end(p2);
                }
                if (!locse_fused) return conv(rel, n + "LFAmlp1", true, true, out, true);
                Tn y = locse_bn_act(pyr->xyz[i], idx, B, N, K, lfa1, out, act16 && out == nullptr, pend1.open ? &pend1 : nullptr);
                if (pend1.open) bn_end(pend1);
                return y;
            };
            // tf.concat([f_neighbours, f_xyz]) (RandLANet.py:328,332): both producers write their column block of the concat buffer
            // directly (no concat copy forward, no split copies backward)
            const int64_t hc = f_pc.C;
            Tn f_agg2;
            const bool narrow = opt.fused_att && ps_op_att_pool_train_supported_ex(K, 2 * hc, opt.mlp_bf16 ? 1 : 0);
            {
                // bfloat16 storage of f_xyz and of LFA mlp2's output (ps_train_options.act_bf16): only where EVERY consumer of the two tensors
                // reads them that way -- the fused LocSE branch, the recompute-form convolution and the split-source pooling kernels
                Tn probe = f_pc;
                probe.p = nullptr;
                act16 = opt.mlp_bf16 && opt.act_bf16 && act_bf16_on && narrow && locse_fused && hc % 8 == 0 && convbn_fused_ok(probe, layer(n + "LFAmlp2"));
            }
            bool wide_split = false;
            Tn f_xyz_s;
            if (!narrow && (att_split_env < 0 ? opt.mlp_bf16 != 0 : att_split_env != 0)) {
                // the wide levels in the split-source form: decided on shapes and on the operands' alignment (pool blocks are 256-byte
                // aligned, so the LocSE rows -- allocated below -- qualify whenever f_pc does)
                Tn probe = f_pc;
                probe.p = nullptr;
                wide_split = att_gemm_split_ok(f_pc, idx, probe, B, N, K);
                if (wide_split) f_xyz_s = locse(nullptr);
            }
            if (narrow || wide_split) {
                // gather_neighbour + concat + att_pooling's core as one kernel per direction
                Tn f_xyz = narrow ? locse(nullptr) : f_xyz_s;
                Tn f_agg = att_split(f_pc, idx, B, N, K, f_xyz, n + "LFAatt_pooling_1", wide_split);
                const LayerP& l2 = layer(n + "LFAmlp2");
                Tn f_xyz2 = convbn_fused_ok(f_xyz, l2) ? conv_bn_fused(f_xyz, l2, true)
                                                       : conv(f_xyz, n + "LFAmlp2", true, true, nullptr, false, /*defer_dgrad: pooling 1's backward stores first*/ true);
                f_agg2 = att_split(f_agg, idx, B, N, K, f_xyz2, n + "LFAatt_pooling_2", wide_split && att_gemm_split_ok(f_agg, idx, f_xyz2, B, N, K));
            } else {
                // (d = 128: the pre-product form measured slower, HBM bound there; bf16 mode: its yardstick rounds the operands of the ONE d x d product)
                const bool pre = !opt.mlp_bf16 && 2 * hc >= 256 && !(opt.fused_att && att_gemm_on && att_gemm_pays(K, 2 * hc));
                Tn cat1 = alloc(B * N * K, 2 * hc);
                Tn right1 = cols(cat1, hc, hc);
                Tn f_xyz = locse(&right1);
                Tn left1 = cols(cat1, 0, hc);
                Tn f_nb = gather(f_pc, idx, B, N, K, &left1);
                Tn fcat1 = concat_views(cat1, f_nb, f_xyz);
                Tn f_agg = pre ? att_pre(f_pc, idx, B, N, K, fcat1, f_xyz, n + "LFAatt_pooling_1") : att(fcat1, n + "LFAatt_pooling_1", K);
                Tn cat2 = alloc(B * N * K, 2 * hc);
                Tn right2 = cols(cat2, hc, hc);
                // (defer_dgrad: f_xyz's other gradient, a column block of pooling 1's dF, arrives later; the convolution adds into it in place)
                const LayerP& l2 = layer(n + "LFAmlp2");
                Tn f_xyz2 = convbn_fused_ok(f_xyz, l2) ? conv_bn_fused(f_xyz, l2, true, &right2) : conv(f_xyz, n + "LFAmlp2", true, true, &right2, false, true);
                Tn left2 = cols(cat2, 0, hc);
                Tn f_nb2 = gather(f_agg, idx, B, N, K, &left2);
                Tn fcat2 = concat_views(cat2, f_nb2, f_xyz2);
                f_agg2 = pre ? att_pre(f_agg, idx, B, N, K, fcat2, f_xyz2, n + "LFAatt_pooling_2") : att(fcat2, n + "LFAatt_pooling_2", K);
            }
            Tn f_enc;
            const LayerP& l_mlp2 = layer(n + "mlp2");
            const LayerP& sl = layer(n + "shortcut");
            if (sync_merge && l_mlp2.gamma >= 0 && sl.gamma >= 0 && l_mlp2.cout == sl.cout) {
                // shared statistics: mlp2 and the shortcut as one pair -- one all-reduce per direction for both (res_pair_sync)
                auto lin = [&](const Tn& x, const LayerP& lp) {
                    return linear(x, Wt(lp), lp.b >= 0 ? params + lp.b : nullptr, gWt(lp), lp.b >= 0 ? grads + lp.b : nullptr, lp.kind == kDeconv);
                };
                Tn xa = lin(f_agg2, l_mlp2);
                Tn xb = lin(feature, sl);
                f_enc = res_pair_sync(xa, l_mlp2, xb, sl);
            } else {
                Tn a = conv(f_agg2, n + "mlp2", true, false);
                // the residual sum + LeakyReLU inside the shortcut's apply pass where that layer runs in the recompute form (levels 0-1)
                const bool fuse = c->tune.train_fuse_residual;  // (A/B knob)
                if (fuse && sl.gamma >= 0 && convbn_rect_ok(feature, sl) && a.contiguous() && a.ld % 4 == 0 && (reinterpret_cast<uintptr_t>(a.p) & 15) == 0 && a.req) {
                    f_enc = conv_bn_rect(feature, sl, false, &a);
                } else {
                    Tn b = conv(feature, n + "shortcut", true, false);
                    f_enc = add_lrelu(a, b);
                }
            }
            f = maxpool(f_enc, pyr->sub_idx[i], idx, B, pyr->n[i + 1], K);
            if (i == 0) enc.push_back(f_enc);
            enc.push_back(f);
        }
        begin_section("decoder");
        f = conv(enc.back(), "decoder_0");
        for (int j = 0; j < L; ++j) {
            // nearest_interpolation + tf.concat([skip, up]) (RandLANet.py:134-143): the gather writes the right column block of the concat
            // buffer, the skip rows are copied into the left one
            const Tn& skip = enc[enc.size() - 2 - j];
            const int lvl = L - 1 - j;
            const int64_t M = pyr->n[lvl];
            Tn cat = alloc(B * M, skip.C + f.C);
            Tn right = cols(cat, skip.C, f.C);
            Tn up = gather(f, pyr->interp_idx[lvl], B, M, 1, &right);
            Tn left = cols(cat, 0, skip.C, false);  // (same id as nothing recorded: the copy below has its own backward)
            left.id = next_id++;
            copy2d(skip, left, false);
            left.req = true;
            {
                const Tn sk = skip;
                record(left, [=](const Tn& dy) { accum(sk, contig(dy)); });  // (consumers such as the max-pool backward ADD into a dense buffer)
            }
            Tn both = concat_views(cat, left, up);
            f = conv(both, "Decoder_layer_" + std::to_string(j));
        }
        begin_section("head");
        f = conv(f, "fc1");
        f = conv(f, "fc2");
        // every rank draws its own mask (N GPUs x 1 cloud behaves like 1 GPU x N clouds, where the clouds sit at different element offsets)
        f = dropout(f, opt.keep_prob, (uint32_t)(0x9e3779b9u * (uint32_t)(step + 1) + 0x85ebca6bu * (uint32_t)rank));
        return conv(f, "fc", false, false);
    }
};

namespace ps {

static void build_layout(ps_trainer* t)
{
    const ps_randla_config& cfg = t->cfg;
    const int L = cfg.num_layers;
    auto add = [&](const std::string& scope, LayerKind kind, int cin, int cout) {
        LayerP lp;
        lp.scope = scope; lp.kind = kind; lp.cin = cin; lp.cout = cout;
        t->by_scope[scope] = (int)t->layers.size();
        t->layers.push_back(lp);
    };
    add("fc0", kDense, cfg.in_channels, 8);
    int d_in = 8;
    for (int i = 0; i < L; ++i) {
        const int d = cfg.d_out[i], h = d / 2;
        const std::string n = "Encoder_layer_" + std::to_string(i);
        add(n + "mlp1", kConv, d_in, h);
        add(n + "LFAmlp1", kConv, 10, h);
        add(n + "LFAatt_pooling_1fc", kDenseNoBias, d, d);
        add(n + "LFAatt_pooling_1mlp", kConv, d, h);
        add(n + "LFAmlp2", kConv, h, h);
        add(n + "LFAatt_pooling_2fc", kDenseNoBias, d, d);
        add(n + "LFAatt_pooling_2mlp", kConv, d, d);
        add(n + "mlp2", kConv, d, 2 * d);
        add(n + "shortcut", kConv, d_in, 2 * d);
        d_in = 2 * d;
    }
    add("decoder_0", kConv, d_in, d_in);
    std::vector<int> chans;  // f_encoder_list channel widths (RandLANet.py:119-127)
    chans.push_back(2 * cfg.d_out[0]);
    for (int i = 0; i < L; ++i) chans.push_back(2 * cfg.d_out[i]);
    int up = d_in;
    for (int j = 0; j < L; ++j) {
        const int skip = chans[chans.size() - 2 - j];
        add("Decoder_layer_" + std::to_string(j), kDeconv, skip + up, skip);
        up = skip;
    }
    add("fc1", kConv, up, 64);
    add("fc2", kConv, 64, 32);
    add("fc", kConvNoBn, 32, cfg.num_classes);

    int64_t off = 0, boff = 0;
    for (LayerP& lp : t->layers) {
        const bool dense = lp.kind == kDense || lp.kind == kDenseNoBias;
        const std::string wname = lp.scope + (dense ? "/kernel" : "/weights"), bname = lp.scope + (dense ? "/bias" : "/biases");
        const std::string bn = lp.kind == kDense ? std::string("batch_normalization") : lp.scope + "/batch_normalization";  // fc0's BN is un-scoped
        lp.w = off;
        const int64_t wr = lp.kind == kDeconv ? lp.cout : lp.cin, wc = lp.kind == kDeconv ? lp.cin : lp.cout;
        t->rows.push_back({wname, off, wr, wc, 0});
        off += (int64_t)lp.cin * lp.cout;
        if (lp.kind != kDenseNoBias) {
            lp.b = off;
            t->rows.push_back({bname, off, 1, lp.cout, 0});
            off += lp.cout;
        }
        if (lp.kind == kDense || lp.kind == kConv || lp.kind == kDeconv) {
            lp.gamma = off;
            t->rows.push_back({bn + "/gamma", off, 1, lp.cout, 0});
            off += lp.cout;
            lp.beta = off;
            t->rows.push_back({bn + "/beta", off, 1, lp.cout, 0});
            off += lp.cout;
            lp.mov_mean = boff;
            t->rows.push_back({bn + "/moving_mean", boff, 1, lp.cout, 1});
            boff += lp.cout;
            lp.mov_var = boff;
            t->rows.push_back({bn + "/moving_variance", boff, 1, lp.cout, 1});
            boff += lp.cout;
        }
    }
    t->n_params = off;
    t->n_buffers = boff;
}

static int run_step(ps_trainer* t, const ps_pyramid* pyr, const float* features, const int32_t* labels, const float* class_weights, float* loss,
                    float* logits_out, bool optimise)
{
    ps_context* c = t->c;
    PS_CHECK(t->params && t->grads && t->buffers, "trainer: ps_trainer_bind has not been called");
    PS_CHECK(!optimise || (t->adam_m && t->adam_v), "ps_randla_train_step: the Adam moment buffers are not bound");
    PS_CHECK(pyr && features && labels && class_weights && loss, "trainer: NULL argument");
    PS_CHECK(pyr->num_layers == t->cfg.num_layers && pyr->K == t->cfg.k_n, "trainer: the pyramid does not match the network (layers / K)");
    PS_HIP(hipSetDevice(c->device));
    const bool bf16 = t->opt.mlp_bf16 != 0;
    const bool was_bf16 = c->train_bf16;
    // (ps_set_train_act_bf16 is a public setter of the op-level surface: a flag left on by an op-level caller of a shared context must not
    //  reach the ops this step calls outside an ActScope -- the row reductions and the split-source pooling read it directly)
    const bool was_act_bf16 = c->train_act_bf16;
    c->train_act_bf16 = false;
    int rc = PS_OK;
    try {
        t->pool.begin_step();
        t->pyramid_vouched = pyr->built != 0 && pyr->built == pyramid_stamp(pyr);
        t->order_of.clear();
        for (int i = 0; i < pyr->num_layers && t->pyramid_vouched; ++i) {  // (only orders ps_pyramid_build wrote are trusted as permutations)
            // (a table's destinations: neigh_idx[i] / sub_idx[i] gather from level i, interp_idx[i] from level i + 1)
            if (pyr->order[i]) t->order_of[pyr->neigh_idx[i]] = pyr->order[i];
            if (pyr->order[i] && pyr->sub_idx[i]) t->order_of[pyr->sub_idx[i]] = pyr->order[i];
            if (i + 1 < pyr->num_layers && pyr->order[i + 1] && pyr->interp_idx[i]) t->order_of[pyr->interp_idx[i]] = pyr->order[i + 1];
        }
        t->forks_used = 0;
        t->coll_calls = t->coll_bytes = 0;
        t->coll_host_ms = 0.0;
        t->next_id = 0;
        t->ops.clear();
        t->grad_of.clear();
        t->deferred.clear();
        t->section_names.clear();
        t->section = 0;
        c->train_bf16 = bf16;
        // weight images: replay the recorded list (ONE packing launch per source file, the products then skip their own), or record it
        if (t->pack.mode == 2) {
            TK(ps::pack_cache_replay(c, t->pack));
        } else {
            ps::pack_cache_clear(t->pack);
            t->pack.mode = 1;
        }
        c->pack_cache = &t->pack;
        Tn logits = t->forward(pyr, features);
        const int64_t R = logits.R, C = logits.C;
        const int32_t* lab = labels;
        Tn mapped;
        if (t->n_label_map) {
            mapped = t->alloc(1, R, false);
            hipLaunchKernelGGL(tr_label_map_kernel, dim3(tr_grid(R)), dim3(256), 0, c->stream, labels, t->label_map.as<int32_t>(), t->n_label_map, R,
                               reinterpret_cast<int32_t*>(mapped.p));
            TK_HIP(hipGetLastError());
            lab = reinterpret_cast<const int32_t*>(mapped.p);
        }
        t->mark("loss");
        Tn dlogits = t->alloc(R, C);
        TK_HIP(hipMemsetAsync(loss, 0, sizeof(float), c->stream));
        TK(ps_op_weighted_ce(c, logits.p, lab, class_weights, R, C, loss, dlogits.p));
        if (logits_out) TK_HIP(hipMemcpyAsync(logits_out, logits.p, sizeof(float) * (size_t)(R * C), hipMemcpyDeviceToDevice, c->stream));
        t->backward(logits, dlogits);
    } catch (const TrainError& e) {
        rc = e.rc;
    } catch (const std::bad_alloc&) {
        ps::set_error("trainer: out of host memory");
        rc = PS_ENOMEM;
    }
    c->train_bf16 = was_bf16;  // the context may be shared with inference-side op calls: never leave the mode on
    c->train_act_bf16 = was_act_bf16;
    c->pack_cache = nullptr;
    if (rc != PS_OK || t->pack.broken || (t->pack.mode == 2 && t->pack.cursor != t->pack.jobs.size())) {
        ps::pack_cache_clear(t->pack);  // another sequence of products than the recorded one (or a failed step): the next step records again
    } else if (t->pack.mode == 1) {
        const int prc = ps::pack_cache_finish_recording(c, t->pack);
        if (prc != PS_OK) ps::pack_cache_clear(t->pack);
    }
    t->ops.clear();
    t->grad_of.clear();
    t->deferred.clear();  // (closures hold tensors: nothing of a failed step may survive it)
    if (rc != PS_OK) {  // (a failed profiled step: its pending events are not read by anybody)
        for (auto& m : t->marks) (void)hipEventDestroy(m.second);
        t->marks.clear();
        for (auto& m : t->coll_marks) {
            (void)hipEventDestroy(m.first);
            (void)hipEventDestroy(m.second);
        }
        t->coll_marks.clear();
    }
    if (t->side_busy) {  // (a failed step: nothing of it may still run when its tensors go back to the pool)
        (void)hipStreamSynchronize(t->side);
        t->side_busy = false;
    }
    t->wjobs.clear();
    t->wkeep.clear();
    t->inv_cache.clear();
    if (rc != PS_OK) return rc;
    if (!optimise) {
        try {
            t->finish_profile();
        } catch (const TrainError& e) {
            return e.rc;
        }
        return PS_OK;
    }
    try {
        t->mark("grad all-reduce");
    } catch (const TrainError& e) {
        return e.rc;
    }
    try {
        if (t->coll_active()) {
            // gradient synchronisation of config 4: ONE all-reduce of the flat fp32 gradient buffer, then the mean over the ranks
            t->allreduce(t->grads, t->n_params, 0);
            if (t->world > 1) {
                Stage st(c, "train_adam", 1);
                hipLaunchKernelGGL(tr_scale_kernel, dim3(tr_grid(t->n_params)), dim3(256), 0, c->stream, t->grads, t->n_params, 1.0f / (float)t->world);
                TK_HIP(hipGetLastError());
            }
        }
    } catch (const TrainError& e) {
        return e.rc;
    }
    t->step += 1;
    try {
        t->mark("adam");
    } catch (const TrainError& e) {
        return e.rc;
    }
    const int arc = ps_op_adam(c, t->params, t->grads, t->adam_m, t->adam_v, t->n_params, t->opt.learning_rate, 0.9f, 0.999f, 1e-8f, t->step);
    try {
        t->finish_profile();
    } catch (const TrainError& e) {
        return e.rc;
    }
    return arc;
}

}  // namespace ps

extern "C" {

int ps_trainer_create(ps_context* c, const ps_randla_config* cfg, const ps_train_options* opt, ps_trainer** out)
{
    PS_CHECK(c && cfg && opt && out, "ps_trainer_create: NULL argument");
    PS_CHECK(cfg->num_layers >= 1 && cfg->num_layers <= PS_MAX_LAYERS && cfg->k_n >= 1 && cfg->num_classes >= 1 && cfg->in_channels >= 1,
             "ps_trainer_create: bad configuration");
    PS_CHECK(opt->keep_prob > 0.f && opt->keep_prob <= 1.f && opt->learning_rate > 0.f, "ps_trainer_create: keep_prob must be in (0, 1], learning_rate > 0");
    PS_CHECK(opt->num_ignored >= 0 && opt->num_ignored <= 8, "ps_trainer_create: at most 8 ignored labels");
    for (int i = 0; i < cfg->num_layers; ++i) PS_CHECK(cfg->d_out[i] >= 2 && cfg->d_out[i] % 2 == 0, "ps_trainer_create: d_out must be even");
    ps_trainer* t = new ps_trainer();
    t->c = c;
    t->act_bf16_on = c->tune.train_act_bf16;
    t->att_split_env = c->tune.train_att_gemm_split;
    t->att_gemm_on = c->tune.train_att_gemm;
    t->merge_syncbn = c->tune.train_merge_syncbn;
    t->cfg = *cfg;
    t->opt = *opt;
    build_layout(t);
    if (opt->num_ignored > 0) {
        // RandLANet.py:77-81: reducing_list = range(C) with a 0 inserted at every ignored index (ascending); ignored entries become -1 here
        std::vector<int32_t> red;
        for (int i = 0; i < cfg->num_classes; ++i) red.push_back(i);
        std::vector<int> ign(opt->ignored_label_inds, opt->ignored_label_inds + opt->num_ignored);
        std::sort(ign.begin(), ign.end());
        for (int v : ign) {
            if (v < 0 || v > (int)red.size()) {
                delete t;
                ps::set_error("ps_trainer_create: ignored label %d out of range", v);
                return PS_EINVAL;
            }
            red.insert(red.begin() + v, -1);
        }
        t->n_label_map = (int)red.size();
        int rc = t->label_map.reserve(sizeof(int32_t) * red.size());
        if (rc != PS_OK) {
            delete t;
            return rc;
        }
        hipError_t e = hipMemcpy(t->label_map.p, red.data(), sizeof(int32_t) * red.size(), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            delete t;
            ps::set_error("ps_trainer_create: label map upload failed: %s", hipGetErrorString(e));
            return PS_EHIP;
        }
    }
    *out = t;
    return PS_OK;
}

int ps_trainer_destroy(ps_trainer* t)
{
    if (!t) return PS_OK;
    t->ops.clear();
    t->grad_of.clear();
    t->deferred.clear();
    t->wkeep.clear();
    t->inv_cache.clear();
    ps::pack_cache_clear(t->pack);
    t->destroy_side();
    t->pool.destroy();
    t->label_map.release();
    delete t;
    return PS_OK;
}

int64_t ps_trainer_param_count(const ps_trainer* t) { return t ? t->n_params : -1; }
int64_t ps_trainer_buffer_count(const ps_trainer* t) { return t ? t->n_buffers : -1; }
int ps_trainer_layout_rows(const ps_trainer* t) { return t ? (int)t->rows.size() : -1; }

int ps_trainer_layout(const ps_trainer* t, int row, char* name, int name_cap, int64_t* offset, int64_t* rows, int64_t* cols, int* is_buffer)
{
    PS_CHECK(t && row >= 0 && row < (int)t->rows.size(), "ps_trainer_layout: row out of range");
    const LayoutRow& r = t->rows[row];
    if (name && name_cap > 0) {
        std::strncpy(name, r.name.c_str(), (size_t)name_cap - 1);
        name[name_cap - 1] = 0;
    }
    if (offset) *offset = r.offset;
    if (rows) *rows = r.rows;
    if (cols) *cols = r.cols;
    if (is_buffer) *is_buffer = r.is_buffer;
    return PS_OK;
}

int ps_trainer_bind(ps_trainer* t, float* params, float* grads, float* adam_m, float* adam_v, float* bn_buffers)
{
    PS_CHECK(t && params && grads && bn_buffers, "ps_trainer_bind: params, grads and bn_buffers are required");
    // the recorded weight-image list holds pointers INTO the parameter buffer and replays its packing launches at the start of the
    // next step, before any product could notice a mismatch: a new buffer (checkpoint reload, buffer swap) drops the recording
    if (params != t->params) ps::pack_cache_clear(t->pack);
    t->params = params;
    t->grads = grads;
    t->adam_m = adam_m;
    t->adam_v = adam_v;
    t->buffers = bn_buffers;
    return PS_OK;
}

int ps_trainer_set_collective(ps_trainer* t, ps_allreduce_fn fn, void* user, int world_size, int rank, int sync_bn)
{
    PS_CHECK(t && world_size >= 1 && rank >= 0 && rank < world_size, "ps_trainer_set_collective: bad world size / rank");
    PS_CHECK(fn || world_size == 1, "ps_trainer_set_collective: a world of %d ranks needs an all-reduce callback", world_size);
    t->coll = fn;
    t->coll_user = user;
    t->world = world_size;
    t->rank = rank;
    t->sync_bn = (sync_bn & 1) != 0;
    t->coll_at_one = (sync_bn & PS_COLLECTIVE_AT_WORLD_ONE) != 0;
    return PS_OK;
}

int ps_trainer_set_step(ps_trainer* t, int64_t step)
{
    PS_CHECK(t && step >= 0, "ps_trainer_set_step: bad argument");
    t->step = step;
    return PS_OK;
}

int64_t ps_trainer_get_step(const ps_trainer* t) { return t ? t->step : -1; }

int ps_trainer_set_options(ps_trainer* t, const ps_train_options* opt)
{
    PS_CHECK(t && opt, "ps_trainer_set_options: NULL argument");
    PS_CHECK(opt->keep_prob > 0.f && opt->keep_prob <= 1.f && opt->learning_rate > 0.f, "ps_trainer_set_options: keep_prob must be in (0, 1], learning_rate > 0");
    PS_CHECK(opt->num_ignored == t->opt.num_ignored, "ps_trainer_set_options: the ignored labels are fixed at creation");
    t->opt = *opt;
    return PS_OK;
}

int64_t ps_trainer_pool_peak_bytes(const ps_trainer* t) { return t ? (int64_t)t->pool.peak : -1; }

int ps_trainer_set_profile(ps_trainer* t, int on)
{
    PS_CHECK(t, "ps_trainer_set_profile: trainer is NULL");
    t->profile = on != 0;
    return PS_OK;
}

int ps_trainer_collective_stats(const ps_trainer* t, int64_t* calls, int64_t* bytes, double* host_ms, double* device_ms)
{
    PS_CHECK(t, "ps_trainer_collective_stats: trainer is NULL");
    if (calls) *calls = t->coll_calls;
    if (bytes) *bytes = t->coll_bytes;
    if (host_ms) *host_ms = t->coll_host_ms;
    if (device_ms) *device_ms = t->coll_device_ms;
    return PS_OK;
}

int ps_trainer_profile(const ps_trainer* t, ps_timing_row* rows, int cap, int* n_rows)
{
    PS_CHECK(t && rows && n_rows && cap >= 0, "ps_trainer_profile: bad argument");
    int n = 0;
    for (const auto& r : t->profile_rows) {
        if (n >= cap) break;
        std::strncpy(rows[n].name, r.first.c_str(), sizeof(rows[n].name) - 1);
        rows[n].name[sizeof(rows[n].name) - 1] = 0;
        rows[n].ms = r.second;
        rows[n].launches = 0;
        ++n;
    }
    *n_rows = n;
    return PS_OK;
}

int ps_randla_backward(ps_trainer* t, const ps_pyramid* pyr, const float* features, const int32_t* labels, const float* class_weights, float* loss,
                       float* logits)
{
    PS_CHECK(t, "ps_randla_backward: trainer is NULL");
    return run_step(t, pyr, features, labels, class_weights, loss, logits, false);
}

int ps_randla_train_step(ps_trainer* t, const ps_pyramid* pyr, const float* features, const int32_t* labels, const float* class_weights, float* loss,
                         float* logits)
{
    PS_CHECK(t, "ps_randla_train_step: trainer is NULL");
    return run_step(t, pyr, features, labels, class_weights, loss, logits, true);
}

}  // extern "C"
