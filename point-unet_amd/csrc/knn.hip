// knn.hip -- exact KNN search kernel + the C-ABI entry points built on it (ps_knn_batch, ps_pyramid_build).
//
// Replaces cpp_knn_batch_omp (PointSegment/utils/nearest_neighbors/knn_.cxx:104-135) and the per-layer loop
// of tf_map (PointSegment/runBraTS.py:147-156).
//
// Kernel shape: one lane per query, the per-query routine of kdtree.h (iterative descent with an explicit
// deferred-node stack, top-K list in registers).  Lanes of a wave are given queries in the *support tree's
// leaf order* whenever the caller can provide it (self-queries and the pyramid's up-sampling queries), so a
// wave walks almost the same root-to-leaf path and its node/leaf loads hit the same cache lines.
// Roofline: neither HBM nor MFMA -- a divergent, latency-bound tree walk; reported in queries/s with the
// algorithmic bytes of SURVEY 8(d) next to it.
#include "common.h"
#include "kdtree.h"
#include "kdtree_build.h"

#include <cfloat>
#include <cstring>

namespace ps {

struct KnnJob {
    TreeView tree;
    const float4* q4;   // queries as (x,y,z,bitcast row) -- e.g. another tree's `pts` -- or NULL
    const float* q3;    // raw [nq,3] queries (row = query position) when q4 == NULL
    int32_t nq;
    int32_t* out;       // [nq, K]
    int32_t* overflow;  // set to 1 if any query overflowed the deferred-node stack
    int32_t* order;     // optional [nq]: order[t] = row of the t-th query (self queries: the tree's leaf order, for ps_pyramid.order)
    int32_t prefix = 0; // != 0: the tree holds exactly the rows [0, tree.n) of the cloud the q4 queries come from (up-sampling queries)
    int32_t sub_m = 0;  // rows [0, sub_m) are ALSO written to sub_out [sub_m, K]: the pooling table of tf_map, pool = neigh_idx[:, :N // r]
    int32_t* sub_out = nullptr;  // (runBraTS.py:150) -- written by the lane that owns the row, no slice launch behind the search
};

static_assert(sizeof(KnnJob) <= 128, "TreeSetPlan::carve reserves 128 bytes per job");

// Deferred-node stack of the search kernels: the kWin most recent entries of every lane live in LDS
// ([slot][word][thread]: a wave's access to one word is one conflict-free row), older ones spill to per-lane scratch
// and come back only when the window has drained.  A pop is the head of the next dependent node load, so its latency
// (LDS ~100 cycles vs a scratch round trip through L2/HBM) is on the critical path of every query; the first descent
// pushes ~log2(n/10) entries, of which the shallow ones -- spilled -- are almost always pruned by their `m` alone.
#ifndef PS_KNN_WIN
#define PS_KNN_WIN 8
#endif
constexpr int kWin = PS_KNN_WIN;  // (a power of two)
// Threads per search workgroup (the LDS window is [slot][word][thread]).  ONE wave: the kernel has no workgroup-level step, and a
// workgroup retires as soon as its own 64 queries are done instead of with the slowest of four waves -- its LDS and wave slot go to the next
// workgroup (or to another lane's kernel) that much earlier.  Measured (round 4, same box): 256 threads 0.182 ms / pipelined step 0.860,
// 128: 0.172 / 0.850, 64: 0.168 / 0.833, 512: 0.211 / 0.880.
#ifndef PS_KNN_THREADS
#define PS_KNN_THREADS 64
#endif
constexpr int kKnnThreads = PS_KNN_THREADS;
// the spill arrays live in their own object: indexed dynamically, they stay in scratch memory, and as members of WindowStack they
// kept its sp / lo counters there too (a scratch store per push, a scratch load in front of every pop)
struct SpillStore {
    int sid[kStackMax];
    float sm[kStackMax], s0[kStackMax], s1[kStackMax], s2[kStackMax];
};
struct WindowStack {
    typedef __attribute__((address_space(3))) float lds_float;
    lds_float* w;  // LDS base of this thread (already offset by threadIdx.x); word stride = kKnnThreads (one word per thread and slot)
    SpillStore* sp_;  // entries [0, lo)
    int sp = 0, lo = 0;  // entries [lo, sp) are in LDS at slot (depth % kWin); [0, lo) in scratch
    __device__ __forceinline__ bool push(int node, float mm, float a, float b, float c)
    {
        if (sp >= kStackMax) return false;
        if (sp - lo == kWin) {  // spill the oldest windowed entry
            const lds_float* e = w + (lo & (kWin - 1)) * 5 * kKnnThreads;
            sp_->sid[lo] = __float_as_int(e[0]); sp_->sm[lo] = e[kKnnThreads]; sp_->s0[lo] = e[2 * kKnnThreads]; sp_->s1[lo] = e[3 * kKnnThreads]; sp_->s2[lo] = e[4 * kKnnThreads];
            ++lo;
        }
        lds_float* e = w + (sp & (kWin - 1)) * 5 * kKnnThreads;
        e[0] = __int_as_float(node); e[kKnnThreads] = mm; e[2 * kKnnThreads] = a; e[3 * kKnnThreads] = b; e[4 * kKnnThreads] = c;
        ++sp;
        return true;
    }
    __device__ __forceinline__ bool pop(float worst, int& node, float& mm, float& a, float& b, float& c)
    {
        while (sp > 0) {
            --sp;
            if (sp >= lo) {
                const lds_float* e = w + (sp & (kWin - 1)) * 5 * kKnnThreads;
                const float em = e[kKnnThreads];
                if (em <= worst) {
                    node = __float_as_int(e[0]); mm = em; a = e[2 * kKnnThreads]; b = e[3 * kKnnThreads]; c = e[4 * kKnnThreads];
                    return true;
                }
            } else {
                lo = sp;  // the window is empty; this entry comes from scratch
                if (sp_->sm[sp] <= worst) {
                    node = sp_->sid[sp]; mm = sp_->sm[sp]; a = sp_->s0[sp]; b = sp_->s1[sp]; c = sp_->s2[sp];
                    return true;
                }
            }
        }
        return false;
    }
};

template <int K>
__device__ __forceinline__ void knn_body(const KnnJob& job, float* win)
{
    // XCD-aware order: workgroups go to the 8 XCDs round-robin, and queries are in leaf order, so workgroup b takes the
    // (b / 8)-th block of the (b % 8)-th eighth of the queries -- every XCD's L2 then serves one contiguous eighth of the tree's
    // leaves (plus the shared top levels) instead of all of it.
    const int per_xcd = (((job.nq + (int)blockDim.x - 1) / (int)blockDim.x) + 7) >> 3;  // of THIS job's workgroups
    if ((int)(blockIdx.x >> 3) >= per_xcd) return;
    const int bx = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    const int t = bx * blockDim.x + threadIdx.x;
    if (t >= job.nq) return;
    // (degenerate cloud: a builder queue overflowed and the tree has holes.  Unreachable for real clouds -- the queues are sized
    // for the worst disjoint-range count -- but whatever runs behind this kernel without a host check must still find valid
    // indices: the rows are filled with index 0 instead of walking a broken tree; the status word reports the failure.)
    const bool broken = gload(job.overflow + 1) != 0;
    float qx, qy, qz;
    int row;
    if (job.q4) {
        const float4 q = gload(job.q4 + t);
        qx = q.x; qy = q.y; qz = q.z;
        row = broken ? t : as_i(q.w);
        if (job.order) gstore(job.order + t, row);
    } else {
        qx = gload(job.q3 + 3 * (size_t)t);
        qy = gload(job.q3 + 3 * (size_t)t + 1);
        qz = gload(job.q3 + 3 * (size_t)t + 2);
        row = t;
    }
    if (broken) {
        for (int j = 0; j < K; ++j) gstore(job.out + (size_t)row * K + j, 0);
        if (row < job.sub_m)
            for (int j = 0; j < K; ++j) gstore(job.sub_out + (size_t)row * K + j, 0);
        return;
    }
    float dist[K];
    int idx[K];
    // Self-queries (the queries ARE this tree's points, lanes in leaf order): the K points around the query in leaf order are K
    // distinct tree points, so the K-th neighbour distance is at most the largest of their distances m.  The list then starts as K
    // phantom entries at a distance just above m instead of +inf: every true neighbour (d <= m) still enters in visit order and
    // pushes the phantoms out -- the result, ties included, is that of the same search on a tree with K extra far points --
    // but sub-trees beyond m are pruned from the first step and the early fill-the-list insertions disappear.
    float seed = FLT_MAX;
    if (K > 1 && job.q4 == job.tree.pts && job.nq >= K) {
        if constexpr (K <= 32) {
            if (job.nq >= 2 * K - 1) {
                // every K-wide window of the (2K-1)-point span around the query bounds the K-th distance by its largest member: take
                // the tightest.  (Leaf order is only locally spatial: a query next to a high split plane has far points on one side,
                // and in a wave the lane with the loosest bound sets the pace.)  Window [s, s+K) = suffix of the left half from s +
                // prefix of the right half up to s+K-1.
                const int w0 = min(max(t - (K - 1), 0), job.nq - (2 * K - 1));
                float d[2 * K - 1];
#pragma unroll
                for (int j = 0; j < 2 * K - 1; ++j) {
                    const float4 p = gload(job.q4 + w0 + j);
                    d[j] = sq_dist(qx, qy, qz, p.x, p.y, p.z);
                }
#pragma unroll
                for (int j = K - 3; j >= 0; --j) d[j] = fmaxf(d[j], d[j + 1]);          // d[j] = max(d[j .. K-2]), j <= K-2
#pragma unroll
                for (int j = K; j < 2 * K - 1; ++j) d[j] = fmaxf(d[j], d[j - 1]);      // d[j] = max(d[K-1 .. j]), j >= K-1
                float m = d[K - 1 + K - 1];                                            // window s = K-1: the right half alone
#pragma unroll
                for (int sft = 0; sft < K - 1; ++sft) m = fminf(m, fmaxf(d[sft], d[sft + K - 1]));
                seed = __fadd_rn(__fadd_rn(m, __fmul_rn(m, 1e-6f)), 1e-30f);  // strictly above m (m >= 0)
            }
        }
        if (seed == FLT_MAX) {
            const int w0 = min(max(t - K / 2, 0), job.nq - K);
            float m = 0.f;
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const float4 p = gload(job.q4 + w0 + j);
                m = fmaxf(m, sq_dist(qx, qy, qz, p.x, p.y, p.z));
            }
            seed = __fadd_rn(__fadd_rn(m, __fmul_rn(m, 1e-6f)), 1e-30f);
        }
    }
    if (K == 1 && job.prefix && job.q4 != nullptr && job.nq >= 32) {
        // up-sampling query (runBraTS.py:151): the queries are the points of level i in THEIR leaf order, the tree holds the prefix
        // subset xyz[:n] of the same cloud.  A leaf-order neighbour of the query whose row is < n is a point of the searched tree, so
        // its distance bounds the 1-NN distance: start from a phantom just above the smallest such distance among 32 neighbours
        // (same argument as above; a query that is in the subset itself finds distance 0).  Measured: the exact bound taken from the
        // query's finished K-NN list is no faster (0.060 vs 0.057 ms) -- what is left is the descent itself.
        const int w0 = min(max(t - 16, 0), job.nq - 32);
        float m = FLT_MAX;
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const float4 p = gload(job.q4 + w0 + j);
            const float dd = sq_dist(qx, qy, qz, p.x, p.y, p.z);
            m = as_i(p.w) < job.tree.n ? fminf(m, dd) : m;
        }
        if (m < FLT_MAX) seed = __fadd_rn(__fadd_rn(m, __fmul_rn(m, 1e-6f)), 1e-30f);
    }
#pragma unroll
    for (int j = 0; j < K; ++j) {
        dist[j] = seed;
        idx[j] = 0;
    }
    SpillStore spill;
    WindowStack st;
    st.sp_ = &spill;
    st.w = (WindowStack::lds_float*)(win + threadIdx.x);
    const bool ok = knn_search_one<K, WindowStack>(job.tree, qx, qy, qz, dist, idx, st);
    if (!ok) gstore(job.overflow, 1);
    int32_t* o = job.out + (size_t)row * K;
    int32_t* o2 = row < job.sub_m ? job.sub_out + (size_t)row * K : nullptr;
    if constexpr (K % 4 == 0) {
#pragma unroll
        for (int j = 0; j < K; j += 4) gstore(reinterpret_cast<int4*>(o + j), make_int4(idx[j], idx[j + 1], idx[j + 2], idx[j + 3]));
        if (o2) {
#pragma unroll
            for (int j = 0; j < K; j += 4) gstore(reinterpret_cast<int4*>(o2 + j), make_int4(idx[j], idx[j + 1], idx[j + 2], idx[j + 3]));
        }
    } else {
#pragma unroll
        for (int j = 0; j < K; ++j) gstore(o + j, idx[j]);
        if (o2) {
#pragma unroll
            for (int j = 0; j < K; ++j) gstore(o2 + j, idx[j]);
        }
    }
}


#ifdef PS_KNN_REFILL_EXP
// ---- EXPERIMENT (measured and rejected, round 6; only compiled with -DPS_KNN_REFILL_EXP, profiles/tools/exp_knn_refill.py) ----------------
// persistent form: a wave owns Q consecutive leaf-order queries and REFILLS the lanes whose query is finished.  Bit-identical tables, but the
// search time is one wave's latency chain, not issue slots: Q = 64 / 128 / 256 / 512 -> 0.183 / 0.317 / 0.53 / 0.91 ms (one query per lane:
// 0.175), i.e. +0.11 ms per 64 queries a wave owns whatever the number of waves on the chip.
// The one-query-per-lane form above runs a wave as long as its slowest lane: over 64 consecutive leaf-order queries the mean lane does
// 0.61-0.66 of the slowest one's leaf iterations, and the kernel is bound by VALU issue -- a third of the issued lane-slots is masked off.
// Here a lane that has finished its query takes the next one of the wave's block as soon as kKnnRefillMin lanes are idle (the
// while-while loop of Aila & Laine's persistent ray traversal): finished lanes hand their rows out, draw the next queries in order
// (rank among the idle lanes), start them (seed, root distances) and join the others in the descent loop.  The arithmetic of a query
// is exactly that of knn_search_one (kdtree.h): same visit order, same insertions, same result.
// The finished rows leave through the lanes' own (drained, hence free) columns of the LDS stack window: K / 4 adjacent lanes write one
// row's 4K contiguous bytes in ONE store instruction (a lane storing its own row as K / 4 16-byte pieces touches 64 different
// lines a quarter-row at a time).
#ifndef PS_KNN_Q
#define PS_KNN_Q 256
#endif
#ifndef PS_KNN_REFILL_MIN
#define PS_KNN_REFILL_MIN 16
#endif
static int knn_q() { const char* e = getenv("PS_KNN_Q"); return e ? atoi(e) : PS_KNN_Q; }                    // TEMPORARY (experiment)
static int knn_refill_min() { const char* e = getenv("PS_KNN_REFILL_MIN"); return e ? atoi(e) : PS_KNN_REFILL_MIN; }

#ifdef PS_KNN_PROF
__device__ unsigned long long g_knn_prof[16];
#define PS_PROF_T() __builtin_readcyclecounter()
#endif

// position of the r-th (0-based) set bit of m; r < popcount(m)
__device__ __forceinline__ int nth_set_bit(unsigned long long m, int r)
{
    unsigned w = (unsigned)m;
    int pos = 0;
    int c = __popc(w);
    if (r >= c) { r -= c; pos = 32; w = (unsigned)(m >> 32); }
#pragma unroll
    for (int s = 16; s >= 1; s >>= 1) {
        c = __popc(w & ((1u << s) - 1u));
        if (r >= c) { r -= c; pos += s; w >>= s; }
    }
    return pos;
}

template <int K>
__device__ __forceinline__ void knn_body_refill(const KnnJob& job, float* win, const int kKnnQ, const int kKnnRefillMin)
{
    static_assert(kKnnThreads == 64, "one wave per workgroup");
    typedef WindowStack::lds_float lds_float;
    const int lane = threadIdx.x;
    const int n_blocks = (job.nq + kKnnQ - 1) / kKnnQ;
    const int per_xcd = (n_blocks + 7) >> 3;  // XCD-aware order, as in knn_body
    if ((int)(blockIdx.x >> 3) >= per_xcd) return;
    const int bx = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    const int q_begin = bx * kKnnQ;
    if (q_begin >= job.nq) return;
    const int q_end = min(q_begin + kKnnQ, job.nq);
    if (gload(job.overflow + 1) != 0) {  // broken tree (builder queue overflow): valid indices, the status word reports the failure
        for (int t = q_begin + lane; t < q_end; t += 64) {
            const int row = t;
            if (job.q4 && job.order) gstore(job.order + t, row);
            for (int j = 0; j < K; ++j) gstore(job.out + (size_t)row * K + j, 0);
            if (row < job.sub_m)
                for (int j = 0; j < K; ++j) gstore(job.sub_out + (size_t)row * K + j, 0);
        }
        return;
    }
    const TreeView tr = job.tree;
    if (tr.n <= 0) {  // nothing to find: rows of zeros (the list's initial indices)
        for (int t = q_begin + lane; t < q_end; t += 64) {
            const int row = job.q4 ? as_i(gload(job.q4 + t).w) : t;
            if (job.q4 && job.order) gstore(job.order + t, row);
            for (int j = 0; j < K; ++j) gstore(job.out + (size_t)row * K + j, 0);
            if (row < job.sub_m)
                for (int j = 0; j < K; ++j) gstore(job.sub_out + (size_t)row * K + j, 0);
        }
        return;
    }
    const TreeMeta mt = *tr.meta;
    const bool self_seed = K > 1 && job.q4 == tr.pts && job.nq >= K;
    const bool wide_seed = self_seed && K <= 32 && job.nq >= 2 * K - 1;
    const bool up_seed = K == 1 && job.prefix && job.q4 != nullptr && job.nq >= 32;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;

    float dist[K];
    int idx[K];
#pragma unroll
    for (int j = 0; j < K; ++j) { dist[j] = FLT_MAX; idx[j] = 0; }
    SpillStore spill;
    WindowStack st;
    st.sp_ = &spill;
    st.w = (lds_float*)(win + lane);
    float qx = 0.f, qy = 0.f, qz = 0.f, m = 0.f, d0 = 0.f, d1 = 0.f, d2 = 0.f;
    int row = 0, cur = 0;
    bool have = false, active = false, ok = true;
    int next = q_begin;  // wave-uniform: first query of the block nobody has taken yet
#ifdef PS_KNN_PROF
    unsigned long long pf_t0 = PS_PROF_T(), pf_refill = 0, pf_desc = 0, pf_leaf = 0, pf_pop = 0, pf_iters = 0, pf_events = 0, pf_lanes = 0, pf_steps = 0;
#endif

    for (;;) {
#ifdef PS_KNN_PROF
        const unsigned long long pf_a = PS_PROF_T();
#endif
        const unsigned long long act = __ballot(active);
        const int n_idle = 64 - __popcll(act);
        if (act == 0ull || (next < q_end && n_idle >= kKnnRefillMin)) {
            // ---- hand out the rows of the finished lanes ----
            const unsigned long long fin = __ballot(!active && have);
            if (fin != 0ull) {
                if constexpr (K == 8 || K == 16 || K == 32) {
                    constexpr int P = K / 4, RPI = 64 / P;  // lanes per row, rows per store instruction
                    if (!active && have) {
#pragma unroll
                        for (int j = 0; j < K; ++j) st.w[j * kKnnThreads] = as_f(idx[j]);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    const int n_fin = __popcll(fin);
                    const lds_float* wbase = (const lds_float*)win;
                    for (int g0 = 0; g0 < n_fin; g0 += RPI) {
                        const int r = g0 + lane / P, piece = lane % P;
                        const int src = nth_set_bit(fin, min(r, n_fin - 1));
                        const int rrow = __shfl(row, src);
                        int4 v;
                        v.x = as_i(wbase[(piece * 4 + 0) * kKnnThreads + src]);
                        v.y = as_i(wbase[(piece * 4 + 1) * kKnnThreads + src]);
                        v.z = as_i(wbase[(piece * 4 + 2) * kKnnThreads + src]);
                        v.w = as_i(wbase[(piece * 4 + 3) * kKnnThreads + src]);
                        if (r < n_fin) {
                            gstore(reinterpret_cast<int4*>(job.out + (size_t)rrow * K + piece * 4), v);
                            if (rrow < job.sub_m) gstore(reinterpret_cast<int4*>(job.sub_out + (size_t)rrow * K + piece * 4), v);
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                } else {
                    if (!active && have) {
                        int32_t* o = job.out + (size_t)row * K;
                        int32_t* o2 = row < job.sub_m ? job.sub_out + (size_t)row * K : nullptr;
                        if constexpr (K % 4 == 0) {
#pragma unroll
                            for (int j = 0; j < K; j += 4) gstore(reinterpret_cast<int4*>(o + j), make_int4(idx[j], idx[j + 1], idx[j + 2], idx[j + 3]));
                            if (o2) {
#pragma unroll
                                for (int j = 0; j < K; j += 4)
                                    gstore(reinterpret_cast<int4*>(o2 + j), make_int4(idx[j], idx[j + 1], idx[j + 2], idx[j + 3]));
                            }
                        } else {
#pragma unroll
                            for (int j = 0; j < K; ++j) gstore(o + j, idx[j]);
                            if (o2) {
#pragma unroll
                                for (int j = 0; j < K; ++j) gstore(o2 + j, idx[j]);
                            }
                        }
                    }
                }
            }
            // ---- the idle lanes draw the next queries of the block, in order ----
            if (!active) {
                const int t = next + __popcll(~act & lt_mask);
                have = t < q_end;
                if (have) {
                    if (job.q4) {
                        const float4 q = gload(job.q4 + t);
                        qx = q.x; qy = q.y; qz = q.z;
                        row = as_i(q.w);
                        if (job.order) gstore(job.order + t, row);
                    } else {
                        qx = gload(job.q3 + 3 * (size_t)t);
                        qy = gload(job.q3 + 3 * (size_t)t + 1);
                        qz = gload(job.q3 + 3 * (size_t)t + 2);
                        row = t;
                    }
                    // the seeded list of knn_body (same bounds, same arithmetic)
                    float seed = FLT_MAX;
                    if constexpr (K > 1 && K <= 32) {
                        if (wide_seed) {
                            const int w0 = min(max(t - (K - 1), 0), job.nq - (2 * K - 1));
                            float d[2 * K - 1];
#pragma unroll
                            for (int j = 0; j < 2 * K - 1; ++j) {
                                const float4 p = gload(job.q4 + w0 + j);
                                d[j] = sq_dist(qx, qy, qz, p.x, p.y, p.z);
                            }
#pragma unroll
                            for (int j = K - 3; j >= 0; --j) d[j] = fmaxf(d[j], d[j + 1]);
#pragma unroll
                            for (int j = K; j < 2 * K - 1; ++j) d[j] = fmaxf(d[j], d[j - 1]);
                            float mm = d[K - 1 + K - 1];
#pragma unroll
                            for (int sft = 0; sft < K - 1; ++sft) mm = fminf(mm, fmaxf(d[sft], d[sft + K - 1]));
                            seed = __fadd_rn(__fadd_rn(mm, __fmul_rn(mm, 1e-6f)), 1e-30f);
                        }
                    }
                    if constexpr (K > 1) {
                        if (self_seed && !wide_seed) {
                            const int w0 = min(max(t - K / 2, 0), job.nq - K);
                            float mm = 0.f;
#pragma unroll
                            for (int j = 0; j < K; ++j) {
                                const float4 p = gload(job.q4 + w0 + j);
                                mm = fmaxf(mm, sq_dist(qx, qy, qz, p.x, p.y, p.z));
                            }
                            seed = __fadd_rn(__fadd_rn(mm, __fmul_rn(mm, 1e-6f)), 1e-30f);
                        }
                    }
                    if constexpr (K == 1) {
                        if (up_seed) {
                            const int w0 = min(max(t - 16, 0), job.nq - 32);
                            float mm = FLT_MAX;
#pragma unroll
                            for (int j = 0; j < 32; ++j) {
                                const float4 p = gload(job.q4 + w0 + j);
                                const float dd = sq_dist(qx, qy, qz, p.x, p.y, p.z);
                                mm = as_i(p.w) < tr.n ? fminf(mm, dd) : mm;
                            }
                            if (mm < FLT_MAX) seed = __fadd_rn(__fadd_rn(mm, __fmul_rn(mm, 1e-6f)), 1e-30f);
                        }
                    }
#pragma unroll
                    for (int j = 0; j < K; ++j) { dist[j] = seed; idx[j] = 0; }
                    // computeInitialDistances (nanoflann.hpp:1045-1061)
                    d0 = 0.f; d1 = 0.f; d2 = 0.f; m = 0.f;
                    if (qx < mt.lo[0]) { d0 = f_mul(f_sub(qx, mt.lo[0]), f_sub(qx, mt.lo[0])); m = f_add(m, d0); }
                    if (qx > mt.hi[0]) { d0 = f_mul(f_sub(qx, mt.hi[0]), f_sub(qx, mt.hi[0])); m = f_add(m, d0); }
                    if (qy < mt.lo[1]) { d1 = f_mul(f_sub(qy, mt.lo[1]), f_sub(qy, mt.lo[1])); m = f_add(m, d1); }
                    if (qy > mt.hi[1]) { d1 = f_mul(f_sub(qy, mt.hi[1]), f_sub(qy, mt.hi[1])); m = f_add(m, d1); }
                    if (qz < mt.lo[2]) { d2 = f_mul(f_sub(qz, mt.lo[2]), f_sub(qz, mt.lo[2])); m = f_add(m, d2); }
                    if (qz > mt.hi[2]) { d2 = f_mul(f_sub(qz, mt.hi[2]), f_sub(qz, mt.hi[2])); m = f_add(m, d2); }
                    cur = mt.root;
                    st.sp = 0;
                    st.lo = 0;
                    active = true;
                }
            }
            next += n_idle;
#ifdef PS_KNN_PROF
            ++pf_events;
#endif
            if (__ballot(active) == 0ull) break;
        }
#ifdef PS_KNN_PROF
        const unsigned long long pf_b = PS_PROF_T();
        pf_refill += pf_b - pf_a;
        ++pf_iters;
        pf_lanes += __popcll(__ballot(active));
#endif
        if (active) {
            // ---- descend to a leaf, deferring the far children (searchLevel, nanoflann.hpp:1372-1406; as knn_search_one) ----
#define PS_KD_VISIT(nd, left_first)                                                                                          \
    {                                                                                                                        \
        const int ax = (int)((unsigned)(nd).x >> 30);                                                                        \
        const int c1 = (nd).x & 0x3fffffff, c2 = (nd).y;                                                                     \
        const float divlow = as_f((nd).z), divhigh = as_f((nd).w);                                                           \
        const float val = ax == 0 ? qx : (ax == 1 ? qy : qz);                                                                \
        const float diff1 = f_sub(val, divlow), diff2 = f_sub(val, divhigh);                                                 \
        left_first = f_add(diff1, diff2) < 0.f;                                                                              \
        const float e = left_first ? diff2 : diff1;                                                                          \
        const float cut = f_mul(e, e);                                                                                       \
        const float dax = ax == 0 ? d0 : (ax == 1 ? d1 : d2);                                                                \
        const float m2 = f_sub(f_add(m, cut), dax);                                                                          \
        if (m2 <= dist[K - 1])                                                                                               \
            ok &= st.push(left_first ? c2 : c1, m2, ax == 0 ? cut : d0, ax == 1 ? cut : d1, ax == 2 ? cut : d2);             \
        cur = left_first ? c1 : c2;                                                                                          \
    }
            if (tr.fat) {
                while (cur & 1) {
#ifdef PS_KNN_PROF
                    ++pf_steps;
#endif
                    const int4* f = tr.fat + 3 * (size_t)cur;
                    const int4 nd = gload(f), kl = gload(f + 1), kr = gload(f + 2);
                    bool left;
                    PS_KD_VISIT(nd, left)
                    if (cur & 1) {
                        int4 kid;
                        kid.x = left ? kl.x : kr.x; kid.y = left ? kl.y : kr.y; kid.z = left ? kl.z : kr.z; kid.w = left ? kl.w : kr.w;
                        bool left2;
                        PS_KD_VISIT(kid, left2)
                    }
                }
            } else {
                while (cur & 1) {
                    const int4 nd = gload(tr.nodes + cur);
                    bool left;
                    PS_KD_VISIT(nd, left)
                }
            }
#undef PS_KD_VISIT
#ifdef PS_KNN_PROF
            const unsigned long long pf_c = PS_PROF_T();
            pf_desc += pf_c - pf_b;
#endif
            // ---- leaf: its points in vind order (nanoflann.hpp:1355-1369) ----
            {
                const int lf_x = (cur & kRefIdMask) >> 1, lf_y = lf_x + (cur >> kRefIdBits);
                float4 pv[kLeafMax];
#pragma unroll
                for (int j = 0; j < kLeafMax; ++j) pv[j] = gload(tr.pts + lf_x + j);
#pragma unroll
                for (int j = 0; j < kLeafMax; ++j) {
                    const float d = lf_x + j < lf_y ? sq_dist(qx, qy, qz, pv[j].x, pv[j].y, pv[j].z) : FLT_MAX;
                    if (__ballot(d < dist[K - 1]) != 0ull) topk_insert<K>(dist, idx, d, as_i(pv[j].w));
                }
            }
#ifdef PS_KNN_PROF
            const unsigned long long pf_d = PS_PROF_T();
            pf_leaf += pf_d - pf_c;
#endif
            // ---- resume at the most recent deferred child that still passes the prune test ----
            if (!st.pop(dist[K - 1], cur, m, d0, d1, d2)) active = false;
#ifdef PS_KNN_PROF
            pf_pop += PS_PROF_T() - pf_d;
#endif
        }
    }
#ifdef PS_KNN_PROF
    if (K > 1 && lane == 0) {
        atomicAdd(&g_knn_prof[0], 1ull);
        atomicAdd(&g_knn_prof[1], PS_PROF_T() - pf_t0);
        atomicAdd(&g_knn_prof[2], pf_refill);
        atomicAdd(&g_knn_prof[3], pf_desc);
        atomicAdd(&g_knn_prof[4], pf_leaf);
        atomicAdd(&g_knn_prof[5], pf_pop);
        atomicAdd(&g_knn_prof[6], pf_iters);
        atomicAdd(&g_knn_prof[7], pf_events);
        atomicAdd(&g_knn_prof[8], pf_lanes);
        atomicAdd(&g_knn_prof[9], pf_steps);
    }
#endif
    if (!ok) gstore(job.overflow, 1);
}

template <int K>
__global__ __launch_bounds__(kKnnThreads) void knn_refill_kernel(const KnnJob* __restrict__ jobs, int q, int rmin)
{
    __shared__ float win[kWin * 5 * kKnnThreads];
    knn_body_refill<K>(jobs[blockIdx.y], win, q, rmin);
}

template <int K>
__global__ __launch_bounds__(kKnnThreads) void knn_pair_refill_kernel(const KnnJob* __restrict__ jobs, int n_first, int q, int rmin)
{
    __shared__ float win[kWin * 5 * kKnnThreads];
    if ((int)blockIdx.y < n_first) knn_body_refill<K>(jobs[blockIdx.y], win, q, rmin);
    else knn_body_refill<1>(jobs[blockIdx.y], win, q, rmin);
}

#endif  // PS_KNN_REFILL_EXP

// (experiment knob: -DPS_KNN_WAVES_PER_EU=n makes the register allocator fit n waves per SIMD)
#ifdef PS_KNN_WAVES_PER_EU
#define PS_KNN_OCC __attribute__((amdgpu_waves_per_eu(PS_KNN_WAVES_PER_EU, PS_KNN_WAVES_PER_EU)))
#else
#define PS_KNN_OCC
#endif

template <int K>
__global__ __launch_bounds__(kKnnThreads) PS_KNN_OCC void knn_kernel(const KnnJob* __restrict__ jobs)
{
    __shared__ float win[kWin * 5 * kKnnThreads];
    knn_body<K>(jobs[blockIdx.y], win);
}

// The pyramid's two searches in ONE launch: jobs [0, n_first) are the K-NN self queries, the rest the 1-NN up-sampling queries.  The
// K-NN search ends with a long tail (its duration is its slowest wave's; the chip holds all of its waves at once), and a second launch
// cannot start under it: here the 1-NN workgroups are dispatched as the K-NN ones retire.
template <int K>
__global__ __launch_bounds__(kKnnThreads) PS_KNN_OCC void knn_pair_kernel(const KnnJob* __restrict__ jobs, int n_first)
{
    __shared__ float win[kWin * 5 * kKnnThreads];
    if ((int)blockIdx.y < n_first) knn_body<K>(jobs[blockIdx.y], win);
    else knn_body<1>(jobs[blockIdx.y], win);
}

__global__ void widen_kernel(const int32_t* __restrict__ in, int64_t* __restrict__ out, size_t count)
{
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < count) out[i] = in[i];
}

#ifdef PS_KNN_REFILL_EXP
static bool knn_refill_on()
{
    const char* e = getenv("PS_KNN_REFILL");  // (experiment build only; uncached: flipped inside one process by exp_knn_refill.py)
    return e ? atoi(e) != 0 : false;
}
#endif

#ifdef PS_KNN_REFILL_EXP
#define PS_KNN_REFILL_LAUNCH(kern, ...) if (refill) hipLaunchKernelGGL(kern, grid, dim3(kKnnThreads), 0, c->stream, __VA_ARGS__); else
#else
#define PS_KNN_REFILL_LAUNCH(kern, ...)
#endif

static int launch_knn(ps_context* c, const KnnJob* d_jobs, int n_jobs, int max_nq, int K)
{
#ifdef PS_KNN_REFILL_EXP
    const bool refill = knn_refill_on();
    const int kq = knn_q(), krm = knn_refill_min();
#else
    constexpr bool refill = false;
    constexpr int kq = kKnnThreads;
#endif
    dim3 grid((ceil_div(max_nq, refill ? kq : kKnnThreads) + 7) & ~7, n_jobs);  // a multiple of 8: the XCD remap in the kernel covers [0, grid) exactly
    if (grid.x == 0 || n_jobs == 0) return PS_OK;
    switch (K) {
#define PS_KCASE(k)                                                             \
    case k:                                                                     \
        PS_KNN_REFILL_LAUNCH(knn_refill_kernel<k>, d_jobs, kq, krm)                                                   \
        hipLaunchKernelGGL(knn_kernel<k>, grid, dim3(kKnnThreads), 0, c->stream, d_jobs); \
        break;
#ifdef PS_KNN_FEW_K
        PS_KCASE(1) PS_KCASE(16) PS_KCASE(32)
#else
        PS_KCASE(1) PS_KCASE(2) PS_KCASE(3) PS_KCASE(4) PS_KCASE(5) PS_KCASE(6) PS_KCASE(7) PS_KCASE(8)
        PS_KCASE(9) PS_KCASE(10) PS_KCASE(11) PS_KCASE(12) PS_KCASE(13) PS_KCASE(14) PS_KCASE(15) PS_KCASE(16)
        PS_KCASE(20) PS_KCASE(24) PS_KCASE(32) PS_KCASE(48) PS_KCASE(64)
#endif
#undef PS_KCASE
        default:
            set_error("ps_knn: K=%d is not a compiled size (1..16, 20, 24, 32, 48, 64)", K);
            return PS_EINVAL;
    }
    PS_HIP(hipGetLastError());
    return PS_OK;
}

static int launch_knn_pair(ps_context* c, const KnnJob* d_jobs, int n_first, int n_jobs, int max_nq, int K)
{
    if (K == 1) return launch_knn(c, d_jobs, n_jobs, max_nq, 1);
#ifdef PS_KNN_REFILL_EXP
    const bool refill = knn_refill_on();
    const int kq = knn_q(), krm = knn_refill_min();
#else
    constexpr bool refill = false;
    constexpr int kq = kKnnThreads;
#endif
    dim3 grid((ceil_div(max_nq, refill ? kq : kKnnThreads) + 7) & ~7, n_jobs);
    if (grid.x == 0 || n_jobs == 0) return PS_OK;
    switch (K) {
#define PS_KCASE(k)                                                                            \
    case k:                                                                                    \
        PS_KNN_REFILL_LAUNCH(knn_pair_refill_kernel<k>, d_jobs, n_first, kq, krm)                                      \
        hipLaunchKernelGGL(knn_pair_kernel<k>, grid, dim3(kKnnThreads), 0, c->stream, d_jobs, n_first); \
        break;
#ifdef PS_KNN_FEW_K
        PS_KCASE(16) PS_KCASE(32)
#else
        PS_KCASE(2) PS_KCASE(3) PS_KCASE(4) PS_KCASE(5) PS_KCASE(6) PS_KCASE(7) PS_KCASE(8)
        PS_KCASE(9) PS_KCASE(10) PS_KCASE(11) PS_KCASE(12) PS_KCASE(13) PS_KCASE(14) PS_KCASE(15) PS_KCASE(16)
        PS_KCASE(20) PS_KCASE(24) PS_KCASE(32) PS_KCASE(48) PS_KCASE(64)
#endif
#undef PS_KCASE
        default:
            set_error("ps_pyramid_build: K=%d is not a compiled size (1..16, 20, 24, 32, 48, 64)", K);
            return PS_EINVAL;
    }
    PS_HIP(hipGetLastError());
    return PS_OK;
}

}  // namespace ps

using namespace ps;

#ifdef PS_KNN_PROF
extern "C" int ps_debug_knn_prof(unsigned long long* out16, int reset)
{
    if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_knn_prof), 128) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_knn_prof), z, 128) != hipSuccess) return -1;
    }
    return 0;
}
#endif

// --------------------------------------------------------------------------------------------------------
// ps_knn_batch
// --------------------------------------------------------------------------------------------------------
static int knn_batch_impl(ps_context* c, const float* support, const float* queries, int64_t B, int64_t n1, int64_t n2,
                          int64_t dim, int64_t K, int32_t* out32, int64_t* out64, int device_ptrs)
{
    PS_CHECK(c != nullptr, "ps_knn_batch: ctx is NULL");
    PS_CHECK(dim == 3, "ps_knn_batch: dim must be 3 (got %lld)", (long long)dim);
    PS_CHECK(B >= 0 && n1 >= 0 && n2 >= 0 && K >= 1, "ps_knn_batch: negative size or K < 1");
    PS_CHECK(n1 < kMaxTreePoints, "ps_knn_batch: n_support too large (limit 2^25 points per cloud)");
    PS_CHECK(out32 != nullptr || out64 != nullptr, "ps_knn_batch: out_idx is NULL");
    if (B == 0 || n2 == 0) return PS_OK;
    PS_CHECK(support != nullptr || n1 == 0, "ps_knn_batch: support is NULL");
    PS_CHECK(queries != nullptr, "ps_knn_batch: queries is NULL");
    PS_HIP(hipSetDevice(c->device));

    const size_t out_count = (size_t)B * n2 * K;
    const float* d_support = support;
    const float* d_queries = queries;
    int32_t* d_out32 = out32;
    int64_t* d_out64 = out64;
    if (!device_ptrs) {
        const size_t sb = (size_t)B * n1 * 3 * sizeof(float), qb = (size_t)B * n2 * 3 * sizeof(float);
        PS_TRY(c->stage_in.reserve(sb + qb + 512));
        PS_TRY(c->stage_out.reserve(out_count * (out64 ? 12 : 4) + 512));
        float* ds = c->stage_in.as<float>();
        float* dq = reinterpret_cast<float*>(c->stage_in.as<char>() + ((sb + 255) & ~size_t(255)));
        if (sb) PS_HIP(hipMemcpyAsync(ds, support, sb, hipMemcpyHostToDevice, c->stream));
        PS_HIP(hipMemcpyAsync(dq, queries, qb, hipMemcpyHostToDevice, c->stream));
        d_support = ds;
        d_queries = dq;
        d_out32 = c->stage_out.as<int32_t>();
        d_out64 = out64 ? reinterpret_cast<int64_t*>(c->stage_out.as<char>() + ((out_count * 4 + 255) & ~size_t(255))) : nullptr;
    } else if (out64) {
        // device int64 output: search into a scratch int32 image first
        PS_TRY(c->stage_out.reserve(out_count * 4 + 256));
        d_out32 = c->stage_out.as<int32_t>();
    }

    // plan + carve the workspace: B trees and the job table
    TreeSetPlan plan;
    for (int64_t b = 0; b < B; ++b) plan.add((int32_t)n1);
    for (int pass = 0; pass < 2; ++pass) {
        c->knn_arena.begin(pass == 0);
        plan.carve(c->knn_arena);
        if (pass == 0) PS_TRY(c->knn_arena.buf.reserve(c->knn_arena.off));
    }
    std::vector<KnnJob> jobs(B);
    for (int64_t b = 0; b < B; ++b) {
        jobs[b].tree = plan.view((int)b);
        jobs[b].q4 = nullptr;
        jobs[b].q3 = d_queries + (size_t)b * n2 * 3;
        jobs[b].nq = (int32_t)n2;
        jobs[b].out = d_out32 + (size_t)b * n2 * K;
        jobs[b].overflow = plan.d_flags;
        jobs[b].order = nullptr;
    }
    std::memcpy(plan.host_jobs(), jobs.data(), sizeof(KnnJob) * B);  // (travels with the builder's own tables: one copy, kdtree_build.h)
    {
        Stage st(c, "kdtree_build", 1);
        for (int64_t b = 0; b < B; ++b) plan.src[b] = d_support + (size_t)b * n1 * 3;
        PS_TRY(build_trees(c, plan));
    }
    int32_t flag[3] = {0, 0, 0};
    {
        Stage st(c, "knn_search", 1);
        PS_TRY(launch_knn(c, reinterpret_cast<const KnnJob*>(plan.d_jobs), (int)B, (int)n2, (int)K));
    }
    if (out64) {
        hipLaunchKernelGGL(widen_kernel, dim3(ceil_div(out_count, 256)), dim3(256), 0, c->stream, d_out32, d_out64 ? d_out64 : out64,
                           out_count);
        PS_HIP(hipGetLastError());
    }
    if (!device_ptrs) {
        if (out64)
            PS_HIP(hipMemcpyAsync(out64, d_out64, out_count * 8, hipMemcpyDeviceToHost, c->stream));
        else
            PS_HIP(hipMemcpyAsync(out32, d_out32, out_count * 4, hipMemcpyDeviceToHost, c->stream));
    }
    PS_HIP(hipMemcpyAsync(flag, plan.d_flags, 12, hipMemcpyDeviceToHost, c->stream));
    // (the job table and the plan's staging live in host memory of this frame: wait before returning)
    PS_HIP(hipStreamSynchronize(c->stream));
    PS_CHECK(flag[1] == 0, "ps_knn_batch: kd-tree builder queue overflow (degenerate cloud)");
    PS_CHECK(flag[0] == 0, "ps_knn_batch: kd-tree deeper than the %d-entry traversal stack", kStackMax);
    return PS_OK;
}

extern "C" int ps_knn_batch(ps_context* c, const float* support, const float* queries, int64_t B, int64_t n1, int64_t n2,
                            int64_t dim, int64_t K, int32_t* out_idx, int device_ptrs)
{
    return knn_batch_impl(c, support, queries, B, n1, n2, dim, K, out_idx, nullptr, device_ptrs);
}

extern "C" int ps_knn_batch_i64(ps_context* c, const float* support, const float* queries, int64_t B, int64_t n1, int64_t n2,
                                int64_t dim, int64_t K, int64_t* out_idx, int device_ptrs)
{
    return knn_batch_impl(c, support, queries, B, n1, n2, dim, K, nullptr, out_idx, device_ptrs);
}

// --------------------------------------------------------------------------------------------------------
// ps_pyramid_build
// --------------------------------------------------------------------------------------------------------
namespace ps {

}  // namespace ps

extern "C" int ps_pyramid_build(ps_context* c, const float* xyz0, int64_t B, int64_t n0, int32_t L, const int32_t* ratios, int32_t K,
                                ps_pyramid* pyr)
{
    PS_CHECK(c && xyz0 && ratios && pyr, "ps_pyramid_build: NULL argument");
    PS_CHECK(L >= 1 && L <= PS_MAX_LAYERS, "ps_pyramid_build: num_layers %d out of range", L);
    PS_CHECK(B >= 1 && n0 >= 1 && n0 < kMaxTreePoints, "ps_pyramid_build: bad B / n0");
    PS_HIP(hipSetDevice(c->device));
    int64_t n[PS_MAX_LAYERS + 1];
    n[0] = n0;
    for (int i = 0; i < L; ++i) {
        PS_CHECK(ratios[i] >= 1, "ps_pyramid_build: ratio[%d] < 1", i);
        n[i + 1] = n[i] / ratios[i];
        PS_CHECK(n[i + 1] >= 1, "ps_pyramid_build: level %d would be empty", i + 1);
    }
    pyr->built = 0;
    pyr->num_layers = L;
    pyr->K = K;
    pyr->B = B;
    for (int i = 0; i <= L; ++i) pyr->n[i] = n[i];
    for (int i = 0; i < L; ++i)
        PS_CHECK(pyr->xyz[i] && pyr->neigh_idx[i] && pyr->sub_idx[i] && pyr->interp_idx[i], "ps_pyramid_build: NULL buffer at layer %d", i);

    // trees: (level l in 0..L) x (cloud b).  Level l's point set is the first n[l] points of every cloud.
    TreeSetPlan plan;
    for (int l = 0; l <= L; ++l)
        for (int64_t b = 0; b < B; ++b) plan.add((int32_t)n[l]);
    plan.extra_jobs = (int)(2 * L * B);
    for (int pass = 0; pass < 2; ++pass) {
        c->knn_arena.begin(pass == 0);
        plan.carve(c->knn_arena);
        if (pass == 0) PS_TRY(c->knn_arena.buf.reserve(c->knn_arena.off));
    }
    // every tree reads its prefix of the caller's cloud; the builders' first pass also writes those rows to ps_pyramid.xyz[level]
    // (xyz[i] = xyz[:, :n_i], runBraTS.py:149; xyz[0] may be the caller's own buffer) -- no slice launch
    plan.copy_dst.assign((size_t)(L + 1) * B, nullptr);
    for (int l = 0; l <= L; ++l)
        for (int64_t b = 0; b < B; ++b) {
            plan.src[l * B + b] = xyz0 + (size_t)b * n0 * 3;
            if (l < L && !(l == 0 && pyr->xyz[0] == xyz0)) plan.copy_dst[l * B + b] = pyr->xyz[l] + (size_t)b * n[l] * 3;
        }

    // jobs: K-NN self queries per level, then 1-NN up-sampling queries per level, both in the query set's own
    // tree (leaf) order so neighbouring lanes walk neighbouring paths.
    std::vector<KnnJob> jobs;
    jobs.reserve(2 * L * B);
    int max_nq = 0;
    for (int l = 0; l < L; ++l)
        for (int64_t b = 0; b < B; ++b) {
            KnnJob j;
            j.tree = plan.view((int)(l * B + b));
            j.q4 = j.tree.pts;
            j.q3 = nullptr;
            j.nq = (int32_t)n[l];
            j.out = pyr->neigh_idx[l] + (size_t)b * n[l] * K;
            j.overflow = plan.d_flags;
            j.order = pyr->order[l] ? pyr->order[l] + (size_t)b * n[l] : nullptr;
            j.sub_m = (int32_t)n[l + 1];
            j.sub_out = pyr->sub_idx[l] + (size_t)b * n[l + 1] * K;
            jobs.push_back(j);
            max_nq = std::max(max_nq, j.nq);
        }
    const size_t n_self = jobs.size();
    for (int l = 0; l < L; ++l)
        for (int64_t b = 0; b < B; ++b) {
            KnnJob j;
            j.tree = plan.view((int)((l + 1) * B + b));
            j.q4 = plan.view((int)(l * B + b)).pts;
            j.q3 = nullptr;
            j.nq = (int32_t)n[l];
            j.out = pyr->interp_idx[l] + (size_t)b * n[l];
            j.overflow = plan.d_flags;
            j.order = nullptr;
            j.prefix = 1;
            jobs.push_back(j);
        }
    std::memcpy(plan.host_jobs(), jobs.data(), sizeof(KnnJob) * jobs.size());  // (uploaded by build_trees with its own tables: one copy)
    {
        Stage st(c, "kdtree_build", 1);
        PS_TRY(build_trees(c, plan));
        st.n = plan.launches;
    }
    const KnnJob* dj = reinterpret_cast<const KnnJob*>(plan.d_jobs);
    int32_t flag[3] = {0, 0, 0};
    {
        {
            Stage st(c, "knn_search", 1);  // K-NN self queries and 1-NN up-sampling queries of every level, one launch
            PS_TRY(launch_knn_pair(c, dj, (int)n_self, (int)jobs.size(), max_nq, (int)K));
        }
        if (c->deferred) {
            // no host synchronisation: the status words go to a pinned slot that ps_synchronize() validates; the host
            // tables behind the asynchronous uploads move into the context's ring so they outlive this frame
            hipEvent_t& ev = c->flag_ev[c->flag_slot];
            if (!ev) PS_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            if (c->pending_mask & (1u << c->flag_slot)) {
                // the slot still holds the status words of the build eight calls ago: wait for THAT copy (long done), not for
                // the stream -- a stream synchronisation here would serialise the host with the GPU every eighth cloud.  A failure
                // it reports is remembered in the context and raised by ps_synchronize: this build's slot, serial and event must
                // line up with the caller's submission count whatever the old build did
                PS_HIP(hipEventSynchronize(ev));
                const int stale_rc = c->check_flag_slot(c->flag_slot);
                if (stale_rc != PS_OK && c->sticky_rc == PS_OK) {  // kept for ps_synchronize (the caller's submit loop never sees it)
                    c->sticky_rc = stale_rc;
                    c->sticky_msg = ps_last_error();
                }
            }
            PS_HIP(hipMemcpyAsync(c->h_flags + 4 * c->flag_slot, plan.d_flags, 12, hipMemcpyDeviceToHost, c->stream));
            PS_HIP(hipEventRecord(ev, c->stream));
            c->pending_mask |= 1u << c->flag_slot;
            c->flag_serial[c->flag_slot] = c->builds;
            ++c->builds;
            c->flag_slot = (c->flag_slot + 1) & 7;
            pyr->built = pyramid_stamp(pyr);
            return PS_OK;  // (all host tables went through the context's pinned upload ring)
        }
        PS_HIP(hipMemcpyAsync(flag, plan.d_flags, 12, hipMemcpyDeviceToHost, c->stream));
        PS_HIP(hipStreamSynchronize(c->stream));
    }
    ++c->builds;
    PS_CHECK(flag[1] == 0, "ps_pyramid_build: kd-tree builder queue overflow (degenerate cloud)");
    PS_CHECK(flag[0] == 0, "ps_pyramid_build: kd-tree deeper than the %d-entry traversal stack", kStackMax);
    pyr->built = pyramid_stamp(pyr);
    return PS_OK;
}
