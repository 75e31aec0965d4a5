// kdtree_build.hip -- builds every kd-tree of a TreeSetPlan ON THE DEVICE, bit-identical (same `vind` permutation,
// same splits, same child order) to nanoflann 1.2.3's recursive builder as used by the reference
// (PointSegment/utils/nearest_neighbors/nanoflann.hpp:916-1043 divideTree / middleSplit_ / planeSplit, :1321-1343
// computeBoundingBox; leaf_max_size 10, knn_.cxx:116).
//
// The recursion is replaced by task queues over three tiers of node size (kernel boundaries are the only global sync):
//   init kernels     points -> (x,y,z,index) records; root bounding boxes by block reduction + ordered-uint atomics
//   chunked levels   nodes above kMid points: every pass of a level (classify + counts, then rank -> swap for each of the two Hoare
//                    sweeps; five in all, see "chunked levels" below) is one small kernel over 2 048-record chunks spread over the chip
//   mid kernel       nodes of kSmall+1 .. kMid points: ONE workgroup loads the node into LDS and advances level by level
//                    over all of its live segments at once, with __syncthreads() only
//   flat kernel      nodes of <= kSmall points: one workgroup, one thread per point position; every round advances all live
//                    segments of the node at once (block-wide flag scans for the ranks of the closed-form Hoare sweeps)
//   straggler kernel (very unbalanced clouds) one workgroup per node still above kMid after the chunked levels finishes
//                    everything above kMid below it depth first -- ONE launch, nothing left for the host to repair
// nanoflann's two Hoare sweeps are reproduced in CLOSED FORM everywhere -- the i-th misplaced element from the left swaps
// with the i-th misplaced element from the right, so ranks from scans of two flag vectors give every swap pair.
// Node ids are position-derived (kdtree.h), so the result does not depend on scheduling.  divlow / divhigh are the
// children's tight extents on the split axis, which the parent can compute at split time as max{v < cut-side} /
// min{v > cut-side} (the children's bounding boxes are never needed otherwise).
//
// Bound: launch latency for the chunked levels (~28 us per level: five dependent launches), instruction issue below; HBM traffic is a few
// passes over 16 B per point per chunked level plus one read and one write per LDS tier.
#include "kdtree_build.h"
#include "wave_ops.h"

#include <algorithm>
#include <cstring>
#include <type_traits>

namespace ps {

constexpr int kSmall = 256;     // nodes up to this many points are finished by one workgroup, one thread per point (build_flat_kernel;
                                // 1 024 was measured: flat kernel 44 -> 88 us, mid kernel 106 -> 73 us, no gain)
constexpr int kMid = 4096;      // nodes up to this many points are split down to <= kSmall by one workgroup in LDS.  Round 4: 8 192 -> 4 096
                                // (one more chunked level, +30 us, for a mid kernel of 57 instead of 104 us: its levels cost a workgroup
                                // time in proportion to its points; 2 048 measured no better: 55 us, another +30 us of chunked passes)
constexpr int kMidThreads = 1024;
constexpr int kBigThreads = 512;
constexpr int kMaxLevels = 96;  // per-level task counters

struct BuildTask {
    int32_t tree, l, r, parent, side, level;
    float lo[3], hi[3];  // incoming bounding box (nanoflann passes the parent's box cut at the plane)
};

struct BuildTree {
    const float* src;
    float4* pts;
    int4* nodes;
    int4* fat;      // [3 * 2n] (TreeView::fat), filled by fatten_kernel
    float* copy_dst;  // optional [n, 3]: the source rows are also written here (ps_pyramid.xyz[level]: no separate slice launch)
    TreeMeta* meta;
    int32_t* posL;  // scratch [n]
    int32_t* posR;  // scratch [n]
    uint8_t* cls;   // scratch [n]: class of a record against the current cut (chunked levels)
    unsigned* bbox_ord;  // [6] ordered-uint min[3], max[3]
    int32_t n;
};

struct BuildQueues {
    BuildTask* q[2];
    BuildTask* mid_q;    // nodes of kSmall+1 .. kMid points (build_mid_kernel)
    BuildTask* small_q;  // nodes of <= kSmall points (build_flat_kernel)
    int32_t* level_cnt;  // [kMaxLevels]
    int32_t* small_cnt;  // small_cnt[0] = pushed, [1] = mid pushed, [2] = mid done, [3] = small done
    int32_t* flags;      // flags[1] = queue overflow
    int32_t q_cap, small_cap;
};

__device__ __forceinline__ unsigned f2ord(float f)
{
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned u)
{
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}
__device__ __forceinline__ float comp(const float4& p, int ax) { return ax == 0 ? p.x : (ax == 1 ? p.y : p.z); }

// Global-address-space view of a device array reached through a descriptor in memory (gmem.h): reads a[i], writes a.set(i, v).
template <class T>
struct GArr {
    T* p;
    __device__ __forceinline__ typename std::remove_const<T>::type operator[](size_t i) const { return gload(static_cast<const T*>(p) + i); }
    __device__ __forceinline__ void set(size_t i, const T& v) const { gstore(p + i, v); }
};

// ---- init -----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void init_points_kernel(const BuildTree* __restrict__ trees, int chunks_x)
{
    __shared__ float s_mn[4][3], s_mx[4][3];
    const BuildTree t = trees[blockIdx.y];
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < t.n; i += chunks_x * 256) {
        const float x = gload(t.src + 3 * (size_t)i), y = gload(t.src + 3 * (size_t)i + 1), z = gload(t.src + 3 * (size_t)i + 2);
        gstore(t.pts + i, make_float4(x, y, z, __int_as_float(i)));
        if (t.copy_dst) {
            gstore(t.copy_dst + 3 * (size_t)i, x);
            gstore(t.copy_dst + 3 * (size_t)i + 1, y);
            gstore(t.copy_dst + 3 * (size_t)i + 2, z);
        }
        mn[0] = fminf(mn[0], x); mx[0] = fmaxf(mx[0], x);
        mn[1] = fminf(mn[1], y); mx[1] = fmaxf(mx[1], y);
        mn[2] = fminf(mn[2], z); mx[2] = fmaxf(mx[2], z);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        {
            mn[a] = wave_min(mn[a]);
            mx[a] = wave_max(mx[a]);
        }
        if ((threadIdx.x & 63) == 0) { s_mn[threadIdx.x >> 6][a] = mn[a]; s_mx[threadIdx.x >> 6][a] = mx[a]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {  // one pair of atomics per axis per workgroup
        const int a = threadIdx.x;
        const float lo = fminf(fminf(s_mn[0][a], s_mn[1][a]), fminf(s_mn[2][a], s_mn[3][a]));
        const float hi = fmaxf(fmaxf(s_mx[0][a], s_mx[1][a]), fmaxf(s_mx[2][a], s_mx[3][a]));
        if (lo <= hi) {
            atomicMin(&t.bbox_ord[a], f2ord(lo));
            atomicMax(&t.bbox_ord[3 + a], f2ord(hi));
        }
    }
}

__device__ __forceinline__ void push_task(const BuildQueues& Q, const BuildTask& t, int next_level)
{
    if (t.r - t.l > kMid) {
        const int slot = atomicAdd(&Q.level_cnt[next_level], 1);
        if (slot < Q.q_cap && next_level < kMaxLevels - 1)
            Q.q[next_level & 1][slot] = t;
        else
            Q.flags[1] = 1;
    } else if (t.r - t.l > kSmall) {
        const int slot = atomicAdd(Q.small_cnt + 1, 1);
        if (slot < Q.q_cap)
            Q.mid_q[slot] = t;
        else
            Q.flags[1] = 1;
    } else {
        const int slot = atomicAdd(Q.small_cnt, 1);
        if (slot < Q.small_cap)
            Q.small_q[slot] = t;
        else
            Q.flags[1] = 1;
    }
}

// ---- the split decision shared by both kernels (middleSplit_, nanoflann.hpp:966-1005) -------------------------
struct SplitChoice {
    int ax;
    float cut;
};
__device__ __forceinline__ SplitChoice choose_split(const float* lo, const float* hi, const float* mn, const float* mx)
{
    float max_span = __fsub_rn(hi[0], lo[0]);
    for (int a = 1; a < 3; ++a) {
        const float s = __fsub_rn(hi[a], lo[a]);
        if (s > max_span) max_span = s;
    }
    const float thresh = __fmul_rn(__fsub_rn(1.0f, 0.00001f), max_span);
    int cutfeat = 0;
    float max_spread = -1.f;
    for (int a = 0; a < 3; ++a) {
        if (__fsub_rn(hi[a], lo[a]) > thresh) {
            const float spread = __fsub_rn(mx[a], mn[a]);
            if (spread > max_spread) {
                cutfeat = a;
                max_spread = spread;
            }
        }
    }
    const float split = __fmul_rn(__fadd_rn(lo[cutfeat], hi[cutfeat]), 0.5f);
    SplitChoice c;
    c.ax = cutfeat;
    c.cut = split < mn[cutfeat] ? mn[cutfeat] : (split > mx[cutfeat] ? mx[cutfeat] : split);
    return c;
}

// The same decision with the chosen axis' values carried along as scalars: in the straggler kernel, whose box lives in LDS, the
// run-time index `lo[cutfeat]` above put the arrays into scratch memory (16 bytes per lane -- and a kernel that needs scratch pays
// for the private-segment set-up on every dispatch).  (Used only there: the same body inside build_flat_kernel crashes hipcc 7.2's
// instruction selection.)
__device__ __forceinline__ SplitChoice choose_split_scalar(const float* lo, const float* hi, const float* mn, const float* mx)
{
    float max_span = __fsub_rn(hi[0], lo[0]);
    for (int a = 1; a < 3; ++a) {
        const float s = __fsub_rn(hi[a], lo[a]);
        if (s > max_span) max_span = s;
    }
    const float thresh = __fmul_rn(__fsub_rn(1.0f, 0.00001f), max_span);
    int cutfeat = 0;
    float max_spread = -1.f, lo_c = lo[0], hi_c = hi[0], mn_c = mn[0], mx_c = mx[0];
    {
        const bool in0 = __fsub_rn(hi[0], lo[0]) > thresh, in1 = __fsub_rn(hi[1], lo[1]) > thresh, in2 = __fsub_rn(hi[2], lo[2]) > thresh;
        const float sp0 = __fsub_rn(mx[0], mn[0]), sp1 = __fsub_rn(mx[1], mn[1]), sp2 = __fsub_rn(mx[2], mn[2]);
        if (in0 && sp0 > max_spread) { cutfeat = 0; max_spread = sp0; }
        if (in1 && sp1 > max_spread) { cutfeat = 1; max_spread = sp1; lo_c = lo[1]; hi_c = hi[1]; mn_c = mn[1]; mx_c = mx[1]; }
        if (in2 && sp2 > max_spread) { cutfeat = 2; max_spread = sp2; lo_c = lo[2]; hi_c = hi[2]; mn_c = mn[2]; mx_c = mx[2]; }
    }
    const float split = __fmul_rn(__fadd_rn(lo_c, hi_c), 0.5f);
    SplitChoice c;
    c.ax = cutfeat;
    c.cut = split < mn_c ? mn_c : (split > mx_c ? mx_c : split);
    return c;
}

// Everything the parent must record once the partition is known.
__device__ __forceinline__ void emit_inner(const BuildQueues& Q, const BuildTree& t, const BuildTask& k, int ax, float cut, int lim1, int lim2,
                                           float maxlt, float mingt, BuildTask* kids /* [2] out */, int* node_id)
{
    const int count = k.r - k.l, half = count / 2;
    const int idx = lim1 > half ? lim1 : (lim2 < half ? lim2 : half);
    const int m = k.l + idx;
    const int id = 2 * m - 1;
    // left = first idx records of [ <cut | ==cut | >cut ]
    const float divlow = idx > lim1 ? cut : maxlt;
    const float divhigh = idx < lim2 ? cut : mingt;
    t.nodes[id] = make_int4((int)((unsigned)ax << 30), 0, __float_as_int(divlow), __float_as_int(divhigh));
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        BuildTask c = k;
        c.parent = id;
        c.side = s;
        c.level = k.level + 1;
        // (compile-time indices only: a run-time index would push the task into scratch memory)
        if (s == 0) {
            c.r = m;
#pragma unroll
            for (int a = 0; a < 3; ++a) c.hi[a] = a == ax ? cut : c.hi[a];
        } else {
            c.l = m;
#pragma unroll
            for (int a = 0; a < 3; ++a) c.lo[a] = a == ax ? cut : c.lo[a];
        }
        kids[s] = c;
    }
    *node_id = id;
}

__device__ __forceinline__ void link_to_parent(const BuildTree& t, const BuildTask& k, int id)
{
    if (k.parent < 0)
        t.meta->root = id;
    else if (k.side == 0)
        atomicOr(&t.nodes[k.parent].x, id);  // low 30 bits were written as 0 by the parent, in an earlier kernel / earlier by this wave
    else
        t.nodes[k.parent].y = id;
}

// ---- level kernel: one workgroup per big node ------------------------------------------------------------------
// Every pass walks the node in slabs of T*E records; thread t owns E CONSECUTIVE records of a slab (a wave covers
// 8 KiB of contiguous memory, all eight 16-byte loads of a lane are independent and in flight together), so the
// order-preserving ranks of the Hoare sweeps need one block scan per slab.
constexpr int kE = 8;

template <int T>
struct BlockRed {
    float mn[T / 64][3], mx[T / 64][3];
    int ia[T / 64], ib[T / 64];
    float fa[T / 64], fb[T / 64];
    int scan[2][T / 64];
};

// Stragglers: a node that is still above kMid points after the chunked levels (lopsided bounding-box-midpoint splits: very
// unbalanced clouds, dense clusters).  One workgroup takes the node and finishes EVERYTHING above kMid points below it depth
// first: after a split it keeps its SMALLER big child and parks the larger one on a stack in LDS; children of <= kMid points go to the
// mid / small queues of the kernels that follow.  One launch therefore completes the top of every tree however unbalanced the
// cloud is -- there is no "unfinished build" state for the host to detect and repair, which is what makes ps_pyramid_build
// safe to run without host synchronisation (deferred checks): the searches and the network that are enqueued right behind
// it always see complete trees and write every index.  Continuing with the smaller child halves the node in hand at
// every push, so at most log2(n / kMid) <= 12 entries are ever parked for a 2^25-point tree; 32 leave room to spare (overflow
// still sets flags[1]).
constexpr int kStragglerStack = 32;

__device__ __forceinline__ void push_small_task(const BuildQueues& Q, const BuildTask& t)
{
    if (t.r - t.l > kSmall) {
        const int slot = atomicAdd(Q.small_cnt + 1, 1);
        if (slot < Q.q_cap)
            Q.mid_q[slot] = t;
        else
            Q.flags[1] = 1;
    } else {
        const int slot = atomicAdd(Q.small_cnt, 1);
        if (slot < Q.small_cap)
            Q.small_q[slot] = t;
        else
            Q.flags[1] = 1;
    }
}

__global__ __launch_bounds__(kBigThreads) void build_level_kernel(const BuildTree* __restrict__ trees, BuildQueues Q, int level)
{
    constexpr int T = kBigThreads, W = T / 64, E = kE;
    __shared__ BlockRed<T> S;
    __shared__ BuildTask s_stack[kStragglerStack];
    __shared__ BuildTask s_cur;
    __shared__ int s_sp;
    const int n_tasks = min(Q.level_cnt[level], Q.q_cap);
    if ((int)blockIdx.x >= n_tasks) return;
    if (threadIdx.x == 0) {
        s_cur = Q.q[level & 1][blockIdx.x];
        s_sp = 0;
    }
    __syncthreads();
  for (;;) {
    BuildTask k;
    k.tree = s_cur.tree; k.l = s_cur.l; k.r = s_cur.r; k.parent = s_cur.parent; k.side = s_cur.side; k.level = s_cur.level;
#pragma unroll
    for (int c = 0; c < 3; ++c) { k.lo[c] = s_cur.lo[c]; k.hi[c] = s_cur.hi[c]; }
    const BuildTree t = trees[k.tree];
    const GArr<float4> a{t.pts + k.l};
    const int count = k.r - k.l;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;

    // ---- pass 1: min / max of the three axes ----
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int base = tid; base < count; base += T * E) {  // coalesced: record base + e*T
        float4 p[E];
#pragma unroll
        for (int e = 0; e < E; ++e) p[e] = a[min(base + e * T, count - 1)];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            mn[0] = fminf(mn[0], p[e].x); mx[0] = fmaxf(mx[0], p[e].x);
            mn[1] = fminf(mn[1], p[e].y); mx[1] = fmaxf(mx[1], p[e].y);
            mn[2] = fminf(mn[2], p[e].z); mx[2] = fmaxf(mx[2], p[e].z);
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        {
            mn[c] = wave_min(mn[c]);
            mx[c] = wave_max(mx[c]);
        }
        if (lane == 0) { S.mn[wave][c] = mn[c]; S.mx[wave][c] = mx[c]; }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float lo = S.mn[0][c], hi = S.mx[0][c];
        for (int w = 1; w < W; ++w) { lo = fminf(lo, S.mn[w][c]); hi = fmaxf(hi, S.mx[w][c]); }
        mn[c] = lo; mx[c] = hi;
    }
    const SplitChoice sc = choose_split_scalar(k.lo, k.hi, mn, mx);
    const int ax = sc.ax;
    const float cut = sc.cut;

    // ---- pass 2: lim1 = #(< cut), lim2 = #(<= cut), max{v < cut}, min{v > cut} ----
    int lt = 0, le = 0;
    float maxlt = -INFINITY, mingt = INFINITY;
    for (int base = tid; base < count; base += T * E) {
        float v[E];
#pragma unroll
        for (int e = 0; e < E; ++e) v[e] = comp(a[min(base + e * T, count - 1)], ax);
#pragma unroll
        for (int e = 0; e < E; ++e)
            if (base + e * T < count) {
                lt += v[e] < cut;
                le += v[e] <= cut;
                if (v[e] < cut) maxlt = fmaxf(maxlt, v[e]);
                if (v[e] > cut) mingt = fminf(mingt, v[e]);
            }
    }
    {
        lt = wave_sum(lt);
        le = wave_sum(le);
        maxlt = wave_max(maxlt);
        mingt = wave_min(mingt);
    }
    if (lane == 0) { S.ia[wave] = lt; S.ib[wave] = le; S.fa[wave] = maxlt; S.fb[wave] = mingt; }
    __syncthreads();
    lt = 0; le = 0; maxlt = -INFINITY; mingt = INFINITY;
    for (int w = 0; w < W; ++w) {
        lt += S.ia[w]; le += S.ib[w];
        maxlt = fmaxf(maxlt, S.fa[w]); mingt = fminf(mingt, S.fb[w]);
    }
    const int lim1 = lt, lim2 = le;

    // ---- the two Hoare sweeps (planeSplit, nanoflann.hpp:1016-1043) in closed form ----
    // sweep 0: [0,count) by (v < cut); sweep 1: [lim1,count) by (v <= cut).  In sweep s the elements left of the
    // boundary that fail the predicate ("misplaced left", ascending) pair with the elements right of it that satisfy
    // it ("misplaced right", descending): i-th with i-th.
    for (int sweep = 0; sweep < 2; ++sweep) {
        const int from = sweep == 0 ? 0 : lim1, bound = sweep == 0 ? lim1 : lim2;
        int offL = 0, offR = 0;
        int slab = 0;
        for (int s0 = from; s0 < count; s0 += T * E, ++slab) {
            // wave w owns records [s0 + w*64*E, +64*E); lane reads record  wbase + e*64 + lane  (coalesced), so the
            // position order inside the wave is e-major and ranks come from E ballots
            const int wbase = s0 + wave * (64 * E);
            unsigned long long bL[E], bR[E];
            {
                float v[E];
#pragma unroll
                for (int e = 0; e < E; ++e) v[e] = comp(a[min(wbase + e * 64 + lane, count - 1)], ax);
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int p = wbase + e * 64 + lane;
                    bool isL = false, isR = false;
                    if (p < count) {
                        const bool keep_left = sweep == 0 ? (v[e] < cut) : (v[e] <= cut);
                        isL = p < bound && !keep_left;
                        isR = p >= bound && keep_left;
                    }
                    bL[e] = __ballot(isL);
                    bR[e] = __ballot(isR);
                }
            }
            int wL = 0, wR = 0;
#pragma unroll
            for (int e = 0; e < E; ++e) { wL += __popcll(bL[e]); wR += __popcll(bR[e]); }
            if (lane == 0) S.scan[slab & 1][wave] = wL | (wR << 16);
            __syncthreads();
            int pre = 0, tot = 0;
            for (int w = 0; w < W; ++w) {
                const int cw = S.scan[slab & 1][w];
                if (w < wave) pre += cw;
                tot += cw;
            }
            int rL = offL + (pre & 0xffff), rR = offR + (pre >> 16);
            const unsigned long long lt_mask = (1ull << lane) - 1ull;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int p = wbase + e * 64 + lane;
                if ((bL[e] >> lane) & 1ull) gstore(t.posL + k.l + rL + __popcll(bL[e] & lt_mask), p);
                if ((bR[e] >> lane) & 1ull) gstore(t.posR + k.l + rR + __popcll(bR[e] & lt_mask), p);
                rL += __popcll(bL[e]);
                rR += __popcll(bR[e]);
            }
            offL += tot & 0xffff;
            offR += tot >> 16;
        }
        __syncthreads();  // posL / posR written by this workgroup are read below by other waves of it
        const int m = offL;  // == offR
        for (int i = tid; i < m; i += T) {
            const int pl = gload(t.posL + k.l + i), pr = gload(t.posR + k.l + m - 1 - i);
            const float4 x = a[pl], y = a[pr];
            a.set(pl, y);
            a.set(pr, x);
        }
        __syncthreads();
    }

    if (tid == 0) {
        BuildTask kids[2];
        int id;
        emit_inner(Q, t, k, ax, cut, lim1, lim2, maxlt, mingt, kids, &id);
        link_to_parent(t, k, id);
        int sp = s_sp;
        bool have = false;
        const int n0 = kids[0].r - kids[0].l, n1 = kids[1].r - kids[1].l;
        if (n0 > kMid && n1 > kMid) {
            // both children are still big: carry on with the SMALLER one and park the larger -- every parked node is then at least
            // as large as everything handled before it is popped, so the stack never holds more than log2(n / kMid) entries
            const int small = n0 <= n1 ? 0 : 1;
            s_cur = kids[small];
            have = true;
            if (sp < kStragglerStack)
                s_stack[sp++] = kids[small ^ 1];
            else
                Q.flags[1] = 1;
        } else {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                if (kids[s].r - kids[s].l <= kMid)
                    push_small_task(Q, kids[s]);
                else {
                    s_cur = kids[s];  // carry on with the one big child
                    have = true;
                }
            }
        }
        if (!have) {
            if (sp > 0)
                s_cur = s_stack[--sp];
            else
                sp = -1;  // this node is finished down to kMid
        }
        s_sp = sp;
    }
    __syncthreads();
    if (s_sp < 0) break;
  }
}

// ---- flat subtree kernel: one workgroup per node of <= kSmall points, ONE THREAD PER POINT POSITION --------------------------------
// A wave-per-node kernel ran a subtree's ~31 splits one after the other (80-90 us for the ~1 400 subtrees of a 180 000-point
// pyramid: a chain of LDS round trips at one or two waves per SIMD).  Here every round advances ALL live segments of the node at
// once: thread i owns position i, carries its segment's descriptor (range, incoming box, parent link) in registers, and the
// per-segment facts (tight extents, counts against the cut, misplaced counts) are LDS accumulators indexed by the segment's first
// position.  The two Hoare sweeps use the same closed form as everywhere else (i-th misplaced-left <-> i-th misplaced-right from the
// right); the order-preserving ranks are one block-wide exclusive scan of the two flags (ballot prefix inside a wave, wave totals
// through LDS) minus the scan value at the segment's first position.  A round is ~12 workgroup barriers whatever the number of
// segments, and a subtree is finished in (height + 1) rounds: 6-7 for the balanced trees of real clouds.
struct FlatAcc {
    unsigned mn[3], mx[3];     // ordered-uint extents
    int lt, le;
    unsigned maxlt, mingt;     // ordered-uint
    int base[2], endL[2];      // per sweep: packed exclusive scan (L | R << 16) at the segment start; inclusive L rank at its last position
};

__global__ __launch_bounds__(kSmall) void build_flat_kernel(const BuildTree* __restrict__ trees, BuildQueues Q)
{
    constexpr int T = kSmall, W = T / 64;
    __shared__ float s_p[4][T];
    __shared__ short s_posR[T];
    __shared__ FlatAcc s_accs[T / 8];  // one per live segment, indexed by start / 8 (live segments are longer than kLeafMax >= 8: unique)
    __shared__ int s_wtot[2][W];
    // (node records go straight to global memory: assembling them in LDS and writing them out once at the end was measured
    //  SLOWER, 53 vs 44 us -- the kernel is bound by the instruction count of its rounds, not by the stores' latency)
    __shared__ int s_live[2];  // "some segment is still live after this round", by round parity (a flag is reset a full round after its last read)
    __shared__ int s_mis[2];   // misplaced records per sweep of the current round
    __shared__ int s_depth;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const int n_tasks = min(Q.small_cnt[0], Q.small_cap);

    for (int ti = blockIdx.x; ti < n_tasks; ti += gridDim.x) {
        const BuildTask k = Q.small_q[ti];
        const BuildTree t = trees[k.tree];
        const int total = k.r - k.l;
        if (tid < total) {
            const float4 p = gload(t.pts + k.l + tid);
            s_p[0][tid] = p.x; s_p[1][tid] = p.y; s_p[2][tid] = p.z; s_p[3][tid] = p.w;
        }
        // this thread's segment (registers): [s, e) relative to the node, incoming box, link to the parent node
        int s = 0, e = total, parent = k.parent, side = k.side, level = k.level;
        float lo[3], hi[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) { lo[c] = k.lo[c]; hi[c] = k.hi[c]; }
        bool live = tid < total;
        int round = 0;
        if (tid == 0) s_depth = 0;
        if (total <= kLeafMax) {  // the node itself is a leaf
            if (tid == 0) {
                t.nodes[2 * k.l] = make_int4(k.l, k.l + total, 0, 0);
                const int ref = leaf_ref(k.l, total);
                if (parent < 0) t.meta->root = ref;
                else if (side == 0) atomicOr(&t.nodes[parent].x, ref);
                else t.nodes[parent].y = ref;
                atomicMax(&t.meta->depth, level);
            }
            __syncthreads();
            continue;
        }
        for (;;) {
            // ---- 0: leaders reset their segment's accumulators ----
            if (live && tid == s) {
                FlatAcc& a = s_accs[s >> 3];
#pragma unroll
                for (int c = 0; c < 3; ++c) { a.mn[c] = 0xffffffffu; a.mx[c] = 0u; }
                a.lt = 0; a.le = 0; a.maxlt = 0u; a.mingt = 0xffffffffu;
            }
            if (tid == 0) { s_live[round & 1] = 0; s_mis[0] = 0; s_mis[1] = 0; }
            __syncthreads();
            // ---- 1: tight extents ----
            // (LDS atomics of a wave to ONE address are served one lane after the other: while a wave lies inside a single segment --
            //  the first rounds, where the segments are long -- it reduces in registers and sends one atomic per quantity)
            const bool uni = __all(live) && __all(s == __builtin_amdgcn_readfirstlane(s));
            float px = 0.f, py = 0.f, pz = 0.f;
            if (live) { px = s_p[0][tid]; py = s_p[1][tid]; pz = s_p[2][tid]; }
            if (uni) {
                const float mnx = wave_min(px), mxx = wave_max(px), mny = wave_min(py), mxy = wave_max(py), mnz = wave_min(pz), mxz = wave_max(pz);
                if (lane < 3) {
                    atomicMin(&s_accs[s >> 3].mn[lane], f2ord(lane == 0 ? mnx : (lane == 1 ? mny : mnz)));
                    atomicMax(&s_accs[s >> 3].mx[lane], f2ord(lane == 0 ? mxx : (lane == 1 ? mxy : mxz)));
                }
            } else if (live) {
                FlatAcc& a = s_accs[s >> 3];
                atomicMin(&a.mn[0], f2ord(px)); atomicMax(&a.mx[0], f2ord(px));
                atomicMin(&a.mn[1], f2ord(py)); atomicMax(&a.mx[1], f2ord(py));
                atomicMin(&a.mn[2], f2ord(pz)); atomicMax(&a.mx[2], f2ord(pz));
            }
            __syncthreads();
            // ---- 2: split choice (every thread of the segment, identical), counts against the cut ----
            int ax = 0;
            float cut = 0.f, v = 0.f;
            if (live) {
                const FlatAcc& a = s_accs[s >> 3];
                float mn[3], mx[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) { mn[c] = ord2f(a.mn[c]); mx[c] = ord2f(a.mx[c]); }
                const SplitChoice sc = choose_split(lo, hi, mn, mx);
                ax = sc.ax;
                cut = sc.cut;
                v = ax == 0 ? px : (ax == 1 ? py : pz);
            }
            if (uni) {
                const int lt = (int)__popcll(__ballot(v < cut)), le = (int)__popcll(__ballot(v <= cut));
                const float ml = wave_max(v < cut ? v : -INFINITY), mg = wave_min(v > cut ? v : INFINITY);
                if (lane == 0) {
                    FlatAcc& w = s_accs[s >> 3];
                    if (lt) { atomicAdd(&w.lt, lt); atomicMax(&w.maxlt, f2ord(ml)); }
                    if (le) atomicAdd(&w.le, le);
                    if (mg < INFINITY) atomicMin(&w.mingt, f2ord(mg));
                }
            } else if (live) {
                FlatAcc& w = s_accs[s >> 3];
                if (v < cut) { atomicAdd(&w.lt, 1); atomicMax(&w.maxlt, f2ord(v)); }
                if (v <= cut) atomicAdd(&w.le, 1);
                if (v > cut) atomicMin(&w.mingt, f2ord(v));
            }
            __syncthreads();
            int lim1 = 0, lim2 = 0;
            float maxlt = 0.f, mingt = 0.f;
            if (live) {
                const FlatAcc& a = s_accs[s >> 3];
                lim1 = a.lt; lim2 = a.le; maxlt = ord2f(a.maxlt); mingt = ord2f(a.mingt);
            }
            const int q = tid - s;
            // ---- the two Hoare sweeps (planeSplit, nanoflann.hpp:1016-1043) in closed form ----
#pragma unroll
            for (int sweep = 0; sweep < 2; ++sweep) {
                const int from = sweep == 0 ? 0 : lim1, bound = sweep == 0 ? lim1 : lim2;
                bool isL = false, isR = false;
                if (live && q >= from) {
                    const bool keep_left = sweep == 0 ? (v < cut) : (v <= cut);
                    isL = q < bound && !keep_left;
                    isR = q >= bound && keep_left;
                }
                const unsigned long long bL = __ballot(isL), bR = __ballot(isR);
                if (lane == 0) {
                    s_wtot[sweep][wave] = __popcll(bL) | (__popcll(bR) << 16);
                    if (bL) atomicAdd(&s_mis[sweep], (int)__popcll(bL));
                }
                __syncthreads();
                if (s_mis[sweep] == 0) continue;  // (uniform) nothing misplaced anywhere in this sweep
                int pre = 0;
#pragma unroll
                for (int w = 0; w < W; ++w) pre += w < wave ? s_wtot[sweep][w] : 0;
                const int excl = pre + (__popcll(bL & lt_mask) | (__popcll(bR & lt_mask) << 16));  // packed exclusive ranks (L | R << 16)
                if (live && tid == s) s_accs[s >> 3].base[sweep] = excl;
                if (live && tid == e - 1) s_accs[s >> 3].endL[sweep] = (excl & 0xffff) + (isL ? 1 : 0);
                __syncthreads();
                int m = 0, baseL = 0;
                if (live) {
                    const FlatAcc& a = s_accs[s >> 3];
                    baseL = a.base[sweep] & 0xffff;
                    m = a.endL[sweep] - baseL;
                    if (isR) s_posR[s + ((excl >> 16) - (a.base[sweep] >> 16))] = (short)tid;
                }
                __syncthreads();
                if (isL) {  // i-th misplaced-left record (ascending) swaps with the i-th misplaced-right record counted from the right
                    const int pr = s_posR[s + m - 1 - ((excl & 0xffff) - baseL)];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float x = s_p[c][tid], y = s_p[c][pr];
                        s_p[c][pr] = x;
                        s_p[c][tid] = y;
                    }
                }
                __syncthreads();
                if (live) v = s_p[ax][tid];  // (the record at this position may have changed)
            }
            // ---- emit the node, descend into the child this position now belongs to ----
            if (live) {
                const int count = e - s, half = count / 2;
                const int idx = lim1 > half ? lim1 : (lim2 < half ? lim2 : half);
                const int id = 2 * (k.l + s + idx) - 1;
                if (tid == s) {
                    const float divlow = idx > lim1 ? cut : maxlt;
                    const float divhigh = idx < lim2 ? cut : mingt;
                    t.nodes[id] = make_int4((int)((unsigned)ax << 30), 0, __float_as_int(divlow), __float_as_int(divhigh));
                    if (parent < 0) t.meta->root = id;
                    else if (side == 0) atomicOr(&t.nodes[parent].x, id);
                    else t.nodes[parent].y = id;
                }
                if (q < idx) {
                    e = s + idx; side = 0;
#pragma unroll
                    for (int c = 0; c < 3; ++c) hi[c] = c == ax ? cut : hi[c];
                } else {
                    s = s + idx; side = 1;
#pragma unroll
                    for (int c = 0; c < 3; ++c) lo[c] = c == ax ? cut : lo[c];
                }
                parent = id;
                level += 1;
            }
            __syncthreads();  // the parents' records are in memory before any child links to them
            if (live && e - s <= kLeafMax) {
                if (tid == s) {
                    t.nodes[2 * (k.l + s)] = make_int4(k.l + s, k.l + e, 0, 0);
                    const int ref = leaf_ref(k.l + s, e - s);
                    if (side == 0) atomicOr(&t.nodes[parent].x, ref);
                    else t.nodes[parent].y = ref;
                    atomicMax(&s_depth, level);
                }
                live = false;
            }
            if (live && tid == s) s_live[round & 1] = 1;
            __syncthreads();
            if (s_live[round & 1] == 0) break;
            ++round;
        }
        if (tid < total) gstore(t.pts + k.l + tid, make_float4(s_p[0][tid], s_p[1][tid], s_p[2][tid], s_p[3][tid]));
        if (tid == 0) atomicMax(&t.meta->depth, s_depth);
        __syncthreads();
    }
}

// ---- mid kernel: one workgroup splits a node of <= kMid points down to <= kSmall-point pieces, entirely in LDS ------
// The node's records are loaded once (128 KiB for 8 192 points) and go back to HBM once at the end; in between the
// workgroup advances LEVEL BY LEVEL over all of the node's live segments (pieces still above kSmall points, at most
// kMid / kSmall of them) at once, with __syncthreads() as the only synchronisation -- instead of one kernel launch and
// several L2 round trips per pass per tree level.  Work unit: a SLAB = up to 512 consecutive positions of ONE segment,
// handled by one wave (lane l owns positions  start + e*64 + l,  e < 8: position order is (e, l)-major, which is what
// the order-preserving ranks of the closed-form Hoare sweeps need).  Same arithmetic as build_level_kernel.
constexpr int kSlab = 512;
constexpr int kMidSegs = kMid / kSmall;            // live segments are disjoint and longer than kSmall
constexpr int kMidSlabs = kMid / kSlab + kMidSegs;  // sum of ceil(len / kSlab)

struct MidSeg {
    int start, end, parent, side, level, slab0;
    float lo[3], hi[3];
    unsigned mn[3], mx[3];  // ordered-uint accumulators (LDS atomics)
    int lt, le;
    unsigned maxlt, mingt;
    int ax;
    float cut;
    int mis[2];  // misplaced records per sweep
};
struct MidSlab {
    int seg, start, end;
    int cnt[2];  // per sweep: misplaced-left | misplaced-right << 16
};

__device__ __forceinline__ void mid_seg_init(MidSeg& g, int start, int end, int parent, int side, int level, int slab0)
{
    g.start = start; g.end = end; g.parent = parent; g.side = side; g.level = level; g.slab0 = slab0;
    for (int a = 0; a < 3; ++a) { g.mn[a] = 0xffffffffu; g.mx[a] = 0u; }
    g.lt = 0; g.le = 0; g.maxlt = 0u; g.mingt = 0xffffffffu;
    g.mis[0] = 0; g.mis[1] = 0;
}

// The node's records live in LDS as FOUR planes (x | y | z | index bits), kMid floats each: a pass that only needs the split-axis
// value reads 4 bytes per record with a conflict-free ds_read_b32 instead of the whole 16-byte record (the passes are LDS-bandwidth
// bound: eight of them per tree level, each over the whole node).
struct MidPlanes {
    float* c[4];
    __device__ __forceinline__ const float* axis(int ax) const { return ax == 0 ? c[0] : (ax == 1 ? c[1] : c[2]); }
};

// misplaced flags of one slab for sweep s as 8 ballots (positions q relative to the segment start g0)
__device__ __forceinline__ void mid_flags(const float* V /* split-axis plane + g0 */, int q0, int q1, int lane, float cut, int sweep, int lim1, int lim2,
                                          unsigned long long (&bL)[8], unsigned long long (&bR)[8])
{
    const int from = sweep == 0 ? 0 : lim1, bound = sweep == 0 ? lim1 : lim2;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int q = q0 + e * 64 + lane;
        bool isL = false, isR = false;
        if (q < q1 && q >= from) {
            const float v = V[q];
            const bool keep_left = sweep == 0 ? (v < cut) : (v <= cut);
            isL = q < bound && !keep_left;
            isR = q >= bound && keep_left;
        }
        bL[e] = __ballot(isL);
        bR[e] = __ballot(isR);
    }
}

__global__ __launch_bounds__(kMidThreads) void build_mid_kernel(const BuildTree* __restrict__ trees, BuildQueues Q)
{
    constexpr int T = kMidThreads, W = T / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char mid_smem[];
    MidPlanes P;
#pragma unroll
    for (int c = 0; c < 4; ++c) P.c[c] = reinterpret_cast<float*>(mid_smem) + c * kMid;                    // 4 x [kMid]
    short* posR = reinterpret_cast<short*>(mid_smem + sizeof(float4) * kMid);                             // [kMid]
    __shared__ MidSeg segs[2][kMidSegs];
    __shared__ MidSlab slabs[2][kMidSlabs];
    __shared__ unsigned long long s_ball[kMidSlabs][16];  // the slab's 8 + 8 flag ballots of the current sweep (written in C, reused in D and E)
    __shared__ int nseg[2], nslab[2], s_mis[2];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const int n_tasks = min(Q.small_cnt[1], Q.q_cap);

    for (int ti = blockIdx.x; ti < n_tasks; ti += gridDim.x) {
        const BuildTask& k = Q.mid_q[ti];
        const int k_l = k.l, k_tree = k.tree;
        const BuildTree& t = trees[k_tree];
        float4* const g_pts = t.pts + k_l;
        const int total = k.r - k_l;
        for (int i = tid; i < total; i += T) {
            const float4 p = gload(g_pts + i);
            P.c[0][i] = p.x; P.c[1][i] = p.y; P.c[2][i] = p.z; P.c[3][i] = p.w;
        }
        if (tid == 0) {
            const int ns = (total + kSlab - 1) / kSlab;
            mid_seg_init(segs[0][0], 0, total, k.parent, k.side, k.level, 0);
            for (int c = 0; c < 3; ++c) { segs[0][0].lo[c] = k.lo[c]; segs[0][0].hi[c] = k.hi[c]; }
            for (int j = 0; j < ns; ++j) {
                MidSlab& sl = slabs[0][j];
                sl.seg = 0; sl.start = j * kSlab; sl.end = min(total, (j + 1) * kSlab);
            }
            nseg[0] = 1; nslab[0] = ns;
            s_mis[0] = 0; s_mis[1] = 0;
        }
        __syncthreads();
        int cur = 0;
        while (nseg[cur] > 0) {
            MidSeg* G = segs[cur];
            const MidSlab* SL = slabs[cur];
            const int n_sl = nslab[cur], n_sg = nseg[cur];
            if (tid == 0) { nseg[cur ^ 1] = 0; nslab[cur ^ 1] = 0; }
            // ---- A: tight extents of every segment ----
            for (int si = wave; si < n_sl; si += W) {
                const int gi = SL[si].seg, s0 = SL[si].start, s1 = SL[si].end;
                float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int i = s0 + e * 64 + lane;
                    if (i < s1) {
                        const float x = P.c[0][i], y = P.c[1][i], z = P.c[2][i];
                        mn[0] = fminf(mn[0], x); mx[0] = fmaxf(mx[0], x);
                        mn[1] = fminf(mn[1], y); mx[1] = fmaxf(mx[1], y);
                        mn[2] = fminf(mn[2], z); mx[2] = fmaxf(mx[2], z);
                    }
                }
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    {
                        mn[c] = wave_min(mn[c]);
                        mx[c] = wave_max(mx[c]);
                    }
                if (lane < 3) {
                    atomicMin(&G[gi].mn[lane], f2ord(lane == 0 ? mn[0] : (lane == 1 ? mn[1] : mn[2])));
                    atomicMax(&G[gi].mx[lane], f2ord(lane == 0 ? mx[0] : (lane == 1 ? mx[1] : mx[2])));
                }
            }
            __syncthreads();
            // ---- B: split choice (recomputed by every wave of the segment, identical), counts against the cut ----
            for (int si = wave; si < n_sl; si += W) {
                const int gi = SL[si].seg, s0 = SL[si].start, s1 = SL[si].end;
                float mn[3], mx[3], blo[3], bhi[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) { mn[c] = ord2f(G[gi].mn[c]); mx[c] = ord2f(G[gi].mx[c]); blo[c] = G[gi].lo[c]; bhi[c] = G[gi].hi[c]; }
                const SplitChoice sc = choose_split(blo, bhi, mn, mx);
                const int ax = sc.ax;
                const float cut = sc.cut;
                const float* V = P.axis(ax);
                int lt = 0, le = 0;
                float maxlt = -INFINITY, mingt = INFINITY;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int i = s0 + e * 64 + lane;
                    if (i < s1) {
                        const float v = V[i];
                        lt += v < cut;
                        le += v <= cut;
                        if (v < cut) maxlt = fmaxf(maxlt, v);
                        if (v > cut) mingt = fminf(mingt, v);
                    }
                }
                {
                    lt = wave_sum(lt);
                    le = wave_sum(le);
                    maxlt = wave_max(maxlt);
                    mingt = wave_min(mingt);
                }
                if (lane == 0) {
                    atomicAdd(&G[gi].lt, lt);
                    atomicAdd(&G[gi].le, le);
                    atomicMax(&G[gi].maxlt, f2ord(maxlt));
                    atomicMin(&G[gi].mingt, f2ord(mingt));
                    G[gi].ax = ax;    // (same value from every slab of the segment)
                    G[gi].cut = cut;
                }
            }
            __syncthreads();
            // ---- the two Hoare sweeps (planeSplit, nanoflann.hpp:1016-1043) in closed form ----
            for (int sweep = 0; sweep < 2; ++sweep) {
                // C: misplaced flags per slab (kept in LDS for D and E) and their counts
                for (int si = wave; si < n_sl; si += W) {
                    const int gi = SL[si].seg;
                    const int g0 = G[gi].start;
                    unsigned long long bL[8], bR[8];
                    mid_flags(P.axis(G[gi].ax) + g0, SL[si].start - g0, SL[si].end - g0, lane, G[gi].cut, sweep, G[gi].lt, G[gi].le, bL, bR);
                    int wL = 0, wR = 0;
#pragma unroll
                    for (int e = 0; e < 8; ++e) { wL += __popcll(bL[e]); wR += __popcll(bR[e]); }
                    if (lane == 0) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) { s_ball[si][e] = bL[e]; s_ball[si][8 + e] = bR[e]; }
                        slabs[cur][si].cnt[sweep] = wL | (wR << 16);
                        if (wL) {
                            atomicAdd(&G[gi].mis[sweep], wL);
                            atomicAdd(&s_mis[sweep], wL);
                        }
                    }
                }
                __syncthreads();
                if (s_mis[sweep] == 0) continue;  // (uniform: nothing misplaced anywhere in this sweep -- the usual case for sweep 1)
                // D: ranks -> positions of the misplaced-right records, in ascending order
                for (int si = wave; si < n_sl; si += W) {
                    const int gi = SL[si].seg;
                    if (G[gi].mis[sweep] == 0) continue;
                    const int g0 = G[gi].start, sl0 = G[gi].slab0;
                    const int mine = sl0 + lane < si ? SL[sl0 + lane].cnt[sweep] : 0;  // slabs of a segment are consecutive, in order
                    int pre = 0;
                    for (int j = 0; j < si - sl0; ++j) pre += __builtin_amdgcn_readlane(mine, j) >> 16;
                    int rR = pre;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const unsigned long long b = s_ball[si][8 + e];
                        if ((b >> lane) & 1ull) posR[g0 + rR + __popcll(b & lt_mask)] = (short)(SL[si].start - g0 + e * 64 + lane);
                        rR += __popcll(b);
                    }
                }
                __syncthreads();
                // E: i-th misplaced-left record (ascending) swaps with the i-th misplaced-right record counted from the right
                for (int si = wave; si < n_sl; si += W) {
                    const int gi = SL[si].seg;
                    const int m = G[gi].mis[sweep];
                    if (m == 0) continue;
                    const int g0 = G[gi].start, sl0 = G[gi].slab0;
                    const int mine = sl0 + lane < si ? SL[sl0 + lane].cnt[sweep] : 0;
                    int rL = 0;
                    for (int j = 0; j < si - sl0; ++j) rL += __builtin_amdgcn_readlane(mine, j) & 0xffff;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const unsigned long long b = s_ball[si][e];
                        if ((b >> lane) & 1ull) {
                            const int q = g0 + SL[si].start - g0 + e * 64 + lane;
                            const int pr = g0 + posR[g0 + m - 1 - (rL + __popcll(b & lt_mask))];
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                const float x = P.c[c][q], y = P.c[c][pr];
                                P.c[c][pr] = x;
                                P.c[c][q] = y;
                            }
                        }
                        rL += __popcll(b);
                    }
                }
                __syncthreads();
            }
            // ---- F: record the nodes, route the children (one thread per segment) ----
            if (tid == 0) { s_mis[0] = 0; s_mis[1] = 0; }
            if (tid < n_sg) {
                const MidSeg& g = G[tid];
                const int count = g.end - g.start, lim1 = g.lt, lim2 = g.le, ax = g.ax;
                const float cut = g.cut;
                const int half = count / 2;
                const int idx = lim1 > half ? lim1 : (lim2 < half ? lim2 : half);
                const int id = 2 * (k_l + g.start + idx) - 1;
                const float divlow = idx > lim1 ? cut : ord2f(g.maxlt);
                const float divhigh = idx < lim2 ? cut : ord2f(g.mingt);
                t.nodes[id] = make_int4((int)((unsigned)ax << 30), 0, __float_as_int(divlow), __float_as_int(divhigh));
                if (g.parent < 0) t.meta->root = id;
                else if (g.side == 0) atomicOr(&t.nodes[g.parent].x, id);
                else t.nodes[g.parent].y = id;
#pragma unroll
                for (int side = 0; side < 2; ++side) {
                    const int cl = side == 0 ? g.start : g.start + idx, cr = side == 0 ? g.start + idx : g.end;
                    float clo[3], chi[3];
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        clo[c] = (side == 1 && c == ax) ? cut : g.lo[c];
                        chi[c] = (side == 0 && c == ax) ? cut : g.hi[c];
                    }
                    if (cr - cl > kSmall) {
                        const int ns = (cr - cl + kSlab - 1) / kSlab;
                        const int slot = atomicAdd(&nseg[cur ^ 1], 1), sl0 = atomicAdd(&nslab[cur ^ 1], ns);
                        MidSeg& ch = segs[cur ^ 1][slot];
                        mid_seg_init(ch, cl, cr, id, side, g.level + 1, sl0);
#pragma unroll
                        for (int c = 0; c < 3; ++c) { ch.lo[c] = clo[c]; ch.hi[c] = chi[c]; }
                        for (int j = 0; j < ns; ++j) {
                            MidSlab& sl = slabs[cur ^ 1][sl0 + j];
                            sl.seg = slot; sl.start = cl + j * kSlab; sl.end = min(cr, cl + (j + 1) * kSlab);
                        }
                    } else {
                        const int slot = atomicAdd(Q.small_cnt, 1);
                        if (slot < Q.small_cap) {
                            BuildTask& b = Q.small_q[slot];
                            b.tree = k_tree; b.l = k_l + cl; b.r = k_l + cr; b.parent = id; b.side = side; b.level = g.level + 1;
#pragma unroll
                            for (int c = 0; c < 3; ++c) { b.lo[c] = clo[c]; b.hi[c] = chi[c]; }
                        } else
                            Q.flags[1] = 1;
                    }
                }
            }
            __syncthreads();
            cur ^= 1;
        }
        for (int i = tid; i < total; i += T) gstore(g_pts + i, make_float4(P.c[0][i], P.c[1][i], P.c[2][i], P.c[3][i]));
        __syncthreads();
    }
}

// ---- chunked levels: the first kHugeLevels levels of nodes above kHuge points ------------------------------------
// One workgroup per node cannot pull more than one CU's bandwidth, which made the top three levels of a 180 000-point
// tree cost ~0.5 ms.  Here every pass of such a node is spread over the chip in 2 048-record chunks, one small
// kernel per pass (kernel boundaries are the global sync).  A dependent launch costs ~5 us whatever it does, so the
// passes of a level are folded down to five:
//   A  classify: the split (from the node's extents, which its PARENT's last pass accumulated; the root's are the cloud's
//      bounding box), one class byte per record (< cut, == cut, > cut), the per-chunk "< cut" count, the node's totals
//   B  sweep-0 ranks: misplaced counts of the earlier chunks follow from the per-chunk counts (only the chunk lim1 falls
//      into needs its class bytes); flags come from the class bytes, not the records; one extra workgroup records the
//      level's nodes, routes their children and plans the next level
//   C  sweep-0 swaps of records AND class bytes; a record that arrives right of lim1 adds its sweep-1 flag to its chunk's count
//   D  sweep-1 ranks (flags from the class bytes)
//   E  sweep-1 swaps + the extents of the two children, accumulated into the children's tasks
// Exactly the same arithmetic and the same closed-form Hoare sweeps as build_level_kernel.
constexpr int kHuge = kMid;       // everything smaller goes to build_mid_kernel
constexpr int kHugeLevels = 8;    // at most; the host launches ceil(log2(n_max / kMid)) of them (HugeState::levels)
constexpr int kChunk = 2048;

struct HugeTask {
    BuildTask k;
    float4* pts;            // the node's first record / class byte / rank-list entries (tree arrays + k.l)
    uint8_t* cls;
    int32_t* posL;
    int32_t* posR;
    int chunk0, nchunks;
    unsigned mn[3], mx[3];  // ordered-uint accumulators (filled by the parent's pass E / from the bounding box)
    int lt, le;
    unsigned maxlt, mingt;  // ordered-uint
    int kid[2];             // slots of the children among the next level's chunked tasks, -1 = not chunked
    int m0, m1;             // swap pairs of sweeps 0 / 1 (passes B / D, by the workgroup of the node's last chunk)
};

// what a workgroup needs to start on its chunk: ONE 32-byte read instead of a chain through the level, task and tree tables
struct alignas(32) ChunkDesc {
    float4* pts;   // first record of the chunk
    uint8_t* cls;  // its class bytes
    int32_t task, first, count, pad;  // records [first, first+count) relative to the node's first record; task < 0 = no chunk
};

struct HugeState {
    HugeTask* tasks[2];
    int32_t* ntasks;       // [kHugeLevels + 1]
    int32_t* nchunks;      // [kHugeLevels + 1] total chunks of the level
    int32_t* c_lt;         // [max_chunks] records < cut of a chunk (pass A)
    int32_t* c_mL;         // [max_chunks] misplaced-left count of a chunk, sweep 1 (passes B + C)
    int32_t* c_mR;
    ChunkDesc* desc[2];    // [max_chunks] by level parity (written by the plan)
    int32_t cap_tasks, cap_chunks;
    int32_t grid_chunks;   // chunk workgroups the host launches per pass
    int32_t levels;        // chunked levels launched by the host
};

// returns the slot among the next level's chunked tasks, or -1 when the task went to the other queues
__device__ __forceinline__ int route_task(const BuildQueues& Q, const HugeState& H, const BuildTree& tr, const BuildTask& t, int next_level)
{
    if (t.r - t.l > kHuge && next_level < H.levels) {
        const int slot = atomicAdd(&H.ntasks[next_level], 1);
        if (slot < H.cap_tasks) {
            HugeTask& h = H.tasks[next_level & 1][slot];
            h.k = t;
            h.pts = tr.pts + t.l;
            h.cls = tr.cls + t.l;
            h.posL = tr.posL + t.l;
            h.posR = tr.posR + t.l;
            for (int a = 0; a < 3; ++a) { h.mn[a] = 0xffffffffu; h.mx[a] = 0u; }
            return slot;
        }
        Q.flags[1] = 1;
        return -1;
    }
    push_task(Q, t, next_level);
    return -1;
}

// Plan of a level, by one workgroup: thread i takes task i (chunk0 = prefix sum of the chunk counts over an LDS copy, accumulator
// reset), then thread c writes the descriptor of chunk c (its task by binary search over the LDS prefix).
__device__ void huge_plan_block(const HugeState& H, int level)
{
    __shared__ int s_nc[256], s_c0[257];
    const int n = min(H.ntasks[level], H.cap_tasks);
    HugeTask* tasks = H.tasks[level & 1];
    ChunkDesc* desc = H.desc[level & 1];
    int carry = 0;
    for (int base = 0; base < n; base += 256) {
        const int i = base + threadIdx.x, nb = min(256, n - base);
        int nc = 0;
        if (i < n) {
            const BuildTask& k = tasks[i].k;
            nc = (k.r - k.l + kChunk - 1) / kChunk;
        }
        __syncthreads();  // the previous round's readers of s_c0 / s_nc are done
        s_nc[threadIdx.x] = nc;
        __syncthreads();
        int c0 = carry, tot = carry;
        for (int j = 0; j < nb; ++j) {
            if (j < (int)threadIdx.x) c0 += s_nc[j];
            tot += s_nc[j];
        }
        s_c0[threadIdx.x] = c0;
        if (threadIdx.x == 0) s_c0[256] = tot;
        if (i < n) {
            HugeTask& t = tasks[i];
            t.chunk0 = c0;
            t.nchunks = nc;
            t.lt = 0; t.le = 0; t.maxlt = 0u; t.mingt = 0xffffffffu;
            t.kid[0] = -1; t.kid[1] = -1;
            t.m0 = 0; t.m1 = 0;
        }
        __syncthreads();
        for (int c = carry + (int)threadIdx.x; c < min(tot, H.cap_chunks); c += 256) {
            int lo = 0, hi = nb - 1;  // last task of the round whose chunk0 <= c
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (s_c0[mid] <= c) lo = mid; else hi = mid - 1;
            }
            // (tasks without chunks cannot exist: a chunked task has > kHuge records)
            const HugeTask& t = tasks[base + lo];
            ChunkDesc d;
            d.first = (c - s_c0[lo]) * kChunk;
            d.count = min(kChunk, t.k.r - t.k.l - d.first);
            d.pts = t.pts + d.first;
            d.cls = t.cls + d.first;
            d.task = base + lo;
            d.pad = 0;
            desc[c] = d;
        }
        carry = tot;
    }
    for (int c = min(carry, H.cap_chunks) + (int)threadIdx.x; c < H.grid_chunks; c += 256) desc[c].task = -1;
    if (threadIdx.x == 0) H.nchunks[level] = min(carry, H.cap_chunks);
}

// Root metadata + routing of every root + the plan of level 0; run by ONE workgroup.
__device__ __forceinline__ void huge_root_block(const BuildTree* __restrict__ trees, int n_trees, const BuildQueues& Q, const HugeState& H)
{
    for (int i = threadIdx.x; i < n_trees; i += blockDim.x) {
        const BuildTree t = trees[i];
        TreeMeta m;
        m.root = 0;
        m.depth = 0;
        for (int a = 0; a < 3; ++a) {
            m.lo[a] = t.n > 0 ? ord2f(t.bbox_ord[a]) : 0.f;
            m.hi[a] = t.n > 0 ? ord2f(t.bbox_ord[3 + a]) : 0.f;
        }
        *t.meta = m;
        if (t.n <= 0) continue;
        BuildTask k;
        k.tree = i; k.l = 0; k.r = t.n; k.parent = -1; k.side = 0; k.level = 0;
        for (int a = 0; a < 3; ++a) { k.lo[a] = m.lo[a]; k.hi[a] = m.hi[a]; }
        const int slot = route_task(Q, H, t, k, 0);
        if (slot >= 0)  // the extents of a root's records ARE its bounding box (same reduction: init_points_kernel)
            for (int a = 0; a < 3; ++a) { H.tasks[0][slot].mn[a] = t.bbox_ord[a]; H.tasks[0][slot].mx[a] = t.bbox_ord[3 + a]; }
    }
    __threadfence();
    __syncthreads();
    huge_plan_block(H, 0);
}

__device__ __forceinline__ ChunkDesc load_desc(const HugeState& H, int level, int c)
{
    const int4* q = reinterpret_cast<const int4*>(H.desc[level & 1] + c);
    const int4 u = gload(q), v = gload(q + 1);
    ChunkDesc d;
    d.pts = reinterpret_cast<float4*>(((unsigned long long)(unsigned)u.y << 32) | (unsigned)u.x);
    d.cls = reinterpret_cast<uint8_t*>(((unsigned long long)(unsigned)u.w << 32) | (unsigned)u.z);
    d.task = v.x; d.first = v.y; d.count = v.z; d.pad = v.w;
    return d;
}

__device__ __forceinline__ SplitChoice huge_split(const HugeTask& t)
{
    float mn[3], mx[3];
    for (int a = 0; a < 3; ++a) { mn[a] = ord2f(t.mn[a]); mx[a] = ord2f(t.mx[a]); }
    return choose_split(t.k.lo, t.k.hi, mn, mx);
}

// sum of one int2 per thread over the workgroup (256 threads), result in every thread; s_red = 8 ints of LDS
__device__ __forceinline__ int2 block_sum2(int x, int y, int* s_red)
{
    x = wave_sum(x);
    y = wave_sum(y);
    __syncthreads();  // s_red free
    if ((threadIdx.x & 63) == 0) { s_red[2 * (threadIdx.x >> 6)] = x; s_red[2 * (threadIdx.x >> 6) + 1] = y; }
    __syncthreads();
    return make_int2(s_red[0] + s_red[2] + s_red[4] + s_red[6], s_red[1] + s_red[3] + s_red[5] + s_red[7]);
}

// pass A
__device__ __forceinline__ void huge_classify_chunk(const HugeState& H, int level, int chunk)
{
    __shared__ int s_i[4][2];
    __shared__ float s_f[4][2];
    const ChunkDesc d = load_desc(H, level, chunk);
    if (d.task < 0) return;
    float4 p[kChunk / 256];
#pragma unroll
    for (int e = 0; e < kChunk / 256; ++e) p[e] = gload(d.pts + min(e * 256 + (int)threadIdx.x, d.count - 1));  // all in flight
    HugeTask& t = H.tasks[level & 1][d.task];
    const SplitChoice sc = huge_split(t);
    int lt = 0, le = 0;
    float maxlt = -INFINITY, mingt = INFINITY;
#pragma unroll
    for (int e = 0; e < kChunk / 256; ++e) {
        const int i = e * 256 + threadIdx.x;
        if (i < d.count) {
            const float v = comp(p[e], sc.ax);
            const bool is_lt = v < sc.cut, is_le = v <= sc.cut;
            lt += is_lt;
            le += is_le;
            if (is_lt) maxlt = fmaxf(maxlt, v);
            if (v > sc.cut) mingt = fminf(mingt, v);
            gstore(d.cls + i, (uint8_t)(is_lt ? 0 : (is_le ? 1 : 2)));
        }
    }
    {
        lt = wave_sum(lt);
        le = wave_sum(le);
        maxlt = wave_max(maxlt);
        mingt = wave_min(mingt);
    }
    if ((threadIdx.x & 63) == 0) {
        s_i[threadIdx.x >> 6][0] = lt; s_i[threadIdx.x >> 6][1] = le;
        s_f[threadIdx.x >> 6][0] = maxlt; s_f[threadIdx.x >> 6][1] = mingt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int clt = s_i[0][0] + s_i[1][0] + s_i[2][0] + s_i[3][0];
        H.c_lt[chunk] = clt;
        atomicAdd(&t.lt, clt);
        atomicAdd(&t.le, s_i[0][1] + s_i[1][1] + s_i[2][1] + s_i[3][1]);
        const float ml = fmaxf(fmaxf(s_f[0][0], s_f[1][0]), fmaxf(s_f[2][0], s_f[3][0]));
        const float mg = fminf(fminf(s_f[0][1], s_f[1][1]), fminf(s_f[2][1], s_f[3][1]));
        if (ml > -INFINITY) atomicMax(&t.maxlt, f2ord(ml));
        if (mg < INFINITY) atomicMin(&t.mingt, f2ord(mg));
    }
}

// Sweep-0 misplaced counts (left, right) of the node's chunks [0, c_to) (indices within the node), summed over the
// workgroup.  A chunk wholly left of lim1 misplaces its records that are not "< cut", one wholly right of it those that are;
// the one chunk lim1 falls into is counted from its class bytes.
__device__ __forceinline__ int2 sweep0_counts(const HugeState& H, const HugeTask& t, int n, int lim1, int chunk0, int c_to, int* s_red)
{
    int mL = 0, mR = 0;
    for (int c = (int)threadIdx.x; c < c_to; c += 256) {
        const int f = c * kChunk, e = min(n, f + kChunk);
        const int clt = H.c_lt[chunk0 + c];
        if (e <= lim1) mL += (e - f) - clt;
        else if (f >= lim1) mR += clt;
    }
    const int cs = lim1 / kChunk;  // the chunk lim1 falls into, if it cuts one
    if (lim1 % kChunk != 0 && lim1 < n && cs < c_to) {
        int cl[kChunk / 256];
#pragma unroll
        for (int j = 0; j < kChunk / 256; ++j) cl[j] = gload(t.cls + min(cs * kChunk + j * 256 + (int)threadIdx.x, n - 1));
#pragma unroll
        for (int j = 0; j < kChunk / 256; ++j) {
            const int p = cs * kChunk + j * 256 + (int)threadIdx.x;
            mL += (p < lim1 && cl[j] != 0);
            mR += (p >= lim1 && p < n && cl[j] == 0);
        }
    }
    return block_sum2(mL, mR, s_red);
}

// class bytes of one chunk, wave w owning records [w*512, w*512+512) of the chunk, striped (-1 past the end)
__device__ __forceinline__ void chunk_classes(const ChunkDesc& d, int wave, int lane, int (&cl)[8])
{
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int i = wave * 512 + e * 64 + lane;
        cl[e] = gload(d.cls + min(i, d.count - 1));
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) cl[e] = wave * 512 + e * 64 + lane < d.count ? cl[e] : -1;
}
// flags of sweep S as 8 ballots per wave
template <int S>
__device__ __forceinline__ void chunk_flags(const int (&cl)[8], const ChunkDesc& d, int lim1, int lim2, int wave, int lane,
                                            unsigned long long (&bL)[8], unsigned long long (&bR)[8])
{
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int p = d.first + wave * 512 + e * 64 + lane;  // within the node
        bool isL, isR;
        if (S == 0) {
            isL = cl[e] > 0 && p < lim1;
            isR = cl[e] == 0 && p >= lim1;
        } else {
            isL = cl[e] == 2 && p >= lim1 && p < lim2;
            isR = cl[e] == 1 && p >= lim2;
        }
        bL[e] = __ballot(isL);
        bR[e] = __ballot(isR);
    }
}

// passes B (S = 0) and D (S = 1)
template <int S>
__device__ __forceinline__ void huge_scatter_chunk(const HugeState& H, int level, int chunk)
{
    __shared__ int s_red[8];
    __shared__ int s_c[4][2];
    const ChunkDesc d = load_desc(H, level, chunk);
    if (d.task < 0) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int cl[8];
    chunk_classes(d, wave, lane, cl);  // (in flight under the prefix below)
    HugeTask& t = H.tasks[level & 1][d.task];
    const int lim1 = t.lt, lim2 = t.le, chunk0 = t.chunk0, nchunks = t.nchunks;
    int32_t* const posL = t.posL;
    int32_t* const posR = t.posR;
    // ranks before this chunk = misplaced counts of the node's earlier chunks
    int2 off;
    if (S == 0)
        off = sweep0_counts(H, t, t.k.r - t.k.l, lim1, chunk0, chunk - chunk0, s_red);
    else {
        int pl = 0, pr = 0;
        for (int c = chunk0 + threadIdx.x; c < chunk; c += 256) { pl += H.c_mL[c]; pr += H.c_mR[c]; }
        off = block_sum2(pl, pr, s_red);
    }
    unsigned long long bL[8], bR[8];
    chunk_flags<S>(cl, d, lim1, lim2, wave, lane, bL, bR);
    int wL = 0, wR = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) { wL += __popcll(bL[e]); wR += __popcll(bR[e]); }
    if (lane == 0) { s_c[wave][0] = wL; s_c[wave][1] = wR; }
    __syncthreads();
    int rL = off.x, rR = off.y;
    if (threadIdx.x == 0 && chunk == chunk0 + nchunks - 1) {
        const int m = off.x + s_c[0][0] + s_c[1][0] + s_c[2][0] + s_c[3][0];
        if (S == 0) t.m0 = m; else t.m1 = m;
    }
    for (int w = 0; w < wave; ++w) { rL += s_c[w][0]; rR += s_c[w][1]; }
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int p = d.first + wave * 512 + e * 64 + lane;
        if ((bL[e] >> lane) & 1ull) gstore(posL + rL + __popcll(bL[e] & lt_mask), p);
        if ((bR[e] >> lane) & 1ull) gstore(posR + rR + __popcll(bR[e] & lt_mask), p);
        rL += __popcll(bL[e]);
        rR += __popcll(bR[e]);
    }
    if (S == 0) {
        // sweep-1 flags of the records sweep 0 leaves in place (right of lim1 and not "< cut"); pass C adds the arrivals
        unsigned long long b1L[8], b1R[8];
        chunk_flags<1>(cl, d, lim1, lim2, wave, lane, b1L, b1R);
        int v1L = 0, v1R = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) { v1L += __popcll(b1L[e]); v1R += __popcll(b1R[e]); }
        __syncthreads();  // s_c read by everyone
        if (lane == 0) { s_c[wave][0] = v1L; s_c[wave][1] = v1R; }
        __syncthreads();
        if (threadIdx.x == 0) {
            H.c_mL[chunk] = s_c[0][0] + s_c[1][0] + s_c[2][0] + s_c[3][0];
            H.c_mR[chunk] = s_c[0][1] + s_c[1][1] + s_c[2][1] + s_c[3][1];
        }
    }
}

// The m swap pairs of a sweep are dealt out evenly over the node's chunks (their workgroups exist anyway): pairs [lo, hi) for
// chunk c of nchunks, in multiples of 64.
__device__ __forceinline__ void pair_share(int m, int c, int nchunks, int* lo, int* hi)
{
    const int per = (((m + nchunks - 1) / nchunks) + 63) & ~63;
    *lo = min(m, c * per);
    *hi = min(m, (c + 1) * per);
}
constexpr int kPairBatch = 2;  // pairs per thread in flight together

// pass C: sweep-0 swaps
__device__ __forceinline__ void huge_swap0_chunk(const HugeState& H, int level, int chunk)
{
    const ChunkDesc d = load_desc(H, level, chunk);
    if (d.task < 0) return;
    const HugeTask& t = H.tasks[level & 1][d.task];
    float4* const a = t.pts;
    uint8_t* const cls = t.cls;
    const int lim2 = t.le, chunk0 = t.chunk0;
    const int m = t.m0;  // (the class bytes change under this pass: the count was taken in pass B)
    int lo, hi;
    pair_share(m, chunk - chunk0, t.nchunks, &lo, &hi);
    for (int base = lo + (int)threadIdx.x; base < hi; base += 256 * kPairBatch) {
        int pl[kPairBatch], pr[kPairBatch], cx[kPairBatch];
        float4 x[kPairBatch], y[kPairBatch];
#pragma unroll
        for (int u = 0; u < kPairBatch; ++u) {
            const int i = min(base + u * 256, hi - 1);
            pl[u] = gload(t.posL + i);
            pr[u] = gload(t.posR + m - 1 - i);
        }
#pragma unroll
        for (int u = 0; u < kPairBatch; ++u) {
            x[u] = gload(a + pl[u]);
            y[u] = gload(a + pr[u]);
            cx[u] = gload(cls + pl[u]);  // 1 or 2; the record at pr was "< cut"
        }
#pragma unroll
        for (int u = 0; u < kPairBatch; ++u) {
            int key = -1;
            if (base + u * 256 < hi) {
                gstore(a + pl[u], y[u]);
                gstore(a + pr[u], x[u]);
                gstore(cls + pl[u], (uint8_t)0);
                gstore(cls + pr[u], (uint8_t)cx[u]);
                // the record now at pr (right of lim1) may be misplaced for sweep 1
                if (pr[u] < lim2) {
                    if (cx[u] == 2) key = 2 * (pr[u] / kChunk);
                } else if (cx[u] == 1)
                    key = 2 * (pr[u] / kChunk) + 1;
            }
            // one atomic per (wave, chunk, side): consecutive pairs land close together
            unsigned long long todo = __ballot(key >= 0);
            while (todo) {
                const int lead = __builtin_ctzll(todo);
                const int k0 = __shfl(key, lead);
                const unsigned long long same = __ballot(key == k0);
                if ((int)(threadIdx.x & 63) == lead) atomicAdd(((k0 & 1) ? H.c_mR : H.c_mL) + chunk0 + (k0 >> 1), __popcll(same));
                todo &= ~same;
            }
        }
    }
}

// pass E: sweep-1 swaps + the extents of the two children
__device__ __forceinline__ void huge_swap1_chunk(const HugeState& H, int level, int chunk)
{
    __shared__ float s_ext[4][12];
    const ChunkDesc d = load_desc(H, level, chunk);
    if (d.task < 0) return;
    // the chunk's records and class bytes (neither is needed when no child is chunked, but the reads are cheap and start the chain early)
    float4 q[kChunk / 256];
    int cl[kChunk / 256];
#pragma unroll
    for (int e = 0; e < kChunk / 256; ++e) {
        const int i = min(e * 256 + (int)threadIdx.x, d.count - 1);
        q[e] = gload(d.pts + i);
        cl[e] = gload(d.cls + i);
    }
    const HugeTask& t = H.tasks[level & 1][d.task];
    float4* const a = t.pts;
    const int n = t.k.r - t.k.l, lim1 = t.lt, lim2 = t.le, half = n / 2;
    const int idx = lim1 > half ? lim1 : (lim2 < half ? lim2 : half);
    const int kid0 = t.kid[0], kid1 = t.kid[1];
    const bool want = kid0 >= 0 || kid1 >= 0;
    const int m = t.m1;
    float lo[2][3], hi[2][3];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int c = 0; c < 3; ++c) { lo[s][c] = INFINITY; hi[s][c] = -INFINITY; }
    auto acc = [&](const float4& p, bool right) {
        const float v[3] = {p.x, p.y, p.z};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            lo[0][c] = fminf(lo[0][c], right ? INFINITY : v[c]);  hi[0][c] = fmaxf(hi[0][c], right ? -INFINITY : v[c]);
            lo[1][c] = fminf(lo[1][c], right ? v[c] : INFINITY);  hi[1][c] = fmaxf(hi[1][c], right ? v[c] : -INFINITY);
        }
    };
    // the records of this chunk that stay where they are (class bytes are not written in this pass; a record that moves is
    // accounted for by the thread that moves it)
#pragma unroll
    for (int e = 0; e < kChunk / 256; ++e) {
        const int i = e * 256 + (int)threadIdx.x, p = d.first + i;
        const bool moves = p >= lim1 && (p < lim2 ? cl[e] == 2 : cl[e] == 1);
        if (i < d.count && !moves) acc(q[e], p >= idx);
    }
    // this chunk's share of the m swap pairs: x (> cut) goes right of lim2, y (== cut) into [lim1, lim2)
    int plo, phi;
    pair_share(m, chunk - t.chunk0, t.nchunks, &plo, &phi);
    for (int base = plo + (int)threadIdx.x; base < phi; base += 256 * kPairBatch) {
        int pl[kPairBatch], pr[kPairBatch];
        float4 x[kPairBatch], y[kPairBatch];
#pragma unroll
        for (int u = 0; u < kPairBatch; ++u) {
            const int i = min(base + u * 256, phi - 1);
            pl[u] = gload(t.posL + i);
            pr[u] = gload(t.posR + m - 1 - i);
        }
#pragma unroll
        for (int u = 0; u < kPairBatch; ++u) {
            x[u] = gload(a + pl[u]);
            y[u] = gload(a + pr[u]);
        }
#pragma unroll
        for (int u = 0; u < kPairBatch; ++u)
            if (base + u * 256 < phi) {
                gstore(a + pl[u], y[u]);
                gstore(a + pr[u], x[u]);
                acc(x[u], pr[u] >= idx);
                acc(y[u], pl[u] >= idx);
            }
    }
    if (!want) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float l = wave_min(lo[s][c]), h = wave_max(hi[s][c]);
            if (lane == 0) { s_ext[wave][s * 6 + c] = l; s_ext[wave][s * 6 + 3 + c] = h; }
        }
    __syncthreads();
    if (threadIdx.x < 12) {
        const int j = threadIdx.x, s = j / 6, c = j % 3;
        const bool is_hi = (j % 6) >= 3;
        const int kid = s == 0 ? kid0 : kid1;
        if (kid >= 0) {
            HugeTask& k = H.tasks[(level + 1) & 1][kid];
            if (is_hi) {
                const float h = fmaxf(fmaxf(s_ext[0][j], s_ext[1][j]), fmaxf(s_ext[2][j], s_ext[3][j]));
                if (h > -INFINITY) atomicMax(&k.mx[c], f2ord(h));
            } else {
                const float l = fminf(fminf(s_ext[0][j], s_ext[1][j]), fminf(s_ext[2][j], s_ext[3][j]));
                if (l < INFINITY) atomicMin(&k.mn[c], f2ord(l));
            }
        }
    }
}

// Records the level's nodes, routes their children and plans the next level's chunk ranges; run by ONE workgroup
// (thread i = task i; the plan is a prefix sum over the handful of next-level tasks).
__device__ __forceinline__ void huge_emit_block(const BuildTree* __restrict__ trees, const BuildQueues& Q, const HugeState& H, int level)
{
    const int n = min(H.ntasks[level], H.cap_tasks);
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        HugeTask& t = H.tasks[level & 1][i];
        const BuildTree tr = trees[t.k.tree];
        const SplitChoice sc = huge_split(t);
        BuildTask kids[2];
        int id;
        emit_inner(Q, tr, t.k, sc.ax, sc.cut, t.lt, t.le, ord2f(t.maxlt), ord2f(t.mingt), kids, &id);
        link_to_parent(tr, t.k, id);
        const int k0 = route_task(Q, H, tr, kids[0], level + 1);
        const int k1 = route_task(Q, H, tr, kids[1], level + 1);
        t.kid[0] = k0;
        t.kid[1] = k1;
    }
    __threadfence();
    __syncthreads();
    if (level + 1 < H.levels) huge_plan_block(H, level + 1);
}

// ---- launch wrappers of the chunked passes ------------------------------------------------------------------------------
// (One cooperative launch with grid-wide barriers between the passes was measured and rejected: every barrier needs an
// agent-scope release/acquire -- an L2 write-back + invalidate per workgroup on this multi-XCD part -- and came to
// ~12 us per pass against ~7 us for a dependent kernel launch.)
__global__ __launch_bounds__(256) void huge_root_kernel(const BuildTree* __restrict__ trees, int n_trees, BuildQueues Q, HugeState H)
{
    huge_root_block(trees, n_trees, Q, H);
}
// PHASE 1 (pass B) carries one extra workgroup (the last one) that records the level's nodes, routes their children and plans
// the next level while the others rank: everything it needs (extents, counts) is final after pass A, and what it writes (node
// records, queues, the next level's task / chunk tables, kid slots) is read by nobody before pass E.
template <int PHASE>
__global__ __launch_bounds__(256) void huge_phase_kernel(const BuildTree* __restrict__ trees, BuildQueues Q, HugeState H, int level)
{
    if (PHASE == 1 && blockIdx.x == gridDim.x - 1) {
        huge_emit_block(trees, Q, H, level);
        return;
    }
    const int c = blockIdx.x;
    if (PHASE == 0) huge_classify_chunk(H, level, c);
    if (PHASE == 1) huge_scatter_chunk<0>(H, level, c);
    if (PHASE == 2) huge_swap0_chunk(H, level, c);
    if (PHASE == 3) huge_scatter_chunk<1>(H, level, c);
    if (PHASE == 4) huge_swap1_chunk(H, level, c);
}

// Last launch of a build: clears the "unfinished" word (kept in the flag layout; the straggler kernel leaves nothing unfinished).
__global__ void build_finish_kernel(BuildQueues Q)
{
    Q.flags[2] = 0;
}

// ---- fat node image (TreeView::fat): every inner node's record next to the records of its two children -------------------------------
// One thread per node id.  Odd ids that no inner node owns (positions inside a leaf) hold whatever the arena held before: their
// slots are never visited by a search, so the only care they need is that a stale child reference is not followed out of bounds.
__global__ __launch_bounds__(256) void fatten_kernel(const BuildTree* __restrict__ trees, int ids_x)
{
    const BuildTree tr = trees[blockIdx.y];
    const int limit = 2 * tr.n;
    for (int id = 2 * (int)(blockIdx.x * 256 + threadIdx.x) + 1; id < limit; id += 2 * ids_x * 256) {
        const int4 nd = gload(tr.nodes + id);
        const int c1 = nd.x & 0x3fffffff, c2 = nd.y;
        int4 k1 = make_int4(0, 0, 0, 0), k2 = k1;
        if ((c1 & 1) && c1 > 0 && c1 < limit) k1 = gload(tr.nodes + c1);
        if ((c2 & 1) && c2 > 0 && c2 < limit) k2 = gload(tr.nodes + c2);
        int4* f = tr.fat + 3 * (size_t)id;
        gstore(f, nd);
        gstore(f + 1, k1);
        gstore(f + 2, k2);
    }
}

static int launch_mid_and_subtrees(ps_context* c, const BuildTree* d_trees, const BuildQueues& Q, size_t tot, size_t T, size_t small_cap)
{
    constexpr size_t mid_lds = sizeof(float4) * kMid + sizeof(short) * kMid;
    if (!c->mid_lds_attr) {  // (per context = per device: function attributes do not carry over to another GPU)
        PS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(build_mid_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)mid_lds));
        c->mid_lds_attr = true;
    }
    hipStream_t st = c->stream;
    const unsigned grid_mid = (unsigned)std::min<size_t>(tot / kSmall + T + 1, 1024);
    hipLaunchKernelGGL(build_mid_kernel, dim3(grid_mid), dim3(kMidThreads), mid_lds, st, d_trees, Q);
    hipLaunchKernelGGL(build_flat_kernel, dim3((unsigned)std::min<size_t>(small_cap, 4096)), dim3(kSmall), 0, st, d_trees, Q);
    PS_HIP(hipGetLastError());
    return PS_OK;
}

// ---- host side -----------------------------------------------------------------------------------------------
void TreeSetPlan::carve(Arena& a)
{
    const size_t T = n.size();
    const size_t tot = total_points();
    for (size_t i = 0; i < T; ++i) {
        d_nodes[i] = a.take<int4>(2 * (size_t)(n[i] > 0 ? n[i] : 1));
        d_fat[i] = a.take<int4>(6 * (size_t)(n[i] > 0 ? n[i] : 1));
        d_pts[i] = a.take<float4>((size_t)(n[i] > 0 ? n[i] : 1) + kLeafMax);  // (+ kLeafMax: the search reads whole leaf slots, kdtree.h)
    }
    d_meta = a.take<TreeMeta>(T);
    {   // the head block (kdtree_build.h): one copy initialises all of it
        size_t hb = 0;
        auto addh = [&](size_t b) { size_t o = hb; hb += (b + 255) & ~size_t(255); return o; };
        addh(128 * (T + (size_t)extra_jobs));                      // job table at offset 0
        off_flags = addh(sizeof(int32_t) * 16);
        off_trees = addh(sizeof(BuildTree) * T);
        off_bbox = addh(sizeof(unsigned) * 8 * T);
        off_cnt = addh(sizeof(int32_t) * (kMaxLevels + 8));
        off_hcnt = addh(sizeof(int32_t) * 2 * (kHugeLevels + 2));
        head_bytes = hb;
        d_head = a.take<char>(hb);
        d_jobs = d_head;
        d_flags = reinterpret_cast<int32_t*>(d_head + off_flags);
        host_blob.assign(hb, 0);
    }
    // builder scratch: posL/posR, queues, chunked-level state (no initial contents needed)
    const size_t q_cap = tot / kSmall + 2 * T + 64, small_cap = tot / 4 + 2 * T + 1024;
    size_t bytes = 0;
    auto add = [&](size_t b) { size_t o = bytes; bytes += (b + 255) & ~size_t(255); return o; };
    add(sizeof(int32_t) * 2 * (tot + T));
    add(sizeof(BuildTask) * q_cap * 3);
    add(sizeof(BuildTask) * small_cap);
    {   // chunked-level state
        const size_t cap_tasks = tot / kHuge + T + 8, cap_chunks = tot / kChunk + cap_tasks + 8;
        add(sizeof(HugeTask) * cap_tasks * 2);
        add(sizeof(int32_t) * 3 * cap_chunks);
        add(sizeof(ChunkDesc) * 2 * cap_chunks);
        add(tot + 16 * T + 16);
    }
    scratch_bytes = bytes;
    d_scratch = a.take<char>(bytes);
}

int build_trees(ps_context* c, TreeSetPlan& plan)
{
    const size_t T = plan.n.size();
    if (T == 0) return PS_OK;
    const size_t tot = plan.total_points();
    const size_t q_cap = tot / kSmall + 2 * T + 64, small_cap = tot / 4 + 2 * T + 1024;
    char* base = static_cast<char*>(plan.d_scratch);
    size_t off = 0;
    auto take = [&](size_t b) { char* p = base + off; off += (b + 255) & ~size_t(255); return p; };
    BuildTree* d_trees = reinterpret_cast<BuildTree*>(plan.d_head + plan.off_trees);
    unsigned* d_bbox = reinterpret_cast<unsigned*>(plan.d_head + plan.off_bbox);
    int32_t* d_cnt = reinterpret_cast<int32_t*>(plan.d_head + plan.off_cnt);
    int32_t* d_hcnt = reinterpret_cast<int32_t*>(plan.d_head + plan.off_hcnt);
    int32_t* d_pos = reinterpret_cast<int32_t*>(take(sizeof(int32_t) * 2 * (tot + T)));
    BuildTask* d_q = reinterpret_cast<BuildTask*>(take(sizeof(BuildTask) * q_cap * 3));
    BuildTask* d_small = reinterpret_cast<BuildTask*>(take(sizeof(BuildTask) * small_cap));
    const size_t cap_tasks = tot / kHuge + T + 8, cap_chunks = tot / kChunk + cap_tasks + 8;
    HugeTask* d_huge = reinterpret_cast<HugeTask*>(take(sizeof(HugeTask) * cap_tasks * 2));
    int32_t* d_cm = reinterpret_cast<int32_t*>(take(sizeof(int32_t) * 3 * cap_chunks));
    ChunkDesc* d_desc = reinterpret_cast<ChunkDesc*>(take(sizeof(ChunkDesc) * 2 * cap_chunks));
    uint8_t* d_cls = reinterpret_cast<uint8_t*>(take(tot + 16 * T + 16));

    // host image of the head block (the caller has written its job table at offset 0; flags and counters are zero from carve())
    PS_CHECK(plan.host_blob.size() == plan.head_bytes, "build_trees: the plan was not carved");
    std::memset(plan.host_blob.data() + plan.off_flags, 0, plan.head_bytes - plan.off_flags);
    BuildTree* h_trees = reinterpret_cast<BuildTree*>(plan.host_blob.data() + plan.off_trees);
    unsigned* h_bbox = reinterpret_cast<unsigned*>(plan.host_blob.data() + plan.off_bbox);
    for (size_t i = 0; i < 8 * T; ++i) h_bbox[i] = (i & 7) < 3 ? 0xffffffffu : 0u;  // ordered-uint min / max accumulators
    size_t pos_off = 0, cls_off = 0;
    int32_t max_n = 0;
    for (size_t i = 0; i < T; ++i) {
        BuildTree& t = h_trees[i];
        t.src = plan.src[i];
        t.pts = plan.d_pts[i];
        t.nodes = plan.d_nodes[i];
        t.fat = plan.d_fat[i];
        t.copy_dst = i < plan.copy_dst.size() ? plan.copy_dst[i] : nullptr;
        t.meta = plan.d_meta + i;
        t.posL = d_pos + pos_off;
        t.posR = d_pos + pos_off + (size_t)plan.n[i] + 1;
        pos_off += 2 * ((size_t)plan.n[i] + 1);
        t.bbox_ord = d_bbox + 8 * i;
        t.cls = d_cls + cls_off;
        cls_off += ((size_t)plan.n[i] + 15) & ~size_t(15);
        t.n = plan.n[i];
        max_n = std::max(max_n, t.n);
    }
    hipStream_t st = c->stream;
    PS_TRY(c->upload_async(plan.d_head, plan.host_blob.data(), plan.head_bytes));  // jobs, status words, tree table, accumulators, counters

    HugeState H;
    H.tasks[0] = d_huge;
    H.tasks[1] = d_huge + cap_tasks;
    H.ntasks = d_hcnt;
    H.nchunks = d_hcnt + (kHugeLevels + 2);
    H.c_mL = d_cm;
    H.c_mR = d_cm + cap_chunks;
    H.c_lt = d_cm + 2 * cap_chunks;
    H.desc[0] = d_desc;
    H.desc[1] = d_desc + cap_chunks;
    H.cap_tasks = (int32_t)cap_tasks;
    H.cap_chunks = (int32_t)cap_chunks;

    BuildQueues Q;
    Q.q[0] = d_q;
    Q.q[1] = d_q + q_cap;
    Q.mid_q = d_q + 2 * q_cap;
    Q.small_q = d_small;
    Q.level_cnt = d_cnt;
    Q.small_cnt = d_cnt + kMaxLevels;
    Q.flags = plan.d_flags;
    Q.q_cap = (int32_t)q_cap;
    Q.small_cap = (int32_t)small_cap;

    // chunked levels until the nodes of a balanced tree are below kMid; whatever is still above kMid after them (lopsided
    // splits) is finished by ONE straggler launch (build_level_kernel, depth first per node)
    int huge_levels = 0;
    while (huge_levels < kHugeLevels && ((size_t)max_n >> huge_levels) > (size_t)kMid) ++huge_levels;
    H.levels = huge_levels;

    const int chunks_x = std::max(1, std::min(ceil_div(max_n, 256 * 4), 256));
    hipLaunchKernelGGL(init_points_kernel, dim3(chunks_x, (unsigned)T), dim3(256), 0, st, d_trees, chunks_x);
    const int grid_big = (int)std::min<size_t>(q_cap, tot / kMid + T + 1);
    const unsigned n_chunks_max = (unsigned)std::min<size_t>(cap_chunks, tot / kChunk + cap_tasks);
    {
        const dim3 gc(std::max(1u, n_chunks_max)), bc(256);
        H.grid_chunks = (int32_t)gc.x;
        hipLaunchKernelGGL(huge_root_kernel, dim3(1), dim3(256), 0, st, d_trees, (int)T, Q, H);
        for (int level = 0; level < huge_levels; ++level) {
            // nodes above kHuge points: every pass spread over the chip (see "chunked levels" above)
            hipLaunchKernelGGL(huge_phase_kernel<0>, gc, bc, 0, st, d_trees, Q, H, level);
            hipLaunchKernelGGL(huge_phase_kernel<1>, dim3(gc.x + 1), bc, 0, st, d_trees, Q, H, level);  // + the emit workgroup
            hipLaunchKernelGGL(huge_phase_kernel<2>, gc, bc, 0, st, d_trees, Q, H, level);
            hipLaunchKernelGGL(huge_phase_kernel<3>, gc, bc, 0, st, d_trees, Q, H, level);
            hipLaunchKernelGGL(huge_phase_kernel<4>, gc, bc, 0, st, d_trees, Q, H, level);
        }
    }
    // (the level queues are indexed by the level a task was pushed FOR: chunked level L pushes for L + 1, the roots for 0)
    hipLaunchKernelGGL(build_level_kernel, dim3(grid_big), dim3(kBigThreads), 0, st, d_trees, Q, huge_levels);
    PS_TRY(launch_mid_and_subtrees(c, d_trees, Q, tot, T, small_cap));
    {
        // the searches' image of the finished trees: every inner node next to its children's records (TreeView::fat)
        const int ids_x = std::max(1, std::min(ceil_div(max_n, 256), 1024));
        hipLaunchKernelGGL(fatten_kernel, dim3(ids_x, (unsigned)T), dim3(256), 0, st, d_trees, ids_x);
        PS_HIP(hipGetLastError());
    }
    plan.launches = 6 + 5 * huge_levels;
    return PS_OK;
}

}  // namespace ps
