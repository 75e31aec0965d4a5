// regchain.hip -- layer chains with the activations in REGISTERS between the layers (see rowgemm.h: rowchain).
//
// The chain is evaluated transposed: per 16-point tile a wave computes Y^T = W^T . X^T on v_mfma_f32_16x16x4_f32, A = a
// weight fragment, B = the activations.  In that orientation the accumulator layout of one layer IS the B-operand layout of the
// next: lane (point p = lane & 15, group g = lane >> 4) ends a layer holding output channels 16t + 4g + q (q = 0..3) of its
// point for every output tile t, and the k-step (t, q) of the next layer needs, from the same lane, exactly that value -- its
// four k-slots are then the channels {16t + q, 16t + 4 + q, 16t + 8 + q, 16t + 12 + q}, an order of the K axis like any other
// (the sum over k does not care; the weight fragments are laid out to match while they are staged into LDS).  So between two
// layers there is only bias + activation on the accumulator registers: no LDS round trip, no wave barrier, no re-layout.
// A lane's four channels of a tile are contiguous in memory, so inputs, appended sources and stored outputs are 16-byte
// accesses.  LDS holds the chain's weights (read once per workgroup, in k-step order: one conflict-free ds_read_b32 per MFMA).
// The tile counts of every layer are template parameters (the network uses seven chain shapes; any other chain takes the
// LDS-staged rowchain kernel), and the next tile's input rows are loaded before the current tile's MFMAs start.
//
// Reference formulation: helper_tf_util.conv2d / conv2d_transpose / tf.layers.dense on [B,N,1,C] tensors
// (PointSegment/helper_tf_util.py:115-250, RandLANet.py:113-151, 314-321), inference-mode BatchNorm folded on the host.
// Bound: HBM (only the chain's input rows and the rows somebody else reads cross it); fp32 MFMA for the widest chains.
#include <algorithm>

#include "mfma_tile.h"
#include "rowgemm.h"

namespace ps {

struct RcLayer {
    const float* wp;    // the layer's standard packed image (PackedLinear::wp, rowgemm.h); re-ordered while staged into LDS
    const float* bias;  // [>= ot*16]
    float* y;           // optional store of this layer's output rows
    const float* ex;    // optional plain-row source appended to the K axis (layer > 0)
    int ca, cb;         // channels of the two K segments: layer 0: s1 | s2; later: previous activations | ex
    int ta, tb;         // their tile counts
    int cout, ot;
    int ks, ntb;        // geometry of wp
    int ldy, ldex, leaky;
    int w_off, b_off;   // float offsets inside the workgroup's LDS image
};
struct RcArgs {
    const float* x1; const int32_t* g1; int ld1, g1m, g1n;
    const float* x2; const int32_t* g2; int ld2, g2m, g2n;
    int n, R;
    RcLayer l[kChainMaxSteps];
    const float* image;  // the chain's finished LDS image (weights in k-step order + biases), or nullptr: re-order while staging
    int image_floats;
};

// Chain shape: N layers; layer l reads IN(l) = (l == 0 ? A0 : O(l-1)) + B(l) sixteen-channel tiles and writes O(l).
template <int N_, int A0_, int B0_, int O0_, int B1_ = 0, int O1_ = 0, int B2_ = 0, int O2_ = 0, int B3_ = 0, int O3_ = 0>
struct RcShape {
    static constexpr int N = N_, A0 = A0_;
    static constexpr int B[4] = {B0_, B1_, B2_, B3_};
    static constexpr int O[4] = {O0_, O1_, O2_, O3_};
    static constexpr int A(int l) { return l == 0 ? A0_ : O[l - 1]; }
    static constexpr int IN(int l) { return A(l) + B[l]; }
    static constexpr int maxt()
    {
        int m = 1;
        for (int l = 0; l < N_; ++l) {
            m = IN(l) > m ? IN(l) : m;
            m = O[l] > m ? O[l] : m;
        }
        return m;
    }
};

// dst[T0 .. T0+NT)[q] <- channels 16(t - T0) + 4g + q of the row at rowp (zero beyond c).  Only the last tile can be partial.
template <int T0, int NT, int MT>
__device__ __forceinline__ void rc_load(float (&dst)[MT][4], const float* __restrict__ rowp, int c, bool vec, int g)
{
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int ch = t * 16 + 4 * g;
        if (vec) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t + 1 < NT || ch < c) v = *reinterpret_cast<const float4*>(rowp + ch);
            dst[T0 + t][0] = v.x; dst[T0 + t][1] = v.y; dst[T0 + t][2] = v.z; dst[T0 + t][3] = v.w;
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) dst[T0 + t][q] = (t + 1 < NT || ch + q < c) ? rowp[ch + q] : 0.f;
        }
    }
}

template <class S, int L>
__device__ __forceinline__ void rc_layer(const RcArgs& a, float (&act)[S::maxt()][4], const float* lds, int lane, int g, int row, int rr, bool ok)
{
    if constexpr (L < S::N) {
        constexpr int IT = S::IN(L), OT = S::O[L], TA = S::A(L), TB = S::B[L];
        const RcLayer& y = a.l[L];
        if constexpr (L > 0 && TB > 0) {
            const bool ve = (y.cb & 3) == 0 && (y.ldex & 3) == 0 && (reinterpret_cast<uintptr_t>(y.ex) & 15) == 0;
            rc_load<TA, TB>(act, y.ex + (size_t)rr * y.ldex, y.cb, ve, g);
        }
        f32x4 acc[OT];
#pragma unroll
        for (int to = 0; to < OT; ++to) acc[to] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (L > 0) __builtin_amdgcn_sched_barrier(0);  // keep this layer's weight reads behind the previous layer's MFMAs
        const float* w = lds + y.w_off + lane;
        // k-steps outermost: consecutive MFMAs go to different accumulators.  Wide layers go four output tiles at a time with a
        // scheduling fence between the groups: the compiler otherwise hoists every weight read of the layer in front of the first
        // MFMA (394 VGPRs, one wave per SIMD, for the 128-channel layer)
#pragma unroll
        for (int t0 = 0; t0 < OT; t0 += 4) {
#pragma unroll
            for (int ti = 0; ti < IT; ++ti)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int to = t0; to < (t0 + 4 < OT ? t0 + 4 : OT); ++to)
                        acc[to] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[((to * IT + ti) * 4 + q) * 64], act[ti][q], acc[to], 0, 0, 0);
            if (t0 + 4 < OT) __builtin_amdgcn_sched_barrier(0);
        }
        const bool vy = y.y && (y.cout & 3) == 0 && (y.ldy & 3) == 0 && (reinterpret_cast<uintptr_t>(y.y) & 15) == 0;
#pragma unroll
        for (int to = 0; to < OT; ++to) {
            const int ch = to * 16 + 4 * g;
            const float4 b = *reinterpret_cast<const float4*>(lds + y.b_off + ch);  // zero on padding channels, like the weights
            float v[4] = {acc[to][0] + b.x, acc[to][1] + b.y, acc[to][2] + b.z, acc[to][3] + b.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (y.leaky) v[q] = leaky02(v[q]);
                act[to][q] = v[q];
            }
            if (y.y && ok) {
                float* o = y.y + (size_t)row * y.ldy + ch;
                if (vy) {
                    if (to + 1 < OT || ch < y.cout) *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (to + 1 < OT || ch + q < y.cout) o[q] = v[q];
                }
            }
        }
        rc_layer<S, L + 1>(a, act, lds, lane, g, row, rr, ok);
    }
}

// The chain's LDS image: per layer dst[w_off + ((t_out * IT + t_in) * 4 + q) * 64 + lane] = W[k(t_in, lane >> 4, q)][16 t_out + (lane & 15)]
// (zero outside the layer), then the biases.  One workgroup of 256 threads.
__device__ __forceinline__ void rc_stage(const RcArgs& a, float* dst)
{
    // ---- stage the weights: LDS[((t_out * IT + t_in) * 4 + q) * 64 + lane] = W[k(t_in, lane >> 4, q)][16 t_out + (lane & 15)] ----
    for (int l = 0; l < a.n; ++l) {
        const RcLayer& y = a.l[l];
        const int it_n = y.ta + y.tb;
        const int ln = threadIdx.x & 63, q = threadIdx.x >> 6;  // one (k-step, lane) slot per thread and (t_in, t_out) pair
        const int i = ln & 15, gg = ln >> 4;
        // eight pairs per round: the eight loads are issued before the first LDS store
        const int npairs = it_n * y.ot;
        for (int base = 0; base < npairs; base += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int pr = base + u;
                v[u] = 0.f;
                if (pr < npairs) {
                    const int ti = pr / y.ot, to = pr - ti * y.ot;
                    int k;
                    bool kok;
                    if (ti < y.ta) { k = ti * 16 + 4 * gg + q; kok = k < y.ca; }
                    else { k = (ti - y.ta) * 16 + 4 * gg + q; kok = k < y.cb; k += y.ca; }
                    // wp[((cblk*KS + s)*64 + lane')*NTB + j] = W[s*4 + (lane' >> 4)][(cblk*NTB + j)*16 + (lane' & 15)]
                    const int cblk = to / y.ntb, j = to - cblk * y.ntb;
                    if (kok && to * 16 + i < y.cout) v[u] = y.wp[(((size_t)cblk * y.ks + (k >> 2)) * 64 + ((k & 3) * 16 + i)) * y.ntb + j];
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int pr = base + u;
                if (pr < npairs) {
                    const int ti = pr / y.ot, to = pr - ti * y.ot;
                    dst[y.w_off + ((to * it_n + ti) * 4 + q) * 64 + ln] = v[u];
                }
            }
        }
        for (int e = threadIdx.x; e < y.ot * 16; e += 256) dst[y.b_off + e] = e < y.cout ? y.bias[e] : 0.f;
    }
}

__global__ __launch_bounds__(256) void rc_pack_kernel(RcArgs a, float* out) { rc_stage(a, out); }

template <class S>
__global__ __launch_bounds__(256) void regchain_kernel(RcArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float rc_lds[];
    constexpr int MT = S::maxt();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int p = lane & 15, g = lane >> 4;
    if (a.image) {  // built once per network by rc_pack_kernel: a plain 16-byte copy
        for (int i = threadIdx.x; i < a.image_floats / 4; i += 256) reinterpret_cast<float4*>(rc_lds)[i] = reinterpret_cast<const float4*>(a.image)[i];
    } else {
        rc_stage(a, rc_lds);
    }
    __syncthreads();

    const bool v1 = (a.l[0].ca & 3) == 0 && (a.ld1 & 3) == 0 && (reinterpret_cast<uintptr_t>(a.x1) & 15) == 0;
    const bool v2 = (a.l[0].cb & 3) == 0 && (a.ld2 & 3) == 0 && (reinterpret_cast<uintptr_t>(a.x2) & 15) == 0;
    const int stride = gridDim.x * 4;
    // rows past the end are computed on a copy of the last row (loads never leave the buffers) and not stored
    auto load_inputs = [&](int tile, float (&dst)[MT][4]) {
        const int rr = min(tile * 16 + p, a.R - 1);
        const int s1 = a.g1 ? (a.g1m ? (rr / a.g1m) * a.g1n : 0) + a.g1[rr] : rr;
        rc_load<0, S::A0>(dst, a.x1 + (size_t)s1 * a.ld1, a.l[0].ca, v1, g);
        if constexpr (S::B[0] > 0) {
            const int s2 = a.g2 ? (a.g2m ? (rr / a.g2m) * a.g2n : 0) + a.g2[rr] : rr;
            rc_load<S::A0, S::B[0]>(dst, a.x2 + (size_t)s2 * a.ld2, a.l[0].cb, v2, g);
        }
    };
    int tile = blockIdx.x * 4 + wave;
    float act[MT][4], nxt[MT][4];
    if (tile * 16 < a.R) load_inputs(tile, nxt);
    for (; tile * 16 < a.R; tile += stride) {
#pragma unroll
        for (int t = 0; t < S::IN(0); ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) act[t][q] = nxt[t][q];
        if ((tile + stride) * 16 < a.R) load_inputs(tile + stride, nxt);  // in flight under this tile's MFMAs
        const int row = tile * 16 + p;
        rc_layer<S, 0>(a, act, rc_lds, lane, g, row, min(row, a.R - 1), row < a.R);
    }
}

static inline int rc_tiles(int c) { return (c + 15) / 16; }

// the chain shapes of the network (BraTS and Pancreas configurations): see the launch plan in randla.hip
using RcFc0 = RcShape<2, 1, 0, 1, 0, 1>;              // fc0 -> mlp1 (level 0): <=16 -> 8 -> 8
using RcEnc0 = RcShape<2, 1, 0, 1, 1, 2>;             // level 0: att2-mlp 16 -> 16, then [16 | shortcut 8] -> 32
using RcOne0 = RcShape<1, 1, 0, 1>;                   // level 0: att1-mlp 16 -> 8 (a single layer: no pre-product at d = 16)
using RcPair1 = RcShape<2, 2, 0, 2, 0, 4>;            // level 1: mlp1 32 -> 32, G = f . Wfc[:32] 32 -> 64
using RcPair1b = RcShape<2, 4, 0, 2, 0, 4>;           // level 1: att1-mlp 64 -> 32, G 32 -> 64
using RcEnc1 = RcShape<2, 4, 0, 4, 2, 8>;             // level 1: att2-mlp 64 -> 64, then [64 | shortcut 32] -> 128
using RcPair2 = RcShape<2, 8, 0, 4, 0, 8>;            // level 2: mlp1 / att1-mlp 128 -> 64, G = f . Wfc[:64] 64 -> 128 (64 KB of weights)
using RcHead = RcShape<4, 2, 2, 2, 0, 4, 0, 2, 0, 1>;  // [skip 32 | up 32] -> 32 -> 64 -> 32 -> classes

template <class S>
static bool rc_matches(const RcArgs& a)
{
    if (a.n != S::N) return false;
    for (int l = 0; l < S::N; ++l)
        if (a.l[l].ta != S::A(l) || a.l[l].tb != S::B[l] || a.l[l].ot != S::O[l]) return false;
    return true;
}

static bool rc_build(const ChainStep* steps, int n_steps, const RowSrc& s1, const RowSrc& s2, RcArgs& a, size_t& lds_bytes)
{
    if (n_steps < 1 || n_steps > kChainMaxSteps) return false;
    a = RcArgs{};
    a.x1 = s1.x; a.g1 = s1.gather; a.ld1 = s1.ld; a.g1m = s1.gm; a.g1n = s1.gn;
    a.x2 = s2.x; a.g2 = s2.gather; a.ld2 = s2.ld; a.g2m = s2.gm; a.g2n = s2.gn;
    a.n = n_steps;
    int off = 0, prev = 0;
    for (int i = 0; i < n_steps; ++i) {
        const PackedLinear* L = steps[i].L;
        if (!L || !L->wp || !L->bias) return false;
        if (steps[i].extra.x && (i == 0 || steps[i].extra.gather)) return false;
        RcLayer& y = a.l[i];
        y.wp = L->wp; y.bias = L->bias; y.y = steps[i].y; y.ldy = steps[i].ldy;
        y.ex = i > 0 ? steps[i].extra.x : nullptr; y.ldex = steps[i].extra.ld;
        y.ca = i == 0 ? s1.c : prev;
        y.cb = i == 0 ? s2.c : (steps[i].extra.x ? steps[i].extra.c : 0);
        if (y.ca + y.cb != L->cin) return false;
        y.ta = rc_tiles(y.ca); y.tb = rc_tiles(y.cb);
        y.cout = L->cout; y.ot = rc_tiles(L->cout);
        y.ks = L->ks; y.ntb = L->ntb; y.leaky = L->leaky;
        y.w_off = off; off += y.ot * (y.ta + y.tb) * 256;
        y.b_off = off; off += y.ot * 16;
        prev = L->cout;
    }
    lds_bytes = (size_t)off * sizeof(float);
    return rc_matches<RcFc0>(a) || rc_matches<RcEnc0>(a) || rc_matches<RcOne0>(a) || rc_matches<RcPair1>(a) || rc_matches<RcPair1b>(a) ||
           rc_matches<RcEnc1>(a) || rc_matches<RcPair2>(a) || rc_matches<RcHead>(a);
}

bool regchain_fits(const ChainStep* steps, int n_steps, const RowSrc& s1, const RowSrc& s2)
{
    RcArgs a;
    size_t lds = 0;
    return rc_build(steps, n_steps, s1, s2, a, lds);
}

template <class S>
static int rc_launch(ps_context* c, const RcArgs& a, size_t lds_bytes, int blocks)
{
    // dynamic LDS above the default limit: raised once per shape and device (a context is bound to one device)
    size_t& have = c->regchain_lds_attr[reinterpret_cast<const void*>(regchain_kernel<S>)];
    if (lds_bytes > 48 * 1024 && lds_bytes > have) {
        PS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(regchain_kernel<S>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        have = lds_bytes;
    }
    hipLaunchKernelGGL(regchain_kernel<S>, dim3(blocks), dim3(256), lds_bytes, c->stream, a);
    return PS_OK;
}

int regchain(ps_context* c, const ChainStep* steps, int n_steps, const RowSrc& s1, const RowSrc& s2, int64_t R, ChainCache* cache)
{
    if (R <= 0) return PS_OK;
    RcArgs a;
    size_t lds_bytes = 0;
    PS_CHECK(rc_build(steps, n_steps, s1, s2, a, lds_bytes), "regchain: not one of the compiled chain shapes");
    PS_CHECK(R < (int64_t)1 << 31, "regchain: too many rows");
    a.R = (int)R;
    if (cache) {
        // the re-ordered weight image depends only on the layers and the split of the first K axis: build it once per network
        ChainCache::Entry* hit = nullptr;
        for (auto& e : cache->entries)
            if (e.wp0 == a.l[0].wp && e.wp_last == a.l[n_steps - 1].wp && e.n == n_steps && e.ca == a.l[0].ca && e.cb == a.l[0].cb) hit = &e;
        if (!hit) {
            cache->entries.emplace_back();
            hit = &cache->entries.back();
            hit->wp0 = a.l[0].wp; hit->wp_last = a.l[n_steps - 1].wp; hit->n = n_steps; hit->ca = a.l[0].ca; hit->cb = a.l[0].cb;
            PS_TRY(hit->img.reserve(lds_bytes));
            hipLaunchKernelGGL(rc_pack_kernel, dim3(1), dim3(256), 0, c->stream, a, hit->img.as<float>());
            PS_HIP(hipGetLastError());
        }
        a.image = hit->img.as<float>();
        a.image_floats = (int)(lds_bytes / sizeof(float));
    }
    const int tiles = (int)((R + 15) / 16);
    const int per_cu = std::max(1, std::min(4, (int)(160 * 1024 / std::max<size_t>(lds_bytes, 1))));  // weights staged once per workgroup
    const int blocks = std::max(1, std::min((tiles + 3) / 4, 256 * per_cu));
    if (rc_matches<RcFc0>(a)) PS_TRY(rc_launch<RcFc0>(c, a, lds_bytes, blocks));
    else if (rc_matches<RcEnc0>(a)) PS_TRY(rc_launch<RcEnc0>(c, a, lds_bytes, blocks));
    else if (rc_matches<RcOne0>(a)) PS_TRY(rc_launch<RcOne0>(c, a, lds_bytes, blocks));
    else if (rc_matches<RcPair1>(a)) PS_TRY(rc_launch<RcPair1>(c, a, lds_bytes, blocks));
    else if (rc_matches<RcPair1b>(a)) PS_TRY(rc_launch<RcPair1b>(c, a, lds_bytes, blocks));
    else if (rc_matches<RcEnc1>(a)) PS_TRY(rc_launch<RcEnc1>(c, a, lds_bytes, blocks));
    else if (rc_matches<RcPair2>(a)) PS_TRY(rc_launch<RcPair2>(c, a, lds_bytes, blocks));
    else PS_TRY(rc_launch<RcHead>(c, a, lds_bytes, blocks));
    PS_HIP(hipGetLastError());
    return PS_OK;
}

}  // namespace ps
