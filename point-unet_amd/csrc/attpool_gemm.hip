// attpool_gemm.hip -- attentive pooling of the TRAINING step at the wide levels (d = 128 / 256 / 512: encoder levels 2-4), forward and
// backward, with the score tensor never in memory.
//
//   att_pooling (PointSegment/RandLANet.py:388-398):   s = F . Wfc   (F = [N*K, d] neighbour set, no bias)
//                                                      p = softmax over the K rows of a point, per channel
//                                                      agg[n, c] = sum_k p[n,k,c] * F[n,k,c]
//
// attpool_train.hip fuses this per POINT with the d x d weights resident in LDS -- which ends at d = 64 in fp32 (d = 128 as bfloat16):
// the wide levels ran op by op (score GEMM -> s [N*K, d] -> softmax-pool kernel; backward: softmax-pool backward -> dS, dF -> input
// gradient GEMM -> weight gradient GEMM: fourteen passes over [N*K, d] tensors per pooling).  Here the two row-side products ride on
// gemm_b3.hip's frame instead -- v_mfma_f32_32x32x16_bf16 over exact three-way bfloat16 splits (P = 3: fp32 results) or one plane of
// rounded operands (P = 1: the bf16-MLP mode), the weight planes streamed L2 -> LDS in 24 KB blocks shared by the four waves of a
// workgroup -- and everything between them stays in the accumulator registers of the wave that owns the rows:
//
//   a wave owns 32 rows = the K = 16 neighbour rows of TWO points, all d columns, 128 columns (a "panel") at a time
//   forward    S = F . W (panel)            accumulator tile: lane = column, registers = rows -> a point's 16 scores of a channel are
//              8 registers of two lanes: softmax = in-lane + ONE v_permlane32_swap; agg = sum p F   -> only agg [N, d] is written
//   backward   S recomputed the same way;   dS = p g (F - agg),  direct = p g   (g = dagg of the point)
//              dS^T through the matrix pipe: T = dS^T . I (one MFMA per plane and 16 rows with an identity operand) turns the tile into
//              lane = row, registers = columns -- which IS the row operand of the next product, register for register (its K axis is
//              simply taken in accumulator order: the W^T image is packed to match, gemm_b3.hip b3_pack_elem<KMAP>) -- and the layout
//              in which a row of dS leaves as 16-byte stores;
//              dF = direct + dS . W^T       accumulated over the panels in registers, seeded with the direct term
//              -> reads F and dagg, writes dF and dS; the weight gradient dW = F^T . dS is the step's ordinary wgrad product over
//                 (F, dS) (wgrad_b3_kernel), whose d x d accumulators per row slab no row-owning wave could hold at d >= 256.
// Passes over [N*K, d] per pooling: forward 1 (was 4), backward 3 + 2 for dW (was 9).
#include "common.h"

#include <type_traits>
#include "mfma_tile.h"
#include "b3_ops.h"
#include "attpool_train.h"
#include "reduce_partials.h"
#include "bf16_io.h"

namespace ps {

struct AttGArgs {
    const float* f; int ld;    // [rows, D] neighbour set, rows = points * 16
    const uint4* w1;           // W planes, natural K order: [D/16][D/32][P][64]
    const uint4* w2;           // W^T planes, accumulator K order (backward)
    const float* dagg;         // [points, D] (backward)
    float* agg;                // [points, D] (forward)
    float* df; int lddf;       // [rows, D] (backward)
    float* ds; int ldds;       // [rows, D] (backward)
    int64_t rows, points;
    int df_accum;              // df += instead of df =
    // split-source form (gather_neighbour + concat folded in, RandLANet.py:326-333): F = [fl[idx] | f] -- `f` / `df` then hold only the
    // right half ([rows, D/2]); the gathered half's gradient leaves as plain rows dfl_rows [rows, D/2] for the gather-reduction
    const float* fl; int ldl;  // [B * n_src, D/2]
    const int32_t* idx;        // [points, 16] cloud-local source rows
    int64_t n_src, n_q;        // rows per cloud of fl / points per cloud
    float* dfl_rows; int ld_rows;
    int f_bf16;                // forward, split form, bf16-MLP mode: the rows of `f` are STORED as bfloat16 (ps_set_train_act_bf16; ld in elements)
};

template <int N, class F>
__device__ __forceinline__ void attg_static_for(F&& f)
{
    if constexpr (N > 0) {
        attg_static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

__device__ __forceinline__ float attg_swap_max(float v)
{
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float attg_swap_sum(float v)
{
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// One stream of weight-plane blocks per workgroup: block = [2 chunks of 16 K][4 column tiles][P planes][64 lanes] uint4 (24 KB at P = 3),
// cut out of an image [K/16][D/32][P][64] at (chunk pair s, tile group g).  Sequence per panel p: the D/32 K-steps of the score product
// (image w1, group p), then per 32-column tile t of the panel and per group ig of four output tiles the blocks of dS . W^T (image w2,
// chunk pair 4 p + t, group ig).
// HALF (backward of d = 512 only): the launch produces the dF columns [256 HALF, 256 HALF + 256) -- NIG = NPAN / 2 groups of four output
// tiles per (panel, tile) instead of NPAN; -1 = all columns in one launch.
template <int D, int P, bool BWD, int HALF = -1>
struct AttGStream {
    static constexpr int NCB = D / 32, NPAN = D / 128, STEPS = D / 32;
    static constexpr int NIG = HALF < 0 ? NPAN : NPAN / 2, IG0 = HALF < 0 ? 0 : HALF * NIG;
    static constexpr int PER = BWD ? STEPS + 4 * NIG : STEPS;
    static constexpr int NB = NPAN * PER;
    struct Blk {
        const uint4* img;
        int s, g;
    };
    static __device__ __forceinline__ Blk blk(const AttGArgs& a, int n)
    {
        const int p = n / PER, m = n - p * PER;
        if (!BWD || m < STEPS) return Blk{a.w1, m, p};
        const int m2 = m - STEPS, t = m2 / NIG, ig = IG0 + (m2 - t * NIG);
        return Blk{a.w2, 4 * p + t, ig};
    }
    static __device__ __forceinline__ const uint4* src(const Blk& b, int i)  // i in [0, 512 P): (u, rest)
    {
        const int u = i / (256 * P), rest = i - u * (256 * P);
        return b.img + ((size_t)((2 * b.s + u) * NCB + 4 * b.g) * P) * 64 + rest;
    }
};

#define PS_ATTG_LOAD_B(n_)                                        \
    do {                                                          \
        const typename S::Blk nb_ = S::blk(a, (n_));              \
        breg0 = *S::src(nb_, 0 * 256 + (int)threadIdx.x);         \
        breg1 = *S::src(nb_, 1 * 256 + (int)threadIdx.x);         \
        if constexpr (P == 3) {                                   \
            breg2 = *S::src(nb_, 2 * 256 + (int)threadIdx.x);     \
            breg3 = *S::src(nb_, 3 * 256 + (int)threadIdx.x);     \
            breg4 = *S::src(nb_, 4 * 256 + (int)threadIdx.x);     \
            breg5 = *S::src(nb_, 5 * 256 + (int)threadIdx.x);     \
        }                                                         \
    } while (0)
#define PS_ATTG_STORE_B(buf_)                                     \
    do {                                                          \
        Bs[(buf_)][0 * 256 + threadIdx.x] = breg0;                \
        Bs[(buf_)][1 * 256 + threadIdx.x] = breg1;                \
        if constexpr (P == 3) {                                   \
            Bs[(buf_)][2 * 256 + threadIdx.x] = breg2;            \
            Bs[(buf_)][3 * 256 + threadIdx.x] = breg3;            \
            Bs[(buf_)][4 * 256 + threadIdx.x] = breg4;            \
            Bs[(buf_)][5 * 256 + threadIdx.x] = breg5;            \
        }                                                         \
    } while (0)

// Row addressing: every global access of a wave is (wave-uniform 64-bit base of its 32 rows) + (32-bit lane offset inside them), so no
// 64-bit address arithmetic sits in vector registers.  A wave whose rows end early (the last workgroup) computes on rows that exist --
// a missing second point reads the first one's rows, a wave without rows reads the first 32 -- and stores nothing for them.
struct AttGRows {
    int64_t rbase;   // first row the wave READS
    int nvalid;      // rows of the wave that exist: 0, 16 or 32 (stores)
    int second;      // row offset of the second point's reads: 16, or 0 when only one point is readable
};
__device__ __forceinline__ AttGRows attg_rows(const AttGArgs& a, int wave)
{
    const int64_t row0 = (int64_t)blockIdx.x * 128 + wave * 32;
    AttGRows r;
    const int64_t left = a.rows - row0;
    r.nvalid = left >= 32 ? 32 : (left >= 16 ? 16 : 0);
    r.rbase = r.nvalid > 0 ? row0 : 0;
    r.second = a.rows - r.rbase >= 32 ? 16 : 0;
    return r;
}

// the score product of one panel: acc[t] (t = 0..3) += F[rows of the wave][:] . W[:, 128 p + 32 t ..]; consumes blocks n .. n + STEPS - 1
// of the stream (block n is in Bs[n & 1] on entry, and block n + STEPS -- if any -- is in Bs[(n + STEPS) & 1] on exit)
#define PS_ATTG_SCORES()                                                                                              \
    do {                                                                                                              \
        load_a_half(0, 0);                                                                                            \
        load_a_half(0, 1);                                                                                            \
        _Pragma("unroll 1") for (int s = 0; s < S::STEPS; ++s, ++n)                                                   \
        {                                                                                                             \
            const int buf = n & 1;                                                                                    \
            const bool more = n + 1 < S::NB;                                                                          \
            _Pragma("unroll") for (int u = 0; u < 2; ++u)                                                             \
            {                                                                                                         \
                const BPlanes<P> ap = a_planes(u, s);                                                                  \
                __builtin_amdgcn_sched_barrier(0);                                                                    \
                if (s + 1 < S::STEPS) load_a_half(s + 1, u);                                                          \
                if (u == 1 && more) PS_ATTG_LOAD_B(n + 1);                                                            \
                __builtin_amdgcn_sched_barrier(0);                                                                    \
                _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                         \
                {                                                                                                     \
                    BPlanes<P> bp;                                                                                    \
                    _Pragma("unroll") for (int pl = 0; pl < P; ++pl) bp.p[pl] = Bs[buf][((u * 4 + t) * P + pl) * 64 + lane]; \
                    acc[t] = b3_mfma6<P>(ap, bp, acc[t]);                                                             \
                }                                                                                                     \
            }                                                                                                         \
            if (more) PS_ATTG_STORE_B(buf ^ 1);                                                                       \
            __syncthreads();                                                                                          \
        }                                                                                                             \
    } while (0)

// value rows of (tile column col, point pi) in accumulator order: element j = row 16 pi + 4 hl + (j & 3) + 8 (j >> 2) of the wave
#define PS_ATTG_VALUES(fv_, pi_)                                                                                      \
    do {                                                                                                              \
        if (SPLIT && col < D / 2) { /* gathered half: the rows' sources (vidx, loaded once per kernel), column col of each */ \
            const float* vb_ = flc[(pi_)] + col;                                                                      \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) (fv_)[j] = vb_[(unsigned)vidx[(pi_)][j] * (unsigned)a.ldl]; \
        } else if constexpr (FRB) { /* rows of bfloat16 (the pointer arithmetic in elements) */                       \
            const __bf16* vb_ = reinterpret_cast<const __bf16*>(a.f) + (size_t)rw.rbase * a.ld + ((pi_) ? rw.second * a.ld : 0) - D / 2; \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) (fv_)[j] = (float)vb_[voff + (unsigned)(((j & 3) + 8 * (j >> 2)) * a.ld)]; \
        } else {                                                                                                      \
            const float* vb_ = fb + ((pi_) ? rw.second * a.ld : 0) - (SPLIT ? D / 2 : 0);                             \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) (fv_)[j] = vb_[voff + (unsigned)(((j & 3) + 8 * (j >> 2)) * a.ld)]; \
        }                                                                                                             \
    } while (0)

// split-source form: the operand row of the lane (its gathered source row for the left half of K, its own row of `f` for the right half),
// and the sources of the value rows in accumulator order (two int4 per point, once per kernel)
#define PS_ATTG_SPLIT_SETUP()                                                                                         \
    const float* pL = nullptr;                                                                                        \
    const float* flc[2] = {nullptr, nullptr};                                                                         \
    int vidx[2][8] = {};                                                                                              \
    if constexpr (SPLIT) {                                                                                            \
        const int64_t row = rw.rbase + (c32 < 16 ? c32 : c32 - 16 + rw.second);                                       \
        pL = a.fl + (size_t)(((row >> 4) / a.n_q) * a.n_src + a.idx[row]) * a.ldl + 8 * hl;                           \
        _Pragma("unroll") for (int pi = 0; pi < 2; ++pi)                                                              \
        {                                                                                                             \
            const int64_t rb = rw.rbase + (pi ? rw.second : 0);                                                       \
            flc[pi] = a.fl + (size_t)(((rb >> 4) / a.n_q) * a.n_src) * a.ldl;                                         \
            const int4 g0 = *reinterpret_cast<const int4*>(a.idx + rb + 4 * hl), g1 = *reinterpret_cast<const int4*>(a.idx + rb + 8 + 4 * hl); \
            vidx[pi][0] = g0.x; vidx[pi][1] = g0.y; vidx[pi][2] = g0.z; vidx[pi][3] = g0.w;                           \
            vidx[pi][4] = g1.x; vidx[pi][5] = g1.y; vidx[pi][6] = g1.z; vidx[pi][7] = g1.w;                           \
        }                                                                                                             \
    }

// FRB (split form, one plane): the rows of `f` are bfloat16 -- eight of them ARE a lane's operand fragment (no split, no rounding: the
// sixteen bytes go to the matrix pipe as loaded)
template <int D, int P, bool SPLIT, bool FRB = false>
__global__ __launch_bounds__(256) void attg_fwd_kernel(AttGArgs a)
{
    static_assert(!FRB || (SPLIT && P == 1), "bfloat16 rows: split-source form of the bf16-MLP mode");
    using S = AttGStream<D, P, false>;
    __shared__ uint4 Bs[2][2 * 4 * P * 64];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int hl = lane >> 5, c32 = lane & 31;
    const AttGRows rw = attg_rows(a, wave);
    const float* fb = a.f + (size_t)rw.rbase * a.ld;  // wave-uniform
    const unsigned xoff = (unsigned)((c32 < 16 ? c32 : c32 - 16 + rw.second) * a.ld + 8 * hl);
    uint4 breg0, breg1, breg2, breg3, breg4, breg5;
    float4 areg[4];
    PS_ATTG_SPLIT_SETUP();
    auto load_a_half = [&](int s, int u) {
        if (SPLIT && 32 * s < D / 2) {  // (a K step lies in one half: D/2 is a multiple of 32)
            areg[2 * u] = *reinterpret_cast<const float4*>(pL + (32 * s + 16 * u));
            areg[2 * u + 1] = *reinterpret_cast<const float4*>(pL + (32 * s + 16 * u + 4));
        } else if constexpr (FRB) {  // (eight bfloat16 = the fragment itself; kept as loaded)
            const unsigned k = (unsigned)(32 * s + 16 * u - D / 2);
            areg[2 * u] = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned short*>(a.f) + (size_t)rw.rbase * a.ld + (xoff + k));
        } else {
            const unsigned k = (unsigned)(32 * s + 16 * u - (SPLIT ? D / 2 : 0));
            areg[2 * u] = *reinterpret_cast<const float4*>(fb + (xoff + k));
            areg[2 * u + 1] = *reinterpret_cast<const float4*>(fb + (xoff + k + 4));
        }
    };
    auto a_planes = [&](int u, int s) -> BPlanes<P> {
        if constexpr (FRB) {
            if (32 * s >= D / 2) {
                BPlanes<P> r;
                r.p[0] = __builtin_bit_cast(uint4, areg[2 * u]);
                return r;
            }
        }
        return b3_split8<P>(areg[2 * u], areg[2 * u + 1]);
    };
    PS_ATTG_LOAD_B(0);
    PS_ATTG_STORE_B(0);
    __syncthreads();
    int n = 0;
#pragma unroll 1
    for (int p = 0; p < S::NPAN; ++p) {
        f32x16 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        PS_ATTG_SCORES();
        // softmax over the 16 rows of each of the wave's two points, weighted sum of the value rows (the tile itself: L1 / L2 hits)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            __builtin_amdgcn_sched_barrier(0);  // (one tile's value loads at a time)
            const int col = 128 * p + 32 * t + c32;
            const unsigned voff = (unsigned)(4 * hl * a.ld + col);
#pragma unroll
            for (int pi = 0; pi < 2; ++pi) {
                float fv[8];
                PS_ATTG_VALUES(fv, pi);
                float m = acc[t][8 * pi];
#pragma unroll
                for (int j = 1; j < 8; ++j) m = fmaxf(m, acc[t][8 * pi + j]);
                m = attg_swap_max(m);
                float z = 0.f, num = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float e = __expf(acc[t][8 * pi + j] - m);
                    z += e;
                    num = __builtin_fmaf(e, fv[j], num);
                }
                z = attg_swap_sum(z);
                num = attg_swap_sum(num);
                if (hl == 0 && 16 * pi < rw.nvalid) a.agg[(size_t)((rw.rbase >> 4) + pi) * D + col] = num * __builtin_amdgcn_rcpf(z);
            }
        }
    }
}

// (d = 128: 64 + 64 accumulators leave room for two waves per SIMD -- asked for, the P = 1 form otherwise hoists its way past 256 registers;
//  d = 256 holds 64 + 128 accumulators: one wave per SIMD, the accumulators of dF in the second half of the register file)
//  d = 512 would hold 64 + 256: it runs as TWO launches, HALF = 0 / 1, each recomputing the scores and dS of every panel and accumulating
//  the dF columns of its half -- 64 + 128 accumulators like d = 256; dS is stored by the first)
template <int D, int P, bool SPLIT, int HALF = -1>
__global__ __launch_bounds__(256, D == 128 ? 2 : 1) void attg_bwd_kernel(AttGArgs a)
{
    static_assert(HALF < 0 || (D == 512 && !SPLIT), "the two-launch form is d = 512's");
    using S = AttGStream<D, P, true, HALF>;
    constexpr int NT = HALF < 0 ? D / 32 : D / 64, IT0 = HALF < 0 ? 0 : HALF * NT;  // dF tiles of this launch: IT0 .. IT0 + NT - 1
    __shared__ uint4 Bs[2][2 * 4 * P * 64];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int hl = lane >> 5, c32 = lane & 31;
    const AttGRows rw = attg_rows(a, wave);
    const float* fb = a.f + (size_t)rw.rbase * a.ld;  // wave-uniform
    const unsigned xoff = (unsigned)((c32 < 16 ? c32 : c32 - 16 + rw.second) * a.ld + 8 * hl);
    uint4 breg0, breg1, breg2, breg3, breg4, breg5;
    float4 areg[4];
    constexpr bool FRB = false;  // (bfloat16 rows: forward only)
    PS_ATTG_SPLIT_SETUP();
    auto load_a_half = [&](int s, int u) {
        if (SPLIT && 32 * s < D / 2) {  // (a K step lies in one half: D/2 is a multiple of 32)
            areg[2 * u] = *reinterpret_cast<const float4*>(pL + (32 * s + 16 * u));
            areg[2 * u + 1] = *reinterpret_cast<const float4*>(pL + (32 * s + 16 * u + 4));
        } else {
            const unsigned k = (unsigned)(32 * s + 16 * u - (SPLIT ? D / 2 : 0));
            areg[2 * u] = *reinterpret_cast<const float4*>(fb + (xoff + k));
            areg[2 * u + 1] = *reinterpret_cast<const float4*>(fb + (xoff + k + 4));
        }
    };
    // identity operand of the transposing product: k-slot (chunk kap, lane half hl, element j) of an accumulator tile is its row
    // 16 kap + 8 (j >> 2) + 4 hl + (j & 3); lane (n = c32, hl) of chunk kap holds 1.0 in the slot whose row is n (one lane half has it)
    uint4 ident[2];
    {
        const bool mine = ((c32 >> 2) & 1) == hl;
        const int j = (c32 & 3) + 4 * ((c32 >> 3) & 1), kap = c32 >> 4;
        const unsigned one = 0x3F80u << (16 * (j & 1));
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const bool on = mine && kap == k;
            ident[k].x = (on && (j >> 1) == 0) ? one : 0u;
            ident[k].y = (on && (j >> 1) == 1) ? one : 0u;
            ident[k].z = (on && (j >> 1) == 2) ? one : 0u;
            ident[k].w = (on && (j >> 1) == 3) ? one : 0u;
        }
    }
    f32x16 acc2[NT];  // dF of the wave's rows: tile it = columns 32 it .. (lane = column, registers = rows)
#pragma unroll
    for (int it = 0; it < NT; ++it)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[it][r] = 0.f;

    auto a_planes = [&](int u, int) -> BPlanes<P> { return b3_split8<P>(areg[2 * u], areg[2 * u + 1]); };
    PS_ATTG_LOAD_B(0);
    PS_ATTG_STORE_B(0);
    __syncthreads();
    int n = 0;
    // (the panels through a compile-time index: `#pragma unroll` alone left the four-panel loop of d = 512 rolled at P = 3, and a run-time
    //  index into acc2 puts the dF accumulators into scratch memory)
    auto panel = [&](auto pc) __attribute__((always_inline)) {
        constexpr int p = decltype(pc)::value;
        f32x16 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        PS_ATTG_SCORES();
        // ---- softmax, direct term p g into the dF accumulators, dS = p g (F - agg) over the scores in place ----
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            __builtin_amdgcn_sched_barrier(0);  // (one tile's value loads at a time)
            const int col = 128 * p + 32 * t + c32;
            const unsigned voff = (unsigned)(4 * hl * a.ld + col);
#pragma unroll
            for (int pi = 0; pi < 2; ++pi) {
                float fv[8];
                PS_ATTG_VALUES(fv, pi);
                const float g = 16 * pi < rw.nvalid ? a.dagg[(size_t)((rw.rbase >> 4) + pi) * D + col] : 0.f;
                float m = acc[t][8 * pi];
#pragma unroll
                for (int j = 1; j < 8; ++j) m = fmaxf(m, acc[t][8 * pi + j]);
                m = attg_swap_max(m);
                float e[8], z = 0.f, num = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    e[j] = __expf(acc[t][8 * pi + j] - m);
                    z += e[j];
                    num = __builtin_fmaf(e[j], fv[j], num);
                }
                z = attg_swap_sum(z);
                num = attg_swap_sum(num);
                const float inv = __builtin_amdgcn_rcpf(z), agg = num * inv, ginv = g * inv;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float pg = e[j] * ginv;
                    if constexpr (4 * p >= IT0 && 4 * p < IT0 + NT) acc2[4 * p + t - IT0][8 * pi + j] += pg;
                    acc[t][8 * pi + j] = pg * (fv[j] - agg);
                }
            }
        }
        // ---- per 32-column tile of the panel: transpose dS on the matrix pipe, store its rows, dF += dS . W^T ----
        float* dsb = a.ds + (size_t)rw.rbase * a.ldds;  // wave-uniform
        const unsigned dsoff = (unsigned)(c32 * a.ldds + 128 * p + 4 * hl);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            __builtin_amdgcn_sched_barrier(0);
            f32x16 T;
#pragma unroll
            for (int r = 0; r < 16; ++r) T[r] = 0.f;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const BPlanes<P> dp = b3_split8<P>(make_float4(acc[t][8 * k], acc[t][8 * k + 1], acc[t][8 * k + 2], acc[t][8 * k + 3]),
                                                   make_float4(acc[t][8 * k + 4], acc[t][8 * k + 5], acc[t][8 * k + 6], acc[t][8 * k + 7]));
#pragma unroll
                for (int pl = P - 1; pl >= 0; --pl) T = b3_mfma(dp.p[pl], ident[k], T);  // (smallest pieces first: every partial sum is exact)
            }
            // T: lane = row c32 of the wave, register r = column 128 p + 32 t + (r & 3) + 8 (r >> 2) + 4 hl
            if (HALF <= 0 && c32 < rw.nvalid) {  // (two-launch form: the first launch stores dS)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4)
                    *reinterpret_cast<float4*>(dsb + (dsoff + (unsigned)(32 * t + 8 * g4))) = make_float4(T[4 * g4], T[4 * g4 + 1], T[4 * g4 + 2], T[4 * g4 + 3]);
            }
            BPlanes<P> tp[2];
            tp[0] = b3_split8<P>(make_float4(T[0], T[1], T[2], T[3]), make_float4(T[4], T[5], T[6], T[7]));
            tp[1] = b3_split8<P>(make_float4(T[8], T[9], T[10], T[11]), make_float4(T[12], T[13], T[14], T[15]));
#pragma unroll
            for (int ig = 0; ig < S::NIG; ++ig, ++n) {
                const int buf = n & 1;
                const bool more = n + 1 < S::NB;
                if (more) PS_ATTG_LOAD_B(n + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int i4 = 0; i4 < 4; ++i4) {
                        BPlanes<P> bp;
#pragma unroll
                        for (int pl = 0; pl < P; ++pl) bp.p[pl] = Bs[buf][((u * 4 + i4) * P + pl) * 64 + lane];
                        acc2[4 * ig + i4] = b3_mfma6<P>(tp[u], bp, acc2[4 * ig + i4]);
                    }
                if (more) PS_ATTG_STORE_B(buf ^ 1);
                __syncthreads();
            }
        }
    };
    attg_static_for<S::NPAN>(panel);
    // ---- dF: register r of tile it = row (r & 3) + 8 (r >> 2) + 4 hl of the wave, column 32 it + c32 ----
#pragma unroll
    for (int it = 0; it < NT; ++it) {
        // split form: the gathered half (tiles below D/2) leaves as plain rows for the gather-reduction, the right half goes to df
        const bool left = SPLIT && 32 * it < D / 2;
        float* dfb = left ? a.dfl_rows + (size_t)rw.rbase * a.ld_rows : a.df + (size_t)rw.rbase * a.lddf;  // wave-uniform
        const int pitch = left ? a.ld_rows : a.lddf;
        const bool accum = !left && a.df_accum != 0;
        const unsigned doff = (unsigned)(4 * hl * pitch + 32 * (IT0 + it) + c32 - (SPLIT && !left ? D / 2 : 0));
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (16 * half < rw.nvalid) {  // (wave-uniform: registers 8 half .. are the rows of point `half`)
                float old[8];
                if (accum) {  // (the reads of a tile in flight before its first store)
#pragma unroll
                    for (int q = 0; q < 8; ++q) old[q] = dfb[doff + (unsigned)((16 * half + (q & 3) + 8 * (q >> 2)) * pitch)];
                }
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    dfb[doff + (unsigned)((16 * half + (q & 3) + 8 * (q >> 2)) * pitch)] = accum ? acc2[it][8 * half + q] + old[q] : acc2[it][8 * half + q];
            }
        }
    }
}
#undef PS_ATTG_VALUES
#undef PS_ATTG_SPLIT_SETUP
#undef PS_ATTG_SCORES
#undef PS_ATTG_LOAD_B
#undef PS_ATTG_STORE_B

__global__ __launch_bounds__(256) void attg_pack_kernel(PackJob j)
{
    // (one matrix, outside a recorded step: the batched kernel's element function through a by-value job)
    PackJob jj = j;
    for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < jj.total; i += (int64_t)gridDim.x * 256) b3_pack_elem_any(jj, i);
}

static int attg_planes(ps_context* c, const float* w, int64_t d, bool transposed_kmap, const uint4** out)
{
    PackJob key = {};
    key.w = w;
    key.sk = transposed_kmap ? 1 : d;
    key.sn = transposed_kmap ? d : 1;
    key.kind = (c->train_bf16 ? 4 : 3) + (transposed_kmap ? 2 : 0);
    key.cin = (int)d; key.cout = (int)d;
    key.total = (int64_t)(d / 16) * (d / 32) * 64;
    bool launch = true;
    void* planes = pack_slot(c, key, (size_t)d * d * 6, launch);
    PS_CHECK(planes != nullptr, "att_pool_gemm: out of device memory for the weight planes");
    if (launch) {
        key.out = planes;
        hipLaunchKernelGGL(attg_pack_kernel, dim3(ceil_div(key.total, 256)), dim3(256), 0, c->stream, key);
        PS_HIP(hipGetLastError());
    }
    *out = static_cast<const uint4*>(planes);
    return PS_OK;
}

template <int D, bool BWD, bool SPLIT>
static int launch_attg_s(ps_context* c, const AttGArgs& a)
{
    const unsigned blocks = (unsigned)((a.rows + 127) / 128);
    if constexpr (!BWD) {
        if constexpr (SPLIT && D == 128) {
            if (a.f_bf16) {
                PS_CHECK(c->train_bf16, "att_pool_gemm: bfloat16 rows belong to the bf16-MLP mode");
                hipLaunchKernelGGL((attg_fwd_kernel<D, 1, true, true>), dim3(blocks), dim3(256), 0, c->stream, a);
                PS_HIP(hipGetLastError());
                return PS_OK;
            }
        }
        PS_CHECK(!a.f_bf16, "att_pool_gemm: bfloat16 rows are taken by the split-source forward at d = 128 only");
        if (c->train_bf16) hipLaunchKernelGGL((attg_fwd_kernel<D, 1, SPLIT>), dim3(blocks), dim3(256), 0, c->stream, a);
        else hipLaunchKernelGGL((attg_fwd_kernel<D, 3, SPLIT>), dim3(blocks), dim3(256), 0, c->stream, a);
    } else if constexpr (D == 512) {
        // 64 + 256 accumulators do not fit a wave: two launches over halves of the dF columns (attg_bwd_kernel, HALF)
        if (c->train_bf16) {
            hipLaunchKernelGGL((attg_bwd_kernel<D, 1, false, 0>), dim3(blocks), dim3(256), 0, c->stream, a);
            hipLaunchKernelGGL((attg_bwd_kernel<D, 1, false, 1>), dim3(blocks), dim3(256), 0, c->stream, a);
        } else {
            hipLaunchKernelGGL((attg_bwd_kernel<D, 3, false, 0>), dim3(blocks), dim3(256), 0, c->stream, a);
            hipLaunchKernelGGL((attg_bwd_kernel<D, 3, false, 1>), dim3(blocks), dim3(256), 0, c->stream, a);
        }
    } else {
        if (c->train_bf16) hipLaunchKernelGGL((attg_bwd_kernel<D, 1, SPLIT>), dim3(blocks), dim3(256), 0, c->stream, a);
        else hipLaunchKernelGGL((attg_bwd_kernel<D, 3, SPLIT>), dim3(blocks), dim3(256), 0, c->stream, a);
    }
    PS_HIP(hipGetLastError());
    return PS_OK;
}
template <int D, bool BWD>
static int launch_attg(ps_context* c, const AttGArgs& a)
{
    if constexpr (D == 512) return launch_attg_s<D, BWD, false>(c, a);  // (forward only, materialised form only)
    else return a.fl ? launch_attg_s<D, BWD, true>(c, a) : launch_attg_s<D, BWD, false>(c, a);
}

// ---- d = 64 (encoder level 1) -----------------------------------------------------------------------------------------------------------
// 5.76 M neighbour rows per batch of 8 clouds: the level where the [N*K, d] tensors are largest and where attpool_train.hip's per-point
// kernels (fp32 MFMA 16x16x4, a wave per point) were bound by the matrix pipe: 0.63 ms forward / 1.71 ms backward per pooling for 0.74 GB
// read / 2.9 GB moved.  Same computation as the wide-level kernels above with what d = 64 allows on top:
//   * both weight images (W planes for the scores, W^T planes in accumulator K order for dF: 24 KB each at P = 3) stay in LDS for the
//     whole kernel -- no stream, no workgroup barrier: the four waves of a workgroup walk their 32-row tiles independently (persistent grid);
//   * the weight gradient dW = F^T . dS (a third product, contraction over the 32 rows of the tile) accumulates in 64 registers per
//     wave: its row operand is the tile's value rows in exactly the register order the softmax epilogue loads them in (lane = column,
//     k-slots = rows), its column operand the planes of dS the transposing product consumes anyway -- per-wave partials, summed by
//     reduce_partials_kernel in a fixed order (deterministic: the grid does not depend on the data);
//   * split-source form (RandLANet.py:326-333: F = [gather(f, idx) | f_xyz]): the gathered half is read through idx (row operand: the
//     lane's own row; value rows: two int4 index loads per point), its gradient leaves as plain rows for the gather-reduction.
//   * per tile the only global reads are the 32 operand rows (the gathered half through one index per lane) and dagg: the rows are
//     requested ONE TILE AHEAD (the index two tiles ahead) and go to a per-wave LDS tile as they are split, which is where the softmax
//     epilogue and the weight-gradient product read their column-wise "value rows" from (the first form read them from global memory a
//     second time, behind the scores: 6.4 us per tile of exposed latency, 0.56 ms per forward pooling).
template <int P, bool BWD, int WAVES, bool FRB>  // FRB: the f_xyz half is stored as bfloat16 (compile-time: a run-time branch around the
                                                 // prefetch made the compiler wait for every load in flight at the join)
__global__ __launch_bounds__(WAVES * 64) void attg64_kernel(AttTrainArgs a, const uint4* __restrict__ w1, const uint4* __restrict__ w2, float* __restrict__ dw_part)
{
    constexpr int NQ = 4, NT = 2, IMG = NQ * NT * P * 64;  // uint4 per weight image
    constexpr int PITCH = 68;                               // floats per row of the value tile: 16-byte rows, the two lane halves on disjoint banks
    extern __shared__ __attribute__((aligned(16))) uint4 Wl[];
    for (int i = threadIdx.x; i < IMG; i += WAVES * 64) {
        Wl[i] = w1[i];
        if constexpr (BWD) Wl[IMG + i] = w2[i];
    }
    __syncthreads();
    const uint4* W1 = Wl;
    const uint4* W2 = Wl + IMG;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int hl = lane >> 5, c32 = lane & 31;
    float* V = reinterpret_cast<float*>(Wl + (BWD ? 2 : 1) * IMG) + wave * (32 * PITCH);
    const bool split = a.fl != nullptr;
    uint4 ident[2];
    {
        const bool mine = ((c32 >> 2) & 1) == hl;
        const int j = (c32 & 3) + 4 * ((c32 >> 3) & 1), kap = c32 >> 4;
        const unsigned one = 0x3F80u << (16 * (j & 1));
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const bool on = mine && kap == k;
            ident[k].x = (on && (j >> 1) == 0) ? one : 0u;
            ident[k].y = (on && (j >> 1) == 1) ? one : 0u;
            ident[k].z = (on && (j >> 1) == 2) ? one : 0u;
            ident[k].w = (on && (j >> 1) == 3) ? one : 0u;
        }
    }
    f32x16 acc3[2][2];  // dW of this wave: tile (it, ct) = rows 32 it .., columns 32 ct .. (lane = column, registers = rows)
    if constexpr (BWD) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc3[i][j][r] = 0.f;
    }
    const int64_t rows = a.R * 16;
    const int64_t ntiles = (rows + 31) >> 5;
    const int64_t stride = (int64_t)gridDim.x * WAVES;
    // the operand row of this lane in tile `tile` (rows is a multiple of 16: the last tile may hold ONE point, whose rows are then read twice)
    auto my_row = [&](int64_t tile) {
        const int64_t row0 = tile << 5;
        [[maybe_unused]] const int second = rows - row0 >= 32 ? 16 : 0;
        return row0 + (c32 < 16 ? c32 : c32 - 16 + second);
    };
    float4 nrow[8];   // the NEXT tile's operand row of this lane: chunks q = 0..3 x two float4
    int nidx = 0;     // split form: the lane's gathered row of the tile after that
    auto fetch_idx = [&](int64_t tile) { if (split && tile < ntiles) nidx = a.idx[my_row(tile)]; };
    auto fetch_row = [&](int64_t tile) {
        if (tile >= ntiles) return;
        const int64_t row = my_row(tile);
        const float *pL, *pR;
        if (split) {
            pL = a.fl + (size_t)(((row >> 4) / a.n_q) * a.n_src + nidx) * a.ldl + 8 * hl;
            pR = a.f + (size_t)row * a.ld + 8 * hl;
        } else {
            pL = a.f + (size_t)row * a.ld + 8 * hl;
            pR = pL + 32;
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (FRB && q >= 2) {  // the f_xyz half stored as bfloat16 (split form only): one 16-byte load for the lane's eight values
                // (kept as loaded: converting here would make the wave wait for its own prefetch; expanded where the chunk is consumed)
                const uint4 u = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(a.f) + ((size_t)row * a.ld + 8 * hl + 16 * (q - 2)));
                nrow[2 * q] = make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
                continue;
            }
            const float* src = q < 2 ? pL + 16 * q : pR + 16 * (q - 2);
            nrow[2 * q] = *reinterpret_cast<const float4*>(src);
            nrow[2 * q + 1] = *reinterpret_cast<const float4*>(src + 4);
        }
    };
    int64_t tile = (int64_t)blockIdx.x * WAVES + wave;
    fetch_idx(tile);
    fetch_row(tile);
    fetch_idx(tile + stride);
#pragma unroll 1
    for (; tile < ntiles; tile += stride) {
        const int64_t row0 = tile << 5;
        const int nvalid = rows - row0 >= 32 ? 32 : 16;
        [[maybe_unused]] const int second = nvalid == 32 ? 16 : 0;
        float gd[2][2];  // dagg of (tile column block t, point pi): requested before the products
        if constexpr (BWD) {
#pragma unroll
            for (int pi = 0; pi < 2; ++pi)
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    gd[t][pi] = 16 * pi < nvalid ? a.dagg[(size_t)((row0 >> 4) + pi) * 64 + 32 * t + c32] : 0.f;
        }
        // ---- scores; the rows go to the value tile as they are split ----
        f32x16 acc[2], acc2[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[t][r] = 0.f;
                acc2[t][r] = 0.f;
            }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            float4 x0 = nrow[2 * q], x1 = nrow[2 * q + 1];
            if (FRB && q >= 2) {
                x1 = unpack_bf16x4(make_uint2(__float_as_uint(x0.z), __float_as_uint(x0.w)));
                x0 = unpack_bf16x4(make_uint2(__float_as_uint(x0.x), __float_as_uint(x0.y)));
            }
            *reinterpret_cast<float4*>(V + c32 * PITCH + 16 * q + 8 * hl) = x0;
            *reinterpret_cast<float4*>(V + c32 * PITCH + 16 * q + 8 * hl + 4) = x1;
            const BPlanes<P> ap = b3_split8<P>(x0, x1);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                BPlanes<P> bp;
#pragma unroll
                for (int pl = 0; pl < P; ++pl) bp.p[pl] = W1[((q * NT + t) * P + pl) * 64 + lane];
                acc[t] = b3_mfma6<P>(ap, bp, acc[t]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        fetch_row(tile + stride);       // (uses the index fetched one iteration ago)
        fetch_idx(tile + 2 * stride);
        __builtin_amdgcn_sched_barrier(0);
        wave_lds_sync();
        // ---- per point: value rows (LDS), softmax, (backward) dS, direct term, dW ----
#pragma unroll
        for (int pi = 0; pi < 2; ++pi) {
            const bool pvalid = 16 * pi < nvalid;
            float fv[2][8];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int j = 0; j < 8; ++j) fv[t][j] = V[(16 * pi + 4 * hl + (j & 3) + 8 * (j >> 2)) * PITCH + 32 * t + c32];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float m = acc[t][8 * pi];
#pragma unroll
                for (int j = 1; j < 8; ++j) m = fmaxf(m, acc[t][8 * pi + j]);
                m = attg_swap_max(m);
                float e[8], z = 0.f, num = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    e[j] = __expf(acc[t][8 * pi + j] - m);
                    z += e[j];
                    num = __builtin_fmaf(e[j], fv[t][j], num);
                }
                z = attg_swap_sum(z);
                num = attg_swap_sum(num);
                const float inv = __builtin_amdgcn_rcpf(z), agg = num * inv;
                if constexpr (!BWD) {
                    if (hl == 0 && pvalid) a.agg[(size_t)((row0 >> 4) + pi) * 64 + 32 * t + c32] = agg;
                } else {
                    const float ginv = gd[t][pi] * inv;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float pg = e[j] * ginv;
                        acc2[t][8 * pi + j] = pg;
                        acc[t][8 * pi + j] = pg * (fv[t][j] - agg);
                    }
                }
            }
            if constexpr (BWD) {
                BPlanes<P> fvp[2], dp[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    fvp[t] = b3_split8<P>(make_float4(fv[t][0], fv[t][1], fv[t][2], fv[t][3]), make_float4(fv[t][4], fv[t][5], fv[t][6], fv[t][7]));
                    dp[t] = b3_split8<P>(make_float4(acc[t][8 * pi], acc[t][8 * pi + 1], acc[t][8 * pi + 2], acc[t][8 * pi + 3]),
                                         make_float4(acc[t][8 * pi + 4], acc[t][8 * pi + 5], acc[t][8 * pi + 6], acc[t][8 * pi + 7]));
                }
#pragma unroll
                for (int it = 0; it < 2; ++it)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) acc3[it][ct] = b3_mfma6<P>(fvp[it], dp[ct], acc3[it][ct]);
            }
        }
        wave_lds_sync();  // (the value tile is rewritten by the next iteration)
        if constexpr (BWD) {
            // ---- dF = direct + dS . W^T: the transposed tile of dS (lane = row, registers = columns) is the row operand, register for register ----
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                __builtin_amdgcn_sched_barrier(0);
                // (the planes of dS a second time rather than a transposed tile kept alive through the loop over the points: 16 registers
                //  for 44 VALU instructions)
                f32x16 T;
#pragma unroll
                for (int r = 0; r < 16; ++r) T[r] = 0.f;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const BPlanes<P> dp = b3_split8<P>(make_float4(acc[ct][8 * k], acc[ct][8 * k + 1], acc[ct][8 * k + 2], acc[ct][8 * k + 3]),
                                                       make_float4(acc[ct][8 * k + 4], acc[ct][8 * k + 5], acc[ct][8 * k + 6], acc[ct][8 * k + 7]));
#pragma unroll
                    for (int pl = P - 1; pl >= 0; --pl) T = b3_mfma(dp.p[pl], ident[k], T);
                }
                BPlanes<P> tp[2];
                tp[0] = b3_split8<P>(make_float4(T[0], T[1], T[2], T[3]), make_float4(T[4], T[5], T[6], T[7]));
                tp[1] = b3_split8<P>(make_float4(T[8], T[9], T[10], T[11]), make_float4(T[12], T[13], T[14], T[15]));
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int it = 0; it < 2; ++it) {
                        BPlanes<P> bp;
#pragma unroll
                        for (int pl = 0; pl < P; ++pl) bp.p[pl] = W2[(((2 * ct + u) * NT + it) * P + pl) * 64 + lane];
                        acc2[it] = b3_mfma6<P>(tp[u], bp, acc2[it]);
                    }
            }
            // ---- stores: register 8 half + q of tile it = row 16 half + 4 hl + (q & 3) + 8 (q >> 2) of the wave, column 32 it + c32 ----
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                float* base;
                int pitch;
                bool accum = false;
                if (!split) {
                    base = a.df + (size_t)row0 * a.lddf + 32 * it + c32;
                    pitch = a.lddf;
                } else if (it == 0) {
                    base = a.dfl_rows + (size_t)row0 * a.ld_rows + c32;
                    pitch = a.ld_rows;
                } else {
                    base = a.df + (size_t)row0 * a.lddf + c32;
                    pitch = a.lddf;
                    accum = a.df_accum != 0;
                }
                if (FRB && split) {
                    // the f_xyz half's gradient in the format of the f_xyz rows (ps_set_train_act_bf16): bfloat16, lddf in elements.  A lane owns
                    // a column; neighbouring lanes swap every second row (one DPP move each), so the even lane holds columns (c, c + 1) of row q
                    // and the odd lane those of row q + 1: 4-byte stores of packed pairs (2-byte stores: 1.07 against 0.53 ms per launch)
                    // (both halves: it == 0 is the gathered half's gradient, plain rows for the gather-reduction -- same format, never accumulated)
                    unsigned* b32 = it == 0 ? reinterpret_cast<unsigned*>(reinterpret_cast<__bf16*>(a.dfl_rows) + (size_t)row0 * a.ld_rows + (c32 & ~1))
                                            : reinterpret_cast<unsigned*>(reinterpret_cast<__bf16*>(a.df) + (size_t)row0 * a.lddf + (c32 & ~1));
                    const bool odd = (c32 & 1) != 0;
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        if (16 * half < nvalid) {
                            unsigned old[4];
                            if (accum) {
#pragma unroll
                                for (int q = 0; q < 8; q += 2)
                                    old[q >> 1] = b32[(unsigned)((16 * half + 4 * hl + ((q + (odd ? 1 : 0)) & 3) + 8 * (q >> 2)) * pitch) >> 1];
                            }
#pragma unroll
                            for (int q = 0; q < 8; q += 2) {
                                const float v0 = acc2[it][8 * half + q], v1 = acc2[it][8 * half + q + 1];
                                const float n0 = __shfl_xor(v0, 1), n1 = __shfl_xor(v1, 1);
                                float lo = odd ? n1 : v0, hi = odd ? v1 : n0;  // (row q + odd: columns c32 & ~1, + 1)
                                if (accum) {
                                    lo += __uint_as_float(old[q >> 1] << 16);
                                    hi += __uint_as_float(old[q >> 1] & 0xffff0000u);
                                }
                                b32[(unsigned)((16 * half + 4 * hl + ((q + (odd ? 1 : 0)) & 3) + 8 * (q >> 2)) * pitch) >> 1] = pack_bf16(lo, hi);
                            }
                        }
                    }
                    continue;
                }
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    if (16 * half < nvalid) {
                        float old[8];
                        if (accum) {
#pragma unroll
                            for (int q = 0; q < 8; ++q) old[q] = base[(unsigned)((16 * half + 4 * hl + (q & 3) + 8 * (q >> 2)) * pitch)];
                        }
#pragma unroll
                        for (int q = 0; q < 8; ++q)
                            base[(unsigned)((16 * half + 4 * hl + (q & 3) + 8 * (q >> 2)) * pitch)] = accum ? acc2[it][8 * half + q] + old[q] : acc2[it][8 * half + q];
                    }
                }
            }
        }
    }
    if constexpr (BWD) {
        float* out = dw_part + (size_t)(blockIdx.x * WAVES + wave) * 4096;
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) out[(32 * it + (r & 3) + 8 * (r >> 2) + 4 * hl) * 64 + 32 * ct + c32] = acc3[it][ct][r];
    }
}

bool att64_gemm_fits(const Tuning& tn, const AttTrainArgs& a, bool backward)
{
    if (!tn.att64_gemm) return false;  // (A/B knob)
    auto al = [](const void* q, int ld) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0 && ld % 4 == 0; };
    if (!al(a.f, a.ld)) return false;
    if (a.fr_bf16 && (!a.fl || a.ld % 8 != 0 || !a.bf16)) return false;  // (bfloat16 rows: split form, 16-byte loads of eight values, bf16-MLP mode)
    if (a.fl) {
        if (!al(a.fl, a.ldl) || (reinterpret_cast<uintptr_t>(a.idx) & 15) != 0 || a.n_q <= 0) return false;
        if (backward && !a.dfl_rows) return false;  // (the float-atomic scatter form: attpool_train.hip)
    }
    return a.R > 0 && a.R * 16 < (1ll << 31);
}

template <int P, bool BWD, int WAVES, bool FRB>
static int launch_attg64_f(ps_context* c, const AttTrainArgs& a, const uint4* w1, const uint4* w2, float* dW)
{
    const size_t smem = (size_t)4 * 2 * P * 64 * 16 * (BWD ? 2 : 1) + sizeof(float) * WAVES * 32 * 68;
    auto kern = attg64_kernel<P, BWD, WAVES, FRB>;
    if (smem > 48 * 1024) PS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    const int per_cu = std::max(1, (int)(160 * 1024 / smem));
    const int64_t tiles = (a.R * 16 + 31) / 32;
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((tiles + WAVES - 1) / WAVES, 256 * per_cu));
    float* part = nullptr;
    if (BWD) {
        PS_TRY(c->red_ws.reserve(sizeof(float) * (size_t)blocks * WAVES * 4096 + 256));
        part = c->red_ws.as<float>();
    }
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(WAVES * 64), smem, c->stream, a, w1, w2, part);
    if (BWD) hipLaunchKernelGGL(reduce_partials_kernel<float>, dim3(ceil_div(4096, 16)), dim3(256), 0, c->stream, static_cast<const float*>(part), blocks * WAVES, 4096, dW);
    PS_HIP(hipGetLastError());
    return PS_OK;
}

template <int P, bool BWD, int WAVES>
static int launch_attg64(ps_context* c, const AttTrainArgs& a, const uint4* w1, const uint4* w2, float* dW)
{
    if constexpr (P == 1) {  // (bfloat16 rows exist in the bf16-MLP mode only)
        if (a.fr_bf16) return launch_attg64_f<P, BWD, WAVES, true>(c, a, w1, w2, dW);
    }
    return launch_attg64_f<P, BWD, WAVES, false>(c, a, w1, w2, dW);
}

int att64_gemm(ps_context* c, AttTrainArgs a, bool backward, float* dW)
{
    const uint4 *w1 = nullptr, *w2 = nullptr;
    PS_TRY(attg_planes(c, a.w, 64, false, &w1));
    if (backward) PS_TRY(attg_planes(c, a.w, 64, true, &w2));
    // forward: twelve waves per workgroup (weights 24 KB + 8.5 KB of value tile per wave: three waves per SIMD); backward: 160 accumulator
    // registers + the prefetched rows -- four waves per workgroup (one per SIMD, no scratch) or eight (two per SIMD with 23 / 57 registers
    // of scratch at P = 1 / 3).  Measured on MI355X, 8 x 45 000 points (profiles/tools/exp_att64.py): fp32 1.349 ms with four, 1.443 with
    // eight; bf16 0.858 with four, 0.713 with eight (attpool_train.hip's kernels: 1.720 / 1.123).  PS_ATT64_OCC = 1 | 2 overrides.
    const int occ_env = c->tune.att64_occ;
    if (c->train_bf16) {
        if (!backward) return launch_attg64<1, false, 12>(c, a, w1, w2, dW);
        return occ_env == 1 ? launch_attg64<1, true, 4>(c, a, w1, w2, dW) : launch_attg64<1, true, 8>(c, a, w1, w2, dW);
    }
    if (!backward) return launch_attg64<3, false, 12>(c, a, w1, w2, dW);
    return occ_env == 2 ? launch_attg64<3, true, 8>(c, a, w1, w2, dW) : launch_attg64<3, true, 4>(c, a, w1, w2, dW);
}

static bool attg_ok(int64_t K, int64_t d, const void* f, int64_t ld)
{
    return K == 16 && (d == 128 || d == 256 || d == 512) && ld % 4 == 0 && ld >= d && (reinterpret_cast<uintptr_t>(f) & 15) == 0;
}

}  // namespace ps

using namespace ps;

extern "C" int ps_op_att_pool_gemm_supported(int64_t K, int64_t d) { return K == 16 && (d == 128 || d == 256 || d == 512) ? 1 : 0; }

extern "C" int ps_op_att_pool_gemm_fwd(ps_context* c, const float* fset, int64_t ld, const float* wfc, int64_t R, int64_t K, int64_t d, float* agg)
{
    PS_CHECK(c && fset && wfc && agg, "ps_op_att_pool_gemm_fwd: NULL argument");
    PS_CHECK(attg_ok(K, d, fset, ld) && R * K < (1ll << 31), "ps_op_att_pool_gemm_fwd: K = 16, d in {128, 256, 512}, rows 16-byte aligned (got K %lld, d %lld, ld %lld)",
             (long long)K, (long long)d, (long long)ld);
    if (R <= 0) return PS_OK;
    PS_HIP(hipSetDevice(c->device));
    Stage st(c, "train_att_gemm_fwd", 2);
    AttGArgs a = {};
    a.f = fset; a.ld = (int)ld; a.agg = agg; a.rows = R * K; a.points = R;
    PS_TRY(attg_planes(c, wfc, d, false, &a.w1));
    switch (d) {
        case 128: return launch_attg<128, false>(c, a);
        case 256: return launch_attg<256, false>(c, a);
        default: return launch_attg<512, false>(c, a);
    }
}

extern "C" int ps_op_att_pool_gemm_bwd(ps_context* c, const float* fset, int64_t ld, const float* wfc, const float* dagg, int64_t R, int64_t K, int64_t d,
                                       float* dfset, int64_t lddf, int accumulate, float* dscores, int64_t ldds)
{
    PS_CHECK(c && fset && wfc && dagg && dfset && dscores, "ps_op_att_pool_gemm_bwd: NULL argument");
    // (d = 512 runs as two launches over halves of the dF columns.  dfset / dscores leave in 16-byte stores: their bases and pitches are
    //  checked like the inputs')
    PS_CHECK(attg_ok(K, d, fset, ld) && R * K < (1ll << 31) && attg_ok(K, d, dfset, lddf) && attg_ok(K, d, dscores, ldds),
             "ps_op_att_pool_gemm_bwd: K = 16, d in {128, 256, 512}, rows of fset / dfset / dscores 16-byte aligned with pitches %% 4 == 0 (got K %lld, d %lld, ld %lld, lddf %lld, ldds %lld)",
             (long long)K, (long long)d, (long long)ld, (long long)lddf, (long long)ldds);
    if (R <= 0) return PS_OK;
    PS_HIP(hipSetDevice(c->device));
    Stage st(c, "train_att_gemm_bwd", 3);
    AttGArgs a = {};
    a.f = fset; a.ld = (int)ld; a.dagg = dagg; a.df = dfset; a.lddf = (int)lddf; a.ds = dscores; a.ldds = (int)ldds;
    a.rows = R * K; a.points = R; a.df_accum = accumulate ? 1 : 0;
    PS_TRY(attg_planes(c, wfc, d, false, &a.w1));
    PS_TRY(attg_planes(c, wfc, d, true, &a.w2));
    return d == 128 ? launch_attg<128, true>(c, a) : (d == 256 ? launch_attg<256, true>(c, a) : launch_attg<512, true>(c, a));
}

/* split-source forms: F = [fl[idx] | fr] is never materialised (include/pointseg_train_ops.h) */
static bool attg_split_ok(const float* fl, int64_t ldl, const int32_t* idx, int64_t B, int64_t n_src, int64_t n_q, const float* fr, int64_t ldr, int64_t K, int64_t d)
{
    return K == 16 && (d == 128 || d == 256) && fl && idx && fr && ldl % 4 == 0 && ldr % 4 == 0 && ldl >= d / 2 && ldr >= d / 2 &&
           ((reinterpret_cast<uintptr_t>(fl) | reinterpret_cast<uintptr_t>(fr) | reinterpret_cast<uintptr_t>(idx)) & 15) == 0 && B >= 0 && n_src > 0 && n_q > 0 &&
           n_src * ldl < (1ll << 31) && B * n_q * K < (1ll << 31);
}

extern "C" int ps_op_att_pool_gemm_fwd_split(ps_context* c, const float* fl, int64_t ldl, const int32_t* idx, int64_t B, int64_t n_src, int64_t n_q,
                                             const float* fr, int64_t ldr, const float* wfc, int64_t K, int64_t d, float* agg)
{
    PS_CHECK(c && wfc && agg && attg_split_ok(fl, ldl, idx, B, n_src, n_q, fr, ldr, K, d),
             "ps_op_att_pool_gemm_fwd_split: K = 16, d in {128, 256}, rows and the index table 16-byte aligned (got K %lld, d %lld)", (long long)K, (long long)d);
    const int64_t R = B * n_q;
    if (R <= 0) return PS_OK;
    PS_HIP(hipSetDevice(c->device));
    Stage st(c, "train_att_gemm_fwd", 2);
    AttGArgs a = {};
    a.f = fr; a.ld = (int)ldr; a.fl = fl; a.ldl = (int)ldl; a.idx = idx; a.n_src = n_src; a.n_q = n_q;
    a.agg = agg; a.rows = R * K; a.points = R;
    a.f_bf16 = c->train_act_bf16 && c->train_bf16 ? 1 : 0;  // (fr as bfloat16 rows, ldr in elements: ps_set_train_act_bf16; d = 128)
    PS_CHECK(!a.f_bf16 || (d == 128 && ldr % 8 == 0), "ps_op_att_pool_gemm_fwd_split: bfloat16 rows: d = 128, ldr %% 8 == 0");
    PS_TRY(attg_planes(c, wfc, d, false, &a.w1));
    return d == 128 ? launch_attg<128, false>(c, a) : launch_attg<256, false>(c, a);
}

extern "C" int ps_op_att_pool_gemm_bwd_split(ps_context* c, const float* fl, int64_t ldl, const int32_t* idx, int64_t B, int64_t n_src, int64_t n_q,
                                             const float* fr, int64_t ldr, const float* wfc, const float* dagg, int64_t K, int64_t d, float* dfl_rows,
                                             int64_t ld_rows, float* dfr, int64_t lddr, int accumulate, float* dscores, int64_t ldds)
{
    PS_CHECK(c && wfc && dagg && dfl_rows && dfr && dscores && attg_split_ok(fl, ldl, idx, B, n_src, n_q, fr, ldr, K, d) && ld_rows >= d / 2 && lddr >= d / 2 &&
                 ld_rows % 4 == 0 && lddr % 4 == 0 && ((reinterpret_cast<uintptr_t>(dfl_rows) | reinterpret_cast<uintptr_t>(dfr)) & 15) == 0 &&
                 attg_ok(K, d, dscores, ldds),
             "ps_op_att_pool_gemm_bwd_split: K = 16, d in {128, 256}, every row array (inputs, dfl_rows, dfr, dscores) and the index table 16-byte aligned with pitches %% 4 == 0 (got K %lld, d %lld)",
             (long long)K, (long long)d);
    const int64_t R = B * n_q;
    if (R <= 0) return PS_OK;
    PS_HIP(hipSetDevice(c->device));
    Stage st(c, "train_att_gemm_bwd", 3);
    AttGArgs a = {};
    a.f = fr; a.ld = (int)ldr; a.fl = fl; a.ldl = (int)ldl; a.idx = idx; a.n_src = n_src; a.n_q = n_q;
    a.dagg = dagg; a.df = dfr; a.lddf = (int)lddr; a.df_accum = accumulate ? 1 : 0; a.dfl_rows = dfl_rows; a.ld_rows = (int)ld_rows;
    a.ds = dscores; a.ldds = (int)ldds; a.rows = R * K; a.points = R;
    PS_TRY(attg_planes(c, wfc, d, false, &a.w1));
    PS_TRY(attg_planes(c, wfc, d, true, &a.w2));
    return d == 128 ? launch_attg<128, true>(c, a) : launch_attg<256, true>(c, a);
}
