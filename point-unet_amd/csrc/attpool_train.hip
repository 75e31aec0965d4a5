// attpool_train.hip -- attentive pooling of the TRAINING step, forward and backward, fused per point.
//
//   att_pooling (PointSegment/RandLANet.py:388-398):   s = F . Wfc   (F = [N*K, d] neighbour set, no bias)
//                                                      p = softmax over the K rows of a point, per channel
//                                                      agg[n, c] = sum_k p[n,k,c] * F[n,k,c]
//
// The op-by-op formulation writes s and p ([N*K, d] each: 1.5 GB at levels 0 and 1 of a batch of 8 x 180 000 points), reads them
// back in the backward together with dscores / dF, and runs the three d x d GEMMs (scores, input gradient, weight gradient) as
// separate passes over those tensors.  Here a wave takes one point (its K x d tile of F goes to LDS once):
//   forward   scores on fp32 MFMA from the LDS tile, softmax with wave shuffles, weighted sum -> only agg is written
//   backward  scores and softmax RECOMPUTED from the tile; with a = agg (recomputed), g = dagg:
//                 dS[k,c] = p[k,c] g[c] (F[k,c] - a[c])
//                 dF      = p . g  +  dS . Wfc^T          (second MFMA product, seeded with the direct term)
//                 dWfc   += F^T . dS                       (third MFMA product, accumulated over the wave's points in registers)
//             -> reads F and dagg, writes dF; per-workgroup dWfc partials go to a workspace that a second tiny kernel sums in
//                a fixed order (deterministic, no float atomics)
// Traffic per attention stage: forward 1 read of F (was: F read twice, s written + read, p written); backward 1 read of F + 1 write
// of dF (was ~8 passes).  The d x d weights (and their transpose) live in LDS; compiled for d = 16, 32, 64 (encoder levels 0 and 1,
// where the [N*K, d] tensors are large); wider levels keep the op-by-op path, whose tensors are small there.
// bf16 mode (Trainer(mlp_dtype="bf16")): the operands of the three products are rounded to bfloat16 (RNE) -- F and Wfc for the
// scores, dS and Wfc^T for the input gradient, F and dS for the weight gradient -- and multiplied on the fp32 MFMA (a product of
// two bfloat16 values is exact in fp32), accumulation in fp32: the same values as a bf16 MFMA with fp32 accumulate.
#include "common.h"
#include "mfma_tile.h"
#include "reduce_partials.h"
#include "attpool_train.h"
#include "bf16_io.h"

#include <map>
#include <mutex>

namespace ps {

__device__ __forceinline__ float round_bf16(float x)
{
    unsigned u = __float_as_uint(x);
    u += 0x7fffu + ((u >> 16) & 1u);  // round to nearest even (finite inputs)
    return __uint_as_float(u & 0xffff0000u);
}

template <int D>
struct AttTrainGeom {
    static constexpr int PW = D + 16 + (D % 32 == 16 ? 16 : 0);  // weight pitch = 16 (mod 32): conflict-free B-fragment reads
    static constexpr int PA = D + 2;                               // tile pitch = 2 (mod 32): conflict-free A-fragment reads
    static constexpr int NT = D / 16;
};

// stages W (row-major [D,D]) and optionally its transpose into LDS with pitch PW, rounded to bf16 when asked
template <int D, int THREADS>
__device__ __forceinline__ void stage_weights(const float* __restrict__ w, float* W, float* WT, bool bf16)
{
    constexpr int PW = AttTrainGeom<D>::PW;
    for (int i = threadIdx.x; i < D * D; i += THREADS) {
        const int r = i / D, c = i - r * D;
        float v = w[i];
        if (bf16) v = round_bf16(v);
        W[r * PW + c] = v;
        if (WT) WT[c * PW + r] = v;
    }
}

// Walks a wave's points (p, p + stride, ...) keeping the cloud of p without a 64-bit division per point (split form: cloud base of fl).
struct PointWalk {
    int p, stride, cloud, rem, n_q;
    __device__ __forceinline__ PointWalk(int first, int stride_, int n_q_) : p(first), stride(stride_), n_q(n_q_ > 0 ? n_q_ : 0x7fffffff)
    {
        cloud = first / n_q;
        rem = first - cloud * n_q;
    }
    __device__ __forceinline__ PointWalk next() const
    {
        PointWalk w = *this;
        w.p += stride;
        w.rem += stride;
        while (w.rem >= n_q) {
            w.rem -= n_q;
            ++w.cloud;
        }
        return w;
    }
};

// The K x D tile of one point as registers: lane e of pass i holds float4 q of row (64 i + e) / (D/4).  fetch = global -> registers
// (issued one point AHEAD: a wave works on one point at a time, and at two to eight waves per SIMD nothing else covers the latency of
// these loads), commit = registers -> LDS.  Split form: columns [0, D/2) from fl[cloud base + idx[p, row]], [D/2, D) from row p*KN + row of f.
template <int D, int KN>
struct TileRegs {
    static constexpr int Q = D / 4, QH = Q / 2, TOT = KN * Q, NV = (TOT + 63) / 64;
    float4 v[NV];
    __device__ __forceinline__ void fetch(const AttTrainArgs& a, const PointWalk& w, int lane)
    {
        // wave-uniform 64-bit bases (scalar unit), 32-bit element offsets per lane (a tile spans KN * ld elements; the gathered rows of one
        // cloud n_src * ldl: both far below 2^31 -- checked on the host)
        const float* frow = a.f + (size_t)w.p * KN * a.ld;
        const float* fsrc = a.fl ? a.fl + (size_t)w.cloud * a.n_src * a.ldl : nullptr;
        const int32_t* irow = a.idx ? a.idx + (size_t)w.p * KN : nullptr;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = 64 * i + lane;
            if (TOT % 64 == 0 || e < TOT) {
                const unsigned row = (unsigned)e / Q, q = (unsigned)e - row * Q;
                if (!a.fl)
                    v[i] = *reinterpret_cast<const float4*>(frow + (row * (unsigned)a.ld + 4u * q));
                else if (q < (unsigned)QH)
                    v[i] = *reinterpret_cast<const float4*>(fsrc + ((unsigned)irow[row] * (unsigned)a.ldl + 4u * q));
                else if (a.fr_bf16) {  // (the f_xyz half stored as bfloat16: 8 bytes for the four values, kept as loaded -- expand() in front of commit)
                    const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(a.f) + ((size_t)w.p * KN * a.ld + (row * (unsigned)a.ld + 4u * (q - QH))));
                    v[i] = make_float4(__uint_as_float(u.x), __uint_as_float(u.y), 0.f, 0.f);
                }
                else
                    v[i] = *reinterpret_cast<const float4*>(frow + (row * (unsigned)a.ld + 4u * (q - QH)));
            }
        }
    }
    // bfloat16-stored right half -> fp32, where the tile is consumed (converting at fetch would make the wave wait for its own prefetch)
    __device__ __forceinline__ void expand(const AttTrainArgs& a, int lane)
    {
        if (!a.fr_bf16) return;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = 64 * i + lane;
            const unsigned q = (unsigned)e % Q;
            if ((TOT % 64 == 0 || e < TOT) && q >= (unsigned)QH) v[i] = unpack_bf16x4(make_uint2(__float_as_uint(v[i].x), __float_as_uint(v[i].y)));
        }
    }
    // fp32 tile (pitch PA); `Ab` receives the bf16-rounded copy when non-null
    template <int PA>
    __device__ __forceinline__ void commit(float* A, float* Ab, int lane) const
    {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = 64 * i + lane;
            if (TOT % 64 == 0 || e < TOT) {
                const int row = e / Q, q = e - row * Q;
                float* dst = A + row * PA + 4 * q;
                dst[0] = v[i].x; dst[1] = v[i].y; dst[2] = v[i].z; dst[3] = v[i].w;
                if (Ab) {
                    float* db = Ab + row * PA + 4 * q;
                    db[0] = round_bf16(v[i].x); db[1] = round_bf16(v[i].y); db[2] = round_bf16(v[i].z); db[3] = round_bf16(v[i].w);
                }
            }
        }
    }
};

// The K x D tile of dF staged in LDS (pitch PA, 8-byte aligned rows) -> global as 16-byte stores: whole rows of df (plain form), or the
// gathered half -> dfl_rows and the f_xyz half -> df (split form).  A point's rows are contiguous in each destination, so the wave writes
// one or two dense spans instead of 4-byte columns of sixteen lanes.  take() reads the staged tile into registers (the LDS tile is then
// free for the next point), put() issues the stores.
template <int D, int KN, int PA>
struct RowStore {
    static constexpr int Q = D / 4, QH = Q / 2, TOT = KN * Q, NV = (TOT + 63) / 64;
    float4 v[NV];
    __device__ __forceinline__ void take(const float* S, int lane)
    {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = 64 * i + lane;
            if (TOT % 64 == 0 || e < TOT) {
                const int row = e / Q, q = e - row * Q;
                const float2 lo = *reinterpret_cast<const float2*>(S + row * PA + 4 * q);
                const float2 hi = *reinterpret_cast<const float2*>(S + row * PA + 4 * q + 2);
                v[i] = make_float4(lo.x, lo.y, hi.x, hi.y);
            }
        }
    }
    __device__ __forceinline__ void put(const AttTrainArgs& a, int64_t p, int lane) const
    {
        float* drow = a.df + (size_t)p * KN * a.lddf;  // (wave-uniform bases, 32-bit lane offsets: as TileRegs::fetch)
        float* lrow = a.dfl_rows ? a.dfl_rows + (size_t)p * KN * a.ld_rows : nullptr;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = 64 * i + lane;
            if (TOT % 64 == 0 || e < TOT) {
                const unsigned row = (unsigned)e / Q, q = (unsigned)e - row * Q;
                float4 o = v[i];
                float4* dst;
                if (!a.fl) {
                    dst = reinterpret_cast<float4*>(drow + (row * (unsigned)a.lddf + 4u * q));
                } else if (q < (unsigned)QH && a.fr_bf16) {
                    // (the gathered half's gradient rows, read back once by the gather-reduction, in the same storage format: ld_rows in elements)
                    *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(a.dfl_rows) + ((size_t)p * KN * a.ld_rows + (row * (unsigned)a.ld_rows + 4u * q))) = pack_bf16x4(o);
                    continue;
                } else if (q < (unsigned)QH) {
                    dst = reinterpret_cast<float4*>(lrow + (row * (unsigned)a.ld_rows + 4u * q));
                } else if (a.fr_bf16) {
                    // the f_xyz half's gradient has the format of the f_xyz rows (ps_set_train_act_bf16): bfloat16, lddf in elements; a second
                    // gradient of the same tensor is added to the stored (rounded) value and the sum is rounded again
                    uint2* d16 = reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(a.df) + ((size_t)p * KN * a.lddf + (row * (unsigned)a.lddf + 4u * (q - QH))));
                    if (a.df_accum) {
                        const float4 h = unpack_bf16x4(*d16);
                        o.x += h.x; o.y += h.y; o.z += h.z; o.w += h.w;
                    }
                    *d16 = pack_bf16x4(o);
                    continue;
                } else {
                    dst = reinterpret_cast<float4*>(drow + (row * (unsigned)a.lddf + 4u * (q - QH)));
                    if (a.df_accum) {
                        const float4 h = *dst;
                        o.x += h.x; o.y += h.y; o.z += h.z; o.w += h.w;
                    }
                }
                *dst = o;
            }
        }
    }
};

// scores of column tile ct: C[k][c] = sum_j X[k][j] W[j][16 ct + c]   (X = tile with pitch PA, W in LDS with pitch PW)
template <int D>
__device__ __forceinline__ f32x4 score_tile(const float* X, const float* W, int ct, int lane)
{
    constexpr int PW = AttTrainGeom<D>::PW, PA = AttTrainGeom<D>::PA;
    const float* xa = X + (lane & 15) * PA + (lane >> 4);
    const float* wb = W + (lane >> 4) * PW + ct * 16 + (lane & 15);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < D / 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[4 * s], wb[4 * s * PW], acc, 0, 0, 0);
    return acc;
}

// all NT column tiles at once, the k steps outermost: NT independent accumulator chains in flight (a tile on its own is a chain of D/4
// dependent MFMAs), one A-fragment read per k step for all of them, and the softmax VALU work of the tiles follows as one block that
// the other waves' matrix work can overlap
template <int D>
__device__ __forceinline__ void score_tiles(const float* X, const float* W, int lane, f32x4 (&acc)[D / 16])
{
    constexpr int PW = AttTrainGeom<D>::PW, PA = AttTrainGeom<D>::PA, NT = D / 16;
    const float* xa = X + (lane & 15) * PA + (lane >> 4);
    const float* wb = W + (lane >> 4) * PW + (lane & 15);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < D / 4; ++s) {
        const float av = xa[4 * s];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, wb[4 * s * PW + 16 * t], acc[t], 0, 0, 0);
    }
}

template <int D, int KN, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void att_train_fwd_kernel(AttTrainArgs a)
{
    static_assert(KN == 16, "one 16-row tile per point");
    constexpr int PW = AttTrainGeom<D>::PW, PA = AttTrainGeom<D>::PA, NT = D / 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* W = smem;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, g = lane >> 4, c16 = lane & 15;  // (wave: an SGPR -- the point walk and every row base derived from it stay on the scalar unit)
    // (fp32 operands only: the bf16-MLP mode has its own kernels below -- the run-time rounding switch these once carried cost a select
    //  chain per element in an issue-bound loop)
    float* A = smem + D * PW + wave * KN * PA;
    stage_weights<D, WAVES * 64>(a.w, W, nullptr, false);
    __syncthreads();
    PointWalk w((int)(blockIdx.x * WAVES + wave), (int)(gridDim.x * WAVES), (int)a.n_q);
    TileRegs<D, KN> regs;
    if (w.p < a.R) regs.fetch(a, w, lane);
    for (; w.p < a.R; w = w.next()) {
        const int64_t p = w.p;
        regs.expand(a, lane), regs.template commit<PA>(A, nullptr, lane);
        wave_lds_sync();
        const PointWalk wn = w.next();
        if (wn.p < a.R) regs.fetch(a, wn, lane);  // the next point's tile travels while this one is worked on
        const float* X = A;
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
            const f32x4 s = score_tile<D>(X, W, ct, lane);  // (tile by tile here: the all-tiles form measured 3 % slower in the forward)
            float m = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
            m = xor_max(m);
            float ssum = 0.f, num = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __expf(s[r] - m);
                ssum += e;
                num = __builtin_fmaf(e, A[(4 * g + r) * PA + ct * 16 + c16], num);
            }
            ssum = xor_sum(ssum);
            num = xor_sum(num);
            if (g == 0) a.agg[(size_t)p * D + ct * 16 + c16] = num * __builtin_amdgcn_rcpf(ssum);  // (v_rcp_f32, 1 ulp: the IEEE division is ten instructions in an issue-bound loop)
        }
        wave_lds_sync();
    }
}

template <int D, int KN, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void att_train_bwd_kernel(AttTrainArgs a)
{
    static_assert(KN == 16, "one 16-row tile per point");
    constexpr int PW = AttTrainGeom<D>::PW, PA = AttTrainGeom<D>::PA, NT = D / 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* W = smem;
    float* WT = smem + D * PW;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, g = lane >> 4, c16 = lane & 15;  // (wave: an SGPR -- the point walk and every row base derived from it stay on the scalar unit)
    constexpr int per_wave = 2 * KN * PA;
    float* A = smem + 2 * D * PW + wave * per_wave;
    float* T = A + KN * PA;                            // dS (rounded in bf16 mode: it only ever feeds the two products)
    constexpr float* Ab = nullptr;  // (fp32 operands only: see the forward kernel)
    stage_weights<D, WAVES * 64>(a.w, W, WT, false);
    __syncthreads();

    f32x4 dw[NT][NT];  // this wave's share of dWfc: tile (ti, tj) = rows 16 ti.., columns 16 tj..
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) dw[ti][tj] = f32x4{0.f, 0.f, 0.f, 0.f};

    // per point: [tile and dagg of the NEXT point requested] -> scores / softmax / dS -> the two products, dF staged in the value tile ->
    // staged rows to registers -> next tile committed -> stores issued.  Loads and stores so have a whole point's work to complete in.
    PointWalk w((int)(blockIdx.x * WAVES + wave), (int)(gridDim.x * WAVES), (int)a.n_q);
    TileRegs<D, KN> regs;
    float gnext[NT];
    if (w.p < a.R) {
        regs.fetch(a, w, lane);
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) gnext[ct] = a.dagg[(size_t)w.p * D + ct * 16 + c16];
        regs.expand(a, lane), regs.template commit<PA>(A, Ab, lane);
    }
    wave_lds_sync();
    for (; w.p < a.R; w = w.next()) {
        const int64_t p = w.p;
        const PointWalk wn = w.next();
        float gcur[NT];
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) gcur[ct] = gnext[ct];
        if (wn.p < a.R) {
            regs.fetch(a, wn, lane);
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) gnext[ct] = a.dagg[(size_t)wn.p * D + ct * 16 + c16];
        }
        const float* X = A;
        f32x4 dfd[NT];  // direct term p * g of every column tile (seeds the second product)
        f32x4 sc_all[NT];
        score_tiles<D>(X, W, lane, sc_all);
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
            const f32x4 s = sc_all[ct];
            float m = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
            m = xor_max(m);
            float e[4], fv[4], ssum = 0.f, num = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                e[r] = __expf(s[r] - m);
                fv[r] = A[(4 * g + r) * PA + ct * 16 + c16];
                ssum += e[r];
                num = __builtin_fmaf(e[r], fv[r], num);
            }
            ssum = xor_sum(ssum);
            num = xor_sum(num);
            const float inv = __builtin_amdgcn_rcpf(ssum), agg = num * inv;
            const float gch = gcur[ct];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pr = e[r] * inv;
                dfd[ct][r] = pr * gch;
                T[(4 * g + r) * PA + ct * 16 + c16] = pr * gch * (fv[r] - agg);
            }
        }
        wave_lds_sync();
        // ---- dWfc += F^T . dS  (contraction over the K = 16 rows: four MFMA steps per tile pair) ----
#pragma unroll
        for (int s = 0; s < KN / 4; ++s) {
            float fa[NT], db[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                fa[t] = X[(4 * s + g) * PA + t * 16 + c16];  // A operand: F^T[i = 16 t + c16][k = 4 s + g]
                db[t] = T[(4 * s + g) * PA + t * 16 + c16];  // B operand: dS[k][j = 16 t + c16]
            }
#pragma unroll
            for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                for (int tj = 0; tj < NT; ++tj) dw[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[ti], db[tj], dw[ti][tj], 0, 0, 0);
        }
        if (a.vec_store) wave_lds_sync();  // the value tile is dead from here on: it stages dF for the 16-byte stores
        // ---- dF = p . g + dS . Wfc^T  -> global  (tile by tile: the all-tiles form of the scores measured no better here) ----
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) {
            const float* xa = T + c16 * PA + g;
            const float* wb = WT + g * PW + tj * 16 + c16;
            f32x4 acc = dfd[tj];
#pragma unroll
            for (int s = 0; s < D / 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[4 * s], wb[4 * s * PW], acc, 0, 0, 0);
            if (a.vec_store) {
#pragma unroll
                for (int r = 0; r < 4; ++r) A[(4 * g + r) * PA + tj * 16 + c16] = acc[r];
            } else if (!a.fl) {
#pragma unroll
                for (int r = 0; r < 4; ++r) a.df[(size_t)(p * KN + 4 * g + r) * a.lddf + tj * 16 + c16] = acc[r];
            } else {
                const int col = tj * 16 + c16;
                if (col >= D / 2 && a.fr_bf16) {  // f_xyz half as bfloat16 rows (the float-atomic scatter form: element by element)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        __bf16* q = reinterpret_cast<__bf16*>(a.df) + (size_t)(p * KN + 4 * g + r) * a.lddf + col - D / 2;
                        *q = (__bf16)(a.df_accum ? (float)*q + acc[r] : acc[r]);
                    }
                } else if (col >= D / 2) {  // f_xyz half: plain rows
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float* q = a.df + (size_t)(p * KN + 4 * g + r) * a.lddf + col - D / 2;
                        *q = a.df_accum ? *q + acc[r] : acc[r];
                    }
                } else if (a.dfl_rows && a.fr_bf16) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) reinterpret_cast<__bf16*>(a.dfl_rows)[(size_t)(p * KN + 4 * g + r) * a.ld_rows + col] = (__bf16)acc[r];
                } else if (a.dfl_rows) {  // gathered half as plain rows: summed per source row by a gather-reduction afterwards
#pragma unroll
                    for (int r = 0; r < 4; ++r) a.dfl_rows[(size_t)(p * KN + 4 * g + r) * a.ld_rows + col] = acc[r];
                } else {             // gathered half: scatter-add onto the source rows
                    const int4 nb = *reinterpret_cast<const int4*>(a.idx + p * KN + 4 * g);
                    const int64_t base = (p / a.n_q) * a.n_src;
                    atomicAdd(a.dfl + (size_t)(base + nb.x) * a.lddl + col, acc[0]);
                    atomicAdd(a.dfl + (size_t)(base + nb.y) * a.lddl + col, acc[1]);
                    atomicAdd(a.dfl + (size_t)(base + nb.z) * a.lddl + col, acc[2]);
                    atomicAdd(a.dfl + (size_t)(base + nb.w) * a.lddl + col, acc[3]);
                }
            }
        }
        wave_lds_sync();
        RowStore<D, KN, PA> out;
        if (a.vec_store) {
            out.take(A, lane);
            wave_lds_sync();
        }
        if (wn.p < a.R) regs.expand(a, lane), regs.template commit<PA>(A, Ab, lane);
        if (a.vec_store) out.put(a, p, lane);
        wave_lds_sync();
    }
    // ---- the workgroup's dWfc partial: waves add up through LDS (fixed order), one plain store per element ----
    __syncthreads();
    float* red = smem;  // weights and tiles are dead: the whole buffer holds the WAVES partials (sized for it on the host)
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int tj = 0; tj < NT; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(size_t)wave * D * D + (ti * 16 + 4 * g + r) * D + tj * 16 + c16] = dw[ti][tj][r];
    __syncthreads();
    for (int i = threadIdx.x; i < D * D; i += WAVES * 64) {
        float sum = 0.f;
        for (int w = 0; w < WAVES; ++w) sum += red[(size_t)w * D * D + i];
        a.dw_part[(size_t)blockIdx.x * D * D + i] = sum;
    }
}

// ---- bf16 mode on the bf16 matrix pipe ---------------------------------------------------------------------------------------------------
// Trainer(mlp_dtype="bf16") rounds the operands of the three products to bfloat16; the kernels above then multiply the rounded values on
// the fp32 MFMA (same results as a bf16 MFMA with fp32 accumulation, but at the fp32 rate: 1/16 of the bf16 pipe).  The kernels below keep
// the rounded operands AS bfloat16 in LDS -- both weight orientations, the tile of F and the tile of dS -- and run the products on
// v_mfma_f32_16x16x16_bf16: lane (i, g) supplies k = 4g .. 4g+3, i.e. ONE 8-byte LDS read per operand and sixteen k per instruction
// (D/16 instructions per 16 x 16 output tile instead of D/4); the output layout is that of the fp32 16x16x4 tile, so the softmax, the
// direct term, the stores and the per-workgroup dWfc reduction are unchanged.  LDS pitches are 32 bytes (mod 128): the 64 lanes' 8-byte
// reads spread over the banks four to a word, the minimum.
typedef short bf16x4s __attribute__((ext_vector_type(4)));

typedef __bf16 bf16x2h __attribute__((ext_vector_type(2)));
typedef float f32x2h __attribute__((ext_vector_type(2)));
// round to nearest even on v_cvt_pk_bf16_f32 (one instruction per PAIR; the bit-twiddling form was five per element in an issue-bound kernel)
__device__ __forceinline__ unsigned bf16_pair(float a, float b)
{
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2h{a, b}, bf16x2h));
}
__device__ __forceinline__ unsigned bf16_bits(float x) { return bf16_pair(x, 0.f) & 0xffffu; }

// (dWfc partials per wave instead of per workgroup when the cross-wave reduction buffer would outgrow LDS)
constexpr bool att_wave_partials(int D, int WAVES) { return (size_t)WAVES * D * D * sizeof(float) > 128 * 1024; }

template <int D>
struct AttBf16Geom {
    static constexpr int PB = D == 16 ? 16 : (D == 32 ? 48 : (D == 64 ? 80 : 144));  // bfloat16 elements per tile / weight row: 32, 96, 160, 288 bytes
    static constexpr int PA = D + 2;                                // fp32 value tile (the weighted sum uses the unrounded F)
    static constexpr int NT = D / 16;
};

// W (row-major [D, D]) -> Wb[i][j] = bf16(W[i][j]) and WTb[j][i] = bf16(W[i][j]); either may be null
template <int D, int THREADS>
__device__ __forceinline__ void stage_weights_bf16(const float* __restrict__ w, unsigned short* Wb, unsigned short* WTb)
{
    constexpr int PB = AttBf16Geom<D>::PB;
    for (int i = threadIdx.x; i < D * D; i += THREADS) {
        const int r = i / D, c = i - r * D;
        const unsigned short v = (unsigned short)bf16_bits(w[i]);
        if (Wb) Wb[r * PB + c] = v;
        if (WTb) WTb[c * PB + r] = v;
    }
}

// a fetched tile -> A (fp32, pitch PA) and Xb (bfloat16, pitch PB)
template <int D, int KN>
__device__ __forceinline__ void commit_tile_bf16(const TileRegs<D, KN>& t, float* A, unsigned short* Xb, int lane)
{
    constexpr int PA = AttBf16Geom<D>::PA, PB = AttBf16Geom<D>::PB, Q = D / 4, TOT = KN * Q;
#pragma unroll
    for (int i = 0; i < TileRegs<D, KN>::NV; ++i) {
        const int e = 64 * i + lane;
        if (TOT % 64 == 0 || e < TOT) {
            const int row = e / Q, q = e - row * Q;
            const float4 v = t.v[i];
            float* dst = A + row * PA + 4 * q;
            dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
            uint2 pk;
            pk.x = bf16_pair(v.x, v.y);
            pk.y = bf16_pair(v.z, v.w);
            *reinterpret_cast<uint2*>(Xb + row * PB + 4 * q) = pk;
        }
    }
}

// C[k][c] = sum_j X[k][j] B[j][16 ct + c] with X rows and the B columns both stored k-contiguous: Xb[row][j], Bt[col][j]
template <int D>
__device__ __forceinline__ f32x4 tile_mma_bf16(const unsigned short* Xb, const unsigned short* Bt, int ct, int lane, f32x4 acc)
{
    constexpr int PB = AttBf16Geom<D>::PB;
    const bf16x4s* xa = reinterpret_cast<const bf16x4s*>(Xb + (lane & 15) * PB + 4 * (lane >> 4));
    const bf16x4s* wb = reinterpret_cast<const bf16x4s*>(Bt + (ct * 16 + (lane & 15)) * PB + 4 * (lane >> 4));
#pragma unroll
    for (int kk = 0; kk < D / 16; ++kk) acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(xa[4 * kk], wb[4 * kk], acc, 0, 0, 0);
    return acc;
}

template <int D, int KN, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void att_train_fwd_bf16_kernel(AttTrainArgs a)
{
    static_assert(KN == 16, "one 16-row tile per point");
    constexpr int PB = AttBf16Geom<D>::PB, PA = AttBf16Geom<D>::PA, NT = D / 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned short* WTb = reinterpret_cast<unsigned short*>(smem);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, g = lane >> 4, c16 = lane & 15;  // (wave: an SGPR -- the point walk and every row base derived from it stay on the scalar unit)
    constexpr int W_FLOATS = D * PB / 2, TILE_FLOATS = KN * PA + KN * PB / 2;
    float* A = smem + W_FLOATS + wave * TILE_FLOATS;
    unsigned short* Xb = reinterpret_cast<unsigned short*>(A + KN * PA);
    stage_weights_bf16<D, WAVES * 64>(a.w, nullptr, WTb);
    __syncthreads();
    PointWalk w((int)(blockIdx.x * WAVES + wave), (int)(gridDim.x * WAVES), (int)a.n_q);
    TileRegs<D, KN> regs;
    if (w.p < a.R) regs.fetch(a, w, lane);
    for (; w.p < a.R; w = w.next()) {
        const int64_t p = w.p;
        regs.expand(a, lane), commit_tile_bf16<D, KN>(regs, A, Xb, lane);
        wave_lds_sync();
        const PointWalk wn = w.next();
        if (wn.p < a.R) regs.fetch(a, wn, lane);  // the next point's tile travels while this one is worked on
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
            const f32x4 s = tile_mma_bf16<D>(Xb, WTb, ct, lane, f32x4{0.f, 0.f, 0.f, 0.f});
            float m = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
            m = xor_max(m);
            float ssum = 0.f, num = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __expf(s[r] - m);
                ssum += e;
                num = __builtin_fmaf(e, A[(4 * g + r) * PA + ct * 16 + c16], num);
            }
            ssum = xor_sum(ssum);
            num = xor_sum(num);
            if (g == 0) a.agg[(size_t)p * D + ct * 16 + c16] = num * __builtin_amdgcn_rcpf(ssum);  // (v_rcp_f32, 1 ulp: the IEEE division is ten instructions in an issue-bound loop)
        }
        wave_lds_sync();
    }
}

template <int D, int KN, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void att_train_bwd_bf16_kernel(AttTrainArgs a)
{
    static_assert(KN == 16, "one 16-row tile per point");
    constexpr int PB = AttBf16Geom<D>::PB, PA = AttBf16Geom<D>::PA, NT = D / 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int W_FLOATS = D * PB / 2, TILE_FLOATS = KN * PA + 2 * (KN * PB / 2);
    unsigned short* Wb = reinterpret_cast<unsigned short*>(smem);
    unsigned short* WTb = reinterpret_cast<unsigned short*>(smem + W_FLOATS);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, g = lane >> 4, c16 = lane & 15;  // (wave: an SGPR -- the point walk and every row base derived from it stay on the scalar unit)
    float* A = smem + 2 * W_FLOATS + wave * TILE_FLOATS;
    unsigned short* Xb = reinterpret_cast<unsigned short*>(A + KN * PA);
    unsigned short* Tb = Xb + KN * PB;  // dS, rounded: it only ever feeds the two products
    stage_weights_bf16<D, WAVES * 64>(a.w, Wb, WTb);
    __syncthreads();

    f32x4 dw[NT][NT];  // this wave's share of dWfc: tile (ti, tj) = rows 16 ti.., columns 16 tj..
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) dw[ti][tj] = f32x4{0.f, 0.f, 0.f, 0.f};

    // per point: [tile and dagg of the NEXT point requested] -> scores / softmax / dS -> the two products, dF staged in the value tile ->
    // staged rows to registers -> next tile committed -> stores issued.  Loads and stores so have a whole point's work to complete in.
    PointWalk w((int)(blockIdx.x * WAVES + wave), (int)(gridDim.x * WAVES), (int)a.n_q);
    TileRegs<D, KN> regs;
    float gnext[NT];
    if (w.p < a.R) {
        regs.fetch(a, w, lane);
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) gnext[ct] = a.dagg[(size_t)w.p * D + ct * 16 + c16];
        regs.expand(a, lane), commit_tile_bf16<D, KN>(regs, A, Xb, lane);
    }
    wave_lds_sync();
    for (; w.p < a.R; w = w.next()) {
        const int64_t p = w.p;
        const PointWalk wn = w.next();
        float gcur[NT];
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) gcur[ct] = gnext[ct];
        if (wn.p < a.R) {
            regs.fetch(a, wn, lane);
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) gnext[ct] = a.dagg[(size_t)wn.p * D + ct * 16 + c16];
        }
        f32x4 dfd[NT];  // direct term p * g of every column tile (seeds the second product)
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
            const f32x4 s = tile_mma_bf16<D>(Xb, WTb, ct, lane, f32x4{0.f, 0.f, 0.f, 0.f});
            float m = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
            m = xor_max(m);
            float e[4], fv[4], ssum = 0.f, num = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                e[r] = __expf(s[r] - m);
                fv[r] = A[(4 * g + r) * PA + ct * 16 + c16];
                ssum += e[r];
                num = __builtin_fmaf(e[r], fv[r], num);
            }
            ssum = xor_sum(ssum);
            num = xor_sum(num);
            const float inv = __builtin_amdgcn_rcpf(ssum), agg = num * inv;
            const float gch = gcur[ct];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pr = e[r] * inv;
                dfd[ct][r] = pr * gch;
                Tb[(4 * g + r) * PB + ct * 16 + c16] = (unsigned short)bf16_bits(pr * gch * (fv[r] - agg));
            }
        }
        wave_lds_sync();
        // ---- dF = p . g + dS . Wfc^T  -> global:  out[row][i] = sum_j dS[row][j] W[i][j]  (B columns = rows of W: Wb) ----
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) {
            const f32x4 acc = tile_mma_bf16<D>(Tb, Wb, tj, lane, dfd[tj]);
            if (a.vec_store) {  // (the fp32 value tile is dead after the softmax loop: it stages dF for the 16-byte stores)
#pragma unroll
                for (int r = 0; r < 4; ++r) A[(4 * g + r) * PA + tj * 16 + c16] = acc[r];
            } else if (!a.fl) {
#pragma unroll
                for (int r = 0; r < 4; ++r) a.df[(size_t)(p * KN + 4 * g + r) * a.lddf + tj * 16 + c16] = acc[r];
            } else {
                const int col = tj * 16 + c16;
                if (col >= D / 2 && a.fr_bf16) {  // f_xyz half as bfloat16 rows (the float-atomic scatter form: element by element)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        __bf16* q = reinterpret_cast<__bf16*>(a.df) + (size_t)(p * KN + 4 * g + r) * a.lddf + col - D / 2;
                        *q = (__bf16)(a.df_accum ? (float)*q + acc[r] : acc[r]);
                    }
                } else if (col >= D / 2) {  // f_xyz half: plain rows
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float* q = a.df + (size_t)(p * KN + 4 * g + r) * a.lddf + col - D / 2;
                        *q = a.df_accum ? *q + acc[r] : acc[r];
                    }
                } else if (a.dfl_rows && a.fr_bf16) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) reinterpret_cast<__bf16*>(a.dfl_rows)[(size_t)(p * KN + 4 * g + r) * a.ld_rows + col] = (__bf16)acc[r];
                } else if (a.dfl_rows) {  // gathered half as plain rows: summed per source row by a gather-reduction afterwards
#pragma unroll
                    for (int r = 0; r < 4; ++r) a.dfl_rows[(size_t)(p * KN + 4 * g + r) * a.ld_rows + col] = acc[r];
                } else {             // gathered half: scatter-add onto the source rows
                    const int4 nb = *reinterpret_cast<const int4*>(a.idx + p * KN + 4 * g);
                    const int64_t base = (p / a.n_q) * a.n_src;
                    atomicAdd(a.dfl + (size_t)(base + nb.x) * a.lddl + col, acc[0]);
                    atomicAdd(a.dfl + (size_t)(base + nb.y) * a.lddl + col, acc[1]);
                    atomicAdd(a.dfl + (size_t)(base + nb.z) * a.lddl + col, acc[2]);
                    atomicAdd(a.dfl + (size_t)(base + nb.w) * a.lddl + col, acc[3]);
                }
            }
        }
        // ---- dWfc += F^T . dS: ONE 16-row contraction per tile pair (lane (i, g) supplies rows 4g .. 4g+3 of column i) ----
        {
            bf16x4s fa[NT], db[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const unsigned short* xp = Xb + (4 * g) * PB + t * 16 + c16;
                const unsigned short* tp = Tb + (4 * g) * PB + t * 16 + c16;
                fa[t] = bf16x4s{(short)xp[0], (short)xp[PB], (short)xp[2 * PB], (short)xp[3 * PB]};
                db[t] = bf16x4s{(short)tp[0], (short)tp[PB], (short)tp[2 * PB], (short)tp[3 * PB]};
            }
#pragma unroll
            for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                for (int tj = 0; tj < NT; ++tj) dw[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(fa[ti], db[tj], dw[ti][tj], 0, 0, 0);
        }
        wave_lds_sync();
        RowStore<D, KN, PA> out;
        if (a.vec_store) {
            out.take(A, lane);
            wave_lds_sync();
        }
        if (wn.p < a.R) regs.expand(a, lane), commit_tile_bf16<D, KN>(regs, A, Xb, lane);
        if (a.vec_store) out.put(a, p, lane);
        wave_lds_sync();
    }
    if constexpr (att_wave_partials(D, WAVES)) {
        // d = 128: WAVES x D x D floats do not fit LDS -- every WAVE stores its own partial (the fixed-order reduction behind this kernel
        // simply sees WAVES times as many rows)
        float* dst = a.dw_part + ((size_t)blockIdx.x * WAVES + wave) * D * D;
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
#pragma unroll
            for (int tj = 0; tj < NT; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[(ti * 16 + 4 * g + r) * D + tj * 16 + c16] = dw[ti][tj][r];
        return;
    }
    // ---- the workgroup's dWfc partial: waves add up through LDS (fixed order), one plain store per element ----
    __syncthreads();
    float* red = smem;  // weights and tiles are dead: the whole buffer holds the WAVES partials (sized for it on the host)
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int tj = 0; tj < NT; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(size_t)wave * D * D + (ti * 16 + 4 * g + r) * D + tj * 16 + c16] = dw[ti][tj][r];
    __syncthreads();
    for (int i = threadIdx.x; i < D * D; i += WAVES * 64) {
        float sum = 0.f;
        for (int w = 0; w < WAVES; ++w) sum += red[(size_t)w * D * D + i];
        a.dw_part[(size_t)blockIdx.x * D * D + i] = sum;
    }
}

// ---- column-split backward (bf16 mode, d = 128): FOUR waves share one point ---------------------------------------------------------------
// At d = 128 the kernel above holds 64 x 4 dWfc accumulators per wave (464 registers: one wave per SIMD) and 17.5 KB of tiles per wave next
// to 74 KB of weights (four waves per CU): every point is one long serial chain -- fetch, scores, softmax, dS, two products, stores -- with
// nothing to hide its latencies.  Here the four waves of a GROUP work on the same point and split the column tiles of every phase: a wave
// computes the scores / softmax / dS of its D/64 column tiles (a 16 x 16 tile holds all K rows of its channels: the softmax stays inside
// the wave), accumulates the matching columns of dWfc (64 x 4 / 4 accumulators), and -- once all of dS is in LDS -- its column tiles of
// dF = p g + dS . Wfc^T.  The group shares ONE set of tiles, so a workgroup of GROUPS groups fits 4 GROUPS waves, each a quarter as long
// per point.  Workgroup barriers separate the phases (all groups take the same number of iterations; a group past the end idles through
// them).  Row outputs only (df / dfl_rows as 16-byte stores): the deterministic step's form.
template <int D, int KN, int GROUPS>
__global__ __launch_bounds__(GROUPS * 256) void att_train_bwd_bf16_cs_kernel(AttTrainArgs a)
{
    static_assert(KN == 16 && D % 64 == 0, "one 16-row tile per point, column tiles split over four waves");
    constexpr int PB = AttBf16Geom<D>::PB, PA = AttBf16Geom<D>::PA, NT = D / 16, NTW = NT / 4;
    constexpr int Q = D / 4, QH = Q / 2, TOT = KN * Q, NVG = TOT / 256;  // float4 per lane of the group's 256 lanes
    static_assert(TOT % 256 == 0, "the tile divides over the group's lanes");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int W_FLOATS = D * PB / 2, TILE_FLOATS = KN * PA + 2 * (KN * PB / 2);
    unsigned short* Wb = reinterpret_cast<unsigned short*>(smem);
    unsigned short* WTb = reinterpret_cast<unsigned short*>(smem + W_FLOATS);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, g = lane >> 4, c16 = lane & 15;  // (wave: an SGPR -- the point walk and every row base derived from it stay on the scalar unit)
    const int grp = wave >> 2, wj = wave & 3, glane = (wj << 6) | lane;
    float* A = smem + 2 * W_FLOATS + grp * TILE_FLOATS;
    unsigned short* Xb = reinterpret_cast<unsigned short*>(A + KN * PA);
    unsigned short* Tb = Xb + KN * PB;
    stage_weights_bf16<D, GROUPS * 256>(a.w, Wb, WTb);

    f32x4 dw[NT][NTW];  // rows 16 ti.., this wave's column tiles wj * NTW + t
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int t = 0; t < NTW; ++t) dw[ti][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int stride = (int)gridDim.x * GROUPS;
    const int first_of_block = (int)blockIdx.x * GROUPS;
    const int iters = first_of_block < a.R ? (int)((a.R - first_of_block + stride - 1) / stride) : 0;  // of the block's FIRST group: the most
    PointWalk w(first_of_block + grp, stride, (int)a.n_q);

    // this lane's share of a point's tile: elements glane, glane + 256, ...
    float4 regs[NVG];
    float gnext[NTW];
    auto fetch = [&](const PointWalk& pw) {
        const int64_t p = pw.p;
        // (wave-uniform bases, 32-bit lane offsets: as TileRegs::fetch)
        const float* frow = a.f + (size_t)p * KN * a.ld;
        const float* fsrc = a.fl ? a.fl + (size_t)pw.cloud * a.n_src * a.ldl : nullptr;
        const int32_t* irow = a.idx ? a.idx + (size_t)p * KN : nullptr;
#pragma unroll
        for (int i = 0; i < NVG; ++i) {
            const unsigned e = 256u * i + glane;
            const unsigned row = e / Q, q = e - row * Q;
            if (!a.fl)
                regs[i] = *reinterpret_cast<const float4*>(frow + (row * (unsigned)a.ld + 4u * q));
            else if (q < (unsigned)QH)
                regs[i] = *reinterpret_cast<const float4*>(fsrc + ((unsigned)irow[row] * (unsigned)a.ldl + 4u * q));
            else if (a.fr_bf16) {  // (kept as loaded; expanded in commit)
                const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(a.f) + ((size_t)p * KN * a.ld + (row * (unsigned)a.ld + 4u * (q - QH))));
                regs[i] = make_float4(__uint_as_float(u.x), __uint_as_float(u.y), 0.f, 0.f);
            }
            else
                regs[i] = *reinterpret_cast<const float4*>(frow + (row * (unsigned)a.ld + 4u * (q - QH)));
        }
        const float* grow = a.dagg + (size_t)p * D;
#pragma unroll
        for (int t = 0; t < NTW; ++t) gnext[t] = grow[(wj * NTW + t) * 16 + c16];
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < NVG; ++i) {
            const int e = 256 * i + glane;
            const int row = e / Q, q = e - row * Q;
            const float4 v = a.fr_bf16 && q >= QH ? unpack_bf16x4(make_uint2(__float_as_uint(regs[i].x), __float_as_uint(regs[i].y))) : regs[i];
            float* dst = A + row * PA + 4 * q;
            dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
            uint2 pk;
            pk.x = bf16_pair(v.x, v.y);
            pk.y = bf16_pair(v.z, v.w);
            *reinterpret_cast<uint2*>(Xb + row * PB + 4 * q) = pk;
        }
    };
    if (w.p < a.R) {
        fetch(w);
        commit();
    }
    __syncthreads();  // weights and the first tiles
    for (int it = 0; it < iters; ++it, w = w.next()) {
        const bool live = w.p < a.R;
        const int64_t p = w.p;
        const PointWalk wn = w.next();
        const bool more = wn.p < a.R;
        float gcur[NTW];
#pragma unroll
        for (int t = 0; t < NTW; ++t) gcur[t] = gnext[t];
        if (more) fetch(wn);
        f32x4 dfd[NTW];
        if (live) {
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
                const int ct = wj * NTW + t;
                const f32x4 s = tile_mma_bf16<D>(Xb, WTb, ct, lane, f32x4{0.f, 0.f, 0.f, 0.f});
                float m = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
                m = xor_max(m);
                float e[4], fv[4], ssum = 0.f, num = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    e[r] = __expf(s[r] - m);
                    fv[r] = A[(4 * g + r) * PA + ct * 16 + c16];
                    ssum += e[r];
                    num = __builtin_fmaf(e[r], fv[r], num);
                }
                ssum = xor_sum(ssum);
                num = xor_sum(num);
                const float inv = __builtin_amdgcn_rcpf(ssum), agg = num * inv;
                const float gch = gcur[t];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pr = e[r] * inv;
                    dfd[t][r] = pr * gch;
                    Tb[(4 * g + r) * PB + ct * 16 + c16] = (unsigned short)bf16_bits(pr * gch * (fv[r] - agg));
                }
            }
            wave_lds_sync();  // (this wave's own dS columns)
            // dWfc columns of this wave += F^T . dS: one 16-row contraction per tile pair
            bf16x4s fa[NT], db[NTW];
#pragma unroll
            for (int ti = 0; ti < NT; ++ti) {
                const unsigned short* xp = Xb + (4 * g) * PB + ti * 16 + c16;
                fa[ti] = bf16x4s{(short)xp[0], (short)xp[PB], (short)xp[2 * PB], (short)xp[3 * PB]};
            }
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
                const unsigned short* tp = Tb + (4 * g) * PB + (wj * NTW + t) * 16 + c16;
                db[t] = bf16x4s{(short)tp[0], (short)tp[PB], (short)tp[2 * PB], (short)tp[3 * PB]};
            }
#pragma unroll
            for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                for (int t = 0; t < NTW; ++t) dw[ti][t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(fa[ti], db[t], dw[ti][t], 0, 0, 0);
        }
        __syncthreads();  // all of dS is in LDS
        if (live) {
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
                const int tj = wj * NTW + t;
                const f32x4 acc = tile_mma_bf16<D>(Tb, Wb, tj, lane, dfd[t]);
#pragma unroll
                for (int r = 0; r < 4; ++r) A[(4 * g + r) * PA + tj * 16 + c16] = acc[r];  // (the value tile is dead: it stages dF)
            }
        }
        __syncthreads();  // dF staged, dS read by everyone
        float4 out[NVG];
        if (live) {
#pragma unroll
            for (int i = 0; i < NVG; ++i) {
                const int e = 256 * i + glane;
                const int row = e / Q, q = e - row * Q;
                const float2 lo = *reinterpret_cast<const float2*>(A + row * PA + 4 * q);
                const float2 hi = *reinterpret_cast<const float2*>(A + row * PA + 4 * q + 2);
                out[i] = make_float4(lo.x, lo.y, hi.x, hi.y);
            }
        }
        __syncthreads();  // the staged tile is in registers: the next point's tile may land
        if (more) commit();
        if (live) {
#pragma unroll
            for (int i = 0; i < NVG; ++i) {
                const int e = 256 * i + glane;
                const int row = e / Q, q = e - row * Q;
                float4 o = out[i];
                float4* dst;
                float* drow = a.df + (size_t)p * KN * a.lddf;
                float* lrow = a.dfl_rows ? a.dfl_rows + (size_t)p * KN * a.ld_rows : nullptr;
                if (!a.fl) {
                    dst = reinterpret_cast<float4*>(drow + ((unsigned)row * (unsigned)a.lddf + 4u * q));
                } else if (q < QH && a.fr_bf16) {  // (the gathered half's gradient rows as bfloat16: RowStore::put)
                    *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(a.dfl_rows) + ((size_t)p * KN * a.ld_rows + ((unsigned)row * (unsigned)a.ld_rows + 4u * q))) = pack_bf16x4(o);
                    continue;
                } else if (q < QH) {
                    dst = reinterpret_cast<float4*>(lrow + ((unsigned)row * (unsigned)a.ld_rows + 4u * q));
                } else if (a.fr_bf16) {  // (the f_xyz half's gradient as bfloat16 rows: RowStore::put)
                    uint2* d16 = reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(a.df) + ((size_t)p * KN * a.lddf + ((unsigned)row * (unsigned)a.lddf + 4u * (q - QH))));
                    if (a.df_accum) {
                        const float4 h = unpack_bf16x4(*d16);
                        o.x += h.x; o.y += h.y; o.z += h.z; o.w += h.w;
                    }
                    *d16 = pack_bf16x4(o);
                    continue;
                } else {
                    dst = reinterpret_cast<float4*>(drow + ((unsigned)row * (unsigned)a.lddf + 4u * (q - QH)));
                    if (a.df_accum) {
                        const float4 h = *dst;
                        o.x += h.x; o.y += h.y; o.z += h.z; o.w += h.w;
                    }
                }
                *dst = o;
            }
        }
        __syncthreads();  // the next tile is visible
    }
    // the group's dWfc partial: its four waves hold disjoint column tiles
    float* dst = a.dw_part + ((size_t)blockIdx.x * GROUPS + grp) * D * D;
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[(ti * 16 + 4 * g + r) * D + (wj * NTW + t) * 16 + c16] = dw[ti][t][r];
}

// Workgroups per launch: every workgroup walks the points with the same stride, so a grid that is not a multiple of what the chip holds at
// once ends with a round at partial occupancy (the d = 16 backward: 77 VGPRs allow three 8-wave workgroups per CU where the LDS footprint
// alone allows four -- 1024 workgroups ran as 768 + 256, the second round as long as the first).  Ask the runtime what fits.
static int att_resident_blocks(const void* kern, int threads, size_t smem, int lds_guess)
{
    static std::mutex mu;
    static std::map<std::pair<const void*, size_t>, int> cache;  // (asked once per kernel and LDS size)
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find({kern, smem});
    if (it != cache.end()) return it->second;
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, threads, smem) != hipSuccess || occ < 1) {
        (void)hipGetLastError();
        occ = lds_guess;
    }
    cache[{kern, smem}] = occ;
    return occ;
}

template <int D>
static int launch_att_train(ps_context* c, AttTrainArgs a, bool backward, float* dW)
{
    constexpr int KN = 16, WAVES = 8;
    constexpr int PW = AttTrainGeom<D>::PW, PA = AttTrainGeom<D>::PA;
    // d = 128 exists on the bf16 matrix pipe only (both weight orientations as bfloat16: 74 KB; in fp32 they are 147 KB): four waves per
    // workgroup in the backward (17.5 KB of tiles per wave, the 64 x 4 dWfc accumulators in AGPRs: one wave per SIMD), eight in the forward
    constexpr int WAVES_F = 8, WAVES_B = D == 128 ? 4 : 8;  // (d = 64 with twelve waves: 168 VGPRs, 34 spilled, 1.85 against 1.12 ms)
    if (backward) {
        auto al = [](const void* q, int ld) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0 && ld % 4 == 0; };
        const bool rows_out = !a.fl || a.dfl_rows;  // (the atomic scatter form keeps its per-element path)
        a.vec_store = rows_out && al(a.df, a.lddf) && (!a.fl || al(a.dfl_rows, a.ld_rows)) ? 1 : 0;
        PS_CHECK(!a.fr_bf16 || a.vec_store || a.bf16, "att_pool_train backward: bfloat16 rows (ps_set_train_act_bf16) belong to the bf16-MLP mode");
    }
    if constexpr (D == 64) {
        // level 1: the 32x32x16 bf16 matrix pipe (exact three-way splits in fp32 mode), weights resident in LDS, dWfc in registers
        // (attpool_gemm.hip: 0.63 -> 0.3 ms forward, 1.71 -> 0.9 ms backward per pooling of 5.76 M rows)
        if (att64_gemm_fits(c->tune, a, backward)) return att64_gemm(c, a, backward, dW);
    }
    if (a.bf16) {  // the bf16-MLP mode: operands kept as bfloat16 in LDS, products on the bf16 matrix pipe
        constexpr int PB = AttBf16Geom<D>::PB;
        const int waves = backward ? WAVES_B : WAVES_F;
        size_t smem = sizeof(float) * ((backward ? 2 : 1) * (size_t)(D * PB / 2) + (size_t)waves * (KN * PA + (backward ? 2 : 1) * (KN * PB / 2)));
        if (backward && !att_wave_partials(D, WAVES_B)) smem = std::max(smem, sizeof(float) * (size_t)WAVES_B * D * D);
        PS_CHECK(smem <= 160 * 1024, "att_pool_train: %zu bytes of LDS needed", smem);
        const int per_cu = std::max(1, std::min(4, (int)(160 * 1024 / smem)));
        if (!backward) {
            auto kern = att_train_fwd_bf16_kernel<D, KN, WAVES_F>;
            if (smem > 48 * 1024) PS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            const int occ = std::min(per_cu, att_resident_blocks(reinterpret_cast<const void*>(kern), WAVES_F * 64, smem, per_cu));
            const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((a.R + WAVES_F - 1) / WAVES_F, 256 * occ));
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(WAVES_F * 64), smem, c->stream, a);
        } else if (D == 128 && a.vec_store && !c->tune.att_no_split) {
            // four waves per point (att_train_bwd_bf16_cs_kernel): three groups per workgroup share the 74 KB of weights: 897 us against
            // 1 192 us for the one-wave-per-point kernel.  (d = 64, four groups: 1 334 us against 1 100 us -- the barriers of a sixteen-wave
            // workgroup cost more than its shorter chains gain: not used there.)
            if constexpr (D == 128) {
                constexpr int GROUPS = 3;
                const size_t sm = sizeof(float) * (2 * (size_t)(D * PB / 2) + (size_t)GROUPS * (KN * PA + 2 * (KN * PB / 2)));
                auto kern = att_train_bwd_bf16_cs_kernel<D, KN, GROUPS>;
                PS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
                const int occ = std::max(1, att_resident_blocks(reinterpret_cast<const void*>(kern), GROUPS * 256, sm, 1));
                const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((a.R + GROUPS - 1) / GROUPS, 256 * occ));
                const int parts = blocks * GROUPS;
                PS_TRY(c->red_ws.reserve(sizeof(float) * (size_t)parts * D * D + 256));
                a.dw_part = c->red_ws.as<float>();
                hipLaunchKernelGGL(kern, dim3(blocks), dim3(GROUPS * 256), sm, c->stream, a);
                hipLaunchKernelGGL(reduce_partials_kernel<float>, dim3(ceil_div(D * D, 16)), dim3(256), 0, c->stream, static_cast<const float*>(a.dw_part), parts, D * D, dW);
            }
        } else {
            auto kern = att_train_bwd_bf16_kernel<D, KN, WAVES_B>;
            if (smem > 48 * 1024) PS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            const int occ = std::min(per_cu, att_resident_blocks(reinterpret_cast<const void*>(kern), WAVES_B * 64, smem, per_cu));
            const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((a.R + WAVES_B - 1) / WAVES_B, 256 * occ));
            const int parts = att_wave_partials(D, WAVES_B) ? blocks * WAVES_B : blocks;
            PS_TRY(c->red_ws.reserve(sizeof(float) * (size_t)parts * D * D + 256));
            a.dw_part = c->red_ws.as<float>();
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(WAVES_B * 64), smem, c->stream, a);
            hipLaunchKernelGGL(reduce_partials_kernel<float>, dim3(ceil_div(D * D, 16)), dim3(256), 0, c->stream, static_cast<const float*>(a.dw_part), parts, D * D, dW);
        }
        PS_HIP(hipGetLastError());
        return PS_OK;
    }
    if constexpr (D == 128) {
        set_error("att_pool_train: d = 128 is compiled for the bf16-MLP mode only (ps_set_train_gemm_bf16)");
        return PS_EINVAL;
    } else {
    const size_t tiles = (size_t)WAVES * (backward ? 2 : 1) * KN * PA;
    size_t smem = sizeof(float) * ((backward ? 2 : 1) * (size_t)D * PW + tiles);
    if (backward) smem = std::max(smem, sizeof(float) * (size_t)WAVES * D * D);
    PS_CHECK(smem <= 160 * 1024, "att_pool_train: %zu bytes of LDS needed", smem);
    const int per_cu = std::max(1, std::min(4, (int)(160 * 1024 / smem)));
    if (!backward) {
        auto kern = att_train_fwd_kernel<D, KN, WAVES>;
        if (smem > 48 * 1024) PS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        const int occ = std::min(per_cu, att_resident_blocks(reinterpret_cast<const void*>(kern), WAVES * 64, smem, per_cu));
        const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((a.R + WAVES - 1) / WAVES, 256 * occ));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(WAVES * 64), smem, c->stream, a);
    } else {
        auto kern = att_train_bwd_kernel<D, KN, WAVES>;
        if (smem > 48 * 1024) PS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        const int occ = std::min(per_cu, att_resident_blocks(reinterpret_cast<const void*>(kern), WAVES * 64, smem, per_cu));
        const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((a.R + WAVES - 1) / WAVES, 256 * occ));
        PS_TRY(c->red_ws.reserve(sizeof(float) * (size_t)blocks * D * D + 256));
        a.dw_part = c->red_ws.as<float>();
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(WAVES * 64), smem, c->stream, a);
        hipLaunchKernelGGL(reduce_partials_kernel<float>, dim3(ceil_div(D * D, 16)), dim3(256), 0, c->stream, static_cast<const float*>(a.dw_part), blocks, D * D, dW);
    }
    PS_HIP(hipGetLastError());
    return PS_OK;
    }
}

static bool att_train_ok(int64_t K, int64_t d, int64_t ld, const void* f)
{
    return K == 16 && (d == 16 || d == 32 || d == 64 || d == 128) && ld % 4 == 0 && (reinterpret_cast<uintptr_t>(f) & 15) == 0;
}

}  // namespace ps

using namespace ps;

extern "C" int ps_op_att_pool_train_supported(int64_t K, int64_t d) { return K == 16 && (d == 16 || d == 32 || d == 64) ? 1 : 0; }
extern "C" int ps_op_att_pool_train_supported_ex(int64_t K, int64_t d, int bf16_mode)
{
    return K == 16 && (d == 16 || d == 32 || d == 64 || (d == 128 && bf16_mode)) ? 1 : 0;
}

extern "C" int ps_op_att_pool_train_fwd(ps_context* c, const float* fset, int64_t ld, const float* wfc, int64_t R, int64_t K, int64_t d, float* agg)
{
    PS_CHECK(c && fset && wfc && agg, "ps_op_att_pool_train_fwd: NULL argument");
    PS_CHECK(att_train_ok(K, d, ld, fset), "ps_op_att_pool_train_fwd: K = 16, d in {16, 32, 64}, rows 16-byte aligned (got K %lld, d %lld, ld %lld)",
             (long long)K, (long long)d, (long long)ld);
    if (R <= 0) return PS_OK;
    PS_HIP(hipSetDevice(c->device));
    Stage st(c, "train_att_fused_fwd", 1);
    AttTrainArgs a = {};
    a.f = fset; a.w = wfc; a.agg = agg; a.R = R; a.ld = (int)ld; a.bf16 = c->train_bf16 ? 1 : 0;
    switch (d) {
        case 16: return launch_att_train<16>(c, a, false, nullptr);
        case 32: return launch_att_train<32>(c, a, false, nullptr);
        case 64: return launch_att_train<64>(c, a, false, nullptr);
        default: return launch_att_train<128>(c, a, false, nullptr);
    }
}

extern "C" int ps_op_att_pool_train_bwd(ps_context* c, const float* fset, int64_t ld, const float* wfc, const float* dagg, int64_t R, int64_t K, int64_t d,
                                        float* dfset, int64_t lddf, float* dwfc)
{
    PS_CHECK(c && fset && wfc && dagg && dfset && dwfc, "ps_op_att_pool_train_bwd: NULL argument");
    PS_CHECK(att_train_ok(K, d, ld, fset) && lddf >= d, "ps_op_att_pool_train_bwd: K = 16, d in {16, 32, 64}, rows 16-byte aligned");
    PS_HIP(hipSetDevice(c->device));
    if (R <= 0) {
        PS_HIP(hipMemsetAsync(dwfc, 0, sizeof(float) * d * d, c->stream));
        return PS_OK;
    }
    Stage st(c, "train_att_fused_bwd", 2);
    AttTrainArgs a = {};
    a.f = fset; a.w = wfc; a.dagg = dagg; a.df = dfset; a.R = R; a.ld = (int)ld; a.lddf = (int)lddf; a.bf16 = c->train_bf16 ? 1 : 0;
    switch (d) {
        case 16: return launch_att_train<16>(c, a, true, dwfc);
        case 32: return launch_att_train<32>(c, a, true, dwfc);
        case 64: return launch_att_train<64>(c, a, true, dwfc);
        default: return launch_att_train<128>(c, a, true, dwfc);
    }
}

/* split-source form: F = [fl[idx] | fr] is never materialised (see include/pointseg.h) */
extern "C" int ps_op_att_pool_train_fwd_split(ps_context* c, const float* fl, int64_t ldl, const int32_t* idx, int64_t B, int64_t n_src, int64_t n_q,
                                              const float* fr, int64_t ldr, const float* wfc, int64_t K, int64_t d, float* agg)
{
    PS_CHECK(c && fl && idx && fr && wfc && agg, "ps_op_att_pool_train_fwd_split: NULL argument");
    PS_CHECK(att_train_ok(K, d, ldr, fr) && ldl % 4 == 0 && (reinterpret_cast<uintptr_t>(fl) & 15) == 0 && n_src * ldl < (1ll << 31) && B >= 0 && n_src > 0 && n_q >= 0,
             "ps_op_att_pool_train_fwd_split: K = 16, d in {16, 32, 64}, rows 16-byte aligned (got K %lld, d %lld)", (long long)K, (long long)d);
    const int64_t R = B * n_q;
    if (R <= 0) return PS_OK;
    PS_HIP(hipSetDevice(c->device));
    Stage st(c, "train_att_fused_fwd", 1);
    AttTrainArgs a = {};
    a.f = fr; a.ld = (int)ldr; a.fl = fl; a.ldl = (int)ldl; a.idx = idx; a.n_src = n_src; a.n_q = n_q;
    a.w = wfc; a.agg = agg; a.R = R; a.bf16 = c->train_bf16 ? 1 : 0;
    a.fr_bf16 = c->train_act_bf16 && c->train_bf16 ? 1 : 0;  // (fr as bfloat16 rows: ps_set_train_act_bf16)
    switch (d) {
        case 16: return launch_att_train<16>(c, a, false, nullptr);
        case 32: return launch_att_train<32>(c, a, false, nullptr);
        case 64: return launch_att_train<64>(c, a, false, nullptr);
        default: return launch_att_train<128>(c, a, false, nullptr);
    }
}

static int att_bwd_split_impl(ps_context* c, const float* fl, int64_t ldl, const int32_t* idx, int64_t B, int64_t n_src, int64_t n_q, const float* fr,
                              int64_t ldr, const float* wfc, const float* dagg, int64_t K, int64_t d, float* dfl, int64_t lddl, float* dfr, int64_t lddr,
                              float* dwfc, float* dfl_rows, int64_t ld_rows);

extern "C" int ps_op_att_pool_train_bwd_split(ps_context* c, const float* fl, int64_t ldl, const int32_t* idx, int64_t B, int64_t n_src, int64_t n_q,
                                              const float* fr, int64_t ldr, const float* wfc, const float* dagg, int64_t K, int64_t d, float* dfl,
                                              int64_t lddl, float* dfr, int64_t lddr, float* dwfc)
{
    PS_CHECK(dfl, "ps_op_att_pool_train_bwd_split: NULL argument");
    return att_bwd_split_impl(c, fl, ldl, idx, B, n_src, n_q, fr, ldr, wfc, dagg, K, d, dfl, lddl, dfr, lddr, dwfc, nullptr, 0);
}

extern "C" int ps_op_att_pool_train_bwd_split_rows(ps_context* c, const float* fl, int64_t ldl, const int32_t* idx, int64_t B, int64_t n_src, int64_t n_q,
                                                   const float* fr, int64_t ldr, const float* wfc, const float* dagg, int64_t K, int64_t d,
                                                   float* dfl_rows, int64_t ld_rows, float* dfr, int64_t lddr, float* dwfc)
{
    PS_CHECK(dfl_rows && ld_rows >= d / 2, "ps_op_att_pool_train_bwd_split_rows: dfl_rows is NULL or its row stride below d/2");
    return att_bwd_split_impl(c, fl, ldl, idx, B, n_src, n_q, fr, ldr, wfc, dagg, K, d, dfl_rows, ld_rows, dfr, lddr, dwfc, dfl_rows, ld_rows);
}

static int att_bwd_split_impl(ps_context* c, const float* fl, int64_t ldl, const int32_t* idx, int64_t B, int64_t n_src, int64_t n_q, const float* fr,
                              int64_t ldr, const float* wfc, const float* dagg, int64_t K, int64_t d, float* dfl, int64_t lddl, float* dfr, int64_t lddr,
                              float* dwfc, float* dfl_rows, int64_t ld_rows)
{
    PS_CHECK(c && fl && idx && fr && wfc && dagg && dfl && dfr && dwfc, "ps_op_att_pool_train_bwd_split: NULL argument");
    PS_CHECK(att_train_ok(K, d, ldr, fr) && ldl % 4 == 0 && (reinterpret_cast<uintptr_t>(fl) & 15) == 0 && n_src * ldl < (1ll << 31) && lddl >= d / 2 && lddr >= d / 2 && n_src > 0,
             "ps_op_att_pool_train_bwd_split: K = 16, d in {16, 32, 64}, rows 16-byte aligned");
    PS_HIP(hipSetDevice(c->device));
    const int64_t R = B * n_q;
    if (R <= 0) {
        PS_HIP(hipMemsetAsync(dwfc, 0, sizeof(float) * d * d, c->stream));
        return PS_OK;
    }
    Stage st(c, "train_att_fused_bwd", 2);
    AttTrainArgs a = {};
    a.f = fr; a.ld = (int)ldr; a.fl = fl; a.ldl = (int)ldl; a.idx = idx; a.n_src = n_src; a.n_q = n_q;
    a.w = wfc; a.dagg = dagg; a.df = dfr; a.lddf = (int)lddr; a.dfl = dfl; a.lddl = (int)lddl; a.R = R; a.bf16 = c->train_bf16 ? 1 : 0;
    a.dfl_rows = dfl_rows; a.ld_rows = (int)ld_rows; a.df_accum = c->att_df_accum ? 1 : 0;
    a.fr_bf16 = c->train_act_bf16 && c->train_bf16 ? 1 : 0;  // (fr as bfloat16 rows: ps_set_train_act_bf16)
    switch (d) {
        case 16: return launch_att_train<16>(c, a, true, dwfc);
        case 32: return launch_att_train<32>(c, a, true, dwfc);
        case 64: return launch_att_train<64>(c, a, true, dwfc);
        default: return launch_att_train<128>(c, a, true, dwfc);
    }
}
