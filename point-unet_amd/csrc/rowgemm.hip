// rowgemm.hip -- see rowgemm.h.  Two kernels, both on v_mfma_f32_16x16x4_f32 with a 16-row x (NTB*16)-column
// output tile per wave:
//
//  rowgemm_direct_kernel (every layer whose channel counts are multiples of 16)
//     no LDS: each lane reads 16 contiguous input channels of its row with four 16-byte loads per 64-channel
//     chunk (next chunk prefetched into registers while the current one feeds 64 MFMAs); the B fragments come from
//     the k-permuted packed weights, one coalesced 4*NTB-byte load per lane per k-step, reused by RT row tiles.
//     SPLITK: for the deep levels (a few hundred rows, 256..1536 input channels) the four waves of a workgroup
//     take interleaved chunks of the SAME tile and reduce through LDS, which quadruples the waves in flight.
//  rowgemm_kernel (generic: level 0 / fc0 shapes such as 7->8, 10->8, 24->32)
//     A staged through a per-wave LDS tile (pitch 66 floats: conflict-free A-fragment reads).
//
// Bound: HBM for the wide-N / narrow-C layers (level 0, decoder tail, head), fp32 MFMA for the deep layers.
#include "rowgemm.h"

#include <algorithm>
#include <cstdlib>

#include "mfma_tile.h"

namespace ps {

constexpr int kKC = 64;      // channels per chunk
constexpr int kPitch = 66;   // LDS row pitch in floats (generic kernel)

struct RowGemmArgs {
    const float* x1; const int32_t* g1; int ld1, c1; int g1m, g1n;
    const float* x2; const int32_t* g2; int ld2, c2; int g2m, g2n;
    const float* wp; const float* bias;
    float* y; int ldy;
    int cin, cout, ks, R, leaky;
    int accum;  // y += act(x.W + b) instead of y = ...  (gradient accumulation of the training step)
};

// ---------------------------------------------------------------------------------------------------------------
template <int NTB, int RT, bool SPLITK>
__global__ __launch_bounds__(256) void rowgemm_direct_kernel(RowGemmArgs a)
{
    using bfrag = typename BFrag<NTB>::type;
    __shared__ float red[SPLITK ? 4 * NTB * 4 * 64 : 1];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int arow = lane & 15, g = lane >> 4;
    const int tile = SPLITK ? blockIdx.x : blockIdx.x * 4 + wave;
    const int row0 = tile * (16 * RT);
    if (!SPLITK && row0 >= a.R) return;
    const int cb = blockIdx.y;
    const int nchunks = (a.cin + 63) >> 6;

    // per-lane row base pointers (both sources), clamped rows for the tail
    const float* p1[RT];
    const float* p2[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int r = min(row0 + rt * 16 + arow, a.R - 1);
        const int s1 = a.g1 ? (a.g1m ? (r / a.g1m) * a.g1n : 0) + a.g1[r] : r;
        p1[rt] = a.x1 + (size_t)s1 * a.ld1;
        if (a.c2) {
            const int s2 = a.g2 ? (a.g2m ? (r / a.g2m) * a.g2n : 0) + a.g2[r] : r;
            p2[rt] = a.x2 + (size_t)s2 * a.ld2 - a.c1;  // indexed by the concatenated channel number
        } else
            p2[rt] = p1[rt];
    }
    auto load_chunk = [&](int c, float4 (&v)[RT][4]) {
        const int k = c * 64 + g * 16;  // this lane's 16 channels (one source: c1 % 16 == 0)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            if (k < a.cin) {
                const float4* src = reinterpret_cast<const float4*>((k < a.c1 ? p1[rt] : p2[rt]) + k);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[rt][i] = src[i];
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[rt][i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };

    f32x4 acc[RT][NTB];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int j = 0; j < NTB; ++j) acc[rt][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const bfrag* wq = reinterpret_cast<const bfrag*>(a.wp) + (size_t)cb * nchunks * 16 * 64 + lane;
    const int c_begin = SPLITK ? wave : 0, c_step = SPLITK ? 4 : 1;
    // (the epilogue's bias values requested in front of the K loop, as in gemm32.hip: one exposed round trip less per launch)
    float bias_r[NTB];
#pragma unroll
    for (int j = 0; j < NTB; ++j) {
        const int col = (cb * NTB + j) * 16 + (lane & 15);
        bias_r[j] = (!SPLITK && a.bias && col < a.cout) ? a.bias[col] : 0.f;
    }
    float4 cur[RT][4], nxt[RT][4];
    if (c_begin < nchunks) load_chunk(c_begin, cur);
    for (int c = c_begin; c < nchunks; c += c_step) {
        if (c + c_step < nchunks) load_chunk(c + c_step, nxt);
        const bfrag* w = wq + (size_t)c * 16 * 64;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bfrag b[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) b[q] = w[(size_t)(i * 4 + q) * 64];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const float av = q == 0 ? cur[rt][i].x : (q == 1 ? cur[rt][i].y : (q == 2 ? cur[rt][i].z : cur[rt][i].w));
#pragma unroll
                    for (int j = 0; j < NTB; ++j)
                        acc[rt][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bfrag_get<NTB>(b[q], j), acc[rt][j], 0, 0, 0);
                }
        }
        if (c + c_step < nchunks) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int i = 0; i < 4; ++i) cur[rt][i] = nxt[rt][i];
        }
    }

    if constexpr (SPLITK) {
        // (RT == 1) reduce the four partial tiles through LDS; wave w finishes n-tiles w, w+4, ...
        static_assert(!SPLITK || RT == 1, "split-K uses one row tile");
#pragma unroll
        for (int j = 0; j < NTB; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[((wave * NTB + j) * 4 + r) * 64 + lane] = acc[0][j][r];
        __syncthreads();
        for (int j = wave; j < NTB; j += 4) {
            const int col = (cb * NTB + j) * 16 + (lane & 15);
            if (col >= a.cout) continue;
            const float bb = a.bias ? a.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + g * 4 + r;
                if (row < a.R) {
                    float v = bb;
#pragma unroll
                    for (int w = 0; w < 4; ++w) v += red[((w * NTB + j) * 4 + r) * 64 + lane];
                    if (a.leaky) v = leaky02(v);
                    if (a.accum) v += a.y[(size_t)row * a.ldy + col];
                    a.y[(size_t)row * a.ldy + col] = v;
                }
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < NTB; ++j) {
            const int col = (cb * NTB + j) * 16 + (lane & 15);
            if (col >= a.cout) continue;
            const float bb = bias_r[j];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = row0 + rt * 16 + g * 4 + r;
                    if (row < a.R) {
                        float v = acc[rt][j][r] + bb;
                        if (a.leaky) v = leaky02(v);
                        if (a.accum) v += a.y[(size_t)row * a.ldy + col];
                        a.y[(size_t)row * a.ldy + col] = v;
                    }
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// bf16 flavour of the direct-load kernel (training step, "bf16 MLPs"): same row ownership and epilogue; a lane's 16 channels of a
// chunk are two 8-wide k-groups of v_mfma_f32_16x16x32_bf16 (lane (row, g) supplies k = 8g..8g+7), rounded to bf16 (RNE,
// v_cvt_pk_bf16_f32) as they leave the load registers; the weights come pre-rounded in the matching order (PackedLinear::wb).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ bf16x8 to_bf16x8(const float4& a, const float4& b)
{
    const bf16x2 p0 = __builtin_convertvector(f32x2v{a.x, a.y}, bf16x2), p1 = __builtin_convertvector(f32x2v{a.z, a.w}, bf16x2);
    const bf16x2 p2 = __builtin_convertvector(f32x2v{b.x, b.y}, bf16x2), p3 = __builtin_convertvector(f32x2v{b.z, b.w}, bf16x2);
    bf16x8 r;
    r[0] = p0[0]; r[1] = p0[1]; r[2] = p1[0]; r[3] = p1[1]; r[4] = p2[0]; r[5] = p2[1]; r[6] = p3[0]; r[7] = p3[1];
    return r;
}

template <int NTB, int RT>
__global__ __launch_bounds__(256) void rowgemm_direct_bf16_kernel(RowGemmArgs a)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int arow = lane & 15, g = lane >> 4;
    const int tile = blockIdx.x * 4 + wave;
    const int row0 = tile * (16 * RT);
    if (row0 >= a.R) return;
    const int cb = blockIdx.y;
    const int nchunks = (a.cin + 63) >> 6;
    const float* p1[RT];
    const float* p2[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int r = min(row0 + rt * 16 + arow, a.R - 1);
        const int s1 = a.g1 ? (a.g1m ? (r / a.g1m) * a.g1n : 0) + a.g1[r] : r;
        p1[rt] = a.x1 + (size_t)s1 * a.ld1;
        if (a.c2) {
            const int s2 = a.g2 ? (a.g2m ? (r / a.g2m) * a.g2n : 0) + a.g2[r] : r;
            p2[rt] = a.x2 + (size_t)s2 * a.ld2 - a.c1;
        } else
            p2[rt] = p1[rt];
    }
    auto load_chunk = [&](int c, float4 (&v)[RT][4]) {
        const int k = c * 64 + g * 16;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            if (k < a.cin) {
                const float4* src = reinterpret_cast<const float4*>((k < a.c1 ? p1[rt] : p2[rt]) + k);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[rt][i] = src[i];
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[rt][i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    f32x4 acc[RT][NTB];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int j = 0; j < NTB; ++j) acc[rt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bf16x8* wb = reinterpret_cast<const bf16x8*>(a.wp) + ((size_t)cb * nchunks * 2 * 64 + lane) * NTB;
    float4 cur[RT][4], nxt[RT][4];
    load_chunk(0, cur);
    for (int c = 0; c < nchunks; ++c) {
        if (c + 1 < nchunks) load_chunk(c + 1, nxt);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 b[NTB];
#pragma unroll
            for (int j = 0; j < NTB; ++j) b[j] = wb[((size_t)(c * 2 + s) * 64) * NTB + j];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const bf16x8 av = to_bf16x8(cur[rt][2 * s], cur[rt][2 * s + 1]);
#pragma unroll
                for (int j = 0; j < NTB; ++j) acc[rt][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b[j], acc[rt][j], 0, 0, 0);
            }
        }
        if (c + 1 < nchunks) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int i = 0; i < 4; ++i) cur[rt][i] = nxt[rt][i];
        }
    }
#pragma unroll
    for (int j = 0; j < NTB; ++j) {
        const int col = (cb * NTB + j) * 16 + (lane & 15);
        if (col >= a.cout) continue;
        const float bb = a.bias ? a.bias[col] : 0.f;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + rt * 16 + g * 4 + r;
                if (row < a.R) {
                    float v = acc[rt][j][r] + bb;
                    if (a.leaky) v = leaky02(v);
                    if (a.accum) v += a.y[(size_t)row * a.ldy + col];
                    a.y[(size_t)row * a.ldy + col] = v;
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Streaming flavour for the training step's [N*K, d] GEMMs (millions of rows, <= 128 input channels, one plain source): these
// are HBM-bound, and the one-shot kernels above (a wave loads 32 rows, multiplies, stores, exits) leave the loads exposed.
// Here a workgroup keeps the column block's packed weights in LDS and its waves walk the row tiles persistently with the
// NEXT tile's rows (and, when accumulating, this tile's old output) in flight under the current tile's MFMAs and stores.
template <int NTB, bool BF16>
__global__ __launch_bounds__(256) void rowgemm_stream_kernel(RowGemmArgs a)
{
    using bfrag = typename BFrag<NTB>::type;
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    __shared__ __attribute__((aligned(16))) float wl[2 * 16 * 64 * 4];  // 32 KiB: fp32 image of <= 128 channels x 64 columns
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int arow = lane & 15, g = lane >> 4;
    const int cb = blockIdx.y;
    const int nchunks = (a.cin + 63) >> 6;  // 1 or 2
    {
        // this column block's weights: fp32 [nchunks][16][64][NTB] floats, bf16 [nchunks][2][64][NTB] x 8 bf16 (= 4 floats)
        const int nf = BF16 ? nchunks * 2 * 64 * NTB * 4 : nchunks * 16 * 64 * NTB;
        const float4* src = reinterpret_cast<const float4*>(a.wp + (size_t)cb * nf);
        for (int i = threadIdx.x; i < nf / 4; i += 256) reinterpret_cast<float4*>(wl)[i] = src[i];
    }
    __syncthreads();
    const int ntiles = (a.R + 15) >> 4;
    const int stride = gridDim.x * 4;
    auto load_tile = [&](int t, float4 (&v)[2][4]) {
        const int r = min(t * 16 + arow, a.R - 1);
        const float* p = a.x1 + (size_t)r * a.ld1 + g * 16;
#pragma unroll
        for (int c = 0; c < 2; ++c)
            if (c < nchunks) {
                if (c * 64 + g * 16 < a.cin) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[c][i] = reinterpret_cast<const float4*>(p + c * 64)[i];
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[c][i] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
    };
    float bias[NTB];
#pragma unroll
    for (int j = 0; j < NTB; ++j) {
        const int col = (cb * NTB + j) * 16 + (lane & 15);
        bias[j] = (a.bias && col < a.cout) ? a.bias[col] : 0.f;
    }
    int t = blockIdx.x * 4 + wave;
    float4 cur[2][4], nxt[2][4];
    if (t < ntiles) load_tile(t, cur);
    for (; t < ntiles; t += stride) {
        const int row0 = t * 16;
        float old[NTB][4];
        if (a.accum) {
#pragma unroll
            for (int j = 0; j < NTB; ++j) {
                const int col = (cb * NTB + j) * 16 + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = row0 + g * 4 + r;
                    old[j][r] = (row < a.R && col < a.cout) ? a.y[(size_t)row * a.ldy + col] : 0.f;
                }
            }
        }
        if (t + stride < ntiles) load_tile(t + stride, nxt);
        f32x4 acc[NTB];
#pragma unroll
        for (int j = 0; j < NTB; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 2; ++c)
            if (c < nchunks) {
                if constexpr (BF16) {
                    const bf16x8* w = reinterpret_cast<const bf16x8*>(wl) + ((size_t)c * 2 * 64 + lane) * NTB;
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        const bf16x8 av = to_bf16x8(cur[c][2 * s2], cur[c][2 * s2 + 1]);
#pragma unroll
                        for (int j = 0; j < NTB; ++j)
                            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, w[(size_t)s2 * 64 * NTB + j], acc[j], 0, 0, 0);
                    }
                } else {
                    const bfrag* w = reinterpret_cast<const bfrag*>(wl) + (size_t)c * 16 * 64 + lane;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const bfrag b = w[(size_t)(i * 4 + q) * 64];
                            const float av = q == 0 ? cur[c][i].x : (q == 1 ? cur[c][i].y : (q == 2 ? cur[c][i].z : cur[c][i].w));
#pragma unroll
                            for (int j = 0; j < NTB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bfrag_get<NTB>(b, j), acc[j], 0, 0, 0);
                        }
                }
            }
#pragma unroll
        for (int j = 0; j < NTB; ++j) {
            const int col = (cb * NTB + j) * 16 + (lane & 15);
            if (col >= a.cout) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + g * 4 + r;
                if (row < a.R) {
                    float v = acc[j][r] + bias[j];
                    if (a.leaky) v = leaky02(v);
                    if (a.accum) v += old[j][r];
                    a.y[(size_t)row * a.ldy + col] = v;
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) cur[c][i] = nxt[c][i];
    }
}

// ---------------------------------------------------------------------------------------------------------------
template <int NTB>
__global__ __launch_bounds__(256) void rowgemm_kernel(RowGemmArgs a)
{
    __shared__ float lds[4][16 * kPitch];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* tile = lds[wave];
    const int row0 = (blockIdx.x * 4 + wave) * 16;
    if (row0 >= a.R) return;  // whole wave out of range (no block-level barriers below)
    const int cb = blockIdx.y;

    // source row numbers of this wave's 16 rows (lane i < 16 resolves row i), broadcast later by shuffle
    int src1 = 0, src2 = 0;
    {
        const int r = row0 + (lane & 15);
        if (r < a.R) {
            src1 = a.g1 ? (a.g1m ? (r / a.g1m) * a.g1n : 0) + a.g1[r] : r;
            src2 = a.g2 ? (a.g2m ? (r / a.g2m) * a.g2n : 0) + a.g2[r] : r;
        }
    }

    f32x4 acc[NTB];
#pragma unroll
    for (int j = 0; j < NTB; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};

    using bfrag = typename BFrag<NTB>::type;
    const bfrag* wp = reinterpret_cast<const bfrag*>(a.wp) + (size_t)cb * a.ks * 64 + lane;

    const int arow = lane & 15, ag = lane >> 4;
    for (int k0 = 0; k0 < a.cin; k0 += kKC) {
        const int kc = min(kKC, a.cin - k0);
        const int kc4 = (kc + 3) & ~3;
        // ---- stage A[16][kc4] (zero padded) ----
        for (int e = lane; e < 16 * kc4; e += 64) {
            const int r = e / kc4, k = e - r * kc4;
            const int kg = k0 + k;
            // (16*kc4 is a multiple of 64, so every lane is active here: the shuffles are convergent)
            const int sr1 = __shfl(src1, r), sr2 = __shfl(src2, r);
            float v = 0.f;
            if (row0 + r < a.R && k < kc)
                v = kg < a.c1 ? a.x1[(size_t)sr1 * a.ld1 + kg] : a.x2[(size_t)sr2 * a.ld2 + (kg - a.c1)];
            tile[r * kPitch + k] = v;
        }
        wave_lds_sync();
        // ---- MFMA over the chunk ----
        const int steps = kc4 >> 2;
        const bfrag* wps = wp + (size_t)(k0 >> 2) * 64;
        for (int s = 0; s < steps; ++s) {  // (run-time trip count: an unroll request here is refused by the compiler -- and warned about)
            const float av = tile[arow * kPitch + s * 4 + ag];
            const bfrag bv = wps[(size_t)s * 64];
#pragma unroll
            for (int j = 0; j < NTB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bfrag_get<NTB>(bv, j), acc[j], 0, 0, 0);
        }
        wave_lds_sync();
    }

    // ---- epilogue: C layout col = lane&15, row = (lane>>4)*4 + r ----
#pragma unroll
    for (int j = 0; j < NTB; ++j) {
        const int col = (cb * NTB + j) * 16 + (lane & 15);
        if (col >= a.cout) continue;
        const float b = a.bias ? a.bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = row0 + (lane >> 4) * 4 + r;
            if (row < a.R) {
                float v = acc[j][r] + b;
                if (a.leaky) v = leaky02(v);
                if (a.accum) v += a.y[(size_t)row * a.ldy + col];
                a.y[(size_t)row * a.ldy + col] = v;
            }
        }
    }
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ---------------------------------------------------------------------------------------------------------------
// rowchain: see rowgemm.h.  One wave per 16-row tile; two LDS tiles per wave (pitch = 2 mod 32: conflict-free A-fragment
// reads) that the layers ping-pong between.
struct ChainArgs {
    const float* x1; const int32_t* g1; int ld1, c1; int g1m, g1n;
    const float* x2; const int32_t* g2; int ld2, c2; int g2m, g2n;
    int n, R;
    struct Lyr {
        const float* wp; const float* bias; float* y; const float* ex;
        int cout, ks, ntb, cblocks, leaky, ldy, ldex, cex;
        int w_off, w_floats, b_off, b_floats;  // this layer's packed weights / bias inside the workgroup's LDS image
    } l[kChainMaxSteps];
    int tile_off;  // float offset of the activation tiles behind the weight image
    int fast_in;   // log2(float4s per input row) when the float4 staging path applies, else 0
};
constexpr int kChainPitch = kChainMaxC + 2;

template <int NTB>
__device__ __forceinline__ void chain_layer(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ wl,
                                            const float* __restrict__ bl, const ChainArgs::Lyr& y, int row0, int R, int lane)
{
    using bfrag = typename BFrag<NTB>::type;
    const int arow = lane & 15, ag = lane >> 4;
    for (int cb = 0; cb < y.cblocks; ++cb) {
        f32x4 acc[NTB];
#pragma unroll
        for (int j = 0; j < NTB; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const bfrag* wp = reinterpret_cast<const bfrag*>(wl) + (size_t)cb * y.ks * 64 + lane;  // B fragments from LDS
        const float* ap = in + arow * kChainPitch + ag;
        int s = 0;
        for (; s + 4 <= y.ks; s += 4) {  // four k-steps' operands in flight before the first MFMA needs them
            float av[4];
            bfrag bv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { av[u] = ap[(s + u) * 4]; bv[u] = wp[(size_t)(s + u) * 64]; }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < NTB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bfrag_get<NTB>(bv[u], j), acc[j], 0, 0, 0);
        }
        for (; s < y.ks; ++s) {
            const float av = ap[s * 4];
            const bfrag bv = wp[(size_t)s * 64];
#pragma unroll
            for (int j = 0; j < NTB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bfrag_get<NTB>(bv, j), acc[j], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NTB; ++j) {
            const int col = (cb * NTB + j) * 16 + (lane & 15);
            if (col >= y.cout) continue;
            const float b = bl[col];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rr = ag * 4 + r;
                float v = acc[j][r] + b;
                if (y.leaky) v = leaky02(v);
                out[rr * kChainPitch + col] = v;
                if (y.y && row0 + rr < R) y.y[(size_t)(row0 + rr) * y.ldy + col] = v;
            }
        }
    }
}

__global__ __launch_bounds__(256) void rowchain_kernel(ChainArgs a)
{
    // LDS: [packed weights + biases of every layer | 4 waves x 2 activation tiles].  The weights are read once per
    // workgroup (a few tens of KB) and serve every row tile it processes: B fragments then cost an LDS read instead of
    // an exposed L2 round trip per k-step.
    extern __shared__ __attribute__((aligned(16))) float chain_lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int l = 0; l < a.n; ++l) {
        const ChainArgs::Lyr& y = a.l[l];
        for (int i = threadIdx.x; i < y.w_floats; i += 256) chain_lds[y.w_off + i] = y.wp[i];
        for (int i = threadIdx.x; i < y.b_floats; i += 256) chain_lds[y.b_off + i] = y.bias[i];
    }
    __syncthreads();
    float* const tiles = chain_lds + a.tile_off + wave * (2 * 16 * kChainPitch);
    for (int tile = blockIdx.x * 4 + wave; tile * 16 < a.R; tile += gridDim.x * 4) {
        const int row0 = tile * 16;
        float* in = tiles;
        float* out = tiles + 16 * kChainPitch;
        // ---- the chain's input rows: [s1 | s2], gathered ----
        int src1 = 0, src2 = 0;
        {
            const int r = row0 + (lane & 15);
            if (r < a.R) {
                src1 = a.g1 ? (a.g1m ? (r / a.g1m) * a.g1n : 0) + a.g1[r] : r;
                src2 = a.g2 ? (a.g2m ? (r / a.g2m) * a.g2n : 0) + a.g2[r] : r;
            }
        }
        int cur = a.c1 + a.c2;
        if (a.fast_in) {
            // both sources are multiples of 4 channels, 16-byte aligned rows, and a row is a power-of-two number of float4s:
            // float4 loads, shifts instead of divisions, all of a lane's loads in flight together
            const int q_shift = a.fast_in, qn = 1 << q_shift;  // float4s per row
            const int c1q = a.c1 >> 2;
            for (int f0 = 0; f0 < 16 * qn; f0 += 4 * 64) {
                float4 v[4];
                int at[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int f = f0 + u * 64 + lane;
                    const bool on = f < 16 * qn;
                    const int r = on ? f >> q_shift : 0, q = f & (qn - 1);
                    const int sr1 = __shfl(src1, r), sr2 = __shfl(src2, r);
                    at[u] = on ? r * kChainPitch + 4 * q : -1;
                    v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (on && row0 + r < a.R)
                        v[u] = q < c1q ? *reinterpret_cast<const float4*>(a.x1 + (size_t)sr1 * a.ld1 + 4 * q)
                                       : *reinterpret_cast<const float4*>(a.x2 + (size_t)sr2 * a.ld2 + 4 * (q - c1q));
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (at[u] >= 0) {
                        float* d = in + at[u];  // (pitch 98: rows are only 8-byte aligned)
                        *reinterpret_cast<float2*>(d) = make_float2(v[u].x, v[u].y);
                        *reinterpret_cast<float2*>(d + 2) = make_float2(v[u].z, v[u].w);
                    }
            }
        } else {
            // eight loads in flight per lane before the first LDS store (a load-store-per-iteration loop would expose one
            // HBM round trip per element); 16 * kc4 is a multiple of 64, so every lane is active at the shuffles
            const int kc4 = (cur + 3) & ~3;
            for (int e0 = 0; e0 < 16 * kc4; e0 += 8 * 64) {
                float v[8];
                int at[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int e = e0 + u * 64 + lane;
                    const bool on = e < 16 * kc4;  // (wave-uniform per u)
                    const int r = on ? e / kc4 : 0, k = on ? e - r * kc4 : 0;
                    const int sr1 = __shfl(src1, r), sr2 = __shfl(src2, r);
                    v[u] = 0.f;
                    at[u] = on ? r * kChainPitch + k : -1;
                    if (on && row0 + r < a.R && k < cur) v[u] = k < a.c1 ? a.x1[(size_t)sr1 * a.ld1 + k] : a.x2[(size_t)sr2 * a.ld2 + (k - a.c1)];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (at[u] >= 0) in[at[u]] = v[u];
            }
        }
        for (int l = 0; l < a.n; ++l) {
            const ChainArgs::Lyr& y = a.l[l];
            if (y.ex) {  // appended plain-row source (+ zero padding of the K axis to a multiple of 4)
                const int w4 = ((cur + y.cex + 3) & ~3) - cur;
                for (int e0 = 0; e0 < 16 * w4; e0 += 8 * 64) {
                    float v[8];
                    int at[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int e = e0 + u * 64 + lane;
                        const bool on = e < 16 * w4;
                        const int r = on ? e / w4 : 0, k = on ? e - r * w4 : 0;
                        at[u] = on ? r * kChainPitch + cur + k : -1;
                        v[u] = (on && row0 + r < a.R && k < y.cex) ? y.ex[(size_t)(row0 + r) * y.ldex + k] : 0.f;
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (at[u] >= 0) in[at[u]] = v[u];
                }
                cur += y.cex;
            } else if (l > 0 && (cur & 3)) {
                for (int e = lane; e < 16 * (4 - (cur & 3)); e += 64) {
                    const int w = 4 - (cur & 3), r = e / w, k = e - r * w;
                    in[r * kChainPitch + cur + k] = 0.f;
                }
            }
            wave_lds_sync();
            switch (y.ntb) {
                case 1: chain_layer<1>(in, out, chain_lds + y.w_off, chain_lds + y.b_off, y, row0, a.R, lane); break;
                case 2: chain_layer<2>(in, out, chain_lds + y.w_off, chain_lds + y.b_off, y, row0, a.R, lane); break;
                default: chain_layer<4>(in, out, chain_lds + y.w_off, chain_lds + y.b_off, y, row0, a.R, lane); break;
            }
            wave_lds_sync();
            float* t = in; in = out; out = t;
            cur = y.cout;
        }
    }
}

// (the network's own chain shapes run in regchain.hip, 1.2-2.4x faster than the LDS-staged rowchain_kernel below, which stays as the
//  form for every other shape; both are parity-tested)

static bool lds_chain_fits(const ChainStep* steps, int n_steps, const RowSrc& s1, const RowSrc& s2);

bool rowchain_fits(const ChainStep* steps, int n_steps, const RowSrc& s1, const RowSrc& s2)
{
    return regchain_fits(steps, n_steps, s1, s2) || lds_chain_fits(steps, n_steps, s1, s2);
}

static bool lds_chain_fits(const ChainStep* steps, int n_steps, const RowSrc& s1, const RowSrc& s2)
{
    if (n_steps < 1 || n_steps > kChainMaxSteps) return false;
    int cur = s1.c + s2.c;
    for (int i = 0; i < n_steps; ++i) {
        const PackedLinear* L = steps[i].L;
        if (!L || !L->wp) return false;
        cur += steps[i].extra.x ? steps[i].extra.c : 0;
        if (cur != L->cin || ((cur + 3) & ~3) > kChainMaxC || L->cout > kChainMaxC) return false;
        if (steps[i].extra.x && steps[i].extra.gather) return false;
        cur = L->cout;
    }
    return true;
}

int rowchain(ps_context* c, const ChainStep* steps, int n_steps, const RowSrc& s1, const RowSrc& s2, int64_t R, ChainCache* cache)
{
    if (R <= 0) return PS_OK;
    if (regchain_fits(steps, n_steps, s1, s2)) return regchain(c, steps, n_steps, s1, s2, R, cache);
    PS_CHECK(lds_chain_fits(steps, n_steps, s1, s2), "rowchain: the layer chain does not fit (channels above %d or mismatched)", kChainMaxC);
    PS_CHECK(R < (int64_t)1 << 31, "rowchain: too many rows");
    ChainArgs a = {};
    a.x1 = s1.x; a.g1 = s1.gather; a.ld1 = s1.ld; a.c1 = s1.c; a.g1m = s1.gm; a.g1n = s1.gn;
    a.x2 = s2.x; a.g2 = s2.gather; a.ld2 = s2.ld; a.c2 = s2.c; a.g2m = s2.gm; a.g2n = s2.gn;
    a.n = n_steps;
    a.R = (int)R;
    int off = 0;
    for (int i = 0; i < n_steps; ++i) {
        const PackedLinear& L = *steps[i].L;
        ChainArgs::Lyr& y = a.l[i];
        y.wp = L.wp; y.bias = L.bias; y.y = steps[i].y; y.ldy = steps[i].ldy;
        y.ex = steps[i].extra.x; y.ldex = steps[i].extra.ld; y.cex = steps[i].extra.x ? steps[i].extra.c : 0;
        y.cout = L.cout; y.ks = L.ks; y.ntb = L.ntb; y.cblocks = L.cblocks; y.leaky = L.leaky;
        y.w_off = off; y.w_floats = (int)L.packed_floats(); off += (y.w_floats + 3) & ~3;
        y.b_off = off; y.b_floats = L.cout_pad(); off += (y.b_floats + 3) & ~3;
    }
    a.tile_off = off;
    {
        const int cin0 = s1.c + s2.c, q = cin0 / 4;
        const bool pow2 = cin0 % 4 == 0 && q >= 2 && (q & (q - 1)) == 0;
        const bool al = s1.c % 4 == 0 && s2.c % 4 == 0 && s1.ld % 4 == 0 && (s2.c == 0 || s2.ld % 4 == 0) && aligned16(s1.x) && (s2.c == 0 || aligned16(s2.x));
        a.fast_in = 0;
        if (pow2 && al)
            while ((1 << a.fast_in) < q) ++a.fast_in;
    }
    const size_t lds_bytes = sizeof(float) * ((size_t)off + 4 * 2 * 16 * kChainPitch);
    PS_CHECK(lds_bytes <= 160 * 1024, "rowchain: weights of the chain do not fit the LDS (%zu bytes)", lds_bytes);
    if (lds_bytes > c->chain_lds_attr) {  // (per context = per device)
        PS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(rowchain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        c->chain_lds_attr = lds_bytes;
    }
    // few, long-lived workgroups: the weight image is loaded once per workgroup
    const int tiles = (int)((R + 15) / 16);
    const int per_cu = std::max(1, (int)(160 * 1024 / lds_bytes));
    const int blocks = std::max(1, std::min((tiles + 3) / 4, 256 * std::min(per_cu, 4)));
    hipLaunchKernelGGL(rowchain_kernel, dim3(blocks), dim3(256), lds_bytes, c->stream, a);
    PS_HIP(hipGetLastError());
    return PS_OK;
}

void pack_weights(const float* W, int cin, int cout, int ntb, float* out)
{
    const int ks = (cin + 3) / 4;
    const int cblocks = (cout + 16 * ntb - 1) / (16 * ntb);
    for (int cb = 0; cb < cblocks; ++cb)
        for (int s = 0; s < ks; ++s)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < ntb; ++j) {
                    const int k = s * 4 + (l >> 4);
                    const int col = (cb * ntb + j) * 16 + (l & 15);
                    const float v = (k < cin && col < cout) ? W[(size_t)k * cout + col] : 0.f;
                    out[(((size_t)cb * ks + s) * 64 + l) * ntb + j] = v;
                }
}

void pack_weights_kperm(const float* W, int cin, int cout, int ntb, float* out)
{
    const int nc = (cin + 63) / 64;
    const int cblocks = (cout + 16 * ntb - 1) / (16 * ntb);
    for (int cb = 0; cb < cblocks; ++cb)
        for (int c = 0; c < nc; ++c)
            for (int s = 0; s < 16; ++s)
                for (int l = 0; l < 64; ++l)
                    for (int j = 0; j < ntb; ++j) {
                        const int k = c * 64 + 16 * (l >> 4) + s;
                        const int col = (cb * ntb + j) * 16 + (l & 15);
                        const float v = (k < cin && col < cout) ? W[(size_t)k * cout + col] : 0.f;
                        out[((((size_t)cb * nc + c) * 16 + s) * 64 + l) * ntb + j] = v;
                    }
}


int rowgemm(ps_context* c, const PackedLinear& L, const RowSrc& s1, const RowSrc& s2, int64_t R, float* y, int ldy)
{
    if (R <= 0) return PS_OK;
    PS_CHECK(s1.c + s2.c == L.cin, "rowgemm: sources give %d channels, layer expects %d", s1.c + s2.c, L.cin);
    if (c->att_bf16x3 && 2.0 * (double)R * L.cin * L.cout >= c->tune.gemm32b_min_flops && gemm32b_fits(L, s1, s2, R, ldy))
        return gemm32b(c, L, s1, s2, R, y, ldy);                                  // ... the large ones on split-bf16 MFMA (gemm32b.hip)
    if (gemm32_fits(L, s1, s2, R, ldy)) return gemm32(c, L, s1, s2, R, y, ldy);  // deep levels: 32x32x2 tiles (gemm32.hip)
    PS_CHECK(R < (1ll << 31), "rowgemm: too many rows");
    RowGemmArgs a;
    a.x1 = s1.x; a.g1 = s1.gather; a.ld1 = s1.ld; a.c1 = s1.c; a.g1m = s1.gm; a.g1n = s1.gn;
    a.x2 = s2.x; a.g2 = s2.gather; a.ld2 = s2.ld; a.c2 = s2.c; a.g2m = s2.gm; a.g2n = s2.gn;
    a.wp = L.wp; a.bias = L.bias;
    a.y = y; a.ldy = ldy;
    a.cin = L.cin; a.cout = L.cout; a.ks = L.ks; a.R = (int)R; a.leaky = L.leaky;
    a.accum = L.accum;

    const bool direct = (L.wq || L.wb) && L.cin % 16 == 0 && s1.c % 16 == 0 && s1.ld % 4 == 0 && aligned16(s1.x) &&
                        (s2.c == 0 || (s2.ld % 4 == 0 && aligned16(s2.x)));
    // the training step's [N*K, d] GEMMs: persistent streaming kernel (never reached by inference-sized calls)
    const bool stream = direct && !s1.gather && s2.c == 0 && L.cin <= 128 && ((R + 15) / 16) * L.cblocks >= 32768;
    if (stream) {
        a.wp = L.wb ? static_cast<const float*>(L.wb) : L.wq;
        const dim3 grid(1024, L.cblocks);  // 4 workgroups per CU, every wave walks ~(tiles / 4096) row tiles
#define PS_STREAM(NTB)                                                                                                      \
    if (L.wb) hipLaunchKernelGGL((rowgemm_stream_kernel<NTB, true>), grid, dim3(256), 0, c->stream, a);                       \
    else hipLaunchKernelGGL((rowgemm_stream_kernel<NTB, false>), grid, dim3(256), 0, c->stream, a)
        switch (L.ntb) {
            case 1: PS_STREAM(1); break;
            case 2: PS_STREAM(2); break;
            case 4: PS_STREAM(4); break;
            default: set_error("rowgemm: bad ntb %d", L.ntb); return PS_EINVAL;
        }
#undef PS_STREAM
    } else if (direct && L.wb) {
        a.wp = static_cast<const float*>(L.wb);
        const int64_t tiles16 = (R + 15) / 16;
        const bool rt2 = tiles16 * L.cblocks >= 8192;
#define PS_LAUNCH_BF(NTB, RT, GX) hipLaunchKernelGGL((rowgemm_direct_bf16_kernel<NTB, RT>), dim3((unsigned)(GX), L.cblocks), dim3(256), 0, c->stream, a)
#define PS_BF_BY_NTB(RT, GX)                                                 \
    switch (L.ntb) {                                                         \
        case 1: PS_LAUNCH_BF(1, RT, GX); break;                              \
        case 2: PS_LAUNCH_BF(2, RT, GX); break;                              \
        case 4: PS_LAUNCH_BF(4, RT, GX); break;                              \
        default: set_error("rowgemm: bad ntb %d", L.ntb); return PS_EINVAL;  \
    }
        if (rt2) {
            PS_BF_BY_NTB(2, (tiles16 + 7) / 8)
        } else {
            PS_BF_BY_NTB(1, (tiles16 + 3) / 4)
        }
#undef PS_BF_BY_NTB
#undef PS_LAUNCH_BF
    } else if (direct) {
        a.wp = L.wq;
        const int64_t tiles16 = (R + 15) / 16;
        const bool splitk = L.cin >= 256 && tiles16 * L.cblocks < 2048;
        const bool rt2 = !splitk && tiles16 * L.cblocks >= 8192;
#define PS_LAUNCH(NTB, RT, SK, GX)                                                                                             \
    hipLaunchKernelGGL((rowgemm_direct_kernel<NTB, RT, SK>), dim3((unsigned)(GX), L.cblocks), dim3(256), 0, c->stream, a)
#define PS_BY_NTB(RT, SK, GX)                                    \
    switch (L.ntb) {                                             \
        case 1: PS_LAUNCH(1, RT, SK, GX); break;                 \
        case 2: PS_LAUNCH(2, RT, SK, GX); break;                 \
        case 4: PS_LAUNCH(4, RT, SK, GX); break;                 \
        default: set_error("rowgemm: bad ntb %d", L.ntb); return PS_EINVAL; \
    }
        if (splitk) {
            PS_BY_NTB(1, true, tiles16)
        } else if (rt2) {
            PS_BY_NTB(2, false, (tiles16 + 7) / 8)
        } else {
            PS_BY_NTB(1, false, (tiles16 + 3) / 4)
        }
#undef PS_BY_NTB
#undef PS_LAUNCH
    } else {
        dim3 grid(ceil_div(R, 64), L.cblocks);
        switch (L.ntb) {
            case 1: hipLaunchKernelGGL(rowgemm_kernel<1>, grid, dim3(256), 0, c->stream, a); break;
            case 2: hipLaunchKernelGGL(rowgemm_kernel<2>, grid, dim3(256), 0, c->stream, a); break;
            case 4: hipLaunchKernelGGL(rowgemm_kernel<4>, grid, dim3(256), 0, c->stream, a); break;
            default: set_error("rowgemm: bad ntb %d", L.ntb); return PS_EINVAL;
        }
    }
    PS_HIP(hipGetLastError());
    return PS_OK;
}

}  // namespace ps
