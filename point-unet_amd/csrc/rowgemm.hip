// rowgemm.hip -- see rowgemm.h.  One wave owns a 16-row x (NTB*16)-column output tile:
//   A (16 rows x 64-channel chunk) is staged row-major into a per-wave LDS tile with a 66-float row pitch, so the
//     MFMA A-fragment read  A[lane&15][4s + (lane>>4)]  hits 32 distinct banks (pitch = 2 mod 32);
//   B comes straight from the host-packed fragment-order weights: one coalesced 4*NTB-byte load per lane per
//     k-step feeds NTB MFMAs (weights are tiny and stay L1/L2 resident);
//   the epilogue adds the bias, applies LeakyReLU(0.2) and stores 64-byte row segments.
// Bound: HBM for the wide-N / narrow-C layers (level 0, decoder tail, head), fp32 MFMA for the deep layers.
#include "rowgemm.h"

namespace ps {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kKC = 64;      // channels staged per chunk
constexpr int kPitch = 66;   // LDS row pitch in floats

// LDS hand-off between the lanes of ONE wave: DS operations of a wave execute in order, so only the compiler
// has to be kept from moving LDS accesses across this point.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int NTB>
struct BFrag;
template <>
struct BFrag<1> { using type = float; };
template <>
struct BFrag<2> { using type = float2; };
template <>
struct BFrag<4> { using type = float4; };

struct RowGemmArgs {
    const float* x1; const int32_t* g1; int ld1, c1; int g1m, g1n;
    const float* x2; const int32_t* g2; int ld2, c2; int g2m, g2n;
    const float* wp; const float* bias;
    float* y; int ldy;
    int cin, cout, ks, R, leaky;
};

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
#pragma unroll 4
        for (int s = 0; s < steps; ++s) {
            const float av = tile[arow * kPitch + s * 4 + ag];
            const bfrag bv = wps[(size_t)s * 64];
            if constexpr (NTB == 1) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[0], 0, 0, 0);
            } else if constexpr (NTB == 2) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv.x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv.y, acc[1], 0, 0, 0);
            } else {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv.x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv.y, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv.z, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv.w, acc[3], 0, 0, 0);
            }
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
                if (a.leaky) v = v >= 0.f ? v : 0.2f * v;
                a.y[(size_t)row * a.ldy + col] = v;
            }
        }
    }
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

int rowgemm(ps_context* c, const PackedLinear& L, const RowSrc& s1, const RowSrc& s2, int64_t R, float* y, int ldy)
{
    if (R <= 0) return PS_OK;
    PS_CHECK(s1.c + s2.c == L.cin, "rowgemm: sources give %d channels, layer expects %d", s1.c + s2.c, L.cin);
    PS_CHECK(R < (1ll << 31), "rowgemm: too many rows");
    RowGemmArgs a;
    a.x1 = s1.x; a.g1 = s1.gather; a.ld1 = s1.ld; a.c1 = s1.c; a.g1m = s1.gm; a.g1n = s1.gn;
    a.x2 = s2.x; a.g2 = s2.gather; a.ld2 = s2.ld; a.c2 = s2.c; a.g2m = s2.gm; a.g2n = s2.gn;
    a.wp = L.wp; a.bias = L.bias;
    a.y = y; a.ldy = ldy;
    a.cin = L.cin; a.cout = L.cout; a.ks = L.ks; a.R = (int)R; a.leaky = L.leaky;
    dim3 grid(ceil_div(R, 64), L.cblocks);
    switch (L.ntb) {
        case 1: hipLaunchKernelGGL(rowgemm_kernel<1>, grid, dim3(256), 0, c->stream, a); break;
        case 2: hipLaunchKernelGGL(rowgemm_kernel<2>, grid, dim3(256), 0, c->stream, a); break;
        case 4: hipLaunchKernelGGL(rowgemm_kernel<4>, grid, dim3(256), 0, c->stream, a); break;
        default: set_error("rowgemm: bad ntb %d", L.ntb); return PS_EINVAL;
    }
    PS_HIP(hipGetLastError());
    return PS_OK;
}

}  // namespace ps
