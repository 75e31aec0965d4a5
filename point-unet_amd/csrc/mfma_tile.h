// mfma_tile.h -- wave-level building blocks on v_mfma_f32_16x16x4_f32 (exact fp32, 64 FLOP/clk/SIMD on gfx950).
//
// Fragment maps (wave64): A[i][k]: lane = i + 16*k   B[k][j]: lane = j + 16*k   C[i][j]: lane = j + 16*(i/4), reg = i%4
// A tiles live in LDS row-major with a pitch = 2 (mod 32) floats so that the A-fragment read is conflict free;
// B fragments are read straight from host-packed weights (rowgemm.h: PackedLinear).
#pragma once

#include <hip/hip_runtime.h>

namespace ps {

using f32x4 = __attribute__((ext_vector_type(4))) float;

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

template <int NTB>
__device__ __forceinline__ float bfrag_get(const typename BFrag<NTB>::type& b, int j);
template <>
__device__ __forceinline__ float bfrag_get<1>(const float& b, int) { return b; }
template <>
__device__ __forceinline__ float bfrag_get<2>(const float2& b, int j) { return j == 0 ? b.x : b.y; }
template <>
__device__ __forceinline__ float bfrag_get<4>(const float4& b, int j) { return j == 0 ? b.x : (j == 1 ? b.y : (j == 2 ? b.z : b.w)); }

constexpr int ntb_for(int cout) { return cout >= 64 ? 4 : (cout >= 32 ? 2 : 1); }

// acc[rt][j] += A_tile[rt] (16 x 4*ksteps, LDS) . W[:, column block]   for RT row tiles sharing each B fragment.
// wp_cb points at the packed weights of this column block, already offset by +lane.
template <int NTB, int RT>
__device__ __forceinline__ void tile_mma(const float* __restrict__ tileA, int pitch, int ksteps,
                                         const typename BFrag<NTB>::type* __restrict__ wp_cb, f32x4 (&acc)[RT][NTB], int lane)
{
    const float* a0 = tileA + (lane & 15) * pitch + (lane >> 4);
#pragma unroll 8
    for (int s = 0; s < ksteps; ++s) {
        const typename BFrag<NTB>::type bv = wp_cb[(size_t)s * 64];
        float av[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) av[rt] = a0[rt * 16 * pitch + s * 4];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int j = 0; j < NTB; ++j)
                acc[rt][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt], bfrag_get<NTB>(bv, j), acc[rt][j], 0, 0, 0);
    }
}

__device__ __forceinline__ float leaky02(float v) { return fmaxf(v, 0.2f * v); }  // == v >= 0 ? v : 0.2 v, one compare less

// reductions across the four 16-lane groups (rows of a C column live in lanes l, l^16, l^32, l^48).
// gfx950's v_permlane16_swap / v_permlane32_swap exchange 16- / 32-lane halves between two registers in one VALU
// instruction: swapping a register with itself leaves {own value, partner's value} in the result pair (which is which
// depends on the lane, but max and + are symmetric), so a butterfly step is one swap and one op, with no LDS round trip
// (ds_bpermute) in the softmax's dependent chain.
// (ds_bpermute forms: fewer issue slots; the level-0 kernels are issue-bound and measured 6 % faster with these)
__device__ __forceinline__ float xor_max_lds(float v)
{
    v = fmaxf(v, __shfl_xor(v, 16));
    return fmaxf(v, __shfl_xor(v, 32));
}
__device__ __forceinline__ float xor_sum_lds(float v)
{
    v += __shfl_xor(v, 16);
    return v + __shfl_xor(v, 32);
}
__device__ __forceinline__ float xor_max(float v)
{
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float xor_sum(float v)
{
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

}  // namespace ps
