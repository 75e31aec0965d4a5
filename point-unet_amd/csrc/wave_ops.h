// wave_ops.h -- all-lanes reductions of a 64-lane wave without LDS traffic.
// `x = op(x, __shfl_xor(x, o))` compiles to ds_bpermute_b32 on gfx950: an LDS-pipe round trip (~100+ cycles) per butterfly step,
// six of them in a dependent chain per reduction.  Here the four in-row steps are DPP modifiers on the VALU instruction itself
// (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror: each pairs complementary lane sets, which is all
// a reduction needs) and the two cross-row steps are v_permlane16_swap / v_permlane32_swap (mfma_tile.h explains the trick).
#pragma once

#include <hip/hip_runtime.h>

namespace ps {

template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, false);
}

template <class Op>
__device__ __forceinline__ unsigned wave_allreduce_u32(unsigned v, Op op)
{
    v = op(v, dpp_u32<0xB1>(v));   // quad_perm [1,0,3,2]
    v = op(v, dpp_u32<0x4E>(v));   // quad_perm [2,3,0,1]
    v = op(v, dpp_u32<0x141>(v));  // row_half_mirror
    v = op(v, dpp_u32<0x140>(v));  // row_mirror
    auto a = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    v = op(a[0], a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return op(b[0], b[1]);
}

__device__ __forceinline__ float wave_min(float v)
{
    return __uint_as_float(wave_allreduce_u32(__float_as_uint(v), [](unsigned a, unsigned b) { return __float_as_uint(fminf(__uint_as_float(a), __uint_as_float(b))); }));
}
__device__ __forceinline__ float wave_max(float v)
{
    return __uint_as_float(wave_allreduce_u32(__float_as_uint(v), [](unsigned a, unsigned b) { return __float_as_uint(fmaxf(__uint_as_float(a), __uint_as_float(b))); }));
}
__device__ __forceinline__ int wave_sum(int v)
{
    return (int)wave_allreduce_u32((unsigned)v, [](unsigned a, unsigned b) { return a + b; });
}

}  // namespace ps
