// bf16_io.h -- bfloat16 STORAGE of activation rows (ps_set_train_act_bf16): 4 / 8 elements per load or store, converted at the register.
// Loads are exact (a bfloat16 is the upper half of an fp32); stores round to nearest even (v_cvt_pk_bf16_f32).
#pragma once

#include <hip/hip_runtime.h>

namespace ps {

typedef __bf16 io_bf16x2 __attribute__((ext_vector_type(2)));
typedef float io_f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi)  // (low half = lo)
{
    return __builtin_bit_cast(unsigned, __builtin_convertvector(io_f32x2{lo, hi}, io_bf16x2));
}
__device__ __forceinline__ float4 unpack_bf16x4(const uint2 u)
{
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
}
__device__ __forceinline__ uint2 pack_bf16x4(const float4 v) { return make_uint2(pack_bf16(v.x, v.y), pack_bf16(v.z, v.w)); }

// four consecutive elements of a row of floats or bfloat16s (`p` is the row's base in its own element type; col = element index, % 4 == 0)
__device__ __forceinline__ float4 load4_any(const float* p, size_t elem, bool b16)
{
    if (b16) return unpack_bf16x4(*reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(p) + elem));
    return *reinterpret_cast<const float4*>(p + elem);
}
__device__ __forceinline__ void store4_any(float* p, size_t elem, const float4 v, bool b16)
{
    if (b16) *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(p) + elem) = pack_bf16x4(v);
    else *reinterpret_cast<float4*>(p + elem) = v;
}

}  // namespace ps
