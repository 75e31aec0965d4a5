// b3_ops.h -- operand helpers of the split-bf16 ("bf16x3") matrix products shared by gemm_b3.hip and attpool_gemm.hip:
// v_mfma_f32_32x32x16_bf16 over exact three-way bfloat16 splits of fp32 operands (P = 3: six piece products per fp32 product, fp32
// accumulate, fp32-level error -- attpool32b.hip explains the split) or over ONE plane of round-to-nearest-even bfloat16 (P = 1: the
// bf16-MLP mode of the training step).
//
// Fragment maps of the instruction (wave64, hl = lane >> 5, c32 = lane & 31):
//   A[m][k]: lane (m = c32, hl) holds k = 8 hl + j, j = 0..7      B[k][n]: lane (n = c32, hl) holds k = 8 hl + j
//   C[m][n]: lane (n = c32, hl), register r holds m = (r & 3) + 8 (r >> 2) + 4 hl
#pragma once

#include <hip/hip_runtime.h>

#include "common.h"

namespace ps {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

__device__ __forceinline__ void b3_split_pair(float x, float y, unsigned& q1, unsigned& q2, unsigned& q3)
{
    const unsigned xu = __float_as_uint(x), yu = __float_as_uint(y);
    const float xr = x - __uint_as_float(xu & 0xffff0000u), yr = y - __uint_as_float(yu & 0xffff0000u);  // exact
    const unsigned xru = __float_as_uint(xr), yru = __float_as_uint(yr);
    const float x3 = xr - __uint_as_float(xru & 0xffff0000u), y3 = yr - __uint_as_float(yru & 0xffff0000u);  // exact, 8 bits
    q1 = __builtin_amdgcn_perm(yu, xu, 0x07060302u);
    q2 = __builtin_amdgcn_perm(yru, xru, 0x07060302u);
    q3 = __builtin_amdgcn_perm(__float_as_uint(y3), __float_as_uint(x3), 0x07060302u);
}
// P = 3: the exact three-way split (fp32 products on the bf16 pipe).  P = 1: ONE plane of round-to-nearest-even bfloat16 -- the bf16-MLP
// mode of the training step (operands rounded to bfloat16, fp32 accumulation), which so runs its large products through the same tiling
// with a sixth of the matrix work: HBM bound ([360k, 256] x [256, 256]: 0.63 ms in rowgemm_direct_bf16 before).
template <int P>
struct BPlanes {
    uint4 p[P];
};
typedef BPlanes<3> B3Planes;
typedef __bf16 b3_bf16x2 __attribute__((ext_vector_type(2)));
typedef float b3_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned b3_rne_pair(float a, float b)  // (low half = a: v_cvt_pk_bf16_f32)
{
    return __builtin_bit_cast(unsigned, __builtin_convertvector(b3_f32x2{a, b}, b3_bf16x2));
}
template <int P>
__device__ __forceinline__ BPlanes<P> b3_split8(const float4& lo, const float4& hi)
{
    BPlanes<P> r;
    if constexpr (P == 3) {
        b3_split_pair(lo.x, lo.y, r.p[0].x, r.p[1].x, r.p[2].x);
        b3_split_pair(lo.z, lo.w, r.p[0].y, r.p[1].y, r.p[2].y);
        b3_split_pair(hi.x, hi.y, r.p[0].z, r.p[1].z, r.p[2].z);
        b3_split_pair(hi.z, hi.w, r.p[0].w, r.p[1].w, r.p[2].w);
    } else {
        r.p[0].x = b3_rne_pair(lo.x, lo.y);
        r.p[0].y = b3_rne_pair(lo.z, lo.w);
        r.p[0].z = b3_rne_pair(hi.x, hi.y);
        r.p[0].w = b3_rne_pair(hi.z, hi.w);
    }
    return r;
}
__device__ __forceinline__ f32x16 b3_mfma(const uint4& a, const uint4& b, f32x16 acc)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
template <int P>
__device__ __forceinline__ f32x16 b3_mfma6(const BPlanes<P>& a, const BPlanes<P>& b, f32x16 acc)
{
    if constexpr (P == 3) {
        acc = b3_mfma(a.p[2], b.p[0], acc);
        acc = b3_mfma(a.p[0], b.p[2], acc);
        acc = b3_mfma(a.p[1], b.p[1], acc);
        acc = b3_mfma(a.p[1], b.p[0], acc);
        acc = b3_mfma(a.p[0], b.p[1], acc);
        acc = b3_mfma(a.p[0], b.p[0], acc);
    } else {
        acc = b3_mfma(a.p[0], b.p[0], acc);
    }
    return acc;
}

// the element function of gemm_b3_pack_kernel for the batched form (PackCache): thread i of job j.
// KMAP: the K axis in the order in which a 32x32 accumulator tile hands its values on as the next product's operand (b3_ops.h:
// k-slot (chunk q, lane half hl, element e) = index 32 (q >> 1) + 16 (q & 1) + 8 (e >> 2) + 4 hl + (e & 3)) -- the image of
// attpool_gemm.hip's second product, whose row operand is the transposed accumulator tile of dS.
template <int P, bool KMAP>
__device__ __forceinline__ void b3_pack_elem(const PackJob& j, int64_t i)
{
    const int ncb = j.cout / 32;
    const int lane = (int)(i & 63), cb = (int)((i >> 6) % ncb), q = (int)((i >> 6) / ncb);
    const int hl = lane >> 5;
    const float* src = j.w + (int64_t)(32 * cb + (lane & 31)) * j.sn;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = KMAP ? 32 * (q >> 1) + 16 * (q & 1) + 8 * (e >> 2) + 4 * hl + (e & 3) : 16 * q + 8 * hl + e;
        v[e] = src[(int64_t)k * j.sk];
    }
    const BPlanes<P> p = b3_split8<P>(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]));
    uint4* dst = static_cast<uint4*>(j.out) + ((size_t)(q * ncb + cb) * P) * 64 + lane;
#pragma unroll
    for (int pl = 0; pl < P; ++pl) dst[64 * pl] = p.p[pl];
}
__device__ __forceinline__ void b3_pack_elem_any(const PackJob& j, int64_t i)
{
    if (j.kind == 3) b3_pack_elem<3, false>(j, i);
    else if (j.kind == 4) b3_pack_elem<1, false>(j, i);
    else if (j.kind == 5) b3_pack_elem<3, true>(j, i);
    else b3_pack_elem<1, true>(j, i);
}

}  // namespace ps
