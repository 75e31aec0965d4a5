// gemm32.hip -- dense layers of the deep levels (few rows, wide channels) on v_mfma_f32_32x32x2_f32.
//
//   Y[r, :] = act([X1[g1[r]] | X2[g2[r]]] . W + b)        R = 351 .. 11 250 rows, cin = 128 .. 1 536, cout = 64 .. 1 024
//
// the 1x1 convolutions of encoder levels 2-4 and of the decoder (helper_tf_util.conv2d / conv2d_transpose,
// PointSegment/helper_tf_util.py:115-250; RandLANet.py:130-141, 315-321), inference-mode BatchNorm folded on the host.
// rowgemm_direct (16x16x4 tiles, one B fragment per MFMA from L2) ran these shapes at 13-20 % of the fp32 MFMA peak:
// 16-row tiles re-read every weight once per 16 rows, and a launch was little more than one exposed load-latency chain.
// Here a wave owns a 32-row x (32*CW)-column block: a weight fragment (one 16-byte read of the pack_p32 image, attpool.h)
// feeds four 64-cycle MFMAs over 32 rows, activations are read 16 bytes per lane straight from the row-major input (K taken
// in the order {8q + 4*half + t}, as in attpool32.hip), eight chunks of loads are in flight ahead of the MFMAs, and when the
// grid would not fill the chip the four waves of a workgroup split the K axis and add their partial blocks through LDS.
#include "attpool.h"
#include "mfma_tile.h"
#include "rowgemm.h"

namespace ps {

using f32x16 = __attribute__((ext_vector_type(16))) float;

struct Gemm32Args {
    const float* x1; const int32_t* g1; int ld1, c1, g1m, g1n;
    const float* x2; const int32_t* g2; int ld2, c2, g2m, g2n;
    const float* wp;    // pack_p32 image of W[cin, cout]
    const float* bias;  // [cout]
    float* y;
    int ldy, R, cin, cout, leaky;
    int rgroups, cgroups;  // workgroup grid: row groups x column groups (see the XCD mapping in the kernel)
};

// waves of a workgroup: SK along K (same output block), 4 / SK consecutive row blocks (SK = 8: eight waves, one row block -- the
// few-row / long-K layers of levels 3-4 and the decoder, whose launch is one exposed chain of loads and MFMAs per wave: half the chain)
template <int CW, int SK>
__global__ __launch_bounds__(SK > 4 ? 64 * SK : 256) void gemm32_kernel(Gemm32Args a)
{
    constexpr int RB = SK > 4 ? 1 : 4 / SK;  // row blocks per workgroup
    __shared__ float red[SK > 1 ? RB * (SK - 1) * CW * 16 * 64 : 1];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int hl = lane >> 5, c32 = lane & 31;
    const int kw = wave % SK, rbw = wave / SK;
    // XCD-aware mapping: workgroups go to the 8 XCDs round-robin (block b -> XCD b % 8).  The (column group, row group) space is
    // walked column-group-major and cut into eight contiguous pieces, one per XCD, so an XCD's L2 holds only its own column
    // groups' weight panels (1/8 of W: the 4 MB matrices of the deepest levels do not fit one 4 MB L2 next to the activations)
    // and consecutive workgroups of an XCD reuse the same panel.
    const int total = a.rgroups * a.cgroups, per_xcd = (total + 7) >> 3;
    const int slot = (int)(blockIdx.x >> 3);
    const int u = (int)(blockIdx.x & 7) * per_xcd + slot;
    if (slot >= per_xcd || u >= total) return;
    const int rb = (u % a.rgroups) * RB + rbw;  // 32-row block
    const int cb = (u / a.rgroups) * CW;        // first 32-column block
    const int nq = a.cin / 8, nq1 = a.c1 / 8;
    const int qa = (nq * kw) / SK, qb = (nq * (kw + 1)) / SK;
    const bool live = rb * 32 < a.R;

    f32x16 acc[CW];
#pragma unroll
    for (int j = 0; j < CW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float bias_r[CW];  // (requested in front of the K loop: behind it the load is one more exposed round trip of a ~9 us launch)
#pragma unroll
    for (int j = 0; j < CW; ++j) bias_r[j] = (kw == 0) ? a.bias[(cb + j) * 32 + c32] : 0.f;
    if (live) {
        const int rr = min(rb * 32 + c32, a.R - 1);
        const int s1 = a.g1 ? (a.g1m ? (rr / a.g1m) * a.g1n : 0) + a.g1[rr] : rr;
        const float* p1 = a.x1 + (size_t)s1 * a.ld1 + 4 * hl;
        const float* p2 = p1;
        if (a.c2) {
            const int s2 = a.g2 ? (a.g2m ? (rr / a.g2m) * a.g2n : 0) + a.g2[rr] : rr;
            p2 = a.x2 + (size_t)s2 * a.ld2 + 4 * hl - (size_t)8 * nq1;
        }
        const float4* wq = reinterpret_cast<const float4*>(a.wp) + (size_t)cb * nq * 64 + lane;
        const size_t wstride = (size_t)nq * 64;  // float4s between consecutive column blocks
        // A ring of PD K-chunks in flight, refilled in place behind its reader (gemm32b.hip's scheme; here PD = 2: a double buffer).  Round 6: the `#pragma unroll 8` loop
        // this replaces was NOT unrolled ("-Wpass-failed: loop not unrolled", silenced by the Makefile's -Wno-pass-failed): every 8-wide K chunk
        // was a load, a wait for it and four MFMAs -- one exposed L2 round trip per chunk, eight to sixteen of them per ~9 us launch.
#ifdef PS_G32_PD
        constexpr int PD = PS_G32_PD;
#else
        constexpr int PD = 2;  // (measured, serial cloud, same box: 2 / 3 chunks ahead 1.306-1.308 ms, 4: 1.325, 8: 1.315 -- the K slices are 4-16 chunks)
#endif
        float4 axr[PD], bwr[PD][CW];
        auto fetch = [&](int slot, int q) __attribute__((always_inline)) {
            q = min(q, qb - 1);  // (past the end: a harmless repeat of the last chunk, never used)
            axr[slot] = *reinterpret_cast<const float4*>((q < nq1 ? p1 : p2) + 8 * q);
#pragma unroll
            for (int j = 0; j < CW; ++j) bwr[slot][j] = wq[(size_t)j * wstride + (size_t)q * 64];
        };
        auto products = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < CW; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(axr[slot].x, bwr[slot][j].x, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(axr[slot].y, bwr[slot][j].y, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(axr[slot].z, bwr[slot][j].z, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(axr[slot].w, bwr[slot][j].w, acc[j], 0, 0, 0);
            }
        };
#pragma unroll
        for (int d = 0; d < PD; ++d) fetch(d, qa + d);
        int q0 = qa;
#pragma unroll 1
        for (; q0 + 2 * PD <= qb; q0 += PD) {  // (groups whose refills all exist: nothing is requested past the end of the slice)
#pragma unroll
            for (int d = 0; d < PD; ++d) {
                products(d);
                __builtin_amdgcn_sched_barrier(0);
                fetch(d, q0 + d + PD);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int d = 0; d < PD; ++d) {  // the ring holds the next PD chunks; fewer than 2 PD are left
            if (q0 + d < qb) {
                products(d);
                if (q0 + d + PD < qb) fetch(d, q0 + d + PD);
            }
        }
#pragma unroll
        for (int d = 0; d < PD - 1; ++d)
            if (q0 + PD + d < qb) products(d);
    }
    if constexpr (SK > 1) {
        // partial blocks of the K slices 1 .. SK-1 go through LDS (register-major: conflict-free), slice 0 adds them up
        if (kw > 0) {
            float* dst = red + ((size_t)(rbw * (SK - 1) + (kw - 1)) * CW * 16) * 64 + lane;
#pragma unroll
            for (int j = 0; j < CW; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) dst[(j * 16 + r) * 64] = acc[j][r];
        }
        __syncthreads();
        if (kw > 0) return;
#pragma unroll
        for (int s = 0; s < SK - 1; ++s) {
            const float* src = red + ((size_t)(rbw * (SK - 1) + s) * CW * 16) * 64 + lane;
#pragma unroll
            for (int j = 0; j < CW; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] += src[(j * 16 + r) * 64];
        }
    }
    if (!live) return;
    // C layout: register r of lane (hl, c32) = row (r & 3) + 8 * (r >> 2) + 4 * hl, column c32 of the block
#pragma unroll
    for (int j = 0; j < CW; ++j) {
        const int col = (cb + j) * 32 + c32;
        const float bb = bias_r[j];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hl;
            float v = acc[j][r] + bb;
            if (a.leaky) v = leaky02(v);
            if (row < a.R) a.y[(size_t)row * a.ldy + col] = v;
        }
    }
}

bool gemm32_fits(const PackedLinear& L, const RowSrc& s1, const RowSrc& s2, int64_t R, int ldy)
{
    return L.w32 && !L.accum && R <= 32768 && L.cin % 8 == 0 && L.cout % 32 == 0 && s1.c % 8 == 0 && s2.c % 8 == 0 && s1.c + s2.c == L.cin &&
           s1.ld % 4 == 0 && (s2.c == 0 || s2.ld % 4 == 0) && (reinterpret_cast<uintptr_t>(s1.x) & 15) == 0 &&
           (s2.c == 0 || (reinterpret_cast<uintptr_t>(s2.x) & 15) == 0) && ldy > 0;
}

int gemm32(ps_context* c, const PackedLinear& L, const RowSrc& s1, const RowSrc& s2, int64_t R, float* y, int ldy)
{
    if (R <= 0) return PS_OK;
    PS_CHECK(gemm32_fits(L, s1, s2, R, ldy), "gemm32: shape / alignment not supported (cin %d, cout %d)", L.cin, L.cout);
    Gemm32Args a;
    a.x1 = s1.x; a.g1 = s1.gather; a.ld1 = s1.ld; a.c1 = s1.c; a.g1m = s1.gm; a.g1n = s1.gn;
    a.x2 = s2.x; a.g2 = s2.gather; a.ld2 = s2.ld; a.c2 = s2.c; a.g2m = s2.gm; a.g2n = s2.gn;
    a.wp = L.w32; a.bias = L.bias; a.y = y; a.ldy = ldy; a.R = (int)R; a.cin = L.cin; a.cout = L.cout; a.leaky = L.leaky;
    const int rblocks = (int)((R + 31) / 32);
    // two column blocks per wave (a row fragment feeds eight MFMAs) once that still leaves a wave for every SIMD
    const int cw = (L.cout % 64 == 0 && (int64_t)rblocks * (L.cout / 64) >= 1024) ? 2 : 1;
    const int cgroups = L.cout / (32 * cw);
    // split K across the waves of a workgroup while the plain grid leaves SIMDs idle (1 024 of them) and the slices stay >= 8 chunks
    const int64_t units = (int64_t)rblocks * cgroups;
    int sk = 1;
    while (sk < 8 && units * sk < 1536 && L.cin / 8 / (sk * 2) >= 8) sk *= 2;
    if (sk == 8 && c->tune.gemm32_no_sk8) sk = 4;
    const dim3 block(sk > 4 ? 64 * sk : 256);
    const int rb_per_wg = sk > 4 ? 1 : 4 / sk;
    a.cgroups = cgroups;
    a.rgroups = (rblocks + rb_per_wg - 1) / rb_per_wg;
    const unsigned grid = 8u * (unsigned)((a.rgroups * a.cgroups + 7) / 8);
#define PS_G32(CW)                                                                                       \
    if (sk == 1) hipLaunchKernelGGL((gemm32_kernel<CW, 1>), dim3(grid), block, 0, c->stream, a);         \
    else if (sk == 2) hipLaunchKernelGGL((gemm32_kernel<CW, 2>), dim3(grid), block, 0, c->stream, a);    \
    else if (sk == 4) hipLaunchKernelGGL((gemm32_kernel<CW, 4>), dim3(grid), block, 0, c->stream, a);    \
    else hipLaunchKernelGGL((gemm32_kernel<CW, 8>), dim3(grid), block, 0, c->stream, a)
    if (cw == 2) { PS_G32(2); } else { PS_G32(1); }
#undef PS_G32
    PS_HIP(hipGetLastError());
    return PS_OK;
}

}  // namespace ps
