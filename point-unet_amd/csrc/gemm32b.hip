// gemm32b.hip -- gemm32.hip's dense layers on v_mfma_f32_32x32x16_bf16 over EXACT three-way bfloat16 splits of both fp32 operands
// ("bf16x3", attpool32b.hip: x = x1 + x2 + x3 with 8 significant bits each, six piece products kept, fp32 accumulate -- fp32-level
// error, 2.7 x less matrix-pipe time than the fp32 MFMA, which runs at 1/16 of the bf16 rate on gfx950).
//
//   Y[r, :] = act([X1[g1[r]] | X2[g2[r]]] . W + b)
//
// the 1x1 convolutions of encoder levels 2-4 and of the decoder (helper_tf_util.conv2d / conv2d_transpose,
// PointSegment/helper_tf_util.py:115-250; RandLANet.py:130-141, 315-321) whose product is large enough to be bound by the matrix
// pipe: [mlp2 ; shortcut] of levels 2-4 (1.47 GFLOP each), decoder_0 and the first decoder steps (0.7-1.1 GFLOP).  On the fp32 MFMA
// (gemm32.hip) those seven launches took 24-35 us each, 23 % of the fp32 peak; here 12-24 us (round 4, rocprofv3 kernel trace: decoder_0 28 -> 22,
// first decoder step 35 -> 26 us by hipEvent pairs).  What bounds them now is not the matrix pipe (13 % busy, SQ_VALU_MFMA_BUSY_CYCLES) but
// the grid: 176-704 workgroups of four waves leave one or two waves per SIMD, so a wave's own sequence -- split (56 VALU instructions
// per 16-wide K chunk), twelve MFMAs, the wait for the next chunk -- is exposed end to end, and 352 workgroups on 256 CUs run as 1 + 1.
//
// A wave owns (32 RW) rows x (32 CW) columns: activations are read 32 bytes per lane straight from the row-major input (row = lane & 31,
// eight consecutive K values per lane half: exactly the operand layout of the instruction) and split in registers; the weights come
// pre-split from the layer's three-plane image (pack_p32b: [column block][K chunk of 16][plane][lane] x 16 bytes).  One split of an
// activation fragment feeds CW column blocks, one weight fragment RW row blocks.  When the grid would not fill the chip the four waves of
// a workgroup split the K axis and add their partial blocks through LDS (fixed order).
#include "attpool.h"
#include "mfma_tile.h"
#include "rowgemm.h"

namespace ps {

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

struct Planes {
    uint4 p[3];
};
__device__ __forceinline__ void split_pair(float x, float y, unsigned& q1, unsigned& q2, unsigned& q3)
{
    const unsigned xu = __float_as_uint(x), yu = __float_as_uint(y);
    const float xr = x - __uint_as_float(xu & 0xffff0000u), yr = y - __uint_as_float(yu & 0xffff0000u);  // exact
    const unsigned xru = __float_as_uint(xr), yru = __float_as_uint(yr);
    const float x3 = xr - __uint_as_float(xru & 0xffff0000u), y3 = yr - __uint_as_float(yru & 0xffff0000u);  // exact, 8 bits
    q1 = __builtin_amdgcn_perm(yu, xu, 0x07060302u);  // [y.hi16 : x.hi16]
    q2 = __builtin_amdgcn_perm(yru, xru, 0x07060302u);
    q3 = __builtin_amdgcn_perm(__float_as_uint(y3), __float_as_uint(x3), 0x07060302u);
}
__device__ __forceinline__ Planes split8(const float4& lo, const float4& hi)
{
    Planes r;
    split_pair(lo.x, lo.y, r.p[0].x, r.p[1].x, r.p[2].x);
    split_pair(lo.z, lo.w, r.p[0].y, r.p[1].y, r.p[2].y);
    split_pair(hi.x, hi.y, r.p[0].z, r.p[1].z, r.p[2].z);
    split_pair(hi.z, hi.w, r.p[0].w, r.p[1].w, r.p[2].w);
    return r;
}
__device__ __forceinline__ f32x16 mfma_b(const uint4& a, const uint4& b, f32x16 acc)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
// the six kept piece products, smallest first
__device__ __forceinline__ f32x16 mfma6(const Planes& a, const Planes& b, f32x16 acc)
{
    acc = mfma_b(a.p[2], b.p[0], acc);
    acc = mfma_b(a.p[0], b.p[2], acc);
    acc = mfma_b(a.p[1], b.p[1], acc);
    acc = mfma_b(a.p[1], b.p[0], acc);
    acc = mfma_b(a.p[0], b.p[1], acc);
    acc = mfma_b(a.p[0], b.p[0], acc);
    return acc;
}

// ... for a whole RW x CW block of accumulators, piece product by piece product: consecutive MFMAs go to DIFFERENT accumulators.  Issued
// accumulator by accumulator (six dependent instructions in a row) a wave spent 55 % of its cycles in issue stalls (SQ_WAIT_INST_ANY,
// round 4 counters): a dependent MFMA waits for its predecessor's last pass, and with one wave per SIMD nothing else fills the gap.
template <int RW, int CW>
__device__ __forceinline__ void mfma6_all(const Planes (&A)[RW], const Planes (&B)[CW], f32x16 (&acc)[RW][CW])
{
    constexpr int pa[6] = {2, 0, 1, 1, 0, 0}, pb[6] = {0, 2, 1, 0, 1, 0};  // smallest products first
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int i = 0; i < RW; ++i)
#pragma unroll
            for (int j = 0; j < CW; ++j) acc[i][j] = mfma_b(A[i].p[pa[t]], B[j].p[pb[t]], acc[i][j]);
}

struct Gemm32bArgs {
    const float* x1; const int32_t* g1; int ld1, c1, g1m, g1n;
    const float* x2; const int32_t* g2; int ld2, c2, g2m, g2n;
    const uint4* wp;    // pack_p32b image of W[cin, cout]
    const float* bias;  // [cout]
    float* y;
    int ldy, R, cin, cout, leaky;
    int rgroups, cgroups;  // workgroup grid: row groups x column groups (XCD mapping as gemm32.hip)
};

// waves of a workgroup: SK along K (same output block), 4 / SK consecutive row units of 32 RW rows
template <int RW, int CW, int SK>
__global__ __launch_bounds__(SK > 4 ? 64 * SK : 256) void gemm32b_kernel(Gemm32bArgs a)
{
    constexpr int RU = SK > 4 ? 1 : 4 / SK;  // row units per workgroup (SK = 8: eight waves on one row unit)
    __shared__ float red[SK > 1 ? RU * (SK - 1) * RW * CW * 16 * 64 : 1];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int hl = lane >> 5, c32 = lane & 31;
    const int kw = wave % SK, ruw = wave / SK;
    // XCD-aware mapping (gemm32.hip): the (column group, row group) space is walked column-group-major and cut into eight contiguous
    // pieces, one per XCD: an XCD's L2 holds only its own column groups' weight planes
    const int total = a.rgroups * a.cgroups, per_xcd = (total + 7) >> 3;
    const int slot = (int)(blockIdx.x >> 3);
    const int u = (int)(blockIdx.x & 7) * per_xcd + slot;
    if (slot >= per_xcd || u >= total) return;
    const int ru = (u % a.rgroups) * RU + ruw;  // row unit: rows [ru * 32 RW, (ru + 1) * 32 RW)
    const int cb = (u / a.rgroups) * CW;        // first 32-column block
    const int nq = a.cin / 16, nq1 = a.c1 / 16;
    const int qa = (nq * kw) / SK, qb = (nq * (kw + 1)) / SK;
    const bool live = ru * 32 * RW < a.R;

    f32x16 acc[RW][CW];
#pragma unroll
    for (int i = 0; i < RW; ++i)
#pragma unroll
        for (int j = 0; j < CW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // the epilogue's bias values are requested now (round 6: loaded behind the K loop they were one more exposed round trip of a launch
    // whose waves live ~15 us)
    float bias_r[CW];
#pragma unroll
    for (int j = 0; j < CW; ++j) bias_r[j] = (kw == 0) ? a.bias[(cb + j) * 32 + c32] : 0.f;
    if (live) {
        const float* p1[RW];
        const float* p2[RW];
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            const int rr = min((ru * RW + i) * 32 + c32, a.R - 1);
            const int s1 = a.g1 ? (a.g1m ? (rr / a.g1m) * a.g1n : 0) + a.g1[rr] : rr;
            p1[i] = a.x1 + (size_t)s1 * a.ld1 + 8 * hl;
            p2[i] = p1[i];
            if (a.c2) {
                const int s2 = a.g2 ? (a.g2m ? (rr / a.g2m) * a.g2n : 0) + a.g2[rr] : rr;
                p2[i] = a.x2 + (size_t)s2 * a.ld2 + 8 * hl - (size_t)16 * nq1;
            }
        }
        const uint4* wq = a.wp + (size_t)cb * nq * 3 * 64 + lane;
        const size_t wstride = (size_t)nq * 3 * 64;  // uint4s between consecutive column blocks
        // PD chunks ahead: a wave's loop is a chain of dependent round trips to L2 / MALL (the weight planes of a layer are read once per
        // 32-row block), and one chunk in flight left the launch latency-bound (23 us for 0.74 GFLOP); the ring is refilled in place
        // behind its last reader
#ifdef PS_G32B_PD
        constexpr int PD = RW * CW >= 4 ? 3 : PS_G32B_PD;  // (experiment: ring depth of the (1, 1) / (1, 2) tiles)
#else
        // Round 6, serial cloud on one box: 2 chunks ahead 1.280-1.286 ms, 3: 1.283-1.296, 4 (rounds 4-5): 1.295-1.306; 6 / 8 take 256 VGPRs
        // and run the enc2-4 dense stages 0.086 / 0.097 / 0.086 and 0.090 / 0.102 / 0.090 ms against 0.078 / 0.077 / 0.075 -- the K slices
        // of these launches are 8-24 chunks, and every chunk fetched past the end of a slice is a wasted request
        constexpr int PD = RW * CW >= 4 ? 3 : 2;
#endif
        float4 xl[PD][RW], xh[PD][RW];
        uint4 bw[PD][CW][3];
        auto fetch = [&](int slot, int q) __attribute__((always_inline)) {
            q = min(q, qb - 1);  // (past the end: a harmless repeat of the last chunk, never used)
#pragma unroll
            for (int i = 0; i < RW; ++i) {
                const float* s = (q < nq1 ? p1[i] : p2[i]) + 16 * q;
                xl[slot][i] = *reinterpret_cast<const float4*>(s);
                xh[slot][i] = *reinterpret_cast<const float4*>(s + 4);
            }
#pragma unroll
            for (int j = 0; j < CW; ++j)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) bw[slot][j][pl] = wq[(size_t)j * wstride + ((size_t)q * 3 + pl) * 64];
        };
#pragma unroll
        for (int d = 0; d < PD; ++d) fetch(d, qa + d);
        int q0 = qa;
        auto products = [&](int d) __attribute__((always_inline)) {
            Planes A[RW], B[CW];
#pragma unroll
            for (int i = 0; i < RW; ++i) A[i] = split8(xl[d][i], xh[d][i]);
#pragma unroll
            for (int j = 0; j < CW; ++j) { B[j].p[0] = bw[d][j][0]; B[j].p[1] = bw[d][j][1]; B[j].p[2] = bw[d][j][2]; }
            mfma6_all<RW, CW>(A, B, acc);
        };
        // full groups of PD chunks whose refills all exist: straight-line code (a branch inside the group made the compiler drain every
        // load at each join).  Round 6: the loop stops one group early -- it used to refill past the end of the slice (a clamped repeat of
        // the last chunk: 2-4 wasted chunk requests of the 8-24 a wave makes)
#pragma unroll 1
        for (; q0 + 2 * PD <= qb; q0 += PD) {
#pragma unroll
            for (int d = 0; d < PD; ++d) {
                products(d);
                // the refill goes into the registers the products just read (issued earlier it needs other registers and a drained copy at
                // the loop's end), in program order: the wait in front of slot d + 1 leaves the PD - 1 younger refills in flight
                __builtin_amdgcn_sched_barrier(0);
                fetch(d, q0 + d + PD);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // the ring holds the next PD chunks; fewer than 2 PD are left.  Slices whose length is a multiple of PD (every layer of the network)
        // need no further request; the others fetch their last few chunks behind the products that free the slot
#pragma unroll
        for (int d = 0; d < PD; ++d) {
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
        // partial blocks of the K slices 1 .. SK-1 go through LDS (register-major: conflict-free), slice 0 adds them up in slice order
        if (kw > 0) {
            float* dst = red + ((size_t)(ruw * (SK - 1) + (kw - 1)) * RW * CW * 16) * 64 + lane;
#pragma unroll
            for (int i = 0; i < RW; ++i)
#pragma unroll
                for (int j = 0; j < CW; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) dst[((i * CW + j) * 16 + r) * 64] = acc[i][j][r];
        }
        __syncthreads();
        if (kw > 0) return;
#pragma unroll
        for (int s = 0; s < SK - 1; ++s) {
            const float* src = red + ((size_t)(ruw * (SK - 1) + s) * RW * CW * 16) * 64 + lane;
#pragma unroll
            for (int i = 0; i < RW; ++i)
#pragma unroll
                for (int j = 0; j < CW; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] += src[((i * CW + j) * 16 + r) * 64];
        }
    }
    if (!live) return;
    // C layout: register r of lane (hl, c32) = row (r & 3) + 8 * (r >> 2) + 4 * hl, column c32 of the block
#pragma unroll
    for (int j = 0; j < CW; ++j) {
        const int col = (cb + j) * 32 + c32;
        const float bb = bias_r[j];
#pragma unroll
        for (int i = 0; i < RW; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (ru * RW + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hl;
                float v = acc[i][j][r] + bb;
                if (a.leaky) v = leaky02(v);
                if (row < a.R) a.y[(size_t)row * a.ldy + col] = v;
            }
    }
}

uint16_t piece_of(float w, int plane)
{
    auto trunc16 = [](float x) {
        uint32_t u;
        __builtin_memcpy(&u, &x, 4);
        u &= 0xffff0000u;
        float r;
        __builtin_memcpy(&r, &u, 4);
        return r;
    };
    volatile float w1 = trunc16(w);
    volatile float r1 = w - w1;
    volatile float w2 = trunc16(r1);
    volatile float w3 = r1 - w2;
    const float pick = plane == 0 ? w1 : (plane == 1 ? w2 : w3);
    uint32_t u;
    __builtin_memcpy(&u, &pick, 4);
    return (uint16_t)(u >> 16);
}

}  // namespace

// [cin, cout] row-major, cin % 16 == 0, cout % 32 == 0  ->  cin * cout * 3 uint16:
//   image[(((cb * nq + q) * 3 + plane) * 64 + lane) * 8 + t] = piece `plane` of W[16 q + 8 (lane >> 5) + t][32 cb + (lane & 31)]
void pack_p32b(const float* W, int cin, int cout, uint16_t* out)
{
    const int nq = cin / 16, ncb = cout / 32;
    for (int cb = 0; cb < ncb; ++cb)
        for (int q = 0; q < nq; ++q)
            for (int pl = 0; pl < 3; ++pl)
                for (int lane = 0; lane < 64; ++lane)
                    for (int t = 0; t < 8; ++t) {
                        const int k = 16 * q + 8 * (lane >> 5) + t, n = 32 * cb + (lane & 31);
                        out[((((size_t)cb * nq + q) * 3 + pl) * 64 + lane) * 8 + t] = piece_of(W[(size_t)k * cout + n], pl);
                    }
}

bool gemm32b_fits(const PackedLinear& L, const RowSrc& s1, const RowSrc& s2, int64_t R, int ldy)
{
    return L.w32b && !L.accum && R <= 32768 && L.cin % 16 == 0 && L.cout % 32 == 0 && s1.c % 16 == 0 && s2.c % 16 == 0 && s1.c + s2.c == L.cin &&
           s1.ld % 4 == 0 && (s2.c == 0 || s2.ld % 4 == 0) && (reinterpret_cast<uintptr_t>(s1.x) & 15) == 0 &&
           (s2.c == 0 || (reinterpret_cast<uintptr_t>(s2.x) & 15) == 0) && ldy > 0;
}

int gemm32b(ps_context* c, const PackedLinear& L, const RowSrc& s1, const RowSrc& s2, int64_t R, float* y, int ldy)
{
    if (R <= 0) return PS_OK;
    PS_CHECK(gemm32b_fits(L, s1, s2, R, ldy), "gemm32b: shape / alignment not supported (cin %d, cout %d)", L.cin, L.cout);
    Gemm32bArgs a;
    a.x1 = s1.x; a.g1 = s1.gather; a.ld1 = s1.ld; a.c1 = s1.c; a.g1m = s1.gm; a.g1n = s1.gn;
    a.x2 = s2.x; a.g2 = s2.gather; a.ld2 = s2.ld; a.c2 = s2.c; a.g2m = s2.gm; a.g2n = s2.gn;
    a.wp = reinterpret_cast<const uint4*>(L.w32b); a.bias = L.bias; a.y = y; a.ldy = ldy; a.R = (int)R; a.cin = L.cin; a.cout = L.cout; a.leaky = L.leaky;
    const int rblocks = (int)((R + 31) / 32);
    // two column blocks per wave (an activation split feeds twelve MFMAs) whenever the layer has them; two row blocks per wave (a weight
    // fragment feeds both: half the weight stream) once that still leaves every SIMD a wave
    int cw = L.cout % 64 == 0 ? 2 : 1;
    int rw = (int64_t)(rblocks / 2) * (L.cout / (32 * cw)) >= 1024 ? 2 : 1;
    // (A/B overrides of the tile shape, PS_GEMM32B_RW / PS_GEMM32B_CW: only the compiled shapes 1 and 2; anything else is ignored)
    if (c->tune.gemm32b_rw == 1 || c->tune.gemm32b_rw == 2) rw = c->tune.gemm32b_rw;
    if ((c->tune.gemm32b_cw == 1 || c->tune.gemm32b_cw == 2) && L.cout % (32 * c->tune.gemm32b_cw) == 0) cw = c->tune.gemm32b_cw;
    const int cgroups = L.cout / (32 * cw);
    const int runits = (rblocks + rw - 1) / rw;
    // split K across the waves of a workgroup while the plain grid leaves SIMDs idle (1 024 of them) and the slices keep >= 4 chunks
    const int64_t units = (int64_t)runits * cgroups;
    int sk = 1;
    // (eight waves per workgroup -- gemm32.hip's form for the long-K layers -- measured SLOWER here: dec1 24.8 -> 30.8 us, pipelined step
    //  0.857 -> 0.875 ms: twice the partial sums through LDS for a chain that the register ring already keeps fed)
    while (sk < 4 && units * sk < 1536 && L.cin / 16 / (sk * 2) >= 4) sk *= 2;
    const int ru_per_wg = 4 / sk;
    a.cgroups = cgroups;
    a.rgroups = (runits + ru_per_wg - 1) / ru_per_wg;
    const unsigned grid = 8u * (unsigned)((a.rgroups * a.cgroups + 7) / 8);
    const dim3 block(256);
#define PS_G32B(RW_, CW_)                                                                                            \
    if (sk == 1) hipLaunchKernelGGL((gemm32b_kernel<RW_, CW_, 1>), dim3(grid), block, 0, c->stream, a);              \
    else if (sk == 2) hipLaunchKernelGGL((gemm32b_kernel<RW_, CW_, 2>), dim3(grid), block, 0, c->stream, a);         \
    else hipLaunchKernelGGL((gemm32b_kernel<RW_, CW_, 4>), dim3(grid), block, 0, c->stream, a)
    if (rw == 2 && cw == 2) { PS_G32B(2, 2); }
    else if (rw == 2) { PS_G32B(2, 1); }
    else if (cw == 2) { PS_G32B(1, 2); }
    else { PS_G32B(1, 1); }
#undef PS_G32B
    PS_HIP(hipGetLastError());
    return PS_OK;
}

}  // namespace ps
