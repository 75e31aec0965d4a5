// attpool32.hip -- fused LocSE + neighbour gather + attentive pooling for d_out >= 64 on v_mfma_f32_32x32x2_f32.
//
// Same computation as attpool.hip (building_block / relative_pos_encoding / gather_neighbour / att_pooling,
// PointSegment/RandLANet.py:323-343, 377-401, up to att_pooling's trailing conv2d; scores in the pre-product form
// G[idx] + f_xyz . Wfc[d/2:, :]), re-tiled:
//   * one MFMA tile = 32 rows = the K neighbours of 32/K points (two points at K = 16, one at K = 32) x 32 channels.  Against
//     the 16x16x4 tiling that is half the MFMA instructions and half the operand fetches per FLOP, the two points of a tile
//     share every weight fragment, and a channel's K scores sit in 8 (16) registers of two lanes: the softmax reductions
//     are in-lane plus ONE v_permlane32_swap instead of two cross-lane steps.
//   * operands move 16 bytes per lane: the K axis of every product is taken in the order {8q + 4*half + t} (q = chunk of 8
//     input channels, half = lane >> 5, t = MFMA step within the chunk), so a lane's operand values of four consecutive
//     MFMAs are one ds_read_b128 of the activation tile and one 16-byte read of the weight image packed to match (pack_p32).
//   * the two small products (LocSE mlp1, LFA mlp2) are evaluated transposed (C[channel][row]): a lane's four consecutive
//     accumulator registers are four consecutive channels of its row and go to the LDS tile as one ds_write_b128.
//   * d <= 128 (levels 1-2: many points): every wave owns a tile, eight waves per workgroup, the level's three weight
//     matrices live in LDS (52 KB at d = 128) -- no weight traffic per point at all; LFA mlp2's output is held in registers
//     until the wave has consumed the mlp1 tile and then overwrites it in place (one tile per wave).
//     d >= 256 (levels 3-4: few points): the waves of a workgroup share one tile and split its output-column blocks; the
//     weights stream from L2 with 16-byte loads (a fragment serves two points).
// LDS tile pitch = H + 4 floats = 4 x odd: conflict-free for the b128 reads and writes above (MI355X_MICROARCH.md, LDS).
//
// Bound: fp32 MFMA.  Per point: (d/32) * (h/2) score MFMAs of 64 cycles per pair of points, e.g. d = 128: 128 + 64 (mlp2) + 10.
#include "attpool.h"
#include "mfma_tile.h"

namespace ps {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

struct Att32Args {
    const float* xyz;
    const int32_t* idx;
    const int32_t* order;
    const float* fg;
    const float* w1; const float* b1;  // LocSE mlp1: [H/32][5][64] image
    const float* w2; const float* b2;  // LFA mlp2:   pack_p32 image of [H, H] (stage 2)
    const float* wb;                   // Wfc[H:, :]: pack_p32 image of [H, D]
    float* agg;
    int n_total, n_cloud;
};

__device__ __forceinline__ float swap32_max(float v)
{
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float swap32_sum(float v)
{
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

template <int D, int STAGE, int KN, int WAVES, bool SPLITN>
__global__ __launch_bounds__(WAVES * 64) void att32_kernel(Att32Args a)
{
    constexpr int H = D / 2, LDF = H + D, PITCH = H + 4, PPT = 32 / KN, RP = 16 / PPT;
    constexpr int CBH = H / 32, CBD = D / 32, NQ = H / 8;
    constexpr int W1F = CBH * 5 * 64, W2F = STAGE == 2 ? H * H : 0, WBF = H * D;
    constexpr int TILE = 32 * PITCH;
    static_assert(H % 32 == 0 && (KN == 16 || KN == 32), "att32: d_out >= 64, K in {16, 32}");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int hl = lane >> 5, c32 = lane & 31;
    // LDS: [weights + biases (wave-per-tile form only)] [per tile owner: 32 neighbour rows | tile(s)]
    float* Ws = smem;
    constexpr int WTOT = SPLITN ? 0 : (W1F + W2F + WBF + 2 * H);
    float* own = smem + WTOT + (SPLITN ? 0 : wave) * (32 + TILE);
    int* NB = reinterpret_cast<int*>(own);
    float* T1 = own + 32;
    float* T2 = SPLITN && STAGE == 2 ? T1 + TILE : T1;
    const float *w1 = a.w1, *w2 = a.w2, *wb = a.wb, *b1 = a.b1, *b2 = a.b2;
    if constexpr (!SPLITN) {
        for (int i = threadIdx.x; i < W1F; i += WAVES * 64) Ws[i] = a.w1[i];
        for (int i = threadIdx.x; i < W2F / 4; i += WAVES * 64) reinterpret_cast<float4*>(Ws + W1F)[i] = reinterpret_cast<const float4*>(a.w2)[i];
        for (int i = threadIdx.x; i < WBF / 4; i += WAVES * 64) reinterpret_cast<float4*>(Ws + W1F + W2F)[i] = reinterpret_cast<const float4*>(a.wb)[i];
        for (int i = threadIdx.x; i < H; i += WAVES * 64) {
            Ws[W1F + W2F + WBF + i] = a.b1[i];
            Ws[W1F + W2F + WBF + H + i] = STAGE == 2 ? a.b2[i] : 0.f;
        }
        __syncthreads();
        w1 = Ws; w2 = Ws + W1F; wb = Ws + W1F + W2F; b1 = Ws + W1F + W2F + WBF; b2 = b1 + H;
    }
    constexpr int CBSTEP = SPLITN ? WAVES : 1;
    const int cb0 = SPLITN ? wave : 0;
    auto phase_sync = [&]() {
        if constexpr (SPLITN) __syncthreads();
        else wave_lds_sync();
    };

    // tiles walk a contiguous eighth of the points per XCD (PointWalk of attpool.hip), PPT consecutive points per tile
    const int per_xcd = ((((a.n_total + 7) >> 3) + PPT - 1) / PPT) * PPT;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const int tiles_per_wg = SPLITN ? 1 : WAVES;
    const int t_end = min(a.n_total, (xcd + 1) * per_xcd);
    // Geometry of a tile = three dependent gathers (leaf order -> neighbour index -> coordinates): the NEXT tile's chain is issued
    // in pieces between the phases of the current tile (gstage 0..2), so none of its latency is exposed.
    const int t_first = xcd * per_xcd + (slot * tiles_per_wg + (SPLITN ? 0 : wave)) * PPT, t_step = slots * tiles_per_wg * PPT;
    int n_pp[PPT], n_nl = 0;
    float n_c[3], n_n[3];
    // (the kernel is VALU-issue bound at d <= 128 -- one VALU instruction per SIMD every four cycles --, so the index arithmetic is
    //  kept lean: no integer division for a single cloud, 24-bit multiplies, 32-bit element offsets from uniform bases)
    const bool one_cloud = a.n_total == a.n_cloud;
    auto cloud_base = [&](int row) { return one_cloud ? 0 : (row / a.n_cloud) * a.n_cloud; };
    auto gstage = [&](int st, int t0n) {
        if (t0n >= t_end) return;
        if (st == 0) {
#pragma unroll
            for (int i = 0; i < PPT; ++i) {
                const int t = min(t0n + i, t_end - 1);
                n_pp[i] = a.order ? cloud_base(t) + a.order[t] : t;
            }
        } else if (st == 1) {
            const unsigned p = PPT == 2 ? (c32 >= KN ? n_pp[PPT - 1] : n_pp[0]) : n_pp[0];
            n_nl = a.idx[p * (unsigned)KN + (unsigned)(c32 & (KN - 1))];
            const float* cp = a.xyz + 3u * p;
            n_c[0] = cp[0]; n_c[1] = cp[1]; n_c[2] = cp[2];
        } else {
            const int p = PPT == 2 ? (c32 >= KN ? n_pp[PPT - 1] : n_pp[0]) : n_pp[0];
            n_nl += cloud_base(p);
            const float* np = a.xyz + 3u * (unsigned)n_nl;
            n_n[0] = np[0]; n_n[1] = np[1]; n_n[2] = np[2];
        }
    };
    gstage(0, t_first);
    gstage(1, t_first);
    gstage(2, t_first);
    for (int t0 = t_first; t0 < t_end; t0 += t_step) {
        // ---- geometry of this lane's row: (point t0 + row / KN, neighbour row % KN) ----
        int pp[PPT];
#pragma unroll
        for (int i = 0; i < PPT; ++i) pp[i] = n_pp[i];
        const int nbr = n_nl;
        const float cx = n_c[0], cy = n_c[1], cz = n_c[2];
        const float nx = n_n[0], ny = n_n[1], nz = n_n[2];
        const float rx = cx - nx, ry = cy - ny, rz = cz - nz;
        const float dis = __builtin_amdgcn_sqrtf(rx * rx + ry * ry + rz * rz);
        // enc10 = [dis, rx, ry, rz, cx, cy, cz, nx, ny, nz]; MFMA step s takes elements 2s (lanes 0-31) and 2s + 1 (lanes 32-63)
        float e[5];
        e[0] = hl ? rx : dis; e[1] = hl ? rz : ry; e[2] = hl ? cy : cx; e[3] = hl ? nx : cz; e[4] = hl ? nz : ny;
        if (hl == 0 && (!SPLITN || wave == 0)) NB[c32] = nbr;
        gstage(0, t0 + t_step);

        // ---- LFA mlp1 (transposed: C[channel][row]): f_xyz1 = lrelu(enc10 . W1 + b1) -> T1 ----
#pragma unroll
        for (int cb = cb0; cb < CBH; cb += CBSTEP) {
            f32x16 acc;  // seeded with the bias (register r of this lane = channel 32 cb + 8 (r >> 2) + 4 hl + (r & 3))
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 bb = *reinterpret_cast<const float4*>(b1 + cb * 32 + g4 * 8 + hl * 4);
                acc[4 * g4] = bb.x; acc[4 * g4 + 1] = bb.y; acc[4 * g4 + 2] = bb.z; acc[4 * g4 + 3] = bb.w;
            }
#pragma unroll
            for (int s = 0; s < 5; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[(cb * 5 + s) * 64 + lane], e[s], acc, 0, 0, 0);
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int ch = cb * 32 + g4 * 8 + hl * 4;
                float4 o;
                o.x = leaky02(acc[4 * g4]); o.y = leaky02(acc[4 * g4 + 1]);
                o.z = leaky02(acc[4 * g4 + 2]); o.w = leaky02(acc[4 * g4 + 3]);
                *reinterpret_cast<float4*>(T1 + c32 * PITCH + ch) = o;
            }
        }
        phase_sync();
        gstage(1, t0 + t_step);
        if constexpr (STAGE == 2) {
            // ---- LFA mlp2 (transposed): f_xyz2 = lrelu(f_xyz1 . W2 + b2) ----
            constexpr int NACC = SPLITN ? 1 : CBH;
            f32x16 acc2[NACC];
#pragma unroll
            for (int cb = cb0; cb < CBH; cb += CBSTEP) {
                f32x16& acc = acc2[SPLITN ? 0 : cb];
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {  // seeded with the bias
                    const float4 bb = *reinterpret_cast<const float4*>(b2 + cb * 32 + g4 * 8 + hl * 4);
                    acc[4 * g4] = bb.x; acc[4 * g4 + 1] = bb.y; acc[4 * g4 + 2] = bb.z; acc[4 * g4 + 3] = bb.w;
                }
                const float4* wq = reinterpret_cast<const float4*>(w2) + (size_t)cb * NQ * 64 + lane;
                const float* xr = T1 + c32 * PITCH + 4 * hl;
#pragma unroll 8
                for (int q = 0; q < NQ; ++q) {
                    const float4 aw = wq[(size_t)q * 64];
                    const float4 bx = *reinterpret_cast<const float4*>(xr + 8 * q);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aw.x, bx.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aw.y, bx.y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aw.z, bx.z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aw.w, bx.w, acc, 0, 0, 0);
                }
                if constexpr (SPLITN) {
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int ch = cb * 32 + g4 * 8 + hl * 4;
                        float4 o;
                        o.x = leaky02(acc[4 * g4]); o.y = leaky02(acc[4 * g4 + 1]);
                        o.z = leaky02(acc[4 * g4 + 2]); o.w = leaky02(acc[4 * g4 + 3]);
                        *reinterpret_cast<float4*>(T2 + c32 * PITCH + ch) = o;
                    }
                }
            }
            if constexpr (!SPLITN) {
                wave_lds_sync();  // every read of the mlp1 tile precedes the writes below (DS operations of a wave execute in order)
#pragma unroll
                for (int cb = 0; cb < CBH; ++cb)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int ch = cb * 32 + g4 * 8 + hl * 4;
                        float4 o;
                        o.x = leaky02(acc2[cb][4 * g4]); o.y = leaky02(acc2[cb][4 * g4 + 1]);
                        o.z = leaky02(acc2[cb][4 * g4 + 2]); o.w = leaky02(acc2[cb][4 * g4 + 3]);
                        *reinterpret_cast<float4*>(T1 + c32 * PITCH + ch) = o;
                    }
            }
            phase_sync();
        }
        const float* TX = T2;

        // ---- scores (C[row][channel]) = G[nbr] + f_xyz . Wfc[H:, :], softmax over the K rows of a point, weighted sum ----
        // accumulator register r of this lane is row (r & 3) + 8 * (r >> 2) + 4 * hl of the tile
        // byte offset of (that row's neighbour, this lane's column of the wave's first column block) in fg: the column blocks that
        // follow are compile-time byte offsets of the loads (instruction immediates): no address arithmetic per gather
        unsigned off[16];
        {
            const unsigned col0 = (unsigned)(cb0 * 32 + c32) * 4u;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int4 nb4 = *reinterpret_cast<const int4*>(NB + 8 * g4 + 4 * hl);
                off[4 * g4] = __umul24(nb4.x, LDF * 4u) + col0; off[4 * g4 + 1] = __umul24(nb4.y, LDF * 4u) + col0;  // rows < 2^24 (att_pool32_fits)
                off[4 * g4 + 2] = __umul24(nb4.z, LDF * 4u) + col0; off[4 * g4 + 3] = __umul24(nb4.w, LDF * 4u) + col0;
            }
        }
        const char* fgb = reinterpret_cast<const char*>(a.fg);
        // The gathers of column block i + 1 (G rows for the scores, f rows for the values) are issued before the MFMAs of block i
        // and consumed after them: their latency hides behind ~2 000 cycles of matrix work.  G is ADDED after the product
        // (instead of seeding the accumulator) for the same reason.
        constexpr int NCB = CBD / CBSTEP;
        f32x2 gq[2][8], v[2][8];  // register pairs: the softmax arithmetic below runs on v_pk_*_f32 (two scores per instruction)
        auto gather = [&](int i, f32x2 (&gdst)[8], f32x2 (&vdst)[8]) {
            const int rel = i * CBSTEP * 32 * 4;  // compile-time (the loop below is fully unrolled)
#pragma unroll
            for (int r = 0; r < 16; ++r) gdst[r >> 1][r & 1] = *reinterpret_cast<const float*>(fgb + off[r] + (rel + H * 4));
            if ((cb0 + i * CBSTEP) * 32 < H) {  // wave-uniform: columns < H take their values from the gathered neighbour features
#pragma unroll
                for (int r = 0; r < 16; ++r) vdst[r >> 1][r & 1] = *reinterpret_cast<const float*>(fgb + off[r] + rel);
            }
        };
        gather(0, gq[0], v[0]);
#pragma unroll
        for (int i = 0; i < NCB; ++i) {
            const int cb = cb0 + i * CBSTEP;
            if (i + 1 < NCB) gather(i + 1, gq[(i + 1) & 1], v[(i + 1) & 1]);
            if (i == 0) gstage(2, t0 + t_step);
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const float4* wq = reinterpret_cast<const float4*>(wb) + (size_t)cb * NQ * 64 + lane;
            const float* xr = TX + c32 * PITCH + 4 * hl;
#pragma unroll 8
            for (int q = 0; q < NQ; ++q) {
                const float4 ax = *reinterpret_cast<const float4*>(xr + 8 * q);
                const float4 bw = wq[(size_t)q * 64];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ax.x, bw.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ax.y, bw.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ax.z, bw.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ax.w, bw.w, acc, 0, 0, 0);
            }
            f32x2 (&vv)[8] = v[i & 1];
            if (cb * 32 >= H) {  // values = f_xyz (LDS tile)
                const float* tv = TX + (cb * 32 - H + c32) + 4 * hl * PITCH;
#pragma unroll
                for (int r = 0; r < 16; ++r) vv[r >> 1][r & 1] = tv[((r & 3) + 8 * (r >> 2)) * PITCH];
            }
            // the Wfc[H:, :] image is pre-multiplied by log2(e) (pack_p32 call in randla.hip), the gathered G joins with the same factor:
            // softmax(s) = exp2(s' - max s') with s' = s log2(e) -- one multiply per score less than expf
            f32x2 sc[8];
            const f32x2 l2e = {1.4426950408889634f, 1.4426950408889634f};
#pragma unroll
            for (int j = 0; j < 8; ++j) sc[j] = __builtin_elementwise_fma(gq[i & 1][j], l2e, f32x2{acc[2 * j], acc[2 * j + 1]});
            constexpr int PP = RP / 2;  // register pairs per point
#pragma unroll
            for (int pi = 0; pi < PPT; ++pi) {
                float m = fmaxf(sc[pi * PP][0], sc[pi * PP][1]);
#pragma unroll
                for (int j = 1; j < PP; ++j) m = fmaxf(m, fmaxf(sc[pi * PP + j][0], sc[pi * PP + j][1]));
                m = swap32_max(m);
                const f32x2 mm = {m, m};
                f32x2 ssum2 = {0.f, 0.f}, num2 = {0.f, 0.f};
#pragma unroll
                for (int j = 0; j < PP; ++j) {
                    const f32x2 dd = sc[pi * PP + j] - mm;
                    const f32x2 ex = {__builtin_amdgcn_exp2f(dd[0]), __builtin_amdgcn_exp2f(dd[1])};
                    ssum2 += ex;
                    num2 = __builtin_elementwise_fma(ex, vv[pi * PP + j], num2);
                }
                const float ssum = swap32_sum(ssum2[0] + ssum2[1]);
                const float num = swap32_sum(num2[0] + num2[1]);
                if (hl == 0 && t0 + pi < t_end) a.agg[__umul24(pp[pi], D) + (unsigned)(cb * 32 + c32)] = num * __builtin_amdgcn_rcpf(ssum);
            }
        }
        phase_sync();  // the tile and the neighbour rows are overwritten by the next tile
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------
void pack_p32(const float* W, int cin, int cout, float* out)
{
    const int nq = cin / 8, cbs = cout / 32;
    for (int cb = 0; cb < cbs; ++cb)
        for (int q = 0; q < nq; ++q)
            for (int l = 0; l < 64; ++l)
                for (int t = 0; t < 4; ++t)
                    out[((((size_t)cb * nq + q) * 64) + l) * 4 + t] = W[(size_t)(8 * q + 4 * (l >> 5) + t) * cout + 32 * cb + (l & 31)];
}

void pack_p32_locse(const float* W1, int cout, float* out)
{
    const int cbs = cout / 32;
    for (int cb = 0; cb < cbs; ++cb)
        for (int s = 0; s < 5; ++s)
            for (int l = 0; l < 64; ++l) out[((size_t)cb * 5 + s) * 64 + l] = W1[(size_t)(2 * s + (l >> 5)) * cout + 32 * cb + (l & 31)];
}

template <int D, int STAGE, int KN>
static int launch_att32(ps_context* c, const Att32Args& a)
{
    constexpr int H = D / 2, PITCH = H + 4, TILE = 32 * PITCH;
    if constexpr (D <= 128) {
        // d = 128: 52 KB of weights + twelve 8.8 KB tiles = 158 KB: one workgroup of twelve waves per CU (three per SIMD);
        // d = 64: 14 KB + eight 4.7 KB tiles: two workgroups of eight waves per CU (register-limited)
        constexpr int WAVES = D == 128 ? 12 : 8;
        constexpr int PER_CU = D == 128 ? 1 : 2;
        constexpr size_t smem = sizeof(float) * ((size_t)(H / 32) * 5 * 64 + (STAGE == 2 ? H * H : 0) + H * D + 2 * H + (size_t)WAVES * (32 + TILE));
        static_assert(smem <= 160 * 1024, "att32: weights + tiles exceed the LDS");
        auto kern = att32_kernel<D, STAGE, KN, WAVES, false>;
        PS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        const int tiles = ceil_div(a.n_total, 32 / KN);
        const int blocks = (std::min(ceil_div(tiles, WAVES), 256 * PER_CU) + 7) & ~7;  // persistent: the weights are staged once per workgroup
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(WAVES * 64), smem, c->stream, a);
    } else {
        constexpr int WAVES = D >= 512 ? 8 : 4;
        constexpr size_t smem = sizeof(float) * (32 + (size_t)TILE * (STAGE == 2 ? 2 : 1));
        auto kern = att32_kernel<D, STAGE, KN, WAVES, true>;
        if (smem > 48 * 1024) PS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        const int tiles = ceil_div(a.n_total, 32 / KN);
        const int blocks = (std::min(tiles, 256 * 16) + 7) & ~7;
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(WAVES * 64), smem, c->stream, a);
    }
    PS_HIP(hipGetLastError());
    return PS_OK;
}

template <int STAGE, int KN>
static int dispatch32(ps_context* c, int d, const Att32Args& a)
{
    switch (d) {
        case 64: return launch_att32<64, STAGE, KN>(c, a);
        case 128: return launch_att32<128, STAGE, KN>(c, a);
        case 256: return launch_att32<256, STAGE, KN>(c, a);
        case 512: return launch_att32<512, STAGE, KN>(c, a);
        default: set_error("att_pool32: d_out %d is not a compiled size (64, 128, 256, 512)", d); return PS_EINVAL;
    }
}

bool att_pool32_fits(const AttStage& s)
{
    // 32-bit byte offsets into fg: rows * (h + d) * 4 bytes must stay below 4 GiB
    return s.p32 && s.wbot && s.d >= 64 && (s.k == 16 || s.k == 32) && s.ldf == s.d / 2 + s.d && s.n_total < (1 << 24) &&
           (uint64_t)s.n_total * (uint64_t)(s.d / 2 + s.d) * 4u < (1ull << 32);
}

int att_pool32_stage(ps_context* c, const AttStage& s)
{
    Att32Args a;
    a.xyz = s.xyz; a.idx = s.idx; a.order = s.order; a.fg = s.fg;
    a.w1 = s.p32->w1; a.b1 = s.lfa1->bias;
    a.w2 = s.lfa2 ? s.p32->w2 : nullptr; a.b2 = s.lfa2 ? s.lfa2->bias : nullptr;
    a.wb = s.lfa2 ? s.p32->wb2 : s.p32->wb1;
    a.agg = s.agg;
    a.n_total = (int)s.n_total; a.n_cloud = (int)s.n_cloud;
    if (s.n_total <= 0) return PS_OK;
    const int stage = s.lfa2 ? 2 : 1;
    if (s.k == 16) return stage == 1 ? dispatch32<1, 16>(c, s.d, a) : dispatch32<2, 16>(c, s.d, a);
    return stage == 1 ? dispatch32<1, 32>(c, s.d, a) : dispatch32<2, 32>(c, s.d, a);
}

}  // namespace ps
