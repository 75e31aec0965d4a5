// attpool32b.hip -- attpool32.hip's computation with every matrix product on v_mfma_f32_32x32x16_bf16 over EXACT three-way
// bfloat16 splits of the fp32 operands ("bf16x3").
//
//   x = x1 + x2 + x3,  x1 = x with the low 16 bits cleared, x2 = (x - x1) with the low 16 bits cleared, x3 = x - x1 - x2:
//   each piece has 8 significant bits, the subtractions are exact, and 8 + 8 + 8 = the 24 bits of an fp32 significand -- the
//   three bfloat16 values add up to x exactly.  A product of two pieces is exact in fp32, the MFMA accumulates in fp32, and of the
//   nine piece products the six of relative size >= 2^-16 are kept:  x.w ~ x1w1 + x1w2 + x2w1 + x1w3 + x3w1 + x2w2 (dropped:
//   x2w3 + x3w2 + x3w3 <= 2^-23 |x||w|, the size of ONE fp32 rounding of the product).  The result carries fp32-level error --
//   the logits parity against the float64 oracle is unchanged (tests/test_gpu_network.py) -- while six 32-cycle bf16 MFMAs cover
//   K = 16 where the fp32 MFMA needs eight 64-cycle ones: 2.7 x less matrix-pipe time.  (The fp32 MFMA is 1/16 of the bf16 rate
//   on gfx950, MI355X_MICROARCH.md.)
//
// What changes against attpool32.hip besides the instruction:
//   * activations never go through LDS as operands: the accumulator registers of a transposed product (C[channel][row]) ARE the
//     next product's operand registers of the same lane (tile row = lane & 31, eight K values per lane half) -- LeakyReLU, split,
//     done; the K axis of every product is simply taken in accumulator order (host images packed to match, pack_b3).  The fp32
//     tile in LDS only serves the value reads of the weighted sum.
//   * weights are stored as three planes of eight bfloat16 per lane (16-byte reads, [block][chunk][plane][lane]).
// d = 64 and 128 (levels 1-2): weights resident in LDS, a wave per tile of two points (att32b_kernel);
// d = 256 and 512 (levels 3-4): a workgroup per tile, weight planes streamed from L2 (att32s_kernel).
#include "attpool.h"
#include "mfma_tile.h"

namespace ps {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

struct Att32bArgs {
    const float* xyz;
    const int32_t* idx;
    const int32_t* order;
    const float* fg;
    const uint4* w1; const float* b1;  // LocSE mlp1: pack_b3_locse image
    const uint4* w2; const float* b2;  // LFA mlp2: pack_b3 image of [H, H] (stage 2)
    const uint4* wb;                   // Wfc[H:, :] (times log2 e): pack_b3 image of [H, D]
    float* agg;
    int n_total, n_cloud;
};

__device__ __forceinline__ float swap32b_max(float v)
{
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float swap32b_sum(float v)
{
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// three bfloat16 planes of eight fp32 values: plane word t = pieces of (x[2t], x[2t+1]), low half = the even element
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
__device__ __forceinline__ Planes split8(const float (&x)[8])
{
    Planes r;
    split_pair(x[0], x[1], r.p[0].x, r.p[1].x, r.p[2].x);
    split_pair(x[2], x[3], r.p[0].y, r.p[1].y, r.p[2].y);
    split_pair(x[4], x[5], r.p[0].z, r.p[1].z, r.p[2].z);
    split_pair(x[6], x[7], r.p[0].w, r.p[1].w, r.p[2].w);
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
__device__ __forceinline__ Planes load_planes(const uint4* img, int slot, int lane)
{
    Planes r;
    r.p[0] = img[(slot * 3 + 0) * 64 + lane];
    r.p[1] = img[(slot * 3 + 1) * 64 + lane];
    r.p[2] = img[(slot * 3 + 2) * 64 + lane];
    return r;
}

template <int D, int STAGE, int KN, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void att32b_kernel(Att32bArgs a)
{
    constexpr int H = D / 2, LDF = H + D, PITCH = H + 4, PPT = 32 / KN, RP = 16 / PPT;
    constexpr int CBH = H / 32, CBD = D / 32, NQ = H / 16;
    constexpr int W1Q = CBH * 3 * 64, W2Q = STAGE == 2 ? CBH * NQ * 3 * 64 : 0, WBQ = CBD * NQ * 3 * 64;  // uint4 counts
    constexpr int TILE = 32 * PITCH;
    static_assert(H % 32 == 0 && (KN == 16 || KN == 32), "att32b: d_out >= 64, K in {16, 32}");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int hl = lane >> 5, c32 = lane & 31;
    // LDS: [weight planes | biases] [per wave: 32 neighbour rows | fp32 tile]
    uint4* Wq = reinterpret_cast<uint4*>(smem);
    constexpr int WTOT = (W1Q + W2Q + WBQ) * 4 + 2 * H;  // floats
    float* own = smem + WTOT + wave * (32 + TILE);
    int* NB = reinterpret_cast<int*>(own);
    float* T1 = own + 32;
    for (int i = threadIdx.x; i < W1Q; i += WAVES * 64) Wq[i] = a.w1[i];
    for (int i = threadIdx.x; i < W2Q; i += WAVES * 64) Wq[W1Q + i] = a.w2[i];
    for (int i = threadIdx.x; i < WBQ; i += WAVES * 64) Wq[W1Q + W2Q + i] = a.wb[i];
    float* bias = smem + (W1Q + W2Q + WBQ) * 4;
    for (int i = threadIdx.x; i < H; i += WAVES * 64) {
        bias[i] = a.b1[i];
        bias[H + i] = STAGE == 2 ? a.b2[i] : 0.f;
    }
    __syncthreads();
    const uint4 *w1 = Wq, *w2 = Wq + W1Q, *wb = Wq + W1Q + W2Q;
    const float *b1 = bias, *b2 = bias + H;

    // tiles walk a contiguous eighth of the points per XCD, PPT consecutive points per tile (as attpool32.hip)
    const int per_xcd = ((((a.n_total + 7) >> 3) + PPT - 1) / PPT) * PPT;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const int t_end = min(a.n_total, (xcd + 1) * per_xcd);
    const int t_first = xcd * per_xcd + (slot * WAVES + wave) * PPT, t_step = slots * WAVES * PPT;
    int n_pp[PPT], n_nl = 0;
    float n_c[3], n_n[3];
    const bool one_cloud = a.n_total == a.n_cloud;
    auto cloud_base = [&](int row) { return one_cloud ? 0 : (row / a.n_cloud) * a.n_cloud; };
    auto gstage = [&](int st, int t0n) {  // the NEXT tile's three dependent gathers, issued in pieces between the phases of this one
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
        int pp[PPT];
#pragma unroll
        for (int i = 0; i < PPT; ++i) pp[i] = n_pp[i];
        const int nbr = n_nl;
        const float cx = n_c[0], cy = n_c[1], cz = n_c[2];
        const float nx = n_n[0], ny = n_n[1], nz = n_n[2];
        const float rx = cx - nx, ry = cy - ny, rz = cz - nz;
        const float dis = __builtin_amdgcn_sqrtf(rx * rx + ry * ry + rz * rz);
        // enc10 = [dis, rx, ry, rz, cx, cy, cz, nx | ny, nz]: the lower lane half holds K values 0..7, the upper 8..15 (10.. are zero)
        const float ev[8] = {hl ? ny : dis, hl ? nz : rx, hl ? 0.f : ry, hl ? 0.f : rz, hl ? 0.f : cx, hl ? 0.f : cy, hl ? 0.f : cz, hl ? 0.f : nx};
        const Planes E = split8(ev);
        if (hl == 0) NB[c32] = nbr;
        gstage(0, t0 + t_step);
        wave_lds_sync();
        // byte offset of (row r's neighbour, this lane's column of the first column block) in fg; the column blocks that follow are
        // compile-time byte offsets of the loads.  accumulator register r of a score tile is row (r & 3) + 8 * (r >> 2) + 4 * hl.
        unsigned off[16];
        {
            const unsigned col0 = (unsigned)c32 * 4u;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int4 nb4 = *reinterpret_cast<const int4*>(NB + 8 * g4 + 4 * hl);
                off[4 * g4] = __umul24(nb4.x, LDF * 4u) + col0; off[4 * g4 + 1] = __umul24(nb4.y, LDF * 4u) + col0;
                off[4 * g4 + 2] = __umul24(nb4.z, LDF * 4u) + col0; off[4 * g4 + 3] = __umul24(nb4.w, LDF * 4u) + col0;
            }
        }
        const char* fgb = reinterpret_cast<const char*>(a.fg);
        f32x2 gq[2][8], v[2][8];
        auto gather = [&](int i, f32x2 (&gdst)[8], f32x2 (&vdst)[8]) {
            const int rel = i * 32 * 4;  // compile-time (the loop below is fully unrolled)
#pragma unroll
            for (int r = 0; r < 16; ++r) gdst[r >> 1][r & 1] = *reinterpret_cast<const float*>(fgb + off[r] + (rel + H * 4));
            if (i * 32 < H) {
#pragma unroll
                for (int r = 0; r < 16; ++r) vdst[r >> 1][r & 1] = *reinterpret_cast<const float*>(fgb + off[r] + rel);
            }
        };
        gather(0, gq[0], v[0]);  // the first block's G and value rows travel under the two small products

        // ---- LFA mlp1 (transposed: C[channel][row]): f_xyz1 = lrelu(enc10 . W1 + b1) ----
        // register r of this lane = channel 32 cb + 8 (r >> 2) + 4 hl + (r & 3) of row c32
        f32x16 f1[CBH];
#pragma unroll
        for (int cb = 0; cb < CBH; ++cb) {
            f32x16 acc;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 bb = *reinterpret_cast<const float4*>(b1 + cb * 32 + g4 * 8 + hl * 4);
                acc[4 * g4] = bb.x; acc[4 * g4 + 1] = bb.y; acc[4 * g4 + 2] = bb.z; acc[4 * g4 + 3] = bb.w;
            }
            acc = mfma6(load_planes(w1, cb, lane), E, acc);
#pragma unroll
            for (int r = 0; r < 16; ++r) f1[cb][r] = leaky02(acc[r]);
        }
        gstage(1, t0 + t_step);
        Planes P[NQ];  // the operand planes of the tile that feeds the scores (chunk q = accumulator registers 8 (q & 1) .. of block q >> 1)
        auto split_block = [&](const f32x16& f, Planes& lo, Planes& hi) {
            const float x0[8] = {f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7]};
            const float x1[8] = {f[8], f[9], f[10], f[11], f[12], f[13], f[14], f[15]};
            lo = split8(x0);
            hi = split8(x1);
        };
        auto store_block = [&](const f32x16& f, int cb) {  // -> fp32 tile (value reads of the weighted sum)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                float4 o;
                o.x = f[4 * g4]; o.y = f[4 * g4 + 1]; o.z = f[4 * g4 + 2]; o.w = f[4 * g4 + 3];
                *reinterpret_cast<float4*>(T1 + c32 * PITCH + cb * 32 + g4 * 8 + hl * 4) = o;
            }
        };
#pragma unroll
        for (int cb = 0; cb < CBH; ++cb) split_block(f1[cb], P[2 * cb], P[2 * cb + 1]);
        if constexpr (STAGE == 1) {
#pragma unroll
            for (int cb = 0; cb < CBH; ++cb) store_block(f1[cb], cb);
        } else {
            // ---- LFA mlp2 (transposed): f_xyz2 = lrelu(f_xyz1 . W2 + b2); the planes of f_xyz1 are the B operands ----
            f32x16 f2[CBH];
#pragma unroll
            for (int cb = 0; cb < CBH; ++cb) {
                f32x16 acc;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const float4 bb = *reinterpret_cast<const float4*>(b2 + cb * 32 + g4 * 8 + hl * 4);
                    acc[4 * g4] = bb.x; acc[4 * g4 + 1] = bb.y; acc[4 * g4 + 2] = bb.z; acc[4 * g4 + 3] = bb.w;
                }
#pragma unroll
                for (int q = 0; q < NQ; ++q) acc = mfma6(load_planes(w2, cb * NQ + q, lane), P[q], acc);
#pragma unroll
                for (int r = 0; r < 16; ++r) f2[cb][r] = leaky02(acc[r]);
            }
#pragma unroll
            for (int cb = 0; cb < CBH; ++cb) {
                split_block(f2[cb], P[2 * cb], P[2 * cb + 1]);
                store_block(f2[cb], cb);
            }
        }
        wave_lds_sync();
        const float* TX = T1;

        // ---- scores (C[row][channel]) = G[nbr] + f_xyz . Wfc[H:, :], softmax over the K rows of a point, weighted sum ----
#pragma unroll
        for (int cb = 0; cb < CBD; ++cb) {
            if (cb + 1 < CBD) gather(cb + 1, gq[(cb + 1) & 1], v[(cb + 1) & 1]);
            if (cb == 0) gstage(2, t0 + t_step);
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int q = 0; q < NQ; ++q) acc = mfma6(P[q], load_planes(wb, cb * NQ + q, lane), acc);
            f32x2 (&vv)[8] = v[cb & 1];
            if (cb * 32 >= H) {  // values = f_xyz (LDS tile)
                const float* tv = TX + (cb * 32 - H + c32) + 4 * hl * PITCH;
#pragma unroll
                for (int r = 0; r < 16; ++r) vv[r >> 1][r & 1] = tv[((r & 3) + 8 * (r >> 2)) * PITCH];
            }
            f32x2 sc[8];
            const f32x2 l2e = {1.4426950408889634f, 1.4426950408889634f};
#pragma unroll
            for (int j = 0; j < 8; ++j) sc[j] = __builtin_elementwise_fma(gq[cb & 1][j], l2e, f32x2{acc[2 * j], acc[2 * j + 1]});
            constexpr int PP = RP / 2;  // register pairs per point
#pragma unroll
            for (int pi = 0; pi < PPT; ++pi) {
                float m = fmaxf(sc[pi * PP][0], sc[pi * PP][1]);
#pragma unroll
                for (int j = 1; j < PP; ++j) m = fmaxf(m, fmaxf(sc[pi * PP + j][0], sc[pi * PP + j][1]));
                m = swap32b_max(m);
                const f32x2 mm = {m, m};
                f32x2 ssum2 = {0.f, 0.f}, num2 = {0.f, 0.f};
#pragma unroll
                for (int j = 0; j < PP; ++j) {
                    const f32x2 dd = sc[pi * PP + j] - mm;
                    const f32x2 ex = {__builtin_amdgcn_exp2f(dd[0]), __builtin_amdgcn_exp2f(dd[1])};
                    ssum2 += ex;
                    num2 = __builtin_elementwise_fma(ex, vv[pi * PP + j], num2);
                }
                const float ssum = swap32b_sum(ssum2[0] + ssum2[1]);
                const float num = swap32b_sum(num2[0] + num2[1]);
                if (hl == 0 && t0 + pi < t_end) a.agg[__umul24(pp[pi], D) + (unsigned)(cb * 32 + c32)] = num * __builtin_amdgcn_rcpf(ssum);
            }
        }
        wave_lds_sync();  // the tile and the neighbour rows are overwritten by the next tile
    }
}


// ---- d >= 256 (levels 3-4: few points): the waves of a workgroup share one tile and split its column blocks; the weight planes
// stream from L2 (16-byte loads, one stage ahead), the tile's operand planes go through LDS ([chunk][plane][lane], 16 bytes per lane:
// every wave needs all chunks but produces only its own blocks).  Each wave runs its two score blocks together, so an operand
// plane read from LDS feeds both.
template <int D, int STAGE, int KN, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void att32s_kernel(Att32bArgs a)
{
    constexpr int H = D / 2, LDF = H + D, PITCH = H + 4, PPT = 32 / KN, RP = 16 / PPT;
    constexpr int CBH = H / 32, CBD = D / 32, NQ = H / 16;
    constexpr int PLQ = NQ * 3 * 64;  // uint4s of one set of operand planes
    constexpr int NB_H = CBH / WAVES, NB_D = CBD / WAVES;  // blocks per wave
    static_assert(CBH % WAVES == 0 && NB_D == 2 && (KN == 16 || KN == 32), "att32s: two score blocks per wave");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;  // (kept in a VGPR here: as an SGPR this kernel measured 8-22 % slower)
    const int hl = lane >> 5, c32 = lane & 31;
    // LDS: [32 neighbour rows] [operand planes A] [operand planes B (stage 2)] [fp32 value tile]
    int* NB = reinterpret_cast<int*>(smem);
    uint4* PA = reinterpret_cast<uint4*>(smem + 32);
    uint4* PB = STAGE == 2 ? PA + PLQ : PA;
    float* TV = reinterpret_cast<float*>(PB + PLQ);

    const int per_xcd = ((((a.n_total + 7) >> 3) + PPT - 1) / PPT) * PPT;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const int t_end = min(a.n_total, (xcd + 1) * per_xcd);
    const int t_first = xcd * per_xcd + slot * PPT, t_step = slots * PPT;
    int n_pp[PPT], n_nl = 0;
    float n_c[3], n_n[3];
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
    auto seed = [&](const float* b, int cb) {
        f32x16 acc;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const float4 bb = *reinterpret_cast<const float4*>(b + cb * 32 + g4 * 8 + hl * 4);
            acc[4 * g4] = bb.x; acc[4 * g4 + 1] = bb.y; acc[4 * g4 + 2] = bb.z; acc[4 * g4 + 3] = bb.w;
        }
        return acc;
    };
    // LeakyReLU of a transposed block -> its two chunks of operand planes (LDS) and, when asked, the fp32 value tile
    auto emit_block = [&](const f32x16& acc, int cb, uint4* planes, bool values) {
        float x0[8], x1[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) { x0[r] = leaky02(acc[r]); x1[r] = leaky02(acc[8 + r]); }
        const Planes lo = split8(x0), hi = split8(x1);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            planes[((2 * cb) * 3 + pl) * 64 + lane] = lo.p[pl];
            planes[((2 * cb + 1) * 3 + pl) * 64 + lane] = hi.p[pl];
        }
        if (values) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float* x = g4 < 2 ? x0 + 4 * g4 : x1 + 4 * (g4 - 2);
                *reinterpret_cast<float4*>(TV + c32 * PITCH + cb * 32 + g4 * 8 + hl * 4) = make_float4(x[0], x[1], x[2], x[3]);
            }
        }
    };
    gstage(0, t_first);
    gstage(1, t_first);
    gstage(2, t_first);
    static_assert(NB_H == 1, "att32s: one transposed block per wave");
    const Planes W1 = load_planes(a.w1, wave, lane);  // this wave's LocSE block: resident for the whole kernel
    const f32x16 seed1 = seed(a.b1, wave);
    constexpr int PD = D >= 512 ? 4 : 2;                // score-weight chunks in flight per block (measured: 2 at d = 256, 4 at d = 512)
    constexpr int PD2 = STAGE == 2 ? (D >= 512 ? 4 : 8) : 1;  // mlp2-weight chunks in flight (measured: all 8 at d = 256, 4 at d = 512)
    const int cbA = wave, cbB = wave + WAVES;
    for (int t0 = t_first; t0 < t_end; t0 += t_step) {
        int pp[PPT];
#pragma unroll
        for (int i = 0; i < PPT; ++i) pp[i] = n_pp[i];
        const int nbr = n_nl;
        const float cx = n_c[0], cy = n_c[1], cz = n_c[2];
        const float nx = n_n[0], ny = n_n[1], nz = n_n[2];
        const float rx = cx - nx, ry = cy - ny, rz = cz - nz;
        const float dis = __builtin_amdgcn_sqrtf(rx * rx + ry * ry + rz * rz);
        const float ev[8] = {hl ? ny : dis, hl ? nz : rx, hl ? 0.f : ry, hl ? 0.f : rz, hl ? 0.f : cx, hl ? 0.f : cy, hl ? 0.f : cz, hl ? 0.f : nx};
        const Planes E = split8(ev);
        if (hl == 0 && wave == 0) NB[c32] = nbr;
        gstage(0, t0 + t_step);
        // the first weight chunks of the next product do not depend on anything: they travel under LocSE and its barrier
        Planes ring[PD2], rA[PD], rB[PD];
        if constexpr (STAGE == 2) {
#pragma unroll
            for (int q = 0; q < PD2; ++q) ring[q] = load_planes(a.w2, wave * NQ + q, lane);
        } else {
#pragma unroll
            for (int q = 0; q < PD; ++q) {
                rA[q] = load_planes(a.wb, cbA * NQ + q, lane);
                rB[q] = load_planes(a.wb, cbB * NQ + q, lane);
            }
        }

        // ---- LFA mlp1 (transposed), this wave's block ----
        emit_block(mfma6(W1, E, seed1), wave, PA, STAGE == 1);
        __syncthreads();
        gstage(1, t0 + t_step);
        // neighbour-row offsets and the gathers of this wave's two score blocks (they travel under the products below)
        unsigned off[16];
        {
            const unsigned col0 = (unsigned)(wave * 32 + c32) * 4u;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int4 nb4 = *reinterpret_cast<const int4*>(NB + 8 * g4 + 4 * hl);
                off[4 * g4] = __umul24(nb4.x, LDF * 4u) + col0; off[4 * g4 + 1] = __umul24(nb4.y, LDF * 4u) + col0;
                off[4 * g4 + 2] = __umul24(nb4.z, LDF * 4u) + col0; off[4 * g4 + 3] = __umul24(nb4.w, LDF * 4u) + col0;
            }
        }
        const char* fgb = reinterpret_cast<const char*>(a.fg);
        if constexpr (STAGE == 2) {
            // ---- LFA mlp2 (transposed): weights (A) stream from L2 PD2 chunks ahead, f_xyz1's planes (B) come from LDS ----
            const int cb = wave;
            f32x16 acc = seed(a.b2, cb);
#pragma unroll 1
            for (int q0 = 0; q0 + PD2 < NQ; q0 += PD2) {
#pragma unroll
                for (int j = 0; j < PD2; ++j) {
                    acc = mfma6(ring[j], load_planes(PA, q0 + j, lane), acc);
                    ring[j] = load_planes(a.w2, cb * NQ + q0 + j + PD2, lane);  // refilled in place, behind its last reader
                }
            }
#pragma unroll
            for (int j = 0; j < PD2; ++j) acc = mfma6(ring[j], load_planes(PA, NQ - PD2 + j, lane), acc);
#pragma unroll
            for (int q = 0; q < PD; ++q) {  // the score product's first chunks, under the split + barrier
                rA[q] = load_planes(a.wb, cbA * NQ + q, lane);
                rB[q] = load_planes(a.wb, cbB * NQ + q, lane);
            }
            emit_block(acc, cb, PB, true);
            __syncthreads();
        }
        gstage(2, t0 + t_step);

        // ---- scores of this wave's two column blocks, softmax over the K rows of a point, weighted sum ----
        f32x16 accA, accB;
#pragma unroll
        for (int r = 0; r < 16; ++r) { accA[r] = 0.f; accB[r] = 0.f; }
        f32x2 gq[2][8], v[2][8];
        {
#pragma unroll 1
            for (int q0 = 0; q0 + PD < NQ; q0 += PD) {
#pragma unroll
                for (int j = 0; j < PD; ++j) {
                    const Planes x = load_planes(PB, q0 + j, lane);
                    accA = mfma6(x, rA[j], accA);
                    accB = mfma6(x, rB[j], accB);
                    rA[j] = load_planes(a.wb, cbA * NQ + q0 + j + PD, lane);  // refilled in place, behind its last reader
                    rB[j] = load_planes(a.wb, cbB * NQ + q0 + j + PD, lane);
                }
            }
            // last pass: nothing is refilled any more; the G rows of both blocks travel under its products
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) gq[i][r >> 1][r & 1] = *reinterpret_cast<const float*>(fgb + off[r] + (i * WAVES * 32 * 4 + H * 4));
#pragma unroll
            for (int j = 0; j < PD; ++j) {
                const Planes x = load_planes(PB, NQ - PD + j, lane);
                accA = mfma6(x, rA[j], accA);
                accB = mfma6(x, rB[j], accB);
            }
        }
        // value rows of the first block (columns < H: gathered neighbour features); the second block (f_xyz columns, LDS) goes first
#pragma unroll
        for (int r = 0; r < 16; ++r) v[0][r >> 1][r & 1] = *reinterpret_cast<const float*>(fgb + off[r]);
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            const int i = 1 - ii;
            const int cb = wave + i * WAVES;
            const f32x16& acc = i == 0 ? accA : accB;
            f32x2 (&vv)[8] = v[i];
            if (cb * 32 >= H) {  // values = f_xyz (LDS tile)
                const float* tv = TV + (cb * 32 - H + c32) + 4 * hl * PITCH;
#pragma unroll
                for (int r = 0; r < 16; ++r) vv[r >> 1][r & 1] = tv[((r & 3) + 8 * (r >> 2)) * PITCH];
            }
            f32x2 sc[8];
            const f32x2 l2e = {1.4426950408889634f, 1.4426950408889634f};
#pragma unroll
            for (int j = 0; j < 8; ++j) sc[j] = __builtin_elementwise_fma(gq[i][j], l2e, f32x2{acc[2 * j], acc[2 * j + 1]});
            constexpr int PP = RP / 2;
#pragma unroll
            for (int pi = 0; pi < PPT; ++pi) {
                float m = fmaxf(sc[pi * PP][0], sc[pi * PP][1]);
#pragma unroll
                for (int j = 1; j < PP; ++j) m = fmaxf(m, fmaxf(sc[pi * PP + j][0], sc[pi * PP + j][1]));
                m = swap32b_max(m);
                const f32x2 mm = {m, m};
                f32x2 ssum2 = {0.f, 0.f}, num2 = {0.f, 0.f};
#pragma unroll
                for (int j = 0; j < PP; ++j) {
                    const f32x2 dd = sc[pi * PP + j] - mm;
                    const f32x2 ex = {__builtin_amdgcn_exp2f(dd[0]), __builtin_amdgcn_exp2f(dd[1])};
                    ssum2 += ex;
                    num2 = __builtin_elementwise_fma(ex, vv[pi * PP + j], num2);
                }
                const float ssum = swap32b_sum(ssum2[0] + ssum2[1]);
                const float num = swap32b_sum(num2[0] + num2[1]);
                if (hl == 0 && t0 + pi < t_end) a.agg[__umul24(pp[pi], D) + (unsigned)(cb * 32 + c32)] = num * __builtin_amdgcn_rcpf(ssum);
            }
        }
        __syncthreads();  // planes, value tile and neighbour rows are overwritten by the next tile
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------
static inline uint16_t piece_of(float w, int plane)
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

// K order of an operand that comes out of a transposed product's accumulators: chunk q, lane half g, element j
static inline int kmap(int q, int g, int j)
{
    const int r = 8 * (q & 1) + j;
    return 32 * (q >> 1) + (r & 3) + 8 * (r >> 2) + 4 * g;
}

void pack_b3(const float* W, int cin, int cout, uint16_t* out)
{
    const int nq = cin / 16, cbs = cout / 32;
    for (int cb = 0; cb < cbs; ++cb)
        for (int q = 0; q < nq; ++q)
            for (int pl = 0; pl < 3; ++pl)
                for (int l = 0; l < 64; ++l)
                    for (int j = 0; j < 8; ++j)
                        out[(((((size_t)cb * nq + q) * 3 + pl) * 64 + l) * 8) + j] = piece_of(W[(size_t)kmap(q, l >> 5, j) * cout + 32 * cb + (l & 31)], pl);
}

void pack_b3_locse(const float* W1, int cout, uint16_t* out)
{
    const int cbs = cout / 32;
    for (int cb = 0; cb < cbs; ++cb)
        for (int pl = 0; pl < 3; ++pl)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    const int k = 8 * (l >> 5) + j;
                    out[((((size_t)cb * 3 + pl) * 64 + l) * 8) + j] = k < 10 ? piece_of(W1[(size_t)k * cout + 32 * cb + (l & 31)], pl) : (uint16_t)0;
                }
}

template <int D, int STAGE, int KN>
static int launch_att32b(ps_context* c, const Att32bArgs& a)
{
    constexpr int H = D / 2, PITCH = H + 4, TILE = 32 * PITCH;
    constexpr int CBH = H / 32, CBD = D / 32, NQ = H / 16;
    // d = 128: 78 KB of weight planes + eight 8.8 KB tiles = 148 KB: one workgroup of eight waves per CU; d = 64: 21 KB + twelve 4.7 KB
    // tiles, one workgroup of twelve waves (134 registers: three waves per SIMD) in stage 2
    // (d = 64, stage 1: sixteen waves -- 126 registers, four waves per SIMD: 43.5 -> 41.2 us; stage 2 needs 134 and spills at sixteen: 50.0 -> 51.0)
    constexpr int WAVES = D == 128 ? 8 : (STAGE == 1 ? 16 : 12);
    constexpr int PER_CU = 1;
    constexpr size_t planes = (size_t)(CBH + (STAGE == 2 ? CBH * NQ : 0) + CBD * NQ) * 3 * 64 * 16;
    constexpr size_t smem = planes + sizeof(float) * (2 * H + (size_t)WAVES * (32 + TILE));
    static_assert(smem <= 160 * 1024, "att32b: weights + tiles exceed the LDS");
    auto kern = att32b_kernel<D, STAGE, KN, WAVES>;
    PS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    const int tiles = ceil_div(a.n_total, 32 / KN);
    const int blocks = (std::min(ceil_div(tiles, WAVES), 256 * PER_CU) + 7) & ~7;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(WAVES * 64), smem, c->stream, a);
    PS_HIP(hipGetLastError());
    return PS_OK;
}

template <int D, int STAGE, int KN>
static int launch_att32s(ps_context* c, const Att32bArgs& a)
{
    constexpr int H = D / 2, PITCH = H + 4, NQ = H / 16;
    constexpr int WAVES = D / 64;  // two score blocks per wave: 4 waves at d = 256, 8 at d = 512
    constexpr size_t smem = sizeof(float) * 32 + (size_t)(STAGE == 2 ? 2 : 1) * NQ * 3 * 64 * 16 + sizeof(float) * 32 * PITCH;
    static_assert(smem <= 160 * 1024, "att32s: planes + value tile exceed the LDS");
    auto kern = att32s_kernel<D, STAGE, KN, WAVES>;
    if (smem > 48 * 1024) PS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    const int tiles = ceil_div(a.n_total, 32 / KN);
    const int blocks = (std::min(tiles, 256 * 16) + 7) & ~7;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(WAVES * 64), smem, c->stream, a);
    PS_HIP(hipGetLastError());
    return PS_OK;
}

bool att_pool32b_fits(const AttStage& s)
{
    return att_pool32_fits(s) && s.p32->w1b && (s.d == 64 || s.d == 128 || s.d == 256 || s.d == 512);
}

int att_pool32b_stage(ps_context* c, const AttStage& s)
{
    Att32bArgs a;
    a.xyz = s.xyz; a.idx = s.idx; a.order = s.order; a.fg = s.fg;
    a.w1 = reinterpret_cast<const uint4*>(s.p32->w1b); a.b1 = s.lfa1->bias;
    a.w2 = s.lfa2 ? reinterpret_cast<const uint4*>(s.p32->w2b) : nullptr; a.b2 = s.lfa2 ? s.lfa2->bias : nullptr;
    a.wb = reinterpret_cast<const uint4*>(s.lfa2 ? s.p32->wb2b : s.p32->wb1b);
    a.agg = s.agg;
    a.n_total = (int)s.n_total; a.n_cloud = (int)s.n_cloud;
    if (s.n_total <= 0) return PS_OK;
    const int stage = s.lfa2 ? 2 : 1;
#define PS_A32B(DD)                                                                                                      \
    if (s.k == 16) return stage == 1 ? launch_att32b<DD, 1, 16>(c, a) : launch_att32b<DD, 2, 16>(c, a);                  \
    return stage == 1 ? launch_att32b<DD, 1, 32>(c, a) : launch_att32b<DD, 2, 32>(c, a)
    if (s.d == 64) { PS_A32B(64); }
    if (s.d == 128) { PS_A32B(128); }
#undef PS_A32B
#define PS_A32S(DD)                                                                                                      \
    if (s.k == 16) return stage == 1 ? launch_att32s<DD, 1, 16>(c, a) : launch_att32s<DD, 2, 16>(c, a);                  \
    return stage == 1 ? launch_att32s<DD, 1, 32>(c, a) : launch_att32s<DD, 2, 32>(c, a)
    if (s.d == 256) { PS_A32S(256); }
    PS_A32S(512);
#undef PS_A32S
}

}  // namespace ps
