// attpool.hip -- fused local spatial encoding + neighbour gather + attentive pooling, one wave per point.
//
// Device form of building_block / relative_pos_encoding / gather_neighbour / att_pooling
// (PointSegment/RandLANet.py:323-343, 377-401) up to (not including) att_pooling's trailing conv2d.
// No [N,K,C] tensor ever reaches HBM: per point the wave
//   1. gathers the K neighbour coordinates and forms the 10-vector [dis, rel, centre, nbr]      (:337-343)
//   2. runs LFA mlp1 (10 -> d/2, BN folded, LeakyReLU) on fp32 MFMA straight from registers      (:326)
//      (stage 2 additionally LFA mlp2 d/2 -> d/2 from the LDS tile)                              (:331)
//   3. computes the attention scores  [f_nb | f_xyz] . Wfc  as  G[idx] + f_xyz . Wfc[d/2:, :]    (:395)
//      where G = f . Wfc[:d/2, :] was produced once per point by the preceding dense layer: gathering commutes
//      with a per-row linear map, which halves the score FLOPs of the reference formulation
//   4. softmax over the K rows of each column with wave shuffles, weighted sum of [f_nb | f_xyz] (:396-398)
// A 16x16 MFMA output tile is exactly (16 neighbours) x (16 channels) of one point.
//
// Bound: fp32 MFMA for d >= 64; latency/L2 for d = 16 (level 0).
#include "attpool.h"
#include "mfma_tile.h"

namespace ps {

struct AttArgs {
    const float* xyz;
    const int32_t* idx;
    const int32_t* order;
    const float* fg;
    const float* w1p; const float* b1;
    const float* w2p; const float* b2;
    const float* wbp;
    const float* wfp;  // full Wfc [d,d] packed (direct formulation)
    float* agg;
    int n_total, n_cloud, ldf;
};

// Point iteration shared by the kernels below.  Unit u = workgroup-slot * units-per-workgroup + unit-in-workgroup walks a
// contiguous EIGHTH of the points per XCD (workgroups go to the 8 XCDs round-robin, so workgroup b serves eighth b % 8); with
// AttArgs::order the t-th point is the t-th in kd-tree leaf order, which makes that eighth a compact region of space: the
// neighbour gathers (coordinates, feature rows) of an XCD then mostly hit its own L2 instead of touching the whole cloud.
struct PointWalk {
    int t, end, stride;
    __device__ __forceinline__ PointWalk(int n_total, int units_per_wg, int unit_in_wg)
    {
        const int per_xcd = (n_total + 7) >> 3;
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;  // the host launches a multiple of 8
        t = xcd * per_xcd + slot * units_per_wg + unit_in_wg;
        end = min(n_total, (xcd + 1) * per_xcd);
        stride = slots * units_per_wg;
    }
};
__device__ __forceinline__ int walk_point(const AttArgs& a, int t)
{
    return a.order ? (t / a.n_cloud) * a.n_cloud + a.order[t] : t;
}

// SPLITN = false: every wave owns a point (its own LDS tiles).  SPLITN = true (deep levels: few points, wide d):
// the WAVES waves of a workgroup share ONE point and one pair of LDS tiles and split the output-column blocks of every
// phase between them -- the per-point dependent MFMA chain gets WAVES times shorter and WAVES times more waves are in
// flight (level 4 has 703 points for 1024 SIMDs).
template <int D, int STAGE, int KN, int WAVES, bool SPLITN>
__global__ __launch_bounds__(WAVES * 64) void att_kernel(AttArgs a)
{
    constexpr int H = D / 2, RT = KN / 16, PITCH = H + 2, LDF = H + D;
    constexpr int NTB_H = ntb_for(H), NTB_D = ntb_for(D);
    constexpr int NT_H = (H + 15) / 16, NT_D = D / 16;
    constexpr int CB_H = (NT_H + NTB_H - 1) / NTB_H, CB_D = (NT_D + NTB_D - 1) / NTB_D;
    constexpr int KS_H = (H + 3) / 4;
    using bfH = typename BFrag<NTB_H>::type;
    using bfD = typename BFrag<NTB_D>::type;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int g = lane >> 4, c16 = lane & 15;
    float* T1 = smem + (SPLITN ? 0 : wave) * (KN * PITCH * (STAGE == 2 ? 2 : 1));
    float* T2 = T1 + KN * PITCH;
    constexpr int CB0_STEP = SPLITN ? WAVES : 1;
    const int cb0 = SPLITN ? wave : 0;
    auto phase_sync = [&]() {
        if constexpr (SPLITN) __syncthreads();
        else wave_lds_sync();
    };

    // d = 64 (level 1, 45 000 points): all three weight matrices fit in ~56 registers per lane; loading them once takes an
    // exposed L2 round trip per MFMA group out of the per-point loop.  Wider levels stream them (L2-resident).
    constexpr bool HOIST = !SPLITN && D <= 64;
    bfH w1r[HOIST ? CB_H : 1][3];
    bfH w2r[HOIST && STAGE == 2 ? CB_H : 1][HOIST && STAGE == 2 ? KS_H : 1];
    bfD wbr[HOIST ? CB_D : 1][HOIST ? KS_H : 1];
    if constexpr (HOIST) {
#pragma unroll
        for (int cb = 0; cb < CB_H; ++cb) {
            const bfH* w = reinterpret_cast<const bfH*>(a.w1p) + (size_t)cb * 3 * 64 + lane;
            w1r[cb][0] = w[0]; w1r[cb][1] = w[64]; w1r[cb][2] = w[128];
            if constexpr (STAGE == 2) {
#pragma unroll
                for (int ks = 0; ks < KS_H; ++ks) w2r[cb][ks] = (reinterpret_cast<const bfH*>(a.w2p) + (size_t)cb * KS_H * 64 + lane)[(size_t)ks * 64];
            }
        }
#pragma unroll
        for (int cb = 0; cb < CB_D; ++cb)
#pragma unroll
            for (int ks = 0; ks < KS_H; ++ks) wbr[cb][ks] = (reinterpret_cast<const bfD*>(a.wbp) + (size_t)cb * KS_H * 64 + lane)[(size_t)ks * 64];
    }
    // A-fragment product from an LDS tile with register-resident B fragments
    auto reg_mma_H = [&](const float* tile, const bfH (&w)[HOIST && STAGE == 2 ? KS_H : 1], f32x4 (&acc)[RT][NTB_H]) {
        const float* t0 = tile + (lane & 15) * PITCH + (lane >> 4);
#pragma unroll
        for (int ks = 0; ks < (HOIST && STAGE == 2 ? KS_H : 1); ++ks)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const float av = t0[rt * 16 * PITCH + ks * 4];
#pragma unroll
                for (int j = 0; j < NTB_H; ++j) acc[rt][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bfrag_get<NTB_H>(w[ks], j), acc[rt][j], 0, 0, 0);
            }
    };
    auto reg_mma_D = [&](const float* tile, const bfD (&w)[HOIST ? KS_H : 1], f32x4 (&acc)[RT][NTB_D]) {
        const float* t0 = tile + (lane & 15) * PITCH + (lane >> 4);
#pragma unroll
        for (int ks = 0; ks < (HOIST ? KS_H : 1); ++ks)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const float av = t0[rt * 16 * PITCH + ks * 4];
#pragma unroll
                for (int j = 0; j < NTB_D; ++j) acc[rt][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bfrag_get<NTB_D>(w[ks], j), acc[rt][j], 0, 0, 0);
            }
    };

    PointWalk walk(a.n_total, SPLITN ? 1 : WAVES, SPLITN ? 0 : wave);
    for (int t = walk.t; t < walk.end; t += walk.stride) {
        const int p = walk_point(a, t);
        const int base = (p / a.n_cloud) * a.n_cloud;
        const float cx = a.xyz[3 * (size_t)p], cy = a.xyz[3 * (size_t)p + 1], cz = a.xyz[3 * (size_t)p + 2];
        int nb[RT];
        float a0[RT], a1[RT], a2[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            nb[rt] = base + a.idx[(size_t)p * KN + rt * 16 + c16];
            const float nx = a.xyz[3 * (size_t)nb[rt]], ny = a.xyz[3 * (size_t)nb[rt] + 1], nz = a.xyz[3 * (size_t)nb[rt] + 2];
            const float rx = cx - nx, ry = cy - ny, rz = cz - nz;
            const float dis = __builtin_amdgcn_sqrtf(rx * rx + ry * ry + rz * rz);
            a0[rt] = g == 0 ? dis : (g == 1 ? rx : (g == 2 ? ry : rz));
            a1[rt] = g == 0 ? cx : (g == 1 ? cy : (g == 2 ? cz : nx));
            a2[rt] = g == 0 ? ny : (g == 1 ? nz : 0.f);
        }
        // ---- LFA mlp1: f_xyz1 = lrelu(enc10 . W1 + b1) -> T1 ----
#pragma unroll
        for (int cb = cb0; cb < CB_H; cb += CB0_STEP) {
            f32x4 acc[RT][NTB_H];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int j = 0; j < NTB_H; ++j) acc[rt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            bfH b0, b1v, b2v;
            if constexpr (HOIST) {
                b0 = w1r[cb][0]; b1v = w1r[cb][1]; b2v = w1r[cb][2];
            } else {
                const bfH* w = reinterpret_cast<const bfH*>(a.w1p) + (size_t)cb * 3 * 64 + lane;
                b0 = w[0]; b1v = w[64]; b2v = w[128];
            }
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int j = 0; j < NTB_H; ++j) {
                    acc[rt][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[rt], bfrag_get<NTB_H>(b0, j), acc[rt][j], 0, 0, 0);
                    acc[rt][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[rt], bfrag_get<NTB_H>(b1v, j), acc[rt][j], 0, 0, 0);
                    acc[rt][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[rt], bfrag_get<NTB_H>(b2v, j), acc[rt][j], 0, 0, 0);
                }
#pragma unroll
            for (int j = 0; j < NTB_H; ++j) {
                const int col = (cb * NTB_H + j) * 16 + c16;
                if (col < H) {
                    const float bb = a.b1[col];
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) T1[(rt * 16 + g * 4 + r) * PITCH + col] = leaky02(acc[rt][j][r] + bb);
                }
            }
        }
        phase_sync();
        const float* TX = T1;
        if constexpr (STAGE == 2) {
            // ---- LFA mlp2: f_xyz2 = lrelu(f_xyz1 . W2 + b2) -> T2 ----
            for (int cb = cb0; cb < CB_H; cb += CB0_STEP) {
                f32x4 acc[RT][NTB_H];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int j = 0; j < NTB_H; ++j) acc[rt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                if constexpr (HOIST) reg_mma_H(T1, w2r[cb], acc);
                else tile_mma<NTB_H, RT>(T1, PITCH, KS_H, reinterpret_cast<const bfH*>(a.w2p) + (size_t)cb * KS_H * 64 + lane, acc, lane);
#pragma unroll
                for (int j = 0; j < NTB_H; ++j) {
                    const int col = (cb * NTB_H + j) * 16 + c16;
                    if (col < H) {
                        const float bb = a.b2[col];
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) T2[(rt * 16 + g * 4 + r) * PITCH + col] = leaky02(acc[rt][j][r] + bb);
                    }
                }
            }
            phase_sync();
            TX = T2;
        }
        // neighbour row of C-layout row (g*4 + r): held by lane (g*4 + r) of group 0
        size_t jr[RT][4];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) jr[rt][r] = (size_t)__shfl(nb[rt], g * 4 + r) * LDF;

        // ---- scores, softmax over the K rows, weighted sum ----
        for (int cb = cb0; cb < CB_D; cb += CB0_STEP) {
            f32x4 acc[RT][NTB_D];
#pragma unroll
            for (int j = 0; j < NTB_D; ++j) {
                const int col = (cb * NTB_D + j) * 16 + c16;
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[rt][j][r] = a.fg[jr[rt][r] + H + col];
            }
            if constexpr (HOIST) reg_mma_D(TX, wbr[cb], acc);
            else tile_mma<NTB_D, RT>(TX, PITCH, KS_H, reinterpret_cast<const bfD*>(a.wbp) + (size_t)cb * KS_H * 64 + lane, acc, lane);
#pragma unroll
            for (int j = 0; j < NTB_D; ++j) {
                const int col = (cb * NTB_D + j) * 16 + c16;
                float m = acc[0][j][0];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) m = fmaxf(m, acc[rt][j][r]);
                m = xor_max(m);
                float ssum = 0.f, num = 0.f;
                // values of the weighted sum: columns < H are the gathered neighbour features (global), the rest f_xyz
                // (LDS).  For H % 16 == 0 a 16-column tile lies on one side: a wave-uniform branch, so the compiler emits
                // global_load / ds_read instead of a flat_load on a selected address.
                float v[RT][4];
                if constexpr (H % 16 == 0) {
                    if ((cb * NTB_D + j) * 16 < H) {
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[rt][r] = a.fg[jr[rt][r] + col];
                    } else {
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[rt][r] = TX[(rt * 16 + g * 4 + r) * PITCH + (col - H)];
                    }
                } else {
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float vg = a.fg[jr[rt][r] + (col < H ? col : 0)];
                            const float vl = TX[(rt * 16 + g * 4 + r) * PITCH + (col < H ? 0 : col - H)];
                            v[rt][r] = col < H ? vg : vl;
                        }
                }
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float e = __expf(acc[rt][j][r] - m);
                        ssum += e;
                        num += e * v[rt][r];
                    }
                ssum = xor_sum(ssum);
                num = xor_sum(num);
                if (g == 0) a.agg[(size_t)p * D + col] = num * __builtin_amdgcn_rcpf(ssum);
            }
        }
        phase_sync();  // T1/T2 are overwritten by the next point
    }
}


// ---- direct formulation (wide levels: d <= 32, weights register-resident) -------------------------------------------
// scores = [f_nb | f_xyz] . Wfc on the full d x d weight, with the gathered neighbour features staged ONCE into the
// LDS tile (they serve both as the A operand and as the values of the weighted sum).  The pre-product formulation above
// gathers 3 x as many bytes per neighbour ([f | G] rows); at levels 0-2 that gather traffic, not the MFMA pipe, was
// the limit (rocprofv3 FETCH_SIZE 617 / 242 / 89 MB per launch against 92 / 92 / 46 MB of feature rows).
// PTS points per wave iteration (two at K = 16: the per-iteration scalar / address / wait overhead -- a quarter of the ~200 wave
// instructions per point of this issue-bound kernel -- is paid once per pair); row tile rt belongs to point rt / (KN / 16).
template <int D, int STAGE, int KN, int WAVES, int PTS>
__global__ __launch_bounds__(WAVES * 64) void att_direct_kernel(AttArgs a)
{
    constexpr int H = D / 2, RPP = KN / 16, RT = PTS * RPP, ROWS = PTS * KN, PA = D + 2, PT = H + 2;
    constexpr int NTB_H = ntb_for(H), NTB_D = ntb_for(D);
    constexpr int NT_H = (H + 15) / 16, NT_D = D / 16;
    constexpr int CB_H = (NT_H + NTB_H - 1) / NTB_H, CB_D = (NT_D + NTB_D - 1) / NTB_D;
    constexpr int KS_H = (H + 3) / 4, KS_D = D / 4;
    constexpr int PER_WAVE = ROWS * PA + (STAGE == 2 ? ROWS * PT : 0);
    using bfH = typename BFrag<NTB_H>::type;
    using bfD = typename BFrag<NTB_D>::type;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int g = lane >> 4, c16 = lane & 15;
    float* A = smem + wave * PER_WAVE;
    float* T1 = A + ROWS * PA;

    // Every B fragment and bias this wave will ever need, loaded ONCE: at d <= 32 the three weight matrices are a few dozen
    // registers, and re-reading them per point put an exposed L2 round trip in front of every MFMA group.
    bfH w1r[CB_H][3];
    float b1r[CB_H][NTB_H];
    bfH w2r[STAGE == 2 ? CB_H : 1][STAGE == 2 ? KS_H : 1];
    float b2r[STAGE == 2 ? CB_H : 1][NTB_H];
    bfD wfr[CB_D][KS_D];
#pragma unroll
    for (int cb = 0; cb < CB_H; ++cb) {
        const bfH* w = reinterpret_cast<const bfH*>(a.w1p) + (size_t)cb * 3 * 64 + lane;
        w1r[cb][0] = w[0]; w1r[cb][1] = w[64]; w1r[cb][2] = w[128];
#pragma unroll
        for (int j = 0; j < NTB_H; ++j) {
            const int col = (cb * NTB_H + j) * 16 + c16;
            b1r[cb][j] = col < H ? a.b1[col] : 0.f;
            if constexpr (STAGE == 2) b2r[cb][j] = col < H ? a.b2[col] : 0.f;
        }
        if constexpr (STAGE == 2) {
#pragma unroll
            for (int ks = 0; ks < KS_H; ++ks) w2r[cb][ks] = (reinterpret_cast<const bfH*>(a.w2p) + (size_t)cb * KS_H * 64 + lane)[(size_t)ks * 64];
        }
    }
#pragma unroll
    for (int cb = 0; cb < CB_D; ++cb)
#pragma unroll
        for (int ks = 0; ks < KS_D; ++ks) wfr[cb][ks] = (reinterpret_cast<const bfD*>(a.wfp) + (size_t)cb * KS_D * 64 + lane)[(size_t)ks * 64];
    const float* a_lane = A + (lane & 15) * PA + (lane >> 4);    // A-fragment base of this lane in the [KN x PA] tile
    const float* t_lane = T1 + (lane & 15) * PT + (lane >> 4);

    // (the kernel is VALU-issue bound -- one VALU instruction per SIMD every four cycles, ~180 of them per point, half of them index
    //  arithmetic --: no integer division for a single cloud, 32-bit element offsets from uniform bases)
    const bool one_cloud = a.n_total == a.n_cloud;
    // PTS consecutive points (of a contiguous eighth of the leaf order per XCD, as PointWalk) per wave iteration
    const int per_xcd = ((((a.n_total + 7) >> 3) + PTS - 1) / PTS) * PTS;
    const int w_end = min(a.n_total, (int)((blockIdx.x & 7) + 1) * per_xcd), w_slots = gridDim.x >> 3;
    for (int t = (int)(blockIdx.x & 7) * per_xcd + ((int)(blockIdx.x >> 3) * WAVES + wave) * PTS; t < w_end; t += w_slots * WAVES * PTS) {
        unsigned p[PTS], base[PTS];
        float ctr[PTS][3];
#pragma unroll
        for (int i = 0; i < PTS; ++i) {
            const int ti = min(t + i, w_end - 1);  // (a pair's second point past the end: recomputes the first, stores nothing)
            p[i] = one_cloud ? (a.order ? (unsigned)a.order[ti] : (unsigned)ti) : (unsigned)walk_point(a, ti);
            base[i] = one_cloud ? 0u : (p[i] / (unsigned)a.n_cloud) * (unsigned)a.n_cloud;
            const float* cp = a.xyz + 3u * p[i];
            ctr[i][0] = cp[0]; ctr[i][1] = cp[1]; ctr[i][2] = cp[2];
        }
        int nb[RT];
        float a0[RT], a1[RT], a2[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int pi = rt / RPP;
            const float cx = ctr[pi][0], cy = ctr[pi][1], cz = ctr[pi][2];
            nb[rt] = (int)(base[pi] + (unsigned)a.idx[p[pi] * (unsigned)KN + (unsigned)((rt % RPP) * 16 + c16)]);
            const float* np = a.xyz + 3u * (unsigned)nb[rt];
            const float nx = np[0], ny = np[1], nz = np[2];
            const float rx = cx - nx, ry = cy - ny, rz = cz - nz;
            const float dis = __builtin_amdgcn_sqrtf(rx * rx + ry * ry + rz * rz);
            a0[rt] = g == 0 ? dis : (g == 1 ? rx : (g == 2 ? ry : rz));
            a1[rt] = g == 0 ? cx : (g == 1 ? cy : (g == 2 ? cz : nx));
            a2[rt] = g == 0 ? ny : (g == 1 ? nz : 0.f);
        }
        // ---- neighbour features -> A[:, 0:H)  (16-byte loads, one row = H*4 contiguous bytes) ----
        {
            constexpr int Q = H / 4, TOT = ROWS * Q;
            static_assert(RT <= 2, "att_direct: one or two row tiles per iteration");
#pragma unroll
            for (int e0 = 0; e0 < TOT; e0 += 64) {
                const int e = e0 + lane;
                const int row = (e < TOT ? e : 0) / Q, q = (e < TOT ? e : 0) % Q;
                int src = __shfl(nb[0], row & 15);
                if constexpr (RT == 2) {
                    const int src1 = __shfl(nb[1], row & 15);
                    src = row >= 16 ? src1 : src;
                }
                if (e < TOT) {
                    const float4 v = *reinterpret_cast<const float4*>(a.fg + (unsigned)src * (unsigned)a.ldf + (unsigned)(4 * q));
                    float* dst = A + row * PA + 4 * q;
                    dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
                }
            }
        }
        // ---- LFA mlp1 -> (stage 1) A[:, H:D)  /  (stage 2) T1 ----
        float* X1 = STAGE == 2 ? T1 : A + H;
        constexpr int P1 = STAGE == 2 ? PT : PA;
#pragma unroll
        for (int cb = 0; cb < CB_H; ++cb) {
            f32x4 acc[RT][NTB_H];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int j = 0; j < NTB_H; ++j) acc[rt][j] = f32x4{b1r[cb][j], b1r[cb][j], b1r[cb][j], b1r[cb][j]};  // bias-seeded: no add in the epilogue
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int j = 0; j < NTB_H; ++j) {
                    acc[rt][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[rt], bfrag_get<NTB_H>(w1r[cb][0], j), acc[rt][j], 0, 0, 0);
                    acc[rt][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[rt], bfrag_get<NTB_H>(w1r[cb][1], j), acc[rt][j], 0, 0, 0);
                    acc[rt][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[rt], bfrag_get<NTB_H>(w1r[cb][2], j), acc[rt][j], 0, 0, 0);
                }
#pragma unroll
            for (int j = 0; j < NTB_H; ++j) {
                const int col = (cb * NTB_H + j) * 16 + c16;
                if (col < H) {
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) X1[(rt * 16 + g * 4 + r) * P1 + col] = leaky02(acc[rt][j][r]);
                }
            }
        }
        wave_lds_sync();
        if constexpr (STAGE == 2) {
#pragma unroll
            for (int cb = 0; cb < CB_H; ++cb) {
                f32x4 acc[RT][NTB_H];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int j = 0; j < NTB_H; ++j) acc[rt][j] = f32x4{b2r[cb][j], b2r[cb][j], b2r[cb][j], b2r[cb][j]};
#pragma unroll
                for (int ks = 0; ks < KS_H; ++ks)
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        const float av = t_lane[rt * 16 * PT + ks * 4];
#pragma unroll
                        for (int j = 0; j < NTB_H; ++j)
                            acc[rt][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bfrag_get<NTB_H>(w2r[cb][ks], j), acc[rt][j], 0, 0, 0);
                    }
#pragma unroll
                for (int j = 0; j < NTB_H; ++j) {
                    const int col = (cb * NTB_H + j) * 16 + c16;
                    if (col < H) {
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) A[(rt * 16 + g * 4 + r) * PA + H + col] = leaky02(acc[rt][j][r]);
                    }
                }
            }
            wave_lds_sync();
        }
        // ---- scores on the full Wfc, softmax over the K rows, weighted sum ----
#pragma unroll
        for (int cb = 0; cb < CB_D; ++cb) {
            f32x4 acc[RT][NTB_D];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int j = 0; j < NTB_D; ++j) acc[rt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS_D; ++ks)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const float av = a_lane[rt * 16 * PA + ks * 4];
#pragma unroll
                    for (int j = 0; j < NTB_D; ++j)
                        acc[rt][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bfrag_get<NTB_D>(wfr[cb][ks], j), acc[rt][j], 0, 0, 0);
                }
#pragma unroll
            for (int j = 0; j < NTB_D; ++j) {
                const int col = (cb * NTB_D + j) * 16 + c16;
#pragma unroll
                for (int pi = 0; pi < PTS; ++pi) {
                    float m = acc[pi * RPP][j][0];
#pragma unroll
                    for (int rt = pi * RPP; rt < (pi + 1) * RPP; ++rt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) m = fmaxf(m, acc[rt][j][r]);
                    m = xor_max_lds(m);
                    float ssum = 0.f, num = 0.f;
#pragma unroll
                    for (int rt = pi * RPP; rt < (pi + 1) * RPP; ++rt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float e = __builtin_amdgcn_exp2f(acc[rt][j][r] - m);  // the Wfc image carries log2(e) (randla.hip)
                            ssum += e;
                            num = __builtin_fmaf(e, A[(rt * 16 + g * 4 + r) * PA + col], num);
                        }
                    ssum = xor_sum_lds(ssum);
                    num = xor_sum_lds(num);
                    if (g == 0 && t + pi < w_end) a.agg[p[pi] * (unsigned)D + (unsigned)col] = num * __builtin_amdgcn_rcpf(ssum);
                }
            }
        }
        wave_lds_sync();  // the tiles are overwritten by the next point
    }
}

template <int D, int STAGE, int KN>
static int launch_att_direct(ps_context* c, const AttArgs& a)
{
    constexpr int H = D / 2;
    constexpr int PTS = KN == 16 ? 2 : 1;
    constexpr size_t per_wave = ((size_t)PTS * KN * (D + 2) + (STAGE == 2 ? (size_t)PTS * KN * (H + 2) : 0)) * sizeof(float);
    constexpr int WAVES = per_wave * 4 <= 160 * 1024 ? 4 : (per_wave * 2 <= 160 * 1024 ? 2 : 1);
    static_assert(per_wave * WAVES <= 160 * 1024, "attention tile does not fit the LDS");
    const size_t smem = per_wave * WAVES;
    auto kern = att_direct_kernel<D, STAGE, KN, WAVES, PTS>;
    if (smem > 48 * 1024) PS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    const int blocks = (std::min(ceil_div(a.n_total, WAVES * PTS), 256 * 8) + 7) & ~7;  // a multiple of 8 (one eighth of the points per XCD)
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(WAVES * 64), smem, c->stream, a);
    PS_HIP(hipGetLastError());
    return PS_OK;
}

template <int STAGE, int KN>
static int dispatch_direct(ps_context* c, int d, const AttArgs& a)
{
    switch (d) {
        case 16: return launch_att_direct<16, STAGE, KN>(c, a);
        case 32: return launch_att_direct<32, STAGE, KN>(c, a);
        default: set_error("att_pool(direct): d_out %d is not a compiled size (16, 32: the weights live in registers)", d); return PS_EINVAL;
    }
}

template <int D, int STAGE, int KN>
static int launch_att(ps_context* c, const AttArgs& a)
{
    constexpr int H = D / 2, PITCH = H + 2;
    constexpr size_t per_wave = (size_t)KN * PITCH * (STAGE == 2 ? 2 : 1) * sizeof(float);
    if (D >= 256 && a.n_total < 16384) {
        // one point per workgroup, the four waves split the column blocks
        constexpr int WAVES = 4;
        const size_t smem = per_wave;
        auto kern = att_kernel<D, STAGE, KN, WAVES, true>;
        if (smem > 48 * 1024) PS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        const int blocks = (std::min(a.n_total, 256 * 16) + 7) & ~7;
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(WAVES * 64), smem, c->stream, a);
        PS_HIP(hipGetLastError());
        return PS_OK;
    }
    constexpr int WAVES = per_wave * 4 <= 160 * 1024 ? 4 : (per_wave * 2 <= 160 * 1024 ? 2 : 1);
    static_assert(per_wave * WAVES <= 160 * 1024, "attention tile does not fit the LDS");
    const size_t smem = per_wave * WAVES;
    auto kern = att_kernel<D, STAGE, KN, WAVES, false>;
    if (smem > 48 * 1024) PS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    const int blocks = (std::min(ceil_div(a.n_total, WAVES), 256 * 8) + 7) & ~7;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(WAVES * 64), smem, c->stream, a);
    PS_HIP(hipGetLastError());
    return PS_OK;
}

template <int STAGE, int KN>
static int dispatch_d(ps_context* c, int d, const AttArgs& a)
{
    switch (d) {
        case 16: return launch_att<16, STAGE, KN>(c, a);
        case 32: return launch_att<32, STAGE, KN>(c, a);
        case 64: return launch_att<64, STAGE, KN>(c, a);
        case 128: return launch_att<128, STAGE, KN>(c, a);
        case 256: return launch_att<256, STAGE, KN>(c, a);
        case 512: return launch_att<512, STAGE, KN>(c, a);
        default: set_error("att_pool: d_out %d is not a compiled size (16,32,64,128,256,512)", d); return PS_EINVAL;
    }
}

int att_pool_stage(ps_context* c, const AttStage& s)
{
    if (c->att_bf16x3 && att_pool32b_fits(s)) return att_pool32b_stage(c, s);
    if (att_pool32_fits(s)) return att_pool32_stage(c, s);
    AttArgs a;
    a.xyz = s.xyz; a.idx = s.idx; a.order = s.order; a.fg = s.fg;
    a.w1p = s.lfa1->wp; a.b1 = s.lfa1->bias;
    a.w2p = s.lfa2 ? s.lfa2->wp : nullptr; a.b2 = s.lfa2 ? s.lfa2->bias : nullptr;
    a.wbp = s.wbot ? s.wbot->wp : nullptr;
    a.wfp = s.wfull ? s.wfull->wp : nullptr;
    a.agg = s.agg;
    a.n_total = (int)s.n_total; a.n_cloud = (int)s.n_cloud;
    a.ldf = s.ldf;
    if (s.n_total <= 0) return PS_OK;
    const int stage = s.lfa2 ? 2 : 1;
    if (s.wfull) {  // direct formulation: fg holds only the features (row stride ldf)
        PS_CHECK((uint64_t)s.n_total * (uint64_t)std::max(s.d, std::max(s.ldf, 3 * 1)) < (1ull << 32) && (uint64_t)s.n_total * s.k < (1ull << 32),
                 "att_pool: level too large for 32-bit element offsets");
        if (s.k == 16) return stage == 1 ? dispatch_direct<1, 16>(c, s.d, a) : dispatch_direct<2, 16>(c, s.d, a);
        if (s.k == 32) return stage == 1 ? dispatch_direct<1, 32>(c, s.d, a) : dispatch_direct<2, 32>(c, s.d, a);
        set_error("att_pool: k_n %d is not a compiled size (16, 32)", s.k);
        return PS_EINVAL;
    }
    if (s.k == 16) return stage == 1 ? dispatch_d<1, 16>(c, s.d, a) : dispatch_d<2, 16>(c, s.d, a);
    if (s.k == 32) return stage == 1 ? dispatch_d<1, 32>(c, s.d, a) : dispatch_d<2, 32>(c, s.d, a);
    set_error("att_pool: k_n %d is not a compiled size (16, 32)", s.k);
    return PS_EINVAL;
}

}  // namespace ps
