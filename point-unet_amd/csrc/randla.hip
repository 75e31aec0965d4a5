// randla.hip -- RandLA-Net inference forward (ps_randla_*): the device form of Network.inference
// (PointSegment/RandLANet.py:110-152) and the blocks it calls (:314-401), inference-mode BatchNorm folded.
//
// Launch plan per encoder level (rows R = B*N_i, h = d/2); "chain" = one rowchain launch (rowgemm.h) where every channel
// count is <= 96, otherwise one rowgemm launch per layer:
//   chain    mlp1 [-> Wfc1[:h,:]]   X[R,d_in]        -> FG1[:, 0:h] [, FG1[:, h:h+d]]   (RandLANet.py:315; the second step is the
//                                                       score pre-product G = f . Wfc[:h,:], only for d >= 64, see attpool.hip;
//                                                       at level 0 mlp1 rides on fc0's chain)
//   att<1>   LocSE+gather+att-pool 1                  -> AGG[R,d]             (:325-329, :337-343, :394-398)
//   chain    att_pooling_1 mlp [-> Wfc2[:h,:]]  AGG   -> FG2[:, 0:h] [, FG2[:, h:h+d]]   (:400)
//   att<2>   LocSE+mlp2+gather+att-pool 2             -> AGG[R,d]             (:331-334)
//   chain    att_pooling_2 mlp -> [mlp2 ; shortcut] over [that | X]   -> ENC_i[R,2d]  (+LeakyReLU)   (:400, :317-321)
//   pool_max random_sample                            -> POOL_i[B*N_{i+1},2d] (:345-360)
// decoder: rowgemm decoder_0 (:130-132); per level rowgemm over [skip | up[interp_idx]] (:137-141); the last decoder step and
// the head (fc1, fc2, fc, :146-151; dropout is the identity in inference) are one chain over the level-0 rows.
#include "attpool.h"
#include "common.h"
#include "rowgemm.h"

#include <memory>

namespace ps {

// ---- random_sample: out[b,m,:] = max_k feature[b, pool_idx[b,m,k], :]  (RandLANet.py:345-360) ----------------
template <int KN>
__global__ __launch_bounds__(256) void pool_max_kernel(const float* __restrict__ feat, const int32_t* __restrict__ idx, const int32_t* __restrict__ order,
                                                       float* __restrict__ out, int rows_out, int m_cloud, int n_cloud, int K, int c4 /* channels / 4 */,
                                                       unsigned char* __restrict__ ties /* optional [rows_out, 4 c4]: how many of the K rows attain the maximum */)
{
    // one thread per (output row, float4 of channels).  Output rows are walked XCD by XCD in `order` (the kd-tree leaf order of
    // the output level, when the pyramid carries it): the K gathered rows of neighbouring outputs overlap and stay in that XCD's L2.
    // KN > 0: K is a compile-time constant -- the K index loads and then the K row loads are all in flight together (with a run-time
    // K the loop waited for every row in turn: 10 us for the 351 rows of the deepest level).
    const int per_xcd = (rows_out + 7) >> 3;
    const int xcd = blockIdx.x & 7;
    const size_t lim = (size_t)min(rows_out, (xcd + 1) * per_xcd) * c4;
    for (size_t t = (size_t)xcd * per_xcd * c4 + (size_t)(blockIdx.x >> 3) * 256 + threadIdx.x; t < lim; t += (size_t)(gridDim.x >> 3) * 256) {
        const int r = (int)(t / c4), q = (int)(t - (size_t)r * c4);
        const int row = order ? (r / m_cloud) * m_cloud + order[r] : r;
        const int base = (row / m_cloud) * n_cloud;
        const int32_t* ix = idx + (size_t)row * K;
        const float4* f4 = reinterpret_cast<const float4*>(feat);
        float4 m;
        if constexpr (KN > 0) {
            int nb[KN];
#pragma unroll
            for (int k = 0; k < KN; k += 4) {
                const int4 i4 = *reinterpret_cast<const int4*>(ix + k);
                nb[k] = i4.x; nb[k + 1] = i4.y; nb[k + 2] = i4.z; nb[k + 3] = i4.w;
            }
            float4 v[KN];
#pragma unroll
            for (int k = 0; k < KN; ++k) v[k] = f4[(size_t)(base + nb[k]) * c4 + q];
            m = v[0];
#pragma unroll
            for (int k = 1; k < KN; ++k) {
                m.x = fmaxf(m.x, v[k].x); m.y = fmaxf(m.y, v[k].y); m.z = fmaxf(m.z, v[k].z); m.w = fmaxf(m.w, v[k].w);
            }
            if (ties) {  // (training: tf.reduce_max's gradient is shared evenly by the rows that attain the maximum)
                int t0 = 0, t1 = 0, t2 = 0, t3 = 0;
#pragma unroll
                for (int k = 0; k < KN; ++k) {
                    t0 += v[k].x == m.x; t1 += v[k].y == m.y; t2 += v[k].z == m.z; t3 += v[k].w == m.w;
                }
                reinterpret_cast<uchar4*>(ties)[(size_t)row * c4 + q] = make_uchar4((unsigned char)t0, (unsigned char)t1, (unsigned char)t2, (unsigned char)t3);
            }
        } else {
            m = f4[(size_t)(base + ix[0]) * c4 + q];
            for (int k = 1; k < K; ++k) {
                const float4 v = f4[(size_t)(base + ix[k]) * c4 + q];
                m.x = fmaxf(m.x, v.x);
                m.y = fmaxf(m.y, v.y);
                m.z = fmaxf(m.z, v.z);
                m.w = fmaxf(m.w, v.w);
            }
            if (ties) {
                int t0 = 0, t1 = 0, t2 = 0, t3 = 0;
                for (int k = 0; k < K; ++k) {
                    const float4 v = f4[(size_t)(base + ix[k]) * c4 + q];
                    t0 += v.x == m.x; t1 += v.y == m.y; t2 += v.z == m.z; t3 += v.w == m.w;
                }
                reinterpret_cast<uchar4*>(ties)[(size_t)row * c4 + q] = make_uchar4((unsigned char)t0, (unsigned char)t1, (unsigned char)t2, (unsigned char)t3);
            }
        }
        reinterpret_cast<float4*>(out)[(size_t)row * c4 + q] = m;
    }
}

int pool_max(ps_context* c, const float* feat, const int32_t* idx, const int32_t* order, float* out, int64_t B, int64_t n, int64_t m, int K, int ch,
             unsigned char* ties)
{
    PS_CHECK(ch % 4 == 0, "pool_max: channel count %d is not a multiple of 4", ch);
    const size_t tot = (size_t)B * m * (ch / 4);
    if (!tot) return PS_OK;
    const unsigned blocks = (unsigned)((std::min<size_t>(ceil_div(tot, 256), 256 * 16) + 7) & ~size_t(7));  // a multiple of 8 (XCD walk)
    const bool al = (reinterpret_cast<uintptr_t>(idx) & 15) == 0;
#define PS_POOL(KN) hipLaunchKernelGGL(pool_max_kernel<KN>, dim3(blocks), dim3(256), 0, c->stream, feat, idx, order, out, (int)(B * m), (int)m, (int)n, K, ch / 4, ties)
    if (K == 16 && al) PS_POOL(16);
    else if (K == 32 && al) PS_POOL(32);
    else PS_POOL(0);
#undef PS_POOL
    PS_HIP(hipGetLastError());
    return PS_OK;
}

struct LayerSpec {
    int cin, cout, leaky;
    size_t w_off, b_off;  // offsets into the host blob (floats)
};

struct EncLevel {
    PackedLinear mlp1, top1, lfa1, bot1, full1, att1mlp, lfa2, top2, bot2, full2, att2mlp, mlp2sc;
    Att32Weights p32;  // d >= 64: weight images of the 32x32x2 attentive-pooling kernels
    bool has_p32 = false;
    int d_in, d;
};

}  // namespace ps

using namespace ps;

struct ps_randla {
    ps_context* ctx = nullptr;
    ps_randla_config cfg;
    std::vector<LayerSpec> specs;  // blob order
    int64_t blob_floats = 0;
    bool have_weights = false;
    DevBuf wbuf;
    ChainCache chains;  // re-ordered weight images of the register-resident layer chains (regchain.hip)
    PackedLinear fc0, decoder0, fc1, fc2, fc;
    std::vector<EncLevel> enc;
    std::vector<PackedLinear> dec;
    // taps of the last forward (device pointers into ctx->net_arena) and their sizes
    struct Tap { int which; const float* p; int64_t count; };
    std::vector<Tap> taps;
    bool keep_taps = false;  // ps_randla_keep_taps: also store the rows only ps_randla_tap reads (last decoder step)
};

namespace {

// blob order: fc0; per level {mlp1, lfa1, att1_fc, att1_mlp, lfa2, att2_fc, att2_mlp, mlp2, shortcut};
// decoder_0; Decoder_layer_j (rows = [skip | interp] input channels); fc1; fc2; fc.  Each entry: W[cin,cout] then b[cout].
void plan_specs(ps_randla* net)
{
    const ps_randla_config& c = net->cfg;
    auto add = [&](int cin, int cout, int leaky) {
        LayerSpec s{cin, cout, leaky, (size_t)net->blob_floats, 0};
        net->blob_floats += (int64_t)cin * cout;
        s.b_off = (size_t)net->blob_floats;
        net->blob_floats += cout;
        net->specs.push_back(s);
    };
    net->specs.clear();
    net->blob_floats = 0;
    add(c.in_channels, 8, 1);
    int d_in = 8;
    for (int i = 0; i < c.num_layers; ++i) {
        const int d = c.d_out[i], h = d / 2;
        add(d_in, h, 1);   // mlp1
        add(10, h, 1);     // LFA mlp1
        add(d, d, 0);      // att_pooling_1 fc (bias slot = zeros)
        add(d, h, 1);      // att_pooling_1 mlp
        add(h, h, 1);      // LFA mlp2
        add(d, d, 0);      // att_pooling_2 fc
        add(d, d, 1);      // att_pooling_2 mlp
        add(d, 2 * d, 0);  // mlp2 (no activation)
        add(d_in, 2 * d, 0);  // shortcut (no activation)
        d_in = 2 * d;
    }
    add(d_in, d_in, 1);  // decoder_0
    // skip channels per decoder step j: f_encoder_list[-j-2]
    std::vector<int> chans;  // [enc0, pool0, ..., pool_{L-1}]
    chans.push_back(2 * c.d_out[0]);
    for (int i = 0; i < c.num_layers; ++i) chans.push_back(2 * c.d_out[i]);
    int up = d_in;
    for (int j = 0; j < c.num_layers; ++j) {
        const int skip = chans[chans.size() - 2 - j];
        add(skip + up, skip, 1);
        up = skip;
    }
    add(up, 64, 1);
    add(64, 32, 1);
    add(32, c.num_classes, 0);
}

}  // namespace

extern "C" int ps_randla_create(ps_context* ctx, const ps_randla_config* cfg, ps_randla** out)
{
    PS_CHECK(ctx && cfg && out, "ps_randla_create: NULL argument");
    PS_CHECK(cfg->num_layers >= 1 && cfg->num_layers <= PS_MAX_LAYERS, "ps_randla_create: num_layers %d out of range", cfg->num_layers);
    PS_CHECK(cfg->k_n == 16 || cfg->k_n == 32, "ps_randla_create: k_n must be 16 or 32 (got %d)", cfg->k_n);
    PS_CHECK(cfg->num_classes >= 1 && cfg->in_channels >= 1, "ps_randla_create: bad channel counts");
    for (int i = 0; i < cfg->num_layers; ++i) {
        const int d = cfg->d_out[i];
        PS_CHECK(d == 16 || d == 32 || d == 64 || d == 128 || d == 256 || d == 512, "ps_randla_create: d_out[%d]=%d unsupported", i, d);
    }
    ps_randla* net = new ps_randla();
    net->ctx = ctx;
    net->cfg = *cfg;
    plan_specs(net);
    *out = net;
    return PS_OK;
}

extern "C" int ps_randla_destroy(ps_randla* net)
{
    if (!net) return PS_OK;
    (void)hipSetDevice(net->ctx->device);
    (void)hipStreamSynchronize(net->ctx->stream);
    net->wbuf.release();
    net->chains.clear();
    delete net;
    return PS_OK;
}

extern "C" int64_t ps_randla_weight_count(const ps_randla* net) { return net ? net->blob_floats : -1; }

extern "C" int ps_randla_set_weights(ps_randla* net, const float* blob, int64_t count)
{
    PS_CHECK(net && blob, "ps_randla_set_weights: NULL argument");
    PS_CHECK(count == net->blob_floats, "ps_randla_set_weights: blob has %lld floats, expected %lld", (long long)count, (long long)net->blob_floats);
    ps_context* c = net->ctx;
    PS_HIP(hipSetDevice(c->device));
    if (!net->chains.entries.empty()) {  // images derived from the old weights: drop them once nothing in flight reads them
        PS_HIP(hipStreamSynchronize(c->stream));
        net->chains.clear();
    }
    std::vector<float> host;  // packed image of everything, then one upload
    struct Pending { PackedLinear* L; size_t wp_off, b_off, wq_off, w32_off, w32b_off; };
    std::vector<Pending> pend;
    auto emit = [&](PackedLinear& L, const float* W, const float* b, int cin, int cout, int leaky) {
        L = PackedLinear();
        L.cin = cin; L.cout = cout; L.leaky = leaky;
        L.ks = (cin + 3) / 4;
        L.ntb = choose_ntb(cout);
        L.cblocks = (cout + 16 * L.ntb - 1) / (16 * L.ntb);
        size_t off = (host.size() + 63) & ~size_t(63);
        host.resize(off + L.packed_floats());
        pack_weights(W, cin, cout, L.ntb, host.data() + off);
        size_t boff = (host.size() + 63) & ~size_t(63);
        host.resize(boff + (size_t)L.cout_pad());
        for (int i = 0; i < L.cout_pad(); ++i) host[boff + i] = (b && i < cout) ? b[i] : 0.f;
        size_t qoff = 0;
        if (cin % 16 == 0) {  // k-permuted image for the direct-load kernel
            qoff = (host.size() + 63) & ~size_t(63);
            host.resize(qoff + L.kperm_floats());
            pack_weights_kperm(W, cin, cout, L.ntb, host.data() + qoff);
        }
        size_t roff = 0;
        if (cin % 8 == 0 && cout % 32 == 0 && (size_t)cin * cout >= 4096) {  // 32x32x2 image for the few-rows / wide-channel shapes (gemm32.hip)
            roff = (host.size() + 63) & ~size_t(63);
            host.resize(roff + (size_t)cin * cout);
            pack_p32(W, cin, cout, host.data() + roff);
        }
        size_t boff3 = 0;
        if (cin % 16 == 0 && cout % 32 == 0 && (size_t)cin * cout >= 32768) {  // three-plane bfloat16 image for the same shapes (gemm32b.hip)
            boff3 = (host.size() + 63) & ~size_t(63);
            host.resize(boff3 + (size_t)cin * cout * 3 / 2);
            pack_p32b(W, cin, cout, reinterpret_cast<uint16_t*>(host.data() + boff3));
        }
        pend.push_back({&L, off, boff, qoff, roff, boff3});
    };
    struct PendingRaw { const float** dst; size_t off; };
    std::vector<PendingRaw> pend_raw;
    auto emit_raw = [&](const float** dst, size_t floats) -> float* {
        const size_t off = (host.size() + 63) & ~size_t(63);
        host.resize(off + floats);
        pend_raw.push_back({dst, off});
        return host.data() + off;  // (valid until the next resize: fill it before emitting anything else)
    };
    size_t si = 0;
    auto W = [&](size_t i) { return blob + net->specs[i].w_off; };
    auto Bv = [&](size_t i) { return blob + net->specs[i].b_off; };
    const ps_randla_config& cfg = net->cfg;
    emit(net->fc0, W(si), Bv(si), cfg.in_channels, 8, 1);
    ++si;
    net->enc.assign(cfg.num_layers, EncLevel());
    int d_in = 8;
    std::vector<float> tmp;
    for (int i = 0; i < cfg.num_layers; ++i) {
        EncLevel& e = net->enc[i];
        const int d = cfg.d_out[i], h = d / 2;
        e.d = d; e.d_in = d_in;
        e.has_p32 = d >= 64;
        if (e.has_p32) {  // (spec order: mlp1, lfa1, att1 fc, att1 mlp, lfa2, att2 fc, ...)
            pack_p32_locse(W(si + 1), h, emit_raw(&e.p32.w1, (size_t)(h / 32) * 5 * 64));
            // the score weights carry log2(e): the kernel's softmax is exp2(s' - max s') (attpool32.hip)
            auto scaled = [&](const float* src, size_t count) {
                tmp.resize(count);
                for (size_t t = 0; t < count; ++t) tmp[t] = (float)((double)src[t] * 1.4426950408889634);
                return tmp.data();
            };
            pack_p32(scaled(W(si + 2) + (size_t)h * d, (size_t)h * d), h, d, emit_raw(&e.p32.wb1, (size_t)h * d));
            pack_p32(W(si + 4), h, h, emit_raw(&e.p32.w2, (size_t)h * h));
            pack_p32(scaled(W(si + 5) + (size_t)h * d, (size_t)h * d), h, d, emit_raw(&e.p32.wb2, (size_t)h * d));
            if (d == 64 || d == 128 || d == 256 || d == 512) {  // three-plane bfloat16 images (attpool32b.hip); 6 bytes per weight
                pack_b3_locse(W(si + 1), h, reinterpret_cast<uint16_t*>(emit_raw(&e.p32.w1b, (size_t)(h / 32) * 3 * 64 * 4)));
                pack_b3(scaled(W(si + 2) + (size_t)h * d, (size_t)h * d), h, d, reinterpret_cast<uint16_t*>(emit_raw(&e.p32.wb1b, (size_t)h * d * 3 / 2)));
                pack_b3(W(si + 4), h, h, reinterpret_cast<uint16_t*>(emit_raw(&e.p32.w2b, (size_t)h * h * 3 / 2)));
                pack_b3(scaled(W(si + 5) + (size_t)h * d, (size_t)h * d), h, d, reinterpret_cast<uint16_t*>(emit_raw(&e.p32.wb2b, (size_t)h * d * 3 / 2)));
            }
        }
        emit(e.mlp1, W(si), Bv(si), d_in, h, 1); ++si;
        emit(e.lfa1, W(si), Bv(si), 10, h, 1); ++si;
        emit(e.top1, W(si), nullptr, h, d, 0);                      // Wfc1[:h, :]
        emit(e.bot1, W(si) + (size_t)h * d, nullptr, h, d, 0);      // Wfc1[h:, :]
        {   // Wfc1 (direct formulation, d <= 32): pre-multiplied by log2(e), att_direct_kernel's softmax is exp2(s' - max s')
            std::vector<float> sc((size_t)d * d);
            for (size_t t = 0; t < sc.size(); ++t) sc[t] = (float)((double)W(si)[t] * 1.4426950408889634);
            emit(e.full1, sc.data(), nullptr, d, d, 0);
        }
        ++si;
        emit(e.att1mlp, W(si), Bv(si), d, h, 1); ++si;
        emit(e.lfa2, W(si), Bv(si), h, h, 1); ++si;
        emit(e.top2, W(si), nullptr, h, d, 0);
        emit(e.bot2, W(si) + (size_t)h * d, nullptr, h, d, 0);
        {
            std::vector<float> sc((size_t)d * d);
            for (size_t t = 0; t < sc.size(); ++t) sc[t] = (float)((double)W(si)[t] * 1.4426950408889634);
            emit(e.full2, sc.data(), nullptr, d, d, 0);
        }
        ++si;
        emit(e.att2mlp, W(si), Bv(si), d, d, 1); ++si;
        // [mlp2 ; shortcut] over the concatenated K axis, biases summed, LeakyReLU on the sum (RandLANet.py:317-321)
        tmp.assign((size_t)(d + d_in) * 2 * d + 2 * d, 0.f);
        std::memcpy(tmp.data(), W(si), sizeof(float) * (size_t)d * 2 * d);
        std::memcpy(tmp.data() + (size_t)d * 2 * d, W(si + 1), sizeof(float) * (size_t)d_in * 2 * d);
        float* bsum = tmp.data() + (size_t)(d + d_in) * 2 * d;
        for (int k = 0; k < 2 * d; ++k) bsum[k] = Bv(si)[k] + Bv(si + 1)[k];
        emit(e.mlp2sc, tmp.data(), bsum, d + d_in, 2 * d, 1);
        si += 2;
        d_in = 2 * d;
    }
    emit(net->decoder0, W(si), Bv(si), d_in, d_in, 1); ++si;
    net->dec.assign(cfg.num_layers, PackedLinear());
    for (int j = 0; j < cfg.num_layers; ++j) {
        emit(net->dec[j], W(si), Bv(si), net->specs[si].cin, net->specs[si].cout, 1);
        ++si;
    }
    emit(net->fc1, W(si), Bv(si), net->specs[si].cin, 64, 1); ++si;
    emit(net->fc2, W(si), Bv(si), 64, 32, 1); ++si;
    emit(net->fc, W(si), Bv(si), 32, cfg.num_classes, 0); ++si;

    PS_TRY(net->wbuf.reserve(host.size() * sizeof(float)));
    PS_HIP(hipMemcpyAsync(net->wbuf.p, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
    PS_HIP(hipStreamSynchronize(c->stream));
    for (auto& p : pend) {
        p.L->wp = net->wbuf.as<float>() + p.wp_off;
        p.L->bias = net->wbuf.as<float>() + p.b_off;
        p.L->wq = p.wq_off ? net->wbuf.as<float>() + p.wq_off : nullptr;
        p.L->w32 = p.w32_off ? net->wbuf.as<float>() + p.w32_off : nullptr;
        p.L->w32b = p.w32b_off ? net->wbuf.as<float>() + p.w32b_off : nullptr;
    }
    for (auto& p : pend_raw) *p.dst = net->wbuf.as<float>() + p.off;
    net->have_weights = true;
    return PS_OK;
}

extern "C" int ps_randla_forward(ps_randla* net, const ps_pyramid* pyr, const float* features, float* logits)
{
    PS_CHECK(net && pyr && features && logits, "ps_randla_forward: NULL argument");
    if (!net->have_weights) {
        set_error("ps_randla_forward: weights not set");
        return PS_ESTATE;
    }
    ps_context* c = net->ctx;
    const ps_randla_config& cfg = net->cfg;
    const int L = cfg.num_layers;
    PS_CHECK(pyr->num_layers == L && pyr->K == cfg.k_n, "ps_randla_forward: pyramid (layers %d, K %d) does not match the network (%d, %d)",
             pyr->num_layers, pyr->K, L, cfg.k_n);
    PS_HIP(hipSetDevice(c->device));
    const int64_t B = pyr->B;
    const int64_t* n = pyr->n;

    // ---- carve activations ----
    Arena& A = c->net_arena;
    float *fc0 = nullptr, *fg = nullptr, *agg = nullptr, *tmp = nullptr, *dec0 = nullptr, *h1 = nullptr, *h2 = nullptr;
    std::vector<float*> encb(L), poolb(L), decb(L);
    for (int pass = 0; pass < 2; ++pass) {
        A.begin(pass == 0);
        fc0 = A.take<float>((size_t)B * n[0] * 8);
        size_t fg_max = 0, agg_max = 0;
        for (int i = 0; i < L; ++i) {
            const int d = cfg.d_out[i];
            fg_max = std::max(fg_max, (size_t)B * n[i] * (d / 2 + d));
            agg_max = std::max(agg_max, (size_t)B * n[i] * d);
            encb[i] = A.take<float>((size_t)B * n[i] * 2 * d);
            poolb[i] = A.take<float>((size_t)B * n[i + 1] * 2 * d);
        }
        fg = A.take<float>(fg_max);
        agg = A.take<float>(agg_max);
        tmp = A.take<float>(agg_max);
        dec0 = A.take<float>((size_t)B * n[L] * 2 * cfg.d_out[L - 1]);
        for (int j = 0; j < L; ++j) decb[j] = A.take<float>((size_t)B * n[L - 1 - j] * net->dec[j].cout);
        h1 = A.take<float>((size_t)B * n[0] * 64);
        h2 = A.take<float>((size_t)B * n[0] * 32);
        if (pass == 0) PS_TRY(A.buf.reserve(A.off));
    }
    net->taps.clear();
    auto tap = [&](int which, const float* p, int64_t count) { net->taps.push_back({which, p, count}); };

    const RowSrc none;
    auto src = [](const float* x, int ld, int ch) {
        RowSrc s;
        s.x = x; s.ld = ld; s.c = ch;
        return s;
    };

    // Layer chains (rowgemm.h: rowchain) wherever the channel counts allow it -- the wide-N / narrow-C end of the network;
    // otherwise one GEMM launch per layer.
    auto step = [](const PackedLinear& Lr, float* y, int ldy) {
        ChainStep st;
        st.L = &Lr; st.y = y; st.ldy = ldy;
        return st;
    };
    const int d0 = net->enc[0].d, ldf0 = d0 >= 64 ? d0 / 2 + d0 : d0 / 2;
    bool mlp1_0_done = false;
    {
        // fc0 -> Encoder_layer_0 mlp1: fc0's rows are kept (shortcut input of the level), mlp1's go to the gather buffer
        ChainStep ch[2] = {step(net->fc0, fc0, 8), step(net->enc[0].mlp1, fg, ldf0)};
        const RowSrc in = src(features, cfg.in_channels, cfg.in_channels);
        if (rowchain_fits(ch, 2, in, none)) {
            Stage st(c, "fc0", 1);
            PS_TRY(rowchain(c, ch, 2, in, none, B * n[0], &net->chains));
            mlp1_0_done = true;
        } else {
            Stage st(c, "fc0", 1);
            PS_TRY(rowgemm(c, net->fc0, in, none, B * n[0], fc0, 8));
        }
    }
    tap(0, fc0, B * n[0] * 8);

    const float* X = fc0;
    int d_in = 8;
    for (int i = 0; i < L; ++i) {
        const EncLevel& e = net->enc[i];
        const int d = e.d, h = d / 2;
        const int64_t R = B * n[i];
        // pre-product formulation (fg = [f | G]) only where the MFMA pipe is the limit; the wide levels gather f alone
        const bool use_g = d >= 64;
        const int ldf = use_g ? h + d : h;
        char nm[48];
        AttStage s;
        s.xyz = pyr->xyz[i]; s.idx = pyr->neigh_idx[i]; s.order = pyr->order[i]; s.fg = fg; s.lfa1 = &e.lfa1; s.agg = agg;
        s.n_total = R; s.n_cloud = n[i]; s.d = d; s.k = cfg.k_n; s.ldf = ldf;
        s.p32 = e.has_p32 ? &e.p32 : nullptr;
        // f = act(x . W + b) [-> G = f . Wfc[:h]]: one chained launch when it fits
        auto feature_rows = [&](const PackedLinear& mlp, const PackedLinear& top, const RowSrc& in) -> int {
            ChainStep ch[2] = {step(mlp, fg, ldf), step(top, fg + h, ldf)};
            const int nsteps = use_g ? 2 : 1;
            if (nsteps == 2 && rowchain_fits(ch, 2, in, none)) return rowchain(c, ch, 2, in, none, R, &net->chains);
            if (nsteps == 1 && regchain_fits(ch, 1, in, none)) return rowchain(c, ch, 1, in, none, R, &net->chains);
            PS_TRY(rowgemm(c, mlp, in, none, R, fg, ldf));
            if (use_g) PS_TRY(rowgemm(c, top, src(fg, ldf, h), none, R, fg + h, ldf));
            return PS_OK;
        };
        if (i == 0 && mlp1_0_done) {
            if (use_g) {  // mlp1's rows came out of the fc0 chain; the score pre-product still has to be formed
                std::snprintf(nm, sizeof nm, "enc%d_dense", i);
                Stage st(c, nm, 1);
                PS_TRY(rowgemm(c, e.top1, src(fg, ldf, h), none, R, fg + h, ldf));
            }
        } else {
            std::snprintf(nm, sizeof nm, "enc%d_dense", i);
            Stage st(c, nm, 1);
            PS_TRY(feature_rows(e.mlp1, e.top1, src(X, d_in, d_in)));
        }
        {
            std::snprintf(nm, sizeof nm, "enc%d_att1", i);
            Stage st(c, nm, 1);
            s.lfa2 = nullptr; s.wbot = use_g ? &e.bot1 : nullptr; s.wfull = use_g ? nullptr : &e.full1;
            PS_TRY(att_pool_stage(c, s));
        }
        {
            std::snprintf(nm, sizeof nm, "enc%d_dense", i);
            Stage st(c, nm, 1);
            PS_TRY(feature_rows(e.att1mlp, e.top2, src(agg, d, d)));
        }
        {
            std::snprintf(nm, sizeof nm, "enc%d_att2", i);
            Stage st(c, nm, 1);
            s.lfa2 = &e.lfa2; s.wbot = use_g ? &e.bot2 : nullptr; s.wfull = use_g ? nullptr : &e.full2;
            PS_TRY(att_pool_stage(c, s));
        }
        {
            std::snprintf(nm, sizeof nm, "enc%d_dense", i);
            Stage st(c, nm, 1);
            // att_pooling_2's mlp -> [mlp2 ; shortcut] over [that | X]
            ChainStep ch[2] = {step(e.att2mlp, nullptr, 0), step(e.mlp2sc, encb[i], 2 * d)};
            ch[1].extra = src(X, d_in, d_in);
            const RowSrc in = src(agg, d, d);
            if (rowchain_fits(ch, 2, in, none)) {
                PS_TRY(rowchain(c, ch, 2, in, none, R, &net->chains));
            } else {
                PS_TRY(rowgemm(c, e.att2mlp, in, none, R, tmp, d));
                PS_TRY(rowgemm(c, e.mlp2sc, src(tmp, d, d), src(X, d_in, d_in), R, encb[i], 2 * d));
            }
        }
        {
            std::snprintf(nm, sizeof nm, "enc%d_pool", i);
            Stage st(c, nm, 1);
            PS_TRY(pool_max(c, encb[i], pyr->sub_idx[i], i + 1 < L ? pyr->order[i + 1] : nullptr, poolb[i], B, n[i], n[i + 1], cfg.k_n, 2 * d));
        }
        tap(10 + i, encb[i], R * 2 * d);
        tap(20 + i, poolb[i], B * n[i + 1] * 2 * d);
        X = poolb[i];
        d_in = 2 * d;
    }

    {
        Stage st(c, "decoder_0", 1);
        PS_TRY(rowgemm(c, net->decoder0, src(X, d_in, d_in), none, B * n[L], dec0, d_in));
    }
    tap(30, dec0, B * n[L] * d_in);

    const float* up = dec0;
    int up_c = d_in;
    for (int j = 0; j < L; ++j) {
        const int lvl = L - 1 - j;  // output lives on level `lvl` points
        const float* skip = lvl == 0 ? encb[0] : poolb[lvl - 1];
        const int skip_c = net->dec[j].cout;
        RowSrc s2 = src(up, up_c, up_c);
        s2.gather = pyr->interp_idx[lvl];
        s2.gm = (int)n[lvl];
        s2.gn = (int)n[lvl + 1];
        char nm[48];
        std::snprintf(nm, sizeof nm, "dec%d", j);
        if (j == L - 1) {
            // last decoder step + the whole head as one chain over the level-0 rows: [skip | up] -> dec -> fc1 -> fc2 -> fc
            // (the decoder step's own rows feed nothing but fc1: they stay in the chain's registers unless somebody wants the tap)
            ChainStep ch[4] = {step(net->dec[j], net->keep_taps ? decb[j] : nullptr, skip_c), step(net->fc1, nullptr, 0), step(net->fc2, nullptr, 0),
                               step(net->fc, logits, cfg.num_classes)};
            const RowSrc s1 = src(skip, skip_c, skip_c);
            if (rowchain_fits(ch, 4, s1, s2)) {
                Stage st(c, "head", 1);
                PS_TRY(rowchain(c, ch, 4, s1, s2, B * n[lvl], &net->chains));
                if (net->keep_taps) tap(40 + j, decb[j], B * n[lvl] * skip_c);
                return PS_OK;
            }
        }
        Stage st(c, nm, 1);
        PS_TRY(rowgemm(c, net->dec[j], src(skip, skip_c, skip_c), s2, B * n[lvl], decb[j], skip_c));
        tap(40 + j, decb[j], B * n[lvl] * skip_c);
        up = decb[j];
        up_c = skip_c;
    }
    {
        Stage st(c, "head", 3);
        PS_TRY(rowgemm(c, net->fc1, src(up, up_c, up_c), none, B * n[0], h1, 64));
        PS_TRY(rowgemm(c, net->fc2, src(h1, 64, 64), none, B * n[0], h2, 32));
        PS_TRY(rowgemm(c, net->fc, src(h2, 32, 32), none, B * n[0], logits, cfg.num_classes));
    }
    return PS_OK;
}

extern "C" int ps_randla_keep_taps(ps_randla* net, int on)
{
    PS_CHECK(net, "ps_randla_keep_taps: net is NULL");
    net->keep_taps = on != 0;
    return PS_OK;
}

extern "C" int ps_randla_tap(ps_randla* net, int which, float* host_out, int64_t count)
{
    PS_CHECK(net && host_out, "ps_randla_tap: NULL argument");
    for (auto& t : net->taps)
        if (t.which == which) {
            PS_CHECK(t.count == count, "ps_randla_tap: tensor %d has %lld floats, caller expects %lld", which, (long long)t.count, (long long)count);
            PS_HIP(hipMemcpyAsync(host_out, t.p, sizeof(float) * (size_t)count, hipMemcpyDeviceToHost, net->ctx->stream));
            PS_HIP(hipStreamSynchronize(net->ctx->stream));
            return PS_OK;
        }
    set_error("ps_randla_tap: no tensor %d (run a forward first; the last decoder step's rows are only kept after ps_randla_keep_taps)", which);
    return PS_EINVAL;
}
