// tests/cxx/train_host.cpp -- a training loop written against include/pointseg.h ONLY (no Python, no torch): the non-Python maintainer's view of
// the boundary.  Builds a pyramid and runs `steps` optimisation steps of the PointSegment RandLA-Net on a synthetic
// cloud read from a raw file, printing one line per step; tests/test_gpu_cxx_host.py compiles it with hipcc, feeds it the same cloud,
// labels and parameters as the Python Trainer and compares the losses.
//
//   train_host <in.bin> <steps>      in.bin: int64 header {B, n0, in_channels, num_classes, n_params, n_buffers, 0, 0}, then
//                                    f32 xyz[B*n0*3], f32 features[B*n0*in_channels], i32 labels[B*n0], f32 class_weights[num_classes],
//                                    f32 params[n_params], f32 buffers[n_buffers]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "pointseg.h"

#define CK(expr)                                                                   \
    do {                                                                           \
        int rc_ = (expr);                                                          \
        if (rc_ != 0) {                                                            \
            std::fprintf(stderr, "%s failed: %s\n", #expr, ps_last_error());       \
            return 10;                                                             \
        }                                                                          \
    } while (0)
#define HK(expr)                                                                   \
    do {                                                                           \
        hipError_t e_ = (expr);                                                    \
        if (e_ != hipSuccess) {                                                    \
            std::fprintf(stderr, "%s failed: %s\n", #expr, hipGetErrorString(e_)); \
            return 11;                                                             \
        }                                                                          \
    } while (0)

template <class T>
static T* to_device(const std::vector<T>& h)
{
    T* d = nullptr;
    if (hipMalloc(&d, sizeof(T) * (h.empty() ? 1 : h.size())) != hipSuccess) return nullptr;
    if (!h.empty() && hipMemcpy(d, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return d;
}

int main(int argc, char** argv)
{
    if (argc != 3) return 2;
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    const int steps = std::atoi(argv[2]);
    int64_t h[8];
    if (std::fread(h, sizeof h, 1, f) != 1) return 3;
    const int64_t B = h[0], n0 = h[1], cin = h[2], classes = h[3];
    std::vector<float> xyz((size_t)(B * n0 * 3)), feats((size_t)(B * n0 * cin)), cw((size_t)classes), params((size_t)h[4]), buffers((size_t)h[5]);
    std::vector<int32_t> labels((size_t)(B * n0));
    if (std::fread(xyz.data(), 4, xyz.size(), f) != xyz.size() || std::fread(feats.data(), 4, feats.size(), f) != feats.size() ||
        std::fread(labels.data(), 4, labels.size(), f) != labels.size() || std::fread(cw.data(), 4, cw.size(), f) != cw.size() ||
        std::fread(params.data(), 4, params.size(), f) != params.size() || std::fread(buffers.data(), 4, buffers.size(), f) != buffers.size())
        return 3;
    std::fclose(f);

    ps_context* ctx = nullptr;
    CK(ps_create(0, &ctx));
    ps_randla_config cfg = {5, 16, (int32_t)classes, (int32_t)cin, {16, 64, 128, 256, 512}};   // helper_tool.py:22-36
    ps_train_options opt = {};
    opt.learning_rate = 1e-3f;
    opt.keep_prob = 1.0f;
    opt.fused_att = 1;
    opt.fused_locse = 1;
    opt.deterministic = 1;
    opt.fused_convbn = 1;
    ps_trainer* tr = nullptr;
    CK(ps_trainer_create(ctx, &cfg, &opt, &tr));
    if (ps_trainer_param_count(tr) != (int64_t)params.size() || ps_trainer_buffer_count(tr) != (int64_t)buffers.size()) {
        std::fprintf(stderr, "layout mismatch: %lld / %lld floats expected\n", (long long)ps_trainer_param_count(tr), (long long)ps_trainer_buffer_count(tr));
        return 4;
    }
    // the named tensors inside the flat buffers (what a checkpoint loader walks)
    char name[128];
    int64_t off, rows, cols;
    int is_buf, n_named = ps_trainer_layout_rows(tr);
    CK(ps_trainer_layout(tr, 0, name, sizeof name, &off, &rows, &cols, &is_buf));
    std::printf("layout %d tensors, first %s [%lld x %lld] at %lld\n", n_named, name, (long long)rows, (long long)cols, (long long)off);

    float* d_params = to_device(params);
    float* d_buffers = to_device(buffers);
    std::vector<float> zeros(params.size(), 0.f);
    float *d_grads = to_device(zeros), *d_m = to_device(zeros), *d_v = to_device(zeros);
    float *d_xyz = to_device(xyz), *d_feats = to_device(feats), *d_cw = to_device(cw);
    int32_t* d_labels = to_device(labels);
    float* d_loss = nullptr;
    HK(hipMalloc(&d_loss, sizeof(float)));
    if (!d_params || !d_buffers || !d_grads || !d_m || !d_v || !d_xyz || !d_feats || !d_cw || !d_labels) return 5;
    CK(ps_trainer_bind(tr, d_params, d_grads, d_m, d_v, d_buffers));

    const int32_t ratios[5] = {4, 4, 4, 4, 2};
    ps_pyramid pyr = {};
    pyr.num_layers = 5;
    pyr.K = 16;
    pyr.B = B;
    int64_t n = n0;
    for (int i = 0; i < 5; ++i) {
        pyr.n[i] = n;
        const int64_t nn = n / ratios[i];
        HK(hipMalloc(&pyr.xyz[i], sizeof(float) * (size_t)(B * n * 3)));
        HK(hipMalloc(&pyr.neigh_idx[i], sizeof(int32_t) * (size_t)(B * n * 16)));
        HK(hipMalloc(&pyr.sub_idx[i], sizeof(int32_t) * (size_t)(B * nn * 16)));
        HK(hipMalloc(&pyr.interp_idx[i], sizeof(int32_t) * (size_t)(B * n)));
        pyr.order[i] = nullptr;
        n = nn;
    }
    pyr.n[5] = n;
    for (int s = 0; s < steps; ++s) {
        CK(ps_pyramid_build(ctx, d_xyz, B, n0, 5, ratios, 16, &pyr));                        // tf_map, runBraTS.py:140-161
        CK(ps_randla_train_step(tr, &pyr, d_feats, d_labels, d_cw, d_loss, nullptr));        // sess.run([train_op, extra_update_ops, loss])
        CK(ps_synchronize(ctx));
        float loss = 0.f;
        HK(hipMemcpy(&loss, d_loss, sizeof loss, hipMemcpyDeviceToHost));
        std::printf("step %lld loss %.9g\n", (long long)ps_trainer_get_step(tr), loss);
    }
    std::vector<float> out(params.size());
    HK(hipMemcpy(out.data(), d_params, sizeof(float) * out.size(), hipMemcpyDeviceToHost));
    double sum = 0.0;
    for (float v : out) sum += (double)v;
    std::printf("param_sum %.9g pool_peak_bytes %lld\n", sum, (long long)ps_trainer_pool_peak_bytes(tr));
    CK(ps_trainer_destroy(tr));
    CK(ps_destroy(ctx));
    return 0;
}
