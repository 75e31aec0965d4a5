"""GPU parity of the RandLA-Net forward: fused HIP path (through the C ABI) vs the float64 NumPy restatement
of the reference graph (oracle/randla_oracle.py).  Tolerance: |logit diff| <= 1e-4 (BASELINE.json north_star:
"segmentation logits within 1e-4 fp32")."""
import numpy as np
import pytest

import netcase

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _run_case(oracle, cfg, xyz, feats, seed=2, taps=True):
    import torch
    from oracle import randla_oracle as ro
    from point_unet_amd import weights
    from point_unet_amd.RandLANet import Network
    from point_unet_amd.pyramid import build_pyramid

    params = weights.init_params(cfg, seed=seed, randomize_bn=True)
    net = Network(cfg, params=params)
    net.keep_taps(taps)
    pyr = build_pyramid(torch.from_numpy(xyz).cuda(), cfg)
    logits = net.inference({"pyramid": pyr, "features": torch.from_numpy(feats).cuda()}).cpu().numpy()
    # oracle on the oracle's own pyramid (bit-exact equality of the two pyramids is test_gpu_knn's job)
    pts, nbr, pool, up = ro.build_pyramid(lambda s, q, k: oracle.knn_batch(s, q, k), xyz, cfg.k_n, cfg.sub_sampling_ratio)
    for i in range(cfg.num_layers):
        assert np.array_equal(pyr.neigh_idx[i].cpu().numpy(), nbr[i])
        assert np.array_equal(pyr.interp_idx[i].cpu().numpy(), up[i])
    tap = {}
    want = ro.inference(params, cfg.num_layers, pts, nbr, pool, up, feats, np.float64, tap=tap)
    report = []
    if taps:
        B = xyz.shape[0]
        names = [(0, "fc0")] + [(10 + i, "enc%d" % i) for i in range(cfg.num_layers)] + [(20 + i, "pool%d" % i) for i in range(cfg.num_layers)]
        names += [(30, "decoder_0")] + [(40 + j, "dec%d" % j) for j in range(cfg.num_layers)]
        for which, nm in names:
            ref = tap[nm]
            got = net.tap(which, ref.shape)
            report.append((nm, float(np.abs(got - ref).max()), float(np.abs(ref).max())))
    err = float(np.abs(logits - want).max())
    return err, float(np.abs(want).max()), report


def test_config1_18k_two_layers(oracle):
    cfg, xyz, feats = netcase.config1()
    err, mag, report = _run_case(oracle, cfg, xyz, feats)
    print("max|logit| %.3f  max err %.3e" % (mag, err), report)
    assert err <= TOL, (err, report)


def test_all_five_widths_small_cloud(oracle):
    cfg, xyz, feats = netcase.small_deep(6000)
    err, mag, report = _run_case(oracle, cfg, xyz, feats)
    print("max|logit| %.3f  max err %.3e" % (mag, err), report)
    assert err <= TOL, (err, report)


def test_split_bf16_attention_is_as_accurate_as_the_fp32_mfma(oracle):
    """Attentive pooling at d_out >= 64 runs on bf16 MFMA over exact three-way bfloat16 splits of the fp32 operands by default
    (csrc/attpool32b.hip); ps_set_att_bf16x3(ctx, 0) selects the fp32 MFMA.  Both must meet the bar against the float64 oracle, and
    the split form must not be the less accurate one by more than a rounding's worth (measured: 4.1e-6 against 3.7e-6)."""
    from point_unet_amd import runtime
    cfg, xyz, feats = netcase.small_deep(6000)
    ctx = runtime.default_context(0)
    errs = {}
    try:
        for on in (True, False):
            ctx.set_att_bf16x3(on)
            errs[on], mag, _ = _run_case(oracle, cfg, xyz, feats, taps=False)
    finally:
        ctx.set_att_bf16x3(True)
    print("max|logit| %.3f  err split-bf16 %.3e  err fp32 MFMA %.3e" % (mag, errs[True], errs[False]))
    assert errs[True] <= TOL and errs[False] <= TOL
    assert errs[True] <= 2 * errs[False] + 2e-6 * max(1.0, mag), errs


def test_batch_of_two_clouds(oracle):
    cfg, xyz, feats = netcase.small_deep(4000, seed=7, B=2)
    err, mag, report = _run_case(oracle, cfg, xyz, feats, taps=False)
    assert err <= TOL, err


def test_k32_pancreas_shape(oracle):
    """BASELINE config 5 shape family: K=32, 4 input channels, 2 classes (small cloud)."""
    cfg, xyz, feats = netcase.small_deep(4096, seed=3, k_n=32, classes=2, mods=1)
    err, mag, report = _run_case(oracle, cfg, xyz, feats)
    assert err <= TOL, (err, report)


def test_forward_is_deterministic(oracle):
    import torch
    from point_unet_amd import weights
    from point_unet_amd.RandLANet import Network
    from point_unet_amd.pyramid import build_pyramid
    cfg, xyz, feats = netcase.small_deep(5000, seed=11)
    net = Network(cfg, params=weights.init_params(cfg, seed=5, randomize_bn=True))
    x, f = torch.from_numpy(xyz).cuda(), torch.from_numpy(feats).cuda()
    a = net.inference({"pyramid": build_pyramid(x, cfg), "features": f}).cpu().numpy()
    b = net.inference({"pyramid": build_pyramid(x, cfg), "features": f}).cpu().numpy()
    assert np.array_equal(a, b)


def _tap_names(cfg):
    names = [(0, "fc0")] + [(10 + i, "enc%d" % i) for i in range(cfg.num_layers)] + [(20 + i, "pool%d" % i) for i in range(cfg.num_layers)]
    return names + [(30, "decoder_0")] + [(40 + j, "dec%d" % j) for j in range(cfg.num_layers)]


def _whole_cloud_vs_oracle(oracle, cfg, xyz, feats_dev, feats_oracle):
    """The forward on the device against the float64 restatement on the WHOLE cloud: logits |diff| <= 1e-4 (north_star), and every
    tapped activation (fc0, enc0..4, pool0..4, decoder_0, dec0..4 -- RandLANet.py:113-141) within 1e-4 of its own magnitude.
    The oracle runs on its own pyramid, which must equal the device's index for index."""
    import os
    import torch
    from oracle import randla_oracle as ro
    from point_unet_amd import weights
    from point_unet_amd.RandLANet import Network
    from point_unet_amd.pyramid import build_pyramid
    params = weights.init_params(cfg, seed=2, randomize_bn=True)
    net = Network(cfg, params=params)
    net.keep_taps(True)
    pyr = build_pyramid(torch.from_numpy(xyz).cuda(), cfg)
    logits = net.inference({"pyramid": pyr, "features": torch.from_numpy(feats_dev).cuda()}).cpu().numpy()
    assert np.isfinite(logits).all()
    th = max(1, min(os.cpu_count() or 1, 32))
    pts, nbr, pool, up = ro.build_pyramid(lambda s, q, k: oracle.knn_batch(s, q, k, threads=th, qpar=True), xyz, cfg.k_n, cfg.sub_sampling_ratio)
    for i in range(cfg.num_layers):
        assert pyr.neigh_idx[i].dtype == torch.int32
        assert np.array_equal(pyr.neigh_idx[i].cpu().numpy(), nbr[i]), i
        assert np.array_equal(pyr.sub_idx[i].cpu().numpy(), pool[i]), i
        assert np.array_equal(pyr.interp_idx[i].cpu().numpy(), up[i]), i
    tap = {}
    want = ro.inference(params, cfg.num_layers, pts, nbr, pool, up, feats_oracle, np.float64, tap=tap)
    report, bad = [], []
    for which, nm in _tap_names(cfg):
        ref = tap.pop(nm)
        got = net.tap(which, ref.shape)
        err, mag = float(np.abs(got - ref).max()), float(np.abs(ref).max())
        report.append((nm, err, mag))
        if not err <= TOL * max(1.0, mag):
            bad.append((nm, err, mag))
    err = float(np.abs(logits - want).max())
    print("max|logit| %.3f  max err %.3e" % (float(np.abs(want).max()), err), report)
    assert err <= TOL, (err, report)
    assert not bad, bad
    return net, pyr, logits


def test_full_size_config2_whole_cloud_vs_oracle(oracle):
    """BASELINE configs[1] at its full size: the 180 000-point BraTS-shaped cloud, K = 16, five levels, ConfigBraTS -- the workload
    bench.py times.  Whole-cloud logits and all 17 taps against the float64 oracle (about half a minute of NumPy on 8 cores): the
    deep levels and the decoder take the branches of THIS size (gemm32 split-K choice, att32s column split, row-count-dependent grids)."""
    from conftest import brats_cloud
    from point_unet_amd.helper_tool import ConfigBraTS
    xyz = brats_cloud(180000, 0)[None]
    feats = np.concatenate([xyz, np.random.default_rng(1).standard_normal((1, 180000, 4)).astype(np.float32)], -1)
    net, _, _ = _whole_cloud_vs_oracle(oracle, ConfigBraTS, xyz, feats, feats)
    net.close()


def test_full_size_config5_whole_cloud_vs_oracle(oracle):
    """BASELINE configs[4] at its full size: 262 144-point cloud, K = 32, 4 input channels (xyz + one CT value, runPancreas.py:118,125),
    2 classes, features handed over as float16, int32 indices.  (1) the pyramid the forward ran on equals the oracle's, index for
    index; (2) whole-cloud logits and all 17 taps against the float64 oracle fed the same float16-rounded features; (3) the float16
    hand-over equals the fp32 path fed the same rounded values, bit for bit."""
    import torch
    from conftest import brats_cloud
    from point_unet_amd.helper_tool import ConfigBraTS

    class cfg(ConfigBraTS):
        k_n, num_classes, in_channels = 32, 2, 4

    n0 = 262144
    xyz = brats_cloud(n0, 0)[None]
    f16 = np.concatenate([xyz, np.random.default_rng(1).standard_normal((1, n0, 1)).astype(np.float32)], -1).astype(np.float16)
    net, pyr, logits = _whole_cloud_vs_oracle(oracle, cfg, xyz, f16, f16.astype(np.float32))
    assert logits.shape == (1, n0, 2)
    same = net.inference({"pyramid": pyr, "features": torch.from_numpy(f16.astype(np.float32)).cuda()}).cpu().numpy()
    assert np.array_equal(logits, same)
    net.close()


def test_pipeline_matches_serial():
    """ForwardPipeline (consecutive clouds on consecutive lanes, one HIP stream + context each, three in flight) returns,
    for every cloud of a sequence, bit-identical logits to the serial one-stream path."""
    import torch
    from point_unet_amd import weights
    from point_unet_amd.RandLANet import Network
    from point_unet_amd.pipeline import ForwardPipeline
    from point_unet_amd.pyramid import build_pyramid
    cfg, _, _ = netcase.small_deep(6000, seed=0, B=1)
    params = weights.init_params(cfg, seed=4, randomize_bn=True)
    clouds = [netcase.small_deep(6000, seed=10 + i, B=1)[1:] for i in range(7)]
    net = Network(cfg, params=params)
    want = []
    for xyz, feats in clouds:
        pyr = build_pyramid(torch.from_numpy(xyz).cuda(), cfg)
        want.append(net.inference({"pyramid": pyr, "features": torch.from_numpy(feats).cuda()}).cpu().numpy())
    pipe = ForwardPipeline(cfg, params=params, lanes=3, coalesce=1)  # (one cloud per launch: the pairs have their own test below)
    dev = [(torch.from_numpy(x).cuda(), torch.from_numpy(f).cuda()) for x, f in clouds]
    torch.cuda.synchronize()
    got = [pipe.submit(x, f) for x, f in dev]  # all seven enqueued before anything is read back
    pipe.synchronize()
    for i, (g, w) in enumerate(zip(got, want)):
        assert np.array_equal(g.cpu().numpy(), w), i
    # a different cloud size re-allocates the slots
    cfg2, xyz2, feats2 = netcase.small_deep(3000, seed=3, B=2)
    out = pipe.submit(torch.from_numpy(xyz2).cuda(), torch.from_numpy(feats2).cuda())
    pipe.synchronize()
    pyr = build_pyramid(torch.from_numpy(xyz2).cuda(), cfg)
    assert np.array_equal(out.cpu().numpy(), net.inference({"pyramid": pyr, "features": torch.from_numpy(feats2).cuda()}).cpu().numpy())
    pipe.close()


def test_pipeline_runs_density_skewed_clouds_between_good_ones(oracle):
    """ForwardPipeline never synchronises the tree build with the host (deferred checks).  Clouds whose bounding-box-midpoint
    splits are very lopsided -- the geometric-progression line and a tumour-dense BraTS-shaped cloud (runBraTS.py:108-110 keeps
    every tumour voxel plus a sparse background) -- submitted between ordinary clouds must come out exactly like the serial,
    host-checked path, and within the logits bar of the oracle; synchronize() reports nothing."""
    import torch
    from oracle import randla_oracle as ro
    from test_gpu_knn import _tumour_dense_cloud
    from point_unet_amd import weights
    from point_unet_amd.RandLANet import Network
    from point_unet_amd.pipeline import ForwardPipeline
    from point_unet_amd.pyramid import build_pyramid
    from conftest import brats_cloud
    n = 40000
    cfg = netcase.make_cfg(5, (16, 64, 128, 256, 512), (4, 4, 4, 4, 2), 16, 4, 7)
    params = weights.init_params(cfg, seed=4, randomize_bn=True)
    rng = np.random.default_rng(5)
    line = np.stack([0.9997 ** np.arange(n), np.zeros(n), np.zeros(n)], 1).astype(np.float32)[rng.permutation(n)]
    good = brats_cloud(n, 21, grid=(80, 80, 60))
    clouds = [good, line, _tumour_dense_cloud(n, 2), good, _tumour_dense_cloud(n, 3)]
    feats = [np.concatenate([c, rng.standard_normal((n, 4)).astype(np.float32)], -1)[None] for c in clouds]
    net = Network(cfg, params=params)
    want = []
    for c, f in zip(clouds, feats):
        pyr = build_pyramid(torch.from_numpy(c[None]).cuda(), cfg)
        want.append(net.inference({"pyramid": pyr, "features": torch.from_numpy(f).cuda()}).cpu().numpy())
    pipe = ForwardPipeline(cfg, params=params, lanes=3, coalesce=1)
    dev = [(torch.from_numpy(c[None]).cuda(), torch.from_numpy(f).cuda()) for c, f in zip(clouds, feats)]
    torch.cuda.synchronize()
    got = [pipe.submit(x, f) for x, f in dev]
    pipe.synchronize()
    for i, (g, w) in enumerate(zip(got, want)):
        assert np.array_equal(g.cpu().numpy(), w), i
    # the tumour-dense cloud against the oracle
    c, f = clouds[2][None], feats[2]
    pts, nbr, pool, up = ro.build_pyramid(lambda s, q, k: oracle.knn_batch(s, q, k), c, cfg.k_n, cfg.sub_sampling_ratio)
    ref = ro.inference(params, cfg.num_layers, pts, nbr, pool, up, f, np.float64)
    assert np.abs(want[2] - ref).max() <= TOL
    pipe.close()


def test_pipeline_coalesces_single_clouds_into_pairs(oracle):
    """ForwardPipeline's opt-in throughput mode (coalesce = 2): consecutive single clouds run two per launch through the same C-ABI
    calls.  Seven clouds of 40 000 points at the true widths: clouds 2k, 2k+1 come out BIT-IDENTICAL to a direct batch-of-two call on
    [cloud 2k, cloud 2k+1]; the seventh, whose partner never comes, is launched alone by synchronize() and equals the batch-1 call; every
    cloud is within 2e-5 of its own batch-1 logits (the split-K dense layers pick their K split by row count: summation order only) and
    the pair form is within the logits bar of the float64 oracle; `launched` tells a caller whether its cloud is still waiting."""
    import torch
    from oracle import randla_oracle as ro
    from point_unet_amd import weights
    from point_unet_amd.RandLANet import Network
    from point_unet_amd.pipeline import ForwardPipeline
    from point_unet_amd.pyramid import build_pyramid
    from conftest import brats_cloud
    n = 40000
    cfg = netcase.make_cfg(5, (16, 64, 128, 256, 512), (4, 4, 4, 4, 2), 16, 4, 7)
    params = weights.init_params(cfg, seed=4, randomize_bn=True)
    rng = np.random.default_rng(6)
    clouds = [brats_cloud(n, 30 + i, grid=(80, 80, 60))[None] for i in range(7)]
    feats = [np.concatenate([c[0], rng.standard_normal((n, 4)).astype(np.float32)], -1)[None] for c in clouds]
    net = Network(cfg, params=params)

    def direct(x, f):
        pyr = build_pyramid(torch.from_numpy(x).cuda(), cfg)
        return net.inference({"pyramid": pyr, "features": torch.from_numpy(f).cuda()}).cpu().numpy()

    single = [direct(c, f) for c, f in zip(clouds, feats)]
    pairs = [direct(np.concatenate(clouds[i:i + 2]), np.concatenate(feats[i:i + 2])) for i in (0, 2, 4)]
    pipe = ForwardPipeline(cfg, params=params, lanes=3, coalesce=2)
    assert pipe.coalesce == 2
    pipe.prime(torch.from_numpy(clouds[0]).cuda(), torch.from_numpy(feats[0]).cuda())
    dev = [(torch.from_numpy(c).cuda(), torch.from_numpy(f).cuda()) for c, f in zip(clouds, feats)]
    torch.cuda.synchronize()
    got, flags = [], []
    for x, f in dev:
        got.append(pipe.submit(x, f))
        flags.append(pipe.launched)
    assert flags == [False, True, False, True, False, True, False]
    pipe.synchronize()
    for i, g in enumerate(got):
        g = g.cpu().numpy()
        assert g.shape == single[i].shape
        if i < 6:
            assert np.array_equal(g[0], pairs[i // 2][i % 2]), i
        else:
            assert np.array_equal(g, single[i]), i
        assert np.abs(g - single[i]).max() <= 2e-5, (i, np.abs(g - single[i]).max())
    pts, nbr, pool, up = ro.build_pyramid(lambda s, q, k: oracle.knn_batch(s, q, k), clouds[1], cfg.k_n, cfg.sub_sampling_ratio)
    ref = ro.inference(params, cfg.num_layers, pts, nbr, pool, up, feats[1], np.float64)
    assert np.abs(got[1].cpu().numpy() - ref).max() <= TOL
    # a pinned lane / a serialised pass / a batch go out on their own, behind whatever was waiting
    a = pipe.submit(*dev[0])
    b = pipe.submit(*dev[1], overlap=False)
    assert pipe.launched
    pipe.synchronize()
    assert np.array_equal(a.cpu().numpy(), single[0]) and np.array_equal(b.cpu().numpy(), single[1])
    pipe.close()


def test_block_methods_reproduce_the_fused_path(oracle):
    """Network.dilated_res_block / building_block (the reference's call sites, RandLANet.py:314-335, composed from the op-level
    kernels) against the fused forward's own encoder output for the same input."""
    import torch
    from point_unet_amd import weights
    from point_unet_amd.RandLANet import Network
    from point_unet_amd.pyramid import build_pyramid
    cfg, xyz, feats = netcase.small_deep(3000, seed=5, B=2)
    params = weights.init_params(cfg, seed=6, randomize_bn=True)
    net = Network(cfg, params=params)
    pyr = build_pyramid(torch.from_numpy(xyz).cuda(), cfg)
    net.inference({"pyramid": pyr, "features": torch.from_numpy(feats).cuda()})
    B, N = xyz.shape[:2]
    fc0 = torch.from_numpy(net.tap(0, (B, N, 8))).cuda()
    want = net.tap(10, (B, N, 2 * cfg.d_out[0]))
    got = net.dilated_res_block(fc0.unsqueeze(2), pyr.xyz[0], pyr.neigh_idx[0], cfg.d_out[0], "Encoder_layer_0")
    assert tuple(got.shape) == (B, N, 1, 2 * cfg.d_out[0])
    assert np.abs(got.squeeze(2).cpu().numpy() - want).max() <= 2e-5 * max(1.0, np.abs(want).max())


def test_half_precision_feature_input(oracle):
    """BASELINE configs[4]: features handed over as float16 (K = 32, 4 input channels, 2 classes) are widened on the device;
    the result equals the fp32 path fed with the same rounded values bit for bit, and the oracle within the bar."""
    import torch
    from oracle import randla_oracle as ro
    from point_unet_amd import weights
    from point_unet_amd.RandLANet import Network
    from point_unet_amd.pyramid import build_pyramid
    cfg, xyz, feats = netcase.small_deep(4096, seed=9, k_n=32, classes=2, mods=1)
    f16 = feats.astype(np.float16)
    params = weights.init_params(cfg, seed=7, randomize_bn=True)
    net = Network(cfg, params=params)
    pyr = build_pyramid(torch.from_numpy(xyz).cuda(), cfg)
    got = net.inference({"pyramid": pyr, "features": torch.from_numpy(f16).cuda()}).cpu().numpy()
    same = net.inference({"pyramid": pyr, "features": torch.from_numpy(f16.astype(np.float32)).cuda()}).cpu().numpy()
    assert np.array_equal(got, same)
    pts, nbr, pool, up = ro.build_pyramid(lambda s, q, k: oracle.knn_batch(s, q, k), xyz, cfg.k_n, cfg.sub_sampling_ratio)
    want = ro.inference(params, cfg.num_layers, pts, nbr, pool, up, f16.astype(np.float32), np.float64)
    assert np.abs(got - want).max() <= TOL


def test_tf_named_checkpoint_variables_run_on_the_device(oracle):
    """N1: a variable dict as `tf.train.load_checkpoint(...)` of the reference model would give it -- names under the
    'layers/' scope with the scope strings concatenated without separators (RandLANet.py:56, 121, 315-334, 395, 400;
    helper_tf_util.py:51,162), conv kernels [1,1,in,out], transposed-conv kernels [1,1,out,in] (helper_tf_util.py:208-212), fc0's
    un-scoped BatchNorm, plus everything else the Saver stores (Adam slots, beta powers, the learning rate: RandLANet.py:87-89,
    101-102) -- goes through weights.from_tf_variables into ps_randla_set_weights / ps_randla_forward.  The oracle is fed the SAME
    dict through a mapping written here (strip the scope, squeeze the 1x1 axes), independent of from_tf_variables."""
    import torch
    from oracle import randla_oracle as ro
    from point_unet_amd import weights
    from point_unet_amd.RandLANet import Network
    from point_unet_amd.pyramid import build_pyramid
    cfg, xyz, feats = netcase.small_deep(6000, seed=31)
    base = weights.init_params(cfg, seed=9, randomize_bn=True)
    rng = np.random.default_rng(0)
    ckpt = {}
    for scope, kind, cin, cout in weights.layer_dims(cfg):
        if kind in ("dense", "dense_nobias"):
            ckpt["layers/%s/kernel" % scope] = base[scope + "/kernel"]
            if kind == "dense":
                ckpt["layers/%s/bias" % scope] = base[scope + "/bias"] + 0.01 * rng.standard_normal(cout).astype(np.float32)
        else:
            w = base[scope + "/weights"]  # [in,out], or [out,in] for the transposed convs
            ckpt["layers/%s/weights" % scope] = w.reshape((1, 1) + w.shape)
            ckpt["layers/%s/biases" % scope] = base[scope + "/biases"] + 0.01 * rng.standard_normal(cout).astype(np.float32)
    for k, v in base.items():
        if "batch_normalization" in k:
            ckpt["layers/" + k] = v
    assert ckpt["layers/Decoder_layer_0/weights"].shape == (1, 1, 512, 1536)      # [1,1,out,in]
    assert ckpt["layers/Encoder_layer_4LFAatt_pooling_2fc/kernel"].shape == (512, 512)
    assert "layers/batch_normalization/gamma" in ckpt                              # fc0's BN, un-scoped (RandLANet.py:115)
    model_vars = dict(ckpt)
    for k, v in model_vars.items():  # what else the Saver holds: optimizer state
        if not k.endswith(("moving_mean", "moving_variance")):
            ckpt["optimizer/" + k + "/Adam"] = np.zeros_like(v)      # slots created inside variable_scope('optimizer') (RandLANet.py:86-89)
            ckpt["optimizer/" + k + "/Adam_1"] = np.zeros_like(v)
            ckpt[k + "/Adam"] = np.ones_like(v)                       # ... and the un-nested spelling some TF versions produce
    ckpt["optimizer/beta1_power"] = np.float32(0.9)
    ckpt["optimizer/beta2_power"] = np.float32(0.999)
    ckpt["optimizer/learning_rate"] = np.float32(1e-4)
    params = weights.from_tf_variables(cfg, ckpt)
    net = Network(cfg, params=params)
    pyr = build_pyramid(torch.from_numpy(xyz).cuda(), cfg)
    logits = net.inference({"pyramid": pyr, "features": torch.from_numpy(feats).cuda()}).cpu().numpy()
    oracle_params = {k[len("layers/"):]: np.squeeze(v, (0, 1)) if np.ndim(v) == 4 else np.asarray(v) for k, v in model_vars.items()}
    pts, nbr, pool, up = ro.build_pyramid(lambda s, q, k: oracle.knn_batch(s, q, k), xyz, cfg.k_n, cfg.sub_sampling_ratio)
    want = ro.inference(oracle_params, cfg.num_layers, pts, nbr, pool, up, feats, np.float64)
    assert np.abs(logits - want).max() <= TOL
    with pytest.raises(KeyError):  # a checkpoint of another architecture: missing variables are an error, not zeros
        weights.from_tf_variables(cfg, {k: v for k, v in ckpt.items() if "Encoder_layer_3mlp2" not in k})
    net.close()
