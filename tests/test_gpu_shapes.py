"""Pyramid + forward parity on awkward configurations: deep levels with fewer points than K, ratios that do not divide the
cloud, unusual width ladders (d_out[0] = 32 / 64, repeated widths), K = 32, batches mixing lattice and uniform clouds,
a 40-point cloud.  Bars as everywhere: indices bit-exact, logits within 1e-4 of the float64 restatement."""
import numpy as np
import pytest

import netcase
from conftest import brats_cloud, uniform_cloud

pytestmark = pytest.mark.gpu

CASES = [
    dict(n0=1000, B=1, L=5, d=(16, 64, 128, 256, 512), r=(4, 4, 4, 4, 2), K=16),   # levels 3-4 have 15 and 3 points (< K)
    dict(n0=1501, B=3, L=4, d=(16, 32, 64, 128), r=(3, 5, 2, 4), K=16),            # odd sizes and ratios
    dict(n0=4099, B=2, L=3, d=(32, 64, 256), r=(4, 4, 4), K=32),                   # d_out[0] = 32, K = 32
    dict(n0=700, B=1, L=2, d=(64, 128), r=(2, 2), K=16),                           # d_out[0] = 64: pre-product formulation at level 0
    dict(n0=2048, B=2, L=5, d=(16, 16, 32, 64, 512), r=(2, 2, 2, 2, 2), K=16, classes=13, mods=1),
    dict(n0=40, B=2, L=2, d=(16, 64), r=(4, 4), K=16),                             # n = 40, 10, 2
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "n%d_B%d_d%s_K%d" % (c["n0"], c["B"], "-".join(map(str, c["d"])), c["K"]))
def test_awkward_configuration(oracle, case):
    import torch
    from oracle import randla_oracle as ro
    from point_unet_amd import weights
    from point_unet_amd.RandLANet import Network
    from point_unet_amd.pyramid import build_pyramid
    mods = case.get("mods", 4)
    cfg = netcase.make_cfg(case["L"], case["d"], case["r"], case["K"], case.get("classes", 4), 3 + mods)
    n0, B = case["n0"], case["B"]
    clouds = []
    for b in range(B):
        lattice = b % 2 == 1 and n0 <= 2500
        clouds.append(brats_cloud(n0, 60 + b, grid=(24, 20, 16)) if lattice else uniform_cloud(n0, 50 + b))
    xyz = np.stack(clouds)
    feats = np.concatenate([xyz, np.random.default_rng(3).standard_normal((B, n0, mods)).astype(np.float32)], -1)
    params = weights.init_params(cfg, seed=5, randomize_bn=True)
    net = Network(cfg, params=params)
    pyr = build_pyramid(torch.from_numpy(xyz).cuda(), cfg)
    got = net.inference({"pyramid": pyr, "features": torch.from_numpy(feats).cuda()}).cpu().numpy()
    pts, nbr, pool, up = ro.build_pyramid(lambda s, q, k: oracle.knn_batch(s, q, k), xyz, cfg.k_n, cfg.sub_sampling_ratio)
    for i in range(cfg.num_layers):
        assert np.array_equal(pyr.neigh_idx[i].cpu().numpy(), nbr[i]), i
        assert np.array_equal(pyr.sub_idx[i].cpu().numpy(), pool[i]), i
        assert np.array_equal(pyr.interp_idx[i].cpu().numpy(), up[i]), i
    want = ro.inference(params, cfg.num_layers, pts, nbr, pool, up, feats, np.float64)
    assert np.abs(got - want).max() <= 1e-4


# (R, c1, c2, cout, gather on the second source, leaky): the deep levels' layers and the decoder's concat steps, ragged row counts,
# one / two / four column blocks, K split over one, two, four and (fp32 form) eight waves
_GEMM32_CASES = [
    (703, 1024, 0, 1024, False, 1),     # decoder_0-like: few rows, long K
    (351, 512, 1024, 512, True, 1),     # first decoder step: [skip | nearest-interpolated], K = 1 536
    (2812, 256, 512, 256, True, 1),
    (11250, 128, 128, 256, False, 1),   # [mlp2 ; shortcut] of level 2 (two plain sources)
    (4097, 64, 0, 96, False, 0),        # three column blocks, no activation
    (33, 16, 0, 32, False, 1),          # a single ragged row block, one K chunk
    (32768, 32, 16, 64, True, 1),       # the row limit of the kernels, K = 48
]


@pytest.mark.parametrize("split_bf16", [1, 0])
@pytest.mark.parametrize("case", _GEMM32_CASES)
def test_deep_level_dense_layers_against_float64(lib, dbg, case, split_bf16):
    """csrc/gemm32b.hip (bf16 MFMA over exact three-way splits of both operands) and csrc/gemm32.hip (fp32 MFMA) on their own, through the
    test-only door ps_debug_gemm32: Y = act([X1 | X2[g]] . W + b) against a float64 product.  Both carry fp32-level error: the bar is
    2e-6 of the row's |x| . |w| sum (an fp32 dot product of K terms in MFMA order), the same for the split form."""
    import ctypes
    import torch
    from point_unet_amd import runtime
    R, c1, c2, cout, gather, leaky = case
    rng = np.random.default_rng(R + cout)
    n2 = max(R // 4, 1)
    x1 = rng.standard_normal((R, c1)).astype(np.float32) * rng.uniform(0.01, 4.0, (1, c1)).astype(np.float32)
    x2 = rng.standard_normal((n2 if gather else R, max(c2, 1))).astype(np.float32)
    g2 = rng.integers(0, n2, R).astype(np.int32) if gather else None
    W = (rng.standard_normal((c1 + c2, cout)) / np.sqrt(c1 + c2)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    d_x1, d_x2 = torch.from_numpy(x1).cuda(), torch.from_numpy(x2).cuda()
    d_g2 = torch.from_numpy(g2).cuda() if gather else None
    y = torch.full((R, cout), float("nan"), dtype=torch.float32, device="cuda")
    ctx = runtime.default_context(0)
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)  # noqa: E731
    torch.cuda.synchronize()
    rc = dbg.ps_debug_gemm32(ctx.handle, split_bf16, p(d_x1), c1, c1, None, p(d_x2) if c2 else None, max(c2, 1), c2, p(d_g2), 0, 0,
                             W.ctypes.data_as(ctypes.c_void_p), b.ctypes.data_as(ctypes.c_void_p), R, cout, leaky, p(y), cout)
    assert rc == 0, lib.ps_last_error()
    torch.cuda.synchronize()
    X = x1.astype(np.float64) if not c2 else np.concatenate([x1.astype(np.float64), (x2[g2] if gather else x2).astype(np.float64)], 1)
    want = X @ W.astype(np.float64) + b.astype(np.float64)
    scale = np.abs(X) @ np.abs(W.astype(np.float64)) + np.abs(b)
    if leaky:
        want = np.where(want >= 0, want, 0.2 * want)
    got = y.cpu().numpy().astype(np.float64)
    assert np.isfinite(got).all()
    rel = float((np.abs(got - want) / scale).max())
    print("R %d K %d N %d split %d: max err / (|x|.|w|) = %.2e" % (R, c1 + c2, cout, split_bf16, rel))
    assert rel <= 2e-6, rel
