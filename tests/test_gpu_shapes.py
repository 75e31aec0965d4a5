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
