"""Round-4 soak (not part of the test-suite): (1) the whole index pyramid against the oracle at point counts around the builder's tier
boundaries (kSmall = 256, kMid = 4 096 and its multiples, the 2 048-record chunks), lattice and uniform clouds, batches; (2) the forward
with EVERY dense layer that fits routed through gemm32b (PS_GEMM32B_MIN_FLOPS=0) at ragged row counts against the float64 oracle.
usage (GPU box): PS_GEMM32B_MIN_FLOPS=0 python profiles/tools/soak_r4.py"""
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import netcase  # noqa: E402
import test_gpu_network as T  # noqa: E402
from conftest import brats_cloud, uniform_cloud  # noqa: E402
from oracle import bindings  # noqa: E402
from oracle import randla_oracle as ro  # noqa: E402
from point_unet_amd.pyramid import build_pyramid  # noqa: E402

bindings.oracle_lib()
th = max(1, min(os.cpu_count() or 1, 32))


class Cfg:
    k_n, num_layers = 16, 5
    sub_sampling_ratio = [4, 4, 4, 4, 2]


sizes = [2049, 4095, 4096, 4097, 8191, 8193, 12289, 16385, 20481, 32769, 40961, 65537, 98305, 131073, 150001]
for n0 in sizes:
    for kind in ("lattice", "uniform"):
        B = 2 if n0 < 20000 else 1
        xyz = np.stack([(brats_cloud(n0, n0 + b) if kind == "lattice" else uniform_cloud(n0, n0 + b)) for b in range(B)])
        pyr = build_pyramid(torch.from_numpy(xyz).cuda(), Cfg)
        pts, nbr, pool, up = ro.build_pyramid(lambda s, q, k: bindings.knn_batch(s, q, k, threads=th, qpar=True), xyz, Cfg.k_n, Cfg.sub_sampling_ratio)
        for i in range(Cfg.num_layers):
            assert np.array_equal(pyr.neigh_idx[i].cpu().numpy(), nbr[i]), (n0, kind, i, "neigh")
            assert np.array_equal(pyr.sub_idx[i].cpu().numpy(), pool[i]), (n0, kind, i, "sub")
            assert np.array_equal(pyr.interp_idx[i].cpu().numpy(), up[i]), (n0, kind, i, "interp")
            assert np.array_equal(pyr.xyz[i].cpu().numpy(), pts[i]), (n0, kind, i, "xyz")
        print("pyramid n0 %6d B %d %-7s exact" % (n0, B, kind), flush=True)

worst = 0.0
for n0, B, k in [(4099, 1, 16), (6007, 1, 16), (10001, 2, 16), (4610, 3, 16), (8193, 1, 32), (5003, 2, 32), (17001, 1, 16)]:  # (netcase.small_deep's 40 x 40 x 30 lattice holds ~18 000 ellipsoid cells: no larger n0)
    cfg, xyz, feats = netcase.small_deep(n0, seed=n0, k_n=k, B=B)
    err, mag, _ = T._run_case(bindings, cfg, xyz, feats, taps=False)
    worst = max(worst, err)
    print("forward n0 %6d B %d K %2d  max|logit| %.3f  err %.3e" % (n0, B, k, mag, err), flush=True)
    assert err <= 1e-4
print("worst", worst, "gemm32b threshold", os.environ.get("PS_GEMM32B_MIN_FLOPS"))
