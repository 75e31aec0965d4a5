"""Shared construction of the network parity cases (BASELINE configs)."""
import numpy as np

from conftest import brats_cloud, uniform_cloud


def make_cfg(num_layers=2, d_out=(16, 64), ratios=(4, 4), k_n=16, num_classes=4, in_channels=7):
    class Cfg:
        pass

    c = Cfg()
    c.num_layers, c.d_out, c.sub_sampling_ratio = num_layers, list(d_out), list(ratios)
    c.k_n, c.num_classes, c.in_channels = k_n, num_classes, in_channels
    return c


def config1():
    """BASELINE config 1: single 18 000-point synthetic cloud, K=16, 2-layer RandLA-Net."""
    cfg = make_cfg()
    xyz = uniform_cloud(18000, 0)[None]
    feats = np.random.default_rng(1).standard_normal((1, 18000, 4)).astype(np.float32)
    return cfg, xyz, np.concatenate([xyz, feats], -1)


def small_deep(n0=6000, seed=0, k_n=16, B=1, classes=4, mods=4):
    """All five encoder widths (16..512) on a small lattice cloud: exercises every compiled kernel shape."""
    cfg = make_cfg(5, (16, 64, 128, 256, 512), (4, 4, 4, 4, 2), k_n, classes, 3 + mods)
    xyz = np.stack([brats_cloud(n0, seed + b, grid=(40, 40, 30)) for b in range(B)])
    feats = np.random.default_rng(seed + 100).standard_normal((B, n0, mods)).astype(np.float32)
    return cfg, xyz, np.concatenate([xyz, feats], -1)
