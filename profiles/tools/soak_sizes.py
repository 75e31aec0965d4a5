"""Ragged sizes through the whole forward against the float64 oracle (not part of the test-suite: a wider sweep of point counts, batch
sizes and K than tests/test_gpu_network.py keeps; netcase.small_deep draws from a 40 x 40 x 30 lattice: keep n0 below ~20 000).  usage (GPU box): python profiles/tools/soak_sizes.py"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import netcase
import test_gpu_network as T
from oracle import bindings
bindings.oracle_lib()
worst = 0.0
for n0, B, k in [(4099, 1, 16), (6007, 1, 16), (10001, 2, 16), (4610, 3, 16), (8193, 1, 32), (5003, 2, 32)]:
    cfg, xyz, feats = netcase.small_deep(n0, seed=n0, k_n=k, B=B)
    err, mag, _ = T._run_case(bindings, cfg, xyz, feats, taps=False)
    worst = max(worst, err)
    print("n0 %6d B %d K %2d  max|logit| %.3f  err %.3e" % (n0, B, k, mag, err), flush=True)
    assert err <= 1e-4
print("worst", worst)
