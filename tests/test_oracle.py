"""CPU tests that PIN the oracle: against every committed golden vector (generated from the real reference C++,
tests/golden/make_golden.py) and, where oracle/_ref is present, against the real reference itself on fresh inputs."""
import glob
import os

import numpy as np
import pytest

from conftest import brats_cloud, uniform_cloud

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_knn_oracle_matches_every_golden_vector(oracle):
    files = sorted(glob.glob(os.path.join(GOLD, "knn_*.npz")))
    assert len(files) >= 8
    for f in files:
        g = np.load(f)
        got = oracle.knn_batch(g["support"], g["queries"], int(g["K"]))
        assert np.array_equal(got, g["idx"]), f


def test_fewer_points_than_k_keeps_zeros(oracle):
    g = np.load(os.path.join(GOLD, "knn_fewer_than_k.npz"))
    assert np.all(g["idx"][:, :, 7:] == 0)  # knn.pyx:93: np.zeros, partially filled


def test_pyramid_golden(oracle):
    from oracle import randla_oracle as ro
    g = np.load(os.path.join(GOLD, "pyramid_brats6000.npz"))
    pts, nbr, pool, up = ro.build_pyramid(lambda s, q, k: oracle.knn_batch(s, q, k), g["xyz"], int(g["K"]), list(g["ratios"]))
    for i in range(5):
        assert np.array_equal(nbr[i], g["neigh_%d" % i])
        assert np.array_equal(pool[i], g["sub_%d" % i])
        assert np.array_equal(up[i], g["interp_%d" % i])


def test_grid_oracle_matches_golden_vectors(oracle):
    g = np.load(os.path.join(GOLD, "grid_all.npz"))
    p, f, l = oracle.canonical_rows(*oracle.grid_subsample(g["points"], g["features"], g["classes"], float(g["sampleDl"])))
    assert np.array_equal(p, g["out_points"]) and np.array_equal(f, g["out_features"]) and np.array_equal(l, g["out_classes"])
    g = np.load(os.path.join(GOLD, "grid_points_only.npz"))
    p, _, _ = oracle.canonical_rows(*oracle.grid_subsample(g["points"], None, None, float(g["sampleDl"])))
    assert np.array_equal(p, g["out_points"])
    g = np.load(os.path.join(GOLD, "grid_negative_coords.npz"))
    p, f, _ = oracle.canonical_rows(*oracle.grid_subsample(g["points"], g["features"], None, float(g["sampleDl"])))
    assert np.array_equal(p, g["out_points"]) and np.array_equal(f, g["out_features"])


def test_network_oracle_regression_pin(oracle):
    import netcase
    from oracle import randla_oracle as ro
    from point_unet_amd import weights
    g = np.load(os.path.join(GOLD, "net_config1.npz"))
    cfg, xyz, feats = netcase.config1()
    params = weights.init_params(cfg, seed=2, randomize_bn=True)
    pts, nbr, pool, up = ro.build_pyramid(lambda s, q, k: oracle.knn_batch(s, q, k), xyz, cfg.k_n, cfg.sub_sampling_ratio)
    assert np.array_equal(nbr[0][0, :256], g["neigh_0_rows"]) and np.array_equal(up[0][0, :256], g["interp_0_rows"])
    tap = {}
    logits = ro.inference(params, cfg.num_layers, pts, nbr, pool, up, feats, np.float64, tap=tap)
    assert np.abs(logits - g["logits"]).max() < 1e-5
    assert np.abs(tap["enc0"][0, :64] - g["enc0_rows"]).max() < 1e-5
    # the float32 evaluation of the same graph stays well inside the 1e-4 parity budget
    l32 = ro.inference(params, cfg.num_layers, pts, nbr, pool, up, feats, np.float32)
    assert np.abs(l32 - logits).max() < 5e-5


needs_ref = pytest.mark.skipif(not __import__("oracle.bindings", fromlist=["x"]).have_ref(),
                               reason="oracle/_ref not built (needs /root/reference: make -C oracle ref)")


@needs_ref
@pytest.mark.parametrize("K", [1, 16, 32])
def test_knn_oracle_vs_real_reference_fresh_inputs(oracle, K):
    for p in (uniform_cloud(9000, 21), brats_cloud(9000, 22, grid=(48, 48, 32))):
        assert np.array_equal(oracle.knn_batch(p[None], p[None], K), oracle.ref_knn_batch(p[None], p[None], K))
        sub = p[:2250]
        assert np.array_equal(oracle.knn_batch(sub[None], p[None], 1), oracle.ref_knn_batch(sub[None], p[None], 1))


@needs_ref
def test_knn_oracle_threading_variants_agree(oracle):
    rng = np.random.default_rng(5)
    s = rng.random((4, 2500, 3), dtype=np.float32)
    q = rng.random((4, 700, 3), dtype=np.float32)
    want = oracle.ref_knn_batch(s, q, 16, omp=False)
    assert np.array_equal(oracle.ref_knn_batch(s, q, 16, omp=True), want)
    assert np.array_equal(oracle.knn_batch(s, q, 16, threads=4), want)
    assert np.array_equal(oracle.knn_batch(s, q, 16, threads=3, qpar=True), want)


@needs_ref
def test_grid_oracle_vs_real_reference_fresh_inputs(oracle):
    rng = np.random.default_rng(3)
    p = rng.random((30000, 3), dtype=np.float32)
    f = rng.standard_normal((30000, 3)).astype(np.float32)
    a = oracle.canonical_rows(*oracle.grid_subsample(p, f, None, 0.05))
    b = oracle.canonical_rows(*oracle.ref_grid_subsample(p, f, None, 0.05))
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
