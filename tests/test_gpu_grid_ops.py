"""GPU parity of grid subsampling (bit-exact after the canonical row sort) and of the op-by-op surface
(Network.* static methods) against the oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _canon(oracle, res, have_f, have_l):
    if not isinstance(res, tuple):
        res = (res,)
    p = res[0]
    f = res[1] if have_f else None
    l = res[-1] if have_l else None
    return oracle.canonical_rows(p, f, l)


def test_grid_golden_vectors(oracle):
    from point_unet_amd.helper_tool import DataProcessing as DP
    g = np.load(os.path.join(GOLD, "grid_all.npz"))
    p, f, l = _canon(oracle, DP.grid_sub_sampling(g["points"], g["features"], g["classes"], float(g["sampleDl"])), True, True)
    assert np.array_equal(p, g["out_points"]) and np.array_equal(f, g["out_features"]) and np.array_equal(l, g["out_classes"])
    g = np.load(os.path.join(GOLD, "grid_points_only.npz"))
    res = DP.grid_sub_sampling(g["points"], grid_size=float(g["sampleDl"]))
    assert isinstance(res, np.ndarray)
    assert np.array_equal(_canon(oracle, res, False, False)[0], g["out_points"])
    g = np.load(os.path.join(GOLD, "grid_negative_coords.npz"))
    p, f, _ = _canon(oracle, DP.grid_sub_sampling(g["points"], g["features"], None, float(g["sampleDl"])), True, False)
    assert np.array_equal(p, g["out_points"]) and np.array_equal(f, g["out_features"])


def test_grid_large_cloud_vs_oracle(oracle):
    """~1e6 points at the reference's BraTS grid size 0.01 (helper_tool.py:27)."""
    from point_unet_amd.helper_tool import DataProcessing as DP
    rng = np.random.default_rng(0)
    p = rng.random((1000000, 3), dtype=np.float32)
    f = rng.standard_normal((1000000, 4)).astype(np.float32)
    got = _canon(oracle, DP.grid_sub_sampling(p, f, None, 0.01), True, False)
    want = oracle.canonical_rows(*oracle.grid_subsample(p, f, None, 0.01))
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    # idempotence-style property: every output point lies in the voxel of its inputs, counts add up
    assert len(got[0]) == len(want[0])


@pytest.mark.parametrize("n", [63, 2049, 100003])
@pytest.mark.parametrize("dl", [0.3, 0.01, 0.0011])
def test_grid_sort_passes_and_ragged_tiles(oracle, n, dl):
    """The cell keys need 6 .. 30 bits over these grid sizes = 1 .. 4 radix passes (the sorted pairs end in either buffer), and the
    sizes leave a ragged last tile / a single partial wave; rows equal the oracle's after the canonical sort, labels included."""
    from point_unet_amd.helper_tool import DataProcessing as DP
    rng = np.random.default_rng(n)
    p = (rng.random((n, 3), dtype=np.float32) * np.array([1.0, 0.7, 1.3], np.float32) - 0.25).astype(np.float32)
    f = rng.standard_normal((n, 2)).astype(np.float32)
    l = rng.integers(0, 4, (n, 1)).astype(np.int32)
    got = _canon(oracle, DP.grid_sub_sampling(p, f, l, dl), True, True)
    want = oracle.canonical_rows(*oracle.grid_subsample(p, f, l, dl))
    assert all(np.array_equal(a, b) for a, b in zip(got, want))


def test_grid_single_point_and_single_voxel(oracle):
    from point_unet_amd.helper_tool import DataProcessing as DP
    one = np.array([[0.3, -0.2, 0.9]], np.float32)
    assert np.array_equal(DP.grid_sub_sampling(one, grid_size=0.1), one)
    blob = np.random.default_rng(1).random((500, 3), dtype=np.float32) * 0.01 + 0.5
    got = DP.grid_sub_sampling(blob, grid_size=10.0)
    want = oracle.grid_subsample(blob, None, None, 10.0)[0]
    assert got.shape == (1, 3) and np.array_equal(got, want)


def test_op_by_op_surface_vs_oracle(oracle):
    import torch
    from oracle import randla_oracle as ro
    from point_unet_amd.RandLANet import Network
    rng = np.random.default_rng(0)
    B, N, M, K, d = 2, 900, 300, 16, 24
    pc = rng.standard_normal((B, N, d)).astype(np.float32)
    xyz = rng.random((B, N, 3), dtype=np.float32)
    idx = rng.integers(0, N, (B, N, K)).astype(np.int32)
    pidx = rng.integers(0, N, (B, M, K)).astype(np.int32)
    iidx = rng.integers(0, M, (B, N, 1)).astype(np.int32)
    c = lambda a: torch.from_numpy(a).cuda()  # noqa: E731
    assert np.array_equal(Network.gather_neighbour(c(pc), c(idx)).cpu().numpy(), ro.gather_neighbour(pc, idx))
    rp = Network.relative_pos_encoding(c(xyz), c(idx)).cpu().numpy()
    want = ro.relative_pos_encoding(xyz, idx)
    assert np.array_equal(rp[..., 1:], want[..., 1:]) and np.abs(rp[..., 0] - want[..., 0]).max() < 1e-6
    assert np.array_equal(Network.random_sample(c(pc[:, :, None]), c(pidx)).cpu().numpy()[:, :, 0], ro.random_sample(pc, pidx))
    sub = pc[:, :M]
    assert np.array_equal(Network.nearest_interpolation(c(sub[:, :, None]), c(iidx)).cpu().numpy()[:, :, 0],
                          ro.nearest_interpolation(sub, iidx))
    # conv2d (BN folded) and att_pooling
    W = rng.standard_normal((d, 40)).astype(np.float32) * 0.2
    b = rng.standard_normal(40).astype(np.float32)
    got = Network.conv2d(c(pc), c(W), c(b), leaky=True).cpu().numpy()
    assert np.abs(got - ro.leaky_relu(pc.astype(np.float64) @ W + b)).max() < 1e-5
    fset = rng.standard_normal((B, 200, K, 32)).astype(np.float32)
    wfc = (rng.standard_normal((32, 32)) * 0.3).astype(np.float32)
    wm = (rng.standard_normal((32, 16)) * 0.3).astype(np.float32)
    bm = rng.standard_normal(16).astype(np.float32)
    got = Network.att_pooling(c(fset), c(wfc), c(wm), c(bm)).cpu().numpy()[:, :, 0]
    f64 = fset.astype(np.float64)
    a = f64 @ wfc
    a = np.exp(a - a.max(2, keepdims=True))
    want = ro.leaky_relu((f64 * a / a.sum(2, keepdims=True)).sum(2) @ wm + bm)
    assert np.abs(got - want).max() < 1e-5


def test_point_to_volume_scatter_matches_the_reference_loop():
    """N3: testBraTS.py:83-101 + 226-231 semantics, including duplicate rows / duplicate voxels (last one wins) and
    points that were not sampled (zeros)."""
    import torch
    from point_unet_amd.postprocess import point2prod
    rng = np.random.default_rng(0)
    Z, X, Y, C = 12, 20, 16, 4
    total, n = 900, 700
    xyz = np.stack([rng.integers(0, X, total), rng.integers(0, Y, total), rng.integers(0, Z, total)], 1).astype(np.int32)  # duplicates likely
    p_idx = rng.integers(0, total, n).astype(np.int32)
    logits = rng.standard_normal((n, C)).astype(np.float32)
    e = np.exp(logits.astype(np.float64) - logits.max(1, keepdims=True))
    probs = (e / e.sum(1, keepdims=True))
    test_probs = np.zeros((total, C))
    test_probs[p_idx] = probs                      # numpy fancy assignment: last duplicate wins
    volume = np.zeros((Z, X, Y, C))
    for i in range(total):                          # the reference's Python loop
        volume[xyz[i][2]][xyz[i][0]][xyz[i][1]] = test_probs[i]
    want = np.moveaxis(volume, 1, 2)
    got = point2prod(torch.from_numpy(logits).cuda(), torch.from_numpy(p_idx).cuda(), torch.from_numpy(xyz).cuda(), (Z, X, Y)).cpu().numpy()
    assert got.shape == want.shape and np.abs(got - want).max() < 1e-6


def test_volume_to_cloud_against_the_reference_functions():
    """tests/golden/volume_to_cloud.npz was produced by the reference's own itensity_normalize_one_volume + convert_pc2ply
    (dataPrepareBraTS.py:33-49, 75-116, run by make_golden.py): same points in the same order, same coordinates / labels /
    voxel indices; normalised intensities within one float32 ulp (float64 sums in a different order, then one cast); the
    sub-cloud equal after the canonical row sort; projection distances equal (indices may differ among exact ties)."""
    import os
    from oracle import bindings as ob
    from point_unet_amd.prepare import prepare_brats_volume, volume_to_cloud
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "volume_to_cloud.npz"))
    xyz, colors, labels, origin = volume_to_cloud(g["raw"], np.where(g["seg"] == 4, 3, g["seg"]))
    assert np.array_equal(xyz, g["xyz"]) and np.array_equal(origin, g["xyz_origin"]) and np.array_equal(labels, g["labels"])
    assert np.abs(colors - g["colors"]).max() <= 2e-7 * np.abs(g["colors"]).max()
    out = prepare_brats_volume(g["raw"], g["seg"], sub_grid_size=float(g["sub_grid_size"]))
    sp, sf, sl = ob.canonical_rows(out["sub_xyz"], out["sub_colors"], out["sub_labels"].reshape(-1, 1).astype(np.int32))
    assert np.array_equal(sp, g["sub_xyz"]) and np.array_equal(sl, g["sub_labels"])
    assert np.abs(sf - g["sub_colors"]).max() <= 1e-6
    d = np.linalg.norm(out["xyz"].astype(np.float64) - out["sub_xyz"][out["proj_idx"]].astype(np.float64), axis=1)
    assert np.abs(d - g["proj_dist"]).max() <= 1e-6
    # count-only call, missing-modality error
    with pytest.raises(Exception, match="no voxel above zero"):
        volume_to_cloud(np.zeros((4, 4, 4, 4), np.float32))


def test_chained_preparation_on_the_device_equals_the_three_hop_form():
    """prepare_brats_volume(chained=True) -- ps_volume_to_cloud_dev -> ps_grid_subsample_dev -> ps_knn_batch on device pointers: the volume goes
    up once and the rows stay in HBM between the ops, as dataPrepareBraTS.py:75-116 is ONE pipeline on the host -- against the three
    host-pointer entry points (three PCIe round trips): every array bit for bit, on the golden volume of make_golden.py (which the
    test above holds against the reference's own functions) and on a synthetic BraTS-sized case (240 x 240 x 155 voxels, an ellipsoid
    of ~1.2 M non-zero voxels: the order of magnitude of dataPrepareBraTS.py:78).  Prints both timings."""
    import os
    import time
    from point_unet_amd.prepare import prepare_brats_volume
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "volume_to_cloud.npz"))
    cases = [("golden", g["raw"], g["seg"], float(g["sub_grid_size"]))]
    rng = np.random.default_rng(7)
    X, Y, Z = 240, 240, 155
    ii, jj, kk = np.meshgrid(np.arange(X), np.arange(Y), np.arange(Z), indexing="ij")
    inside = ((ii - 120) / 70.0) ** 2 + ((jj - 120) / 85.0) ** 2 + ((kk - 77) / 60.0) ** 2 <= 1.0
    raw = np.zeros((4, X, Y, Z), np.float32)
    for m in range(4):
        raw[m][inside] = (200.0 + 50.0 * m + 40.0 * rng.standard_normal(int(inside.sum()))).clip(1.0, None).astype(np.float32)
    seg = np.zeros((X, Y, Z), np.int32)
    seg[inside] = rng.integers(0, 5, int(inside.sum()))
    cases.append(("240x240x155", raw, seg, 0.01))
    for name, vol, sg, grid in cases:
        out = {}
        for chained in (False, True):
            prepare_brats_volume(vol, sg, sub_grid_size=grid, chained=chained)  # (warm-up: workspaces, pinned staging)
            t0 = time.perf_counter()
            out[chained] = prepare_brats_volume(vol, sg, sub_grid_size=grid, chained=chained)
            out[chained]["seconds"] = time.perf_counter() - t0
        a, b = out[True], out[False]
        for k in ("xyz", "colors", "labels", "xyz_origin", "sub_xyz", "sub_colors", "sub_labels", "proj_idx"):
            assert a[k].dtype == b[k].dtype and np.array_equal(a[k], b[k]), (name, k)
        print("%s: %d points -> %d sub-cloud points; chained on the device %.3f s, three host-pointer hops %.3f s" % (
            name, len(a["xyz"]), len(a["sub_xyz"]), a["seconds"], b["seconds"]))
