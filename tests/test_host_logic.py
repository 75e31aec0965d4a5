"""CPU tests of the product's host-side logic (no GPU): the kd-tree construction rules and the per-query search
routine that the HIP kernel instantiates (run on the host through the ps_debug_* doors), the MFMA weight packing,
and the BatchNorm folding / blob layout (checked by replaying the device's launch plan in NumPy)."""
import ctypes
import os

import numpy as np
import pytest

import netcase
from conftest import brats_cloud, uniform_cloud


def _host_knn(dbg, lib, s, q, K):
    s = np.ascontiguousarray(s, np.float32)
    q = np.ascontiguousarray(q, np.float32)
    out = np.zeros((s.shape[0], q.shape[1], K), np.int32)
    rc = dbg.ps_debug_knn_host(s.ctypes.data, q.ctypes.data, s.shape[0], s.shape[1], q.shape[1], K, out.ctypes.data)
    assert rc == 0, lib.ps_last_error()
    return out


@pytest.mark.parametrize("K", [1, 16, 32])
@pytest.mark.parametrize("kind", ["uniform", "lattice"])
def test_search_routine_matches_oracle(dbg, lib, oracle, kind, K):
    p = uniform_cloud(6000, 1) if kind == "uniform" else brats_cloud(6000, 1, grid=(40, 40, 30))
    assert np.array_equal(_host_knn(dbg, lib, p[None], p[None], K), oracle.knn_batch(p[None], p[None], K))


def test_search_routine_upsampling_and_small(dbg, lib, oracle):
    p = brats_cloud(8000, 2, grid=(40, 40, 30))
    sub = p[:2000]
    assert np.array_equal(_host_knn(dbg, lib, sub[None], p[None], 1), oracle.knn_batch(sub[None], p[None], 1))
    rng = np.random.default_rng(0)
    for n in (1, 3, 10, 11, 25):
        q = rng.random((n, 3), dtype=np.float32)
        for K in (1, 5, 16):
            assert np.array_equal(_host_knn(dbg, lib, q[None], q[None], K), oracle.knn_batch(q[None], q[None], K)), (n, K)


def test_tree_layout_against_oracle_tree(dbg, lib, oracle):
    """Same permutation, same splits, same child order as the oracle's tree, in the product's id scheme."""
    p = brats_cloud(5000, 4, grid=(32, 32, 24))
    n = len(p)
    vind = np.zeros(n, np.int32)
    nodes = np.zeros((2 * n, 4), np.int32)
    pts = np.zeros((n, 4), np.float32)
    rd = np.zeros(2, np.int32)
    bbox = np.zeros(6, np.float32)
    assert dbg.ps_debug_kdtree_host(p.ctypes.data, n, vind.ctypes.data, nodes.ctypes.data, pts.ctypes.data, rd.ctypes.data,
                                    bbox.ctypes.data) == 0
    t = oracle.kdtree_export(p)
    assert np.array_equal(vind, t["vind"])
    assert np.array_equal(bbox, t["bbox"])
    assert np.array_equal(pts[:, :3], p[vind]) and np.array_equal(pts[:, 3].view(np.int32), vind)

    def walk(oid, ref, lo, hi):
        pid, cnt = ref & ((1 << 26) - 1), ref >> 26  # reference = node id | leaf point count << 26 (csrc/kdtree.h)
        if t["axis"][oid] < 0:
            assert cnt == hi - lo and pid % 2 == 0 and pid // 2 == t["a"][oid] == lo and nodes[pid, 0] == lo and nodes[pid, 1] == t["b"][oid] == hi
            return 1
        assert cnt == 0 and pid % 2 == 1
        m = (pid + 1) // 2
        ax = (int(nodes[pid, 0]) & 0xffffffff) >> 30
        assert ax == t["axis"][oid]
        assert nodes[pid, 2:].view(np.float32)[0] == t["lo"][oid] and nodes[pid, 2:].view(np.float32)[1] == t["hi"][oid]
        c1, c2 = int(nodes[pid, 0]) & 0x3fffffff, int(nodes[pid, 1])
        return 1 + max(walk(t["a"][oid], c1, lo, m), walk(t["b"][oid], c2, m, hi))

    depth = walk(0, int(rd[0]), 0, n)
    assert depth - 1 == rd[1]


def test_weight_packing_is_the_mfma_b_fragment_order(dbg, lib):
    rng = np.random.default_rng(0)
    for cin, cout, ntb in [(7, 8, 1), (10, 32, 2), (96, 128, 4), (24, 32, 2)]:
        W = rng.standard_normal((cin, cout)).astype(np.float32)
        ks, cb = (cin + 3) // 4, (cout + 16 * ntb - 1) // (16 * ntb)
        out = np.zeros(cb * ks * 64 * ntb, np.float32)
        assert dbg.ps_debug_pack_weights(W.ctypes.data, cin, cout, ntb, out.ctypes.data) == 0
        out = out.reshape(cb, ks, 64, ntb)
        Wp = np.zeros((ks * 4, cb * ntb * 16), np.float32)
        Wp[:cin, :cout] = W
        for l in (0, 5, 17, 33, 63):
            for s in range(ks):
                for c in range(cb):
                    for j in range(ntb):
                        assert out[c, s, l, j] == Wp[s * 4 + (l >> 4), (c * ntb + j) * 16 + (l & 15)]


def test_split_bf16_planes_add_up_exactly_and_follow_the_accumulator_order(dbg, lib):
    """pack_b3 (csrc/attpool32b.hip): every fp32 weight is stored as three bfloat16 pieces whose sum IS the weight (8 + 8 + 8
    significant bits, exact subtractions), in the K order of an operand that comes out of a transposed product's accumulators: chunk q,
    lane half g, element j <-> channel 32 (q >> 1) + (r & 3) + 8 (r >> 2) + 4 g with r = 8 (q & 1) + j."""
    rng = np.random.default_rng(4)
    for cin, cout in [(32, 32), (64, 128), (128, 64)]:
        W = (rng.standard_normal((cin, cout)) * np.exp(rng.uniform(-20, 20, (cin, cout)))).astype(np.float32)
        W[0, 0], W[1, 1] = 0.0, -1.0
        nq, cbs = cin // 16, cout // 32
        out = np.zeros(cbs * nq * 3 * 64 * 8, np.uint16)
        assert dbg.ps_debug_pack_b3(W.ctypes.data, cin, cout, out.ctypes.data) == 0
        planes = (out.astype(np.uint32) << 16).view(np.float32).reshape(cbs, nq, 3, 64, 8)
        seen = np.zeros((cin, cout), np.int32)
        for cb in range(cbs):
            for q in range(nq):
                for l in range(64):
                    for j in range(8):
                        r = 8 * (q & 1) + j
                        k = 32 * (q >> 1) + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)
                        col = 32 * cb + (l & 31)
                        p1, p2, p3 = (np.float64(planes[cb, q, pl, l, j]) for pl in range(3))
                        assert p1 + p2 + p3 == np.float64(W[k, col]), (cin, cout, k, col)
                        assert abs(p2) <= abs(p1) * 2.0 ** -7 + 1e-45 and abs(p3) <= abs(p1) * 2.0 ** -15 + 1e-45
                        seen[k, col] += 1
        assert np.all(seen == 1)


def _replay_device_plan(cfg, blob, xyz, nbr, pool, up, feats):
    """NumPy float64 replay of csrc/randla.hip's launch plan from the folded blob: same layer order, the
    G = f.Wfc[:h] pre-product, [mlp2;shortcut] as one GEMM over the concatenated K axis."""
    from oracle import randla_oracle as ro
    from point_unet_amd import weights
    pos = 0
    layers = []
    for scope, kind, cin, cout in weights.layer_dims(cfg):
        W = blob[pos:pos + cin * cout].reshape(cin, cout).astype(np.float64)
        pos += cin * cout
        b = blob[pos:pos + cout].astype(np.float64)
        pos += cout
        layers.append((W, b))
    assert pos == blob.size
    it = iter(layers)
    lrelu = ro.leaky_relu

    def dense(x, Wb, act):
        y = x @ Wb[0] + Wb[1]
        return lrelu(y) if act else y

    def att(f, G, f_xyz, Wbot):
        scores = ro.gather_neighbour(G, idx) + f_xyz @ Wbot
        scores = scores - scores.max(2, keepdims=True)
        e = np.exp(scores)
        fset = np.concatenate([ro.gather_neighbour(f, idx), f_xyz], -1)
        return (e * fset).sum(2) / e.sum(2)

    X = dense(feats.astype(np.float64), next(it), True)
    enc = []
    for i in range(cfg.num_layers):
        d = cfg.d_out[i]
        h = d // 2
        idx = nbr[i]
        mlp1, lfa1, fc1, a1mlp, lfa2, fc2, a2mlp, mlp2, sc = [next(it) for _ in range(9)]
        enc10 = ro.relative_pos_encoding(xyz[i].astype(np.float64), idx)
        f_pc = dense(X, mlp1, True)
        f_xyz1 = dense(enc10, lfa1, True)
        agg1 = att(f_pc, f_pc @ fc1[0][:h], f_xyz1, fc1[0][h:])
        f_agg1 = dense(agg1, a1mlp, True)
        f_xyz2 = dense(f_xyz1, lfa2, True)
        agg2 = att(f_agg1, f_agg1 @ fc2[0][:h], f_xyz2, fc2[0][h:])
        tmp = dense(agg2, a2mlp, True)
        f_enc = lrelu(np.concatenate([tmp, X], -1) @ np.concatenate([mlp2[0], sc[0]], 0) + (mlp2[1] + sc[1]))
        X = ro.random_sample(f_enc, pool[i])
        if i == 0:
            enc.append(f_enc)
        enc.append(X)
    f = dense(X, next(it), True)
    for j in range(cfg.num_layers):
        f = dense(np.concatenate([enc[-j - 2], ro.nearest_interpolation(f, up[-j - 1])], -1), next(it), True)
    f = dense(f, next(it), True)
    f = dense(f, next(it), True)
    return dense(f, next(it), False)


def test_folded_blob_and_launch_plan_reproduce_the_reference_graph(oracle):
    from oracle import randla_oracle as ro
    from point_unet_amd import weights
    cfg, xyz, feats = netcase.small_deep(1500, seed=1)
    cfg.d_out = [16, 32, 64, 32, 16]  # keep the float64 replay fast; the layout logic is width independent
    params = weights.init_params(cfg, seed=3, randomize_bn=True)
    blob = weights.fold_to_blob(cfg, params)
    pts, nbr, pool, up = ro.build_pyramid(lambda s, q, k: oracle.knn_batch(s, q, k), xyz, cfg.k_n, cfg.sub_sampling_ratio)
    want = ro.inference(params, cfg.num_layers, pts, nbr, pool, up, feats, np.float64)
    got = _replay_device_plan(cfg, blob, pts, nbr, pool, up, feats)
    # the blob is fp32 (folded in float64, rounded once): agreement to fp32 rounding of the weights
    assert np.abs(got - want).max() < 2e-5 * max(1.0, np.abs(want).max())


def test_param_count_matches_the_reference_model():
    from point_unet_amd import weights
    from point_unet_amd.helper_tool import ConfigBraTS, ConfigPancreas
    assert weights.num_params(ConfigBraTS) == 4992852
    assert weights.num_params(ConfigPancreas) == 4992762


def test_facade_validates_arguments():
    from point_unet_amd.utils.cpp_wrappers.cpp_subsampling import grid_subsampling
    from point_unet_amd.utils.nearest_neighbors.lib.python import nearest_neighbors as nn
    with pytest.raises(ValueError):
        nn.knn_batch(np.zeros((1, 5, 2), np.float32), np.zeros((1, 5, 2), np.float32), 3)
    with pytest.raises(ValueError):
        nn.knn_batch(np.zeros((2, 5, 3), np.float32), np.zeros((1, 5, 3), np.float32), 3)
    with pytest.raises(RuntimeError, match="points.shape is not"):
        grid_subsampling.compute(np.zeros((5, 2), np.float32))
    with pytest.raises(RuntimeError, match="features.shape is not"):
        grid_subsampling.compute(np.zeros((5, 3), np.float32), features=np.zeros((4, 2), np.float32))
    with pytest.raises(RuntimeError, match="classes.shape is not"):
        grid_subsampling.compute(np.zeros((5, 3), np.float32), classes=np.zeros((4,), np.int32))


def test_tf_variable_importer_round_trip():
    """N1: TF-1.x checkpoint variable names/shapes -> parameter dict (names per SURVEY 8f N1)."""
    from point_unet_amd import weights
    from point_unet_amd.helper_tool import ConfigBraTS
    p = weights.init_params(ConfigBraTS, seed=1, randomize_bn=True)
    tf_vars = {}
    for k, v in p.items():
        if k.endswith("/weights"):
            v = v.reshape((1, 1) + v.shape)  # conv kernels are 4-D in the checkpoint
        tf_vars["layers/" + k] = v
        if not k.endswith(("moving_mean", "moving_variance")):
            tf_vars["layers/" + k + "/Adam"] = np.zeros_like(v)  # optimizer slots are saved too (RandLANet.py:101-102)
    back = weights.from_tf_variables(ConfigBraTS, tf_vars)
    assert set(back) == set(p)
    for k in p:
        assert np.array_equal(back[k], p[k]), k
    assert np.array_equal(weights.fold_to_blob(ConfigBraTS, back), weights.fold_to_blob(ConfigBraTS, p))


def test_ply_reader_and_writer_against_the_reference_file(tmp_path):
    """tests/golden/brats_example.ply was written by the reference's helper_ply.write_ply (make_golden.py): this package's
    reader returns the same fields, and this package's writer reproduces the file byte for byte."""
    import os
    from point_unet_amd import dataset, helper_ply
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g = np.load(os.path.join(here, "ply_example.npz"))
    data = helper_ply.read_ply(os.path.join(here, "brats_example.ply"))
    assert data.dtype.names == ("x", "y", "z", "t1ce", "t1", "flair", "t2", "class")
    assert np.array_equal(np.stack([data["x"], data["y"], data["z"]], 1), g["xyz"]) and np.array_equal(data["class"], g["cls"])
    assert np.array_equal(np.stack([data[m] for m in dataset.BRATS_MODALITIES], 1), g["mods"])
    out = str(tmp_path / "mine")
    assert helper_ply.write_ply(out, [g["xyz"], g["mods"], g["cls"]], ["x", "y", "z", "t1ce", "t1", "flair", "t2", "class"])
    assert open(out + ".ply", "rb").read() == open(os.path.join(here, "brats_example.ply"), "rb").read()
    assert helper_ply.write_ply(out, [g["xyz"], g["cls"][:10]], ["x", "y", "z", "class"]) is False  # ragged fields
    # mesh variant round trip
    tri = np.array([[0, 1, 2], [2, 3, 4]], np.int32)
    assert helper_ply.write_ply(out + "_m.ply", g["xyz"][:5], ["x", "y", "z"], triangular_faces=tri)
    v, f = helper_ply.read_ply(out + "_m.ply", triangular_mesh=True)
    assert np.array_equal(f, tri) and np.array_equal(v["z"], g["xyz"][:5, 2])


def test_brats_sampler_keeps_every_tumour_voxel_and_shuffles():
    """runBraTS.py:107-114: all tumour voxels + a background sample up to num_points, shuffled; features = [xyz | modalities]."""
    import os
    from point_unet_amd import dataset, helper_ply
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    data = helper_ply.read_ply(os.path.join(here, "brats_example.ply"))
    xyz, mods, labels, idx = dataset.sample_brats_cloud(data, 300, np.random.default_rng(0))
    assert xyz.shape == (300, 3) and mods.shape == (300, 4) and idx.dtype == np.int32 and len(np.unique(idx)) == 300
    assert set(np.flatnonzero(data["class"] > 0)) <= set(idx.tolist())
    assert np.array_equal(labels, data["class"][idx]) and np.array_equal(xyz[:, 0], data["x"][idx])
    assert not np.array_equal(idx, np.sort(idx))
    assert dataset.network_features(xyz, mods).shape == (300, 7)
    with pytest.raises(ValueError):
        dataset.sample_brats_cloud(data, 501)


def test_iou_from_confusions_matches_the_reference_formula():
    from point_unet_amd.helper_tool import DataProcessing as DP
    rng = np.random.default_rng(0)
    c = rng.integers(0, 50, (3, 4, 4))
    c[1, 2, :] = 0  # class 2 absent from matrix 1: takes the mean IoU of the present classes
    got = DP.IoU_from_confusions(c)
    tp = np.diagonal(c, axis1=-2, axis2=-1).astype(np.float64)
    fn_tp, fp_tp = c.sum(-1), c.sum(-2)
    iou = tp / (fp_tp + fn_tp - tp + 1e-6)
    mask = fn_tp < 1e-3
    want = iou + mask * (iou.sum(-1, keepdims=True) / ((1 - mask).sum(-1, keepdims=True) + 1e-6))
    assert np.allclose(got, want, rtol=1e-12, atol=1e-12)
    assert np.allclose(DP.get_class_weights("BraTS_Block64"), 1 / (np.array([1403, 22, 80, 11]) / 1516.0 + 0.02))


def test_package_import_sets_the_pipeline_environment_contract():
    """point_unet_amd/__init__.py: a fresh interpreter WITHOUT GPU_MAX_HW_QUEUES gets 6 (the lanes of ForwardPipeline need a hardware queue
    each; HIP reads the variable once, when its runtime starts), a user's own value is left alone, and ForwardPipeline's check warns when
    the runtime has fewer queues than lanes + 2 (pipeline.py::_check_environment).  Nothing else in the environment is touched by the
    import (HSA_ENABLE_IPC_MODE_LEGACY is the launcher's setting: bench.py, tests/conftest.py)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = "import os, sys; sys.path.insert(0, %r); import point_unet_amd; print(os.environ['GPU_MAX_HW_QUEUES'], os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', 'unset'))" % root
    env = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "HSA_ENABLE_IPC_MODE_LEGACY")}
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout.split()
    assert out == ["6", "unset"]
    env["GPU_MAX_HW_QUEUES"] = "5"
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout.split()
    assert out[0] == "5"
    import warnings
    from point_unet_amd import pipeline
    old = os.environ.get("GPU_MAX_HW_QUEUES")
    try:
        os.environ["GPU_MAX_HW_QUEUES"] = "4"
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            pipeline._check_environment(4, False)
        assert len(w) == 1 and "GPU_MAX_HW_QUEUES" in str(w[0].message)
        os.environ["GPU_MAX_HW_QUEUES"] = "6"
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            pipeline._check_environment(4, False)
        assert not w
    finally:
        if old is None:
            os.environ.pop("GPU_MAX_HW_QUEUES", None)
        else:
            os.environ["GPU_MAX_HW_QUEUES"] = old


def test_the_default_library_never_reads_the_environment():
    """the A/B knobs of profiles/tools/* are ONE struct (csrc/common.h, ps::Tuning) that only a library built with
    -DPS_TUNING_ENV fills from PS_* variables, once, in ps_create.  The default build holds constants: every getenv in csrc/ sits inside an
    #ifdef of an experiment macro, there is no function-static cache of a knob, and the shipped .so does not even contain the variables'
    names -- PS_INV_BUCKET=0 in the environment of the default build cannot change what it runs."""
    import re
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "point-unet_amd", "csrc")
    allowed_macros = ("PS_TUNING_ENV", "PS_KNN_REFILL_EXP", "PS_KNN_PROF")
    for name in sorted(os.listdir(csrc)):
        if not name.endswith((".hip", ".h")):
            continue
        depth_of = []  # stack of the macro names of the open #if blocks
        for ln, line in enumerate(open(os.path.join(csrc, name)), 1):
            t = line.strip()
            if t.startswith(("#ifdef", "#ifndef", "#if ")):
                depth_of.append(t)
            elif t.startswith("#endif") and depth_of:
                depth_of.pop()
            code = t.split("//")[0]
            if "getenv" in code:
                assert any(any(m in d for m in allowed_macros) for d in depth_of), "%s:%d reads the environment in the default build: %s" % (name, ln, t)
            assert not re.search(r"static\s+const\s+\w+\s+\w+\s*=.*getenv", code), "%s:%d caches a knob in a function-static" % (name, ln)
    so = open(os.path.join(os.path.dirname(csrc), "libpointseg_hip.so"), "rb").read()
    for var in (b"PS_INV_BUCKET", b"PS_WGRAD_WGS", b"PS_INV_TILE", b"PS_TRAIN_ATT_GEMM", b"PS_GEMM32B_RW", b"PS_KNN_REFILL"):
        assert var not in so, var


def test_unroll_failures_stay_visible_in_the_build():
    """Round 6: `#pragma unroll 8` had been silently refused in gemm32.hip's K loop (one exposed round trip per 8-wide chunk, nine launches of
    a cloud 20 % slower than they had to be) behind -Wno-pass-failed.  The flag is gone: a "loop not unrolled" warning is a performance bug
    report -- and the recipe keeps -Wall, so it is printed by every build (the tree compiles without one)."""
    mk = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "point-unet_amd", "csrc", "Makefile")).read()
    flags = [ln for ln in mk.splitlines() if ln.startswith("FLAGS")][0]
    assert "-Wno-pass-failed" not in flags and "-Wall" in flags
