"""The reference's own import route for the two native modules (PointSegment/helper_tool.py:13-17):

    sys.path.append(os.path.join(BASE_DIR, 'utils'))
    import cpp_wrappers.cpp_subsampling.grid_subsampling as cpp_subsampling
    import nearest_neighbors.lib.python.nearest_neighbors as nearest_neighbors

with BASE_DIR = this package's directory.  Run in a fresh interpreter that has ONLY `<package>/utils` added to sys.path (not the
repo root), so a relative import that climbs out of the top-level package would fail exactly as it would for a reference user."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
UTILS = os.path.join(ROOT, "point-unet_amd", "utils")

IMPORTS = textwrap.dedent("""
    import os, sys
    BASE_DIR = %r
    sys.path.append(BASE_DIR)
    sys.path.append(os.path.join(BASE_DIR, 'utils'))
    import cpp_wrappers.cpp_subsampling.grid_subsampling as cpp_subsampling
    import nearest_neighbors.lib.python.nearest_neighbors as nearest_neighbors
""") % os.path.join(ROOT, "point-unet_amd")


def _run(body, *args):
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    return subprocess.run([sys.executable, "-c", IMPORTS + textwrap.dedent(body)] + list(args), cwd="/tmp", env=env, capture_output=True, text=True,
                          timeout=600)


def test_reference_import_lines_work_from_a_fresh_interpreter():
    r = _run("""
        assert callable(nearest_neighbors.knn_batch) and callable(nearest_neighbors.knn) and callable(cpp_subsampling.compute)
        import inspect
        assert list(inspect.signature(nearest_neighbors.knn_batch).parameters) == ["pts", "queries", "K", "omp"]      # knn.pyx:71
        assert list(inspect.signature(cpp_subsampling.compute).parameters)[:4] == ["points", "features", "classes", "sampleDl"]  # wrapper.cpp:64-65
        print("ok", nearest_neighbors.__name__, cpp_subsampling.__name__)
    """)
    assert r.returncode == 0, r.stderr
    assert r.stdout.split() == ["ok", "nearest_neighbors.lib.python.nearest_neighbors", "cpp_wrappers.cpp_subsampling.grid_subsampling"]


def test_both_routes_share_one_binding():
    """Package route and reference route in one interpreter: one libpointseg_hip.so handle, one context table."""
    r = subprocess.run([sys.executable, "-c", textwrap.dedent("""
        import sys
        sys.path.insert(0, %r)
        import point_unet_amd.helper_tool as ht
        sys.path.append(%r)
        import nearest_neighbors.lib.python.nearest_neighbors as nn
        import cpp_wrappers.cpp_subsampling.grid_subsampling as gs
        assert nn._lib is ht.nearest_neighbors._lib is gs._lib and nn.runtime is gs.runtime
        print("ok")
    """) % (ROOT, UTILS)], cwd="/tmp", capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stderr


@pytest.mark.gpu
def test_knn_batch_and_grid_through_the_reference_route_match_the_goldens(tmp_path, oracle):
    """knn_batch(..., omp=True) exactly as DataProcessing.knn_search calls it (helper_tool.py:91) and cpp_subsampling.compute
    as grid_sub_sampling calls it (helper_tool.py:133-143), both through the reference's module paths, against goldens
    produced by the real reference (tests/golden/make_golden.py)."""
    g = os.path.join(ROOT, "tests", "golden")
    out = str(tmp_path / "res.npz")
    r = _run("""
        import numpy as np
        g, out = sys.argv[1], sys.argv[2]
        k = np.load(os.path.join(g, "knn_lattice_k16.npz"))
        idx = nearest_neighbors.knn_batch(k["support"], k["queries"], int(k["K"]), omp=True)
        u = np.load(os.path.join(g, "knn_upsample_k1.npz"))
        up = nearest_neighbors.knn_batch(u["support"], u["queries"], 1, omp=True).astype(np.int32)
        a = np.load(os.path.join(g, "grid_all.npz"))
        p, f, c = cpp_subsampling.compute(a["points"], features=a["features"], classes=a["classes"], sampleDl=float(a["sampleDl"]), verbose=0)
        np.savez(out, idx=idx, up=up, p=p, f=f, c=c)
    """, g, out)
    assert r.returncode == 0, r.stderr
    res = np.load(out)
    k = np.load(os.path.join(g, "knn_lattice_k16.npz"))
    assert res["idx"].dtype == np.int64 and np.array_equal(res["idx"], k["idx"])
    assert np.array_equal(res["up"], np.load(os.path.join(g, "knn_upsample_k1.npz"))["idx"].astype(np.int32))
    a = np.load(os.path.join(g, "grid_all.npz"))
    p, f, c = oracle.canonical_rows(res["p"], res["f"], res["c"])  # the reference emits unordered_map order: compare after the row sort
    assert np.array_equal(p, a["out_points"]) and np.array_equal(f, a["out_features"]) and np.array_equal(c, a["out_classes"])
