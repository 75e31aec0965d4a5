"""Host sanitizer legs (GPU sanitizers are not available on the pool): the C restatement (oracle/knn_oracle.c,
oracle/grid_oracle.c) and the product's host-side kd-tree code (csrc/kdtree_host.hip + the search routine of csrc/kdtree.h that the
kernels instantiate) are compiled with -fsanitize=address,undefined (every report fatal) behind raw-file drivers and run on the
inputs of the committed golden vectors; the outputs must equal the fixtures' expected arrays -- a clean exit with the right answer
= no memory error, no undefined behaviour on those paths."""
import glob
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", OMP_NUM_THREADS="2")


@pytest.fixture(scope="module")
def oracle_asan():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"])
    return os.path.join(ROOT, "oracle", "asan_check")


@pytest.fixture(scope="module")
def host_asan():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "point-unet_amd", "csrc"), "-s", "asan-host"])
    return os.path.join(ROOT, "point-unet_amd", "csrc", "build", "host_asan_check")


def _run(exe, infile, outfile):
    p = subprocess.run([exe, str(infile), str(outfile)], env=ENV, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, "sanitizer build failed (rc %d):\n%s" % (p.returncode, p.stderr[-4000:])
    assert "ERROR: AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr, p.stderr[-4000:]


def _write_knn(path, support, queries, K, threads=1, qpar=0):
    s, q = np.ascontiguousarray(support, np.float32), np.ascontiguousarray(queries, np.float32)
    hdr = np.array([1, s.shape[0], s.shape[1], q.shape[1], K, threads, qpar, 0], np.int64)
    with open(path, "wb") as f:
        f.write(hdr.tobytes())
        f.write(s.tobytes())
        f.write(q.tobytes())
    return (s.shape[0], q.shape[1], K)


def _knn_files():
    files = sorted(glob.glob(os.path.join(GOLD, "knn_*.npz")))
    assert len(files) >= 8
    return files


def test_knn_restatement_under_asan_ubsan(oracle_asan, tmp_path):
    for f in _knn_files():
        g = np.load(f)
        shape = _write_knn(tmp_path / "in.bin", g["support"], g["queries"], int(g["K"]))
        _run(oracle_asan, tmp_path / "in.bin", tmp_path / "out.bin")
        got = np.fromfile(tmp_path / "out.bin", np.int64).reshape(shape)
        assert np.array_equal(got, g["idx"]), f
    # the OpenMP forms (over the batch, over the queries) on a batched case
    rng = np.random.default_rng(0)
    s, q = rng.random((3, 1500, 3), dtype=np.float32), rng.random((3, 400, 3), dtype=np.float32)
    outs = []
    for threads, qpar in ((1, 0), (2, 0), (2, 1)):
        shape = _write_knn(tmp_path / "in.bin", s, q, 16, threads, qpar)
        _run(oracle_asan, tmp_path / "in.bin", tmp_path / "out.bin")
        outs.append(np.fromfile(tmp_path / "out.bin", np.int64).reshape(shape))
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])


def test_grid_restatement_under_asan_ubsan(oracle_asan, oracle, tmp_path):
    for name, use_f, use_c in (("grid_all.npz", True, True), ("grid_points_only.npz", False, False), ("grid_negative_coords.npz", True, False)):
        g = np.load(os.path.join(GOLD, name))
        p = np.ascontiguousarray(g["points"], np.float32)
        f = np.ascontiguousarray(g["features"], np.float32) if use_f else None
        c = np.ascontiguousarray(g["classes"], np.int32).reshape(len(p), -1) if use_c else None
        fdim, ldim = (f.shape[1] if use_f else 0), (c.shape[1] if use_c else 0)
        with open(tmp_path / "in.bin", "wb") as fh:
            fh.write(np.array([2, len(p), fdim, ldim, 0, 0, 0, 0], np.int64).tobytes())
            fh.write(np.float32(g["sampleDl"]).tobytes())
            fh.write(p.tobytes())
            if use_f:
                fh.write(f.tobytes())
            if use_c:
                fh.write(c.tobytes())
        _run(oracle_asan, tmp_path / "in.bin", tmp_path / "out.bin")
        raw = open(tmp_path / "out.bin", "rb").read()
        M = int(np.frombuffer(raw[:8], np.int64)[0])
        off = 8
        op = np.frombuffer(raw[off:off + 12 * M], np.float32).reshape(M, 3)
        off += 12 * M
        of = np.frombuffer(raw[off:off + 4 * M * fdim], np.float32).reshape(M, fdim) if use_f else None
        off += 4 * M * fdim
        oc = np.frombuffer(raw[off:off + 4 * M * ldim], np.int32).reshape(M, ldim) if use_c else None
        a = oracle.canonical_rows(op, of, oc)
        assert np.array_equal(a[0], g["out_points"]), name
        if use_f:
            assert np.array_equal(a[1], g["out_features"]), name
        if use_c:
            assert np.array_equal(a[2].reshape(g["out_classes"].shape), g["out_classes"]), name


def test_product_host_tree_code_under_asan_ubsan(host_asan, tmp_path):
    """csrc/kdtree_host.hip (construction) + csrc/kdtree.h knn_search_one (the routine the HIP kernel instantiates), host-only build."""
    ran = 0
    for f in _knn_files():
        g = np.load(f)
        K = int(g["K"])
        if K not in (1, 5, 7, 16, 32):
            continue
        shape = _write_knn(tmp_path / "in.bin", g["support"], g["queries"], K)
        _run(host_asan, tmp_path / "in.bin", tmp_path / "out.bin")
        got = np.fromfile(tmp_path / "out.bin", np.int64).reshape(shape)
        assert np.array_equal(got, g["idx"]), f
        ran += 1
    assert ran >= 6
