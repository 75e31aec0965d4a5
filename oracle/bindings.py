"""ctypes bindings for the checkers (TEST INFRASTRUCTURE ONLY).

  * liboracle.so        -- our C restatement (oracle/knn_oracle.c, oracle/grid_oracle.c)
  * _ref/libknn_ref.so  -- the REAL reference KNN (PointSegment/utils/nearest_neighbors/knn_.cxx compiled
                           unmodified, C++-mangled symbols)
  * _ref/libgrid_ref.so -- the REAL reference grid_subsampling.cpp + cloud.cpp behind oracle/grid_ref_shim.cpp

Nothing here reads /root/reference at run time; the _ref libraries are prebuilt by `make -C oracle ref`.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_f32p = ctypes.POINTER(ctypes.c_float)
_i64p = ctypes.POINTER(ctypes.c_int64)
_i32p = ctypes.POINTER(ctypes.c_int32)


def build(ref=True):
    """Compile the checkers (called by __graft_entry__.build())."""
    subprocess.check_call(["make", "-C", _HERE, "-s"])
    if ref and os.path.isdir("/root/reference/PointSegment"):
        subprocess.check_call(["make", "-C", _HERE, "-s", "ref"])


def _load(path):
    if not os.path.exists(path):
        return None
    return ctypes.CDLL(path)


_oracle = None


def oracle_lib():
    global _oracle
    if _oracle is None:
        p = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(p):
            build(ref=False)
        _oracle = ctypes.CDLL(p)
        _oracle.oracle_knn_batch.argtypes = [_f32p, _f32p] + [ctypes.c_int64] * 4 + [_i64p, ctypes.c_int]
        _oracle.oracle_knn_batch.restype = None
        _oracle.oracle_knn_batch_qpar.argtypes = _oracle.oracle_knn_batch.argtypes
        _oracle.oracle_knn_batch_qpar.restype = None
        _oracle.oracle_kdtree_export.argtypes = [_f32p, ctypes.c_int64, _i32p, _i32p, _i32p, _i32p, _f32p,
                                                 _f32p, _f32p, ctypes.c_int64]
        _oracle.oracle_kdtree_export.restype = ctypes.c_int64
        _oracle.oracle_grid_subsample.argtypes = [_f32p, ctypes.c_int64, _f32p, ctypes.c_int64, _i32p,
                                                  ctypes.c_int64, ctypes.c_float, _f32p, _f32p, _i32p]
        _oracle.oracle_grid_subsample.restype = ctypes.c_int64
    return _oracle


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a, t):
    return a.ctypes.data_as(t) if a is not None else None


def knn_batch(support, queries, K, threads=1, qpar=False):
    """Oracle KNN. support [B,N1,3], queries [B,N2,3] -> int64 [B,N2,K] (np.zeros + fill, knn.pyx:93)."""
    s, q = _f32(support), _f32(queries)
    assert s.ndim == 3 and q.ndim == 3 and s.shape[2] == 3 and q.shape[2] == 3 and s.shape[0] == q.shape[0]
    B, n1, n2 = s.shape[0], s.shape[1], q.shape[1]
    out = np.zeros((B, n2, K), dtype=np.int64)
    fn = oracle_lib().oracle_knn_batch_qpar if qpar else oracle_lib().oracle_knn_batch
    fn(_ptr(s, _f32p), _ptr(q, _f32p), B, n1, n2, K, _ptr(out, _i64p), int(threads))
    return out


def kdtree_export(support):
    """Flat arrays of the oracle's kd-tree for one cloud [N,3] (white-box tests of the device builder)."""
    s = _f32(support)
    n = s.shape[0]
    cap = 2 * n + 8
    vind = np.zeros(n, np.int32)
    a = np.zeros(cap, np.int32)
    b = np.zeros(cap, np.int32)
    ax = np.zeros(cap, np.int32)
    lo = np.zeros(cap, np.float32)
    hi = np.zeros(cap, np.float32)
    bbox = np.zeros(6, np.float32)
    nn = oracle_lib().oracle_kdtree_export(_ptr(s, _f32p), n, _ptr(vind, _i32p), _ptr(a, _i32p), _ptr(b, _i32p),
                                           _ptr(ax, _i32p), _ptr(lo, _f32p), _ptr(hi, _f32p), _ptr(bbox, _f32p),
                                           cap)
    assert nn >= 0
    return dict(vind=vind, a=a[:nn], b=b[:nn], axis=ax[:nn], lo=lo[:nn], hi=hi[:nn], bbox=bbox)


def _grid_common(run, fetch, points, features, classes, dl):
    p = _f32(points)
    n = p.shape[0]
    f = _f32(features) if features is not None else None
    c = np.ascontiguousarray(classes, dtype=np.int32) if classes is not None else None
    fdim = f.shape[1] if f is not None else 0
    if c is not None and c.ndim == 1:
        c = c.reshape(n, 1)
    ldim = c.shape[1] if c is not None else 0
    return run(p, n, f, fdim, c, ldim, dl, fetch)


def grid_subsample(points, features=None, classes=None, sampleDl=0.1):
    """Oracle grid subsampling; rows in ascending cell-key order. Returns (pts, feats|None, labels|None)."""

    def run(p, n, f, fdim, c, ldim, dl, _):
        lib = oracle_lib()
        M = lib.oracle_grid_subsample(_ptr(p, _f32p), n, _ptr(f, _f32p), fdim, _ptr(c, _i32p), ldim, dl, None,
                                      None, None)
        op = np.zeros((M, 3), np.float32)
        of = np.zeros((M, fdim), np.float32) if fdim else None
        oc = np.zeros((M, ldim), np.int32) if ldim else None
        lib.oracle_grid_subsample(_ptr(p, _f32p), n, _ptr(f, _f32p), fdim, _ptr(c, _i32p), ldim, dl,
                                  _ptr(op, _f32p), _ptr(of, _f32p), _ptr(oc, _i32p))
        return op, of, oc

    return _grid_common(run, None, points, features, classes, float(sampleDl))


# ------------------------------------------------------------------------------------------------
# the real reference (prebuilt under oracle/_ref; absent => returns None / raises)
# ------------------------------------------------------------------------------------------------
_ref_knn = None
_ref_grid = None


def have_ref():
    return os.path.exists(os.path.join(_HERE, "_ref", "libknn_ref.so")) and os.path.exists(
        os.path.join(_HERE, "_ref", "libgrid_ref.so"))


def ref_knn_batch(support, queries, K, omp=True):
    """The REAL cpp_knn_batch(_omp) (knn_.cxx:72-135) exactly as knn.pyx:71-109 drives it."""
    global _ref_knn
    if _ref_knn is None:
        _ref_knn = ctypes.CDLL(os.path.join(_HERE, "_ref", "libknn_ref.so"))
        for name in ("_Z17cpp_knn_batch_ompPKfmmmS0_mmPl", "_Z13cpp_knn_batchPKfmmmS0_mmPl"):
            fn = getattr(_ref_knn, name)
            fn.argtypes = [_f32p] + [ctypes.c_size_t] * 3 + [_f32p] + [ctypes.c_size_t] * 2 + [
                ctypes.POINTER(ctypes.c_long)]
            fn.restype = None
    s, q = _f32(support), _f32(queries)
    B, n1, n2 = s.shape[0], s.shape[1], q.shape[1]
    out = np.zeros((B, n2, K), dtype=np.int64)
    fn = getattr(_ref_knn, "_Z17cpp_knn_batch_ompPKfmmmS0_mmPl" if omp else "_Z13cpp_knn_batchPKfmmmS0_mmPl")
    fn(_ptr(s, _f32p), B, n1, 3, _ptr(q, _f32p), n2, K, out.ctypes.data_as(ctypes.POINTER(ctypes.c_long)))
    return out


def ref_grid_subsample(points, features=None, classes=None, sampleDl=0.1):
    """The REAL grid_subsampling() (grid_subsampling.cpp:5-106); rows in unordered_map order."""
    global _ref_grid
    if _ref_grid is None:
        _ref_grid = ctypes.CDLL(os.path.join(_HERE, "_ref", "libgrid_ref.so"))
        _ref_grid.ref_grid_subsample_run.argtypes = [_f32p, ctypes.c_long, _f32p, ctypes.c_long, _i32p,
                                                     ctypes.c_long, ctypes.c_float]
        _ref_grid.ref_grid_subsample_run.restype = ctypes.c_long
        _ref_grid.ref_grid_subsample_fetch.argtypes = [_f32p, _f32p, _i32p]
        _ref_grid.ref_grid_subsample_fetch.restype = None

    def run(p, n, f, fdim, c, ldim, dl, _):
        M = _ref_grid.ref_grid_subsample_run(_ptr(p, _f32p), n, _ptr(f, _f32p), fdim, _ptr(c, _i32p), ldim, dl)
        op = np.zeros((M, 3), np.float32)
        of = np.zeros((M, fdim), np.float32) if fdim else None
        oc = np.zeros((M, ldim), np.int32) if ldim else None
        _ref_grid.ref_grid_subsample_fetch(_ptr(op, _f32p), _ptr(of, _f32p), _ptr(oc, _i32p))
        return op, of, oc

    return _grid_common(run, None, points, features, classes, float(sampleDl))


def canonical_rows(pts, feats=None, labels=None):
    """Lexicographic row sort used to compare grid-subsampling outputs."""
    order = np.lexsort((pts[:, 2], pts[:, 1], pts[:, 0]))
    return (pts[order], feats[order] if feats is not None else None,
            labels[order] if labels is not None else None)
