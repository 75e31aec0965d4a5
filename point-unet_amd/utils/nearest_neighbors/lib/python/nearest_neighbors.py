"""Drop-in for the reference's Cython module `nearest_neighbors` (PointSegment/utils/nearest_neighbors/knn.pyx),
imported by the reference as `nearest_neighbors.lib.python.nearest_neighbors` (PointSegment/helper_tool.py:17).

    knn(pts, queries, K, omp=False)        -> int64 [N2, K]       knn.pyx:33-69
    knn_batch(pts, queries, K, omp=False)  -> int64 [B, N2, K]    knn.pyx:71-109

Same argument meaning and return dtype; `omp` is accepted and ignored (the search runs on the GPU).  Inputs are
made contiguous float32 like knn.pyx:95-96 does.  Results are index-for-index those of the reference, including
the order among equal distances.
"""
import ctypes

import numpy as np

def _bind(levels_up):
    """The ctypes binding and the runtime of the package this file ships in.  Works on both import routes: as
    `point_unet_amd.utils.nearest_neighbors.lib.python.nearest_neighbors` and -- the reference's own route, PointSegment/helper_tool.py:13-17:
    `sys.path.append(<package>/utils)` + `import nearest_neighbors.lib.python.nearest_neighbors` -- as a top-level module, where a relative import would
    climb out of the top-level package: the package is then loaded by its path."""
    import importlib
    import importlib.util
    import os
    import sys
    if "point_unet_amd" not in sys.modules:
        pkg = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), *[os.pardir] * levels_up))
        spec = importlib.util.spec_from_file_location("point_unet_amd", os.path.join(pkg, "__init__.py"), submodule_search_locations=[pkg])
        mod = importlib.util.module_from_spec(spec)
        sys.modules["point_unet_amd"] = mod
        spec.loader.exec_module(mod)
    return importlib.import_module("point_unet_amd._lib"), importlib.import_module("point_unet_amd.runtime")


_lib, runtime = _bind(4)


def _run(pts, queries, K):
    pts = np.ascontiguousarray(pts, dtype=np.float32)
    queries = np.ascontiguousarray(queries, dtype=np.float32)
    if pts.ndim != 3 or queries.ndim != 3:
        raise ValueError("knn_batch expects [B,N,dim] arrays")
    if pts.shape[0] != queries.shape[0]:
        raise ValueError("knn_batch: batch sizes differ (%d vs %d)" % (pts.shape[0], queries.shape[0]))
    if pts.shape[2] != 3 or queries.shape[2] != 3:
        raise ValueError("knn_batch: only dim == 3 is supported")
    if pts.shape[1] == 0:
        raise ValueError("knn_batch: empty support set")  # the reference asserts npts != 0 (KDTreeTableAdaptor.h:136)
    B, n1, n2 = pts.shape[0], pts.shape[1], queries.shape[1]
    out = np.zeros((B, n2, int(K)), dtype=np.int64)
    ctx = runtime.default_context(0)
    _lib.check(_lib.lib().ps_knn_batch_i64(ctx.handle, runtime.ptr(pts), runtime.ptr(queries), B, n1, n2, 3, int(K),
                                           runtime.ptr(out), 0))
    return out


def knn_batch(pts, queries, K, omp=False):
    return _run(pts, queries, K)


def knn(pts, queries, K, omp=False):
    pts = np.asarray(pts)
    queries = np.asarray(queries)
    if pts.ndim != 2 or queries.ndim != 2:
        raise ValueError("knn expects [N,dim] arrays")
    return _run(pts[None], queries[None], K)[0]
