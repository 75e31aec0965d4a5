"""Drop-in for the reference's CPython module `grid_subsampling`
(PointSegment/utils/cpp_wrappers/cpp_subsampling/wrapper.cpp:58-286), imported by the reference as
`cpp_wrappers.cpp_subsampling.grid_subsampling` (PointSegment/helper_tool.py:16).

    compute(points, features=None, classes=None, sampleDl=0.1, method="barycenters", verbose=0)

returns (points), (points, features), (points, classes) or (points, features, classes) exactly as
wrapper.cpp:269-276 does, raising RuntimeError with the reference's messages on malformed input
(wrapper.cpp:76-190).  Rows come out in ascending voxel-key order (the reference emits unordered_map order).
"""
import ctypes

import numpy as np

def _bind(levels_up):
    """The ctypes binding and the runtime of the package this file ships in.  Works on both import routes: as
    `point_unet_amd.utils.cpp_wrappers.cpp_subsampling.grid_subsampling` and -- the reference's own route, PointSegment/helper_tool.py:13-17:
    `sys.path.append(<package>/utils)` + `import cpp_wrappers.cpp_subsampling.grid_subsampling` -- as a top-level module, where a relative import would
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


_lib, runtime = _bind(3)


def compute(points, features=None, classes=None, sampleDl=0.1, method="barycenters", verbose=0):
    points = np.ascontiguousarray(points, dtype=np.float32)
    if points.ndim != 2:
        raise RuntimeError("Wrong dimensions : points.shape is not (N, 3)")  # wrapper.cpp:118-123
    if points.shape[1] != 3:
        raise RuntimeError("Wrong dimensions : points.shape is not (N, 3)")
    n = points.shape[0]
    use_feature = features is not None
    use_classes = classes is not None
    fdim = ldim = 0
    if use_feature:
        features = np.ascontiguousarray(features, dtype=np.float32)
        if features.ndim != 2:
            raise RuntimeError("Wrong dimensions : features.shape is not (N, d)")  # wrapper.cpp:126-131
        if features.shape[0] != n:
            raise RuntimeError("Wrong dimensions : features.shape is not (N, d)")  # wrapper.cpp:150-157
        fdim = features.shape[1]
    if use_classes:
        classes = np.ascontiguousarray(classes, dtype=np.int32)
        if classes.ndim > 2:
            raise RuntimeError("Wrong dimensions : classes.shape is not (N,) or (N, d)")  # wrapper.cpp:134-139
        if classes.shape[0] != n:
            raise RuntimeError("Wrong dimensions : classes.shape is not (N,) or (N, d)")  # wrapper.cpp:160-167
        ldim = 1 if classes.ndim == 1 else classes.shape[1]
    if n == 0:
        raise RuntimeError("Error")  # wrapper.cpp:225-229: empty result
    ctx = runtime.default_context(0)
    L = _lib.lib()
    M = ctypes.c_int64(0)
    args = (ctx.handle, runtime.ptr(points), n, runtime.ptr(features) if use_feature else None, fdim,
            runtime.ptr(classes) if use_classes else None, ldim, ctypes.c_float(sampleDl))
    _lib.check(L.ps_grid_subsample(*args, ctypes.byref(M), None, None, None))
    if M.value < 1:
        raise RuntimeError("Error")
    out_p = np.zeros((M.value, 3), np.float32)
    out_f = np.zeros((M.value, fdim), np.float32) if use_feature else None
    out_c = np.zeros((M.value, ldim), np.int32) if use_classes else None
    _lib.check(L.ps_grid_subsample(*args, ctypes.byref(M), runtime.ptr(out_p), runtime.ptr(out_f), runtime.ptr(out_c)))
    if use_feature and use_classes:
        return out_p, out_f, out_c
    if use_feature:
        return out_p, out_f
    if use_classes:
        return out_p, out_c
    return out_p
