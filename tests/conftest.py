import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import bindings
    bindings.oracle_lib()  # builds liboracle.so on demand
    return bindings


@pytest.fixture(scope="session")
def lib():
    from point_unet_amd import _lib
    return _lib.lib()


@pytest.fixture(scope="session")
def dbg(lib):
    """The TEST-ONLY library point-unet_amd/libpointseg_debug.so (csrc/debug_hooks.h): the product's own host logic behind C doors.
    It links against the product library (loaded first, by the `lib` fixture); the package itself never loads it."""
    import ctypes
    c_vp, i64 = ctypes.c_void_p, ctypes.c_int64
    h = ctypes.CDLL(os.path.join(ROOT, "point-unet_amd", "libpointseg_debug.so"))
    protos = {
        "ps_debug_knn_host": [c_vp, c_vp, i64, i64, i64, i64, c_vp],
        "ps_debug_kdtree_host": [c_vp, i64, c_vp, c_vp, c_vp, c_vp, c_vp],
        "ps_debug_kdtree_device": [c_vp, c_vp, i64, c_vp, c_vp, c_vp, c_vp, c_vp],
        "ps_debug_pack_weights": [c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_vp],
        "ps_debug_pack_b3": [c_vp, ctypes.c_int, ctypes.c_int, c_vp],
        "ps_debug_gemm32": [c_vp, ctypes.c_int, c_vp, ctypes.c_int, ctypes.c_int, c_vp, c_vp, ctypes.c_int, ctypes.c_int, c_vp, ctypes.c_int, ctypes.c_int,
                            c_vp, c_vp, i64, ctypes.c_int, ctypes.c_int, c_vp, ctypes.c_int],
    }
    for name, args in protos.items():
        fn = getattr(h, name)
        fn.restype = ctypes.c_int
        fn.argtypes = args
    return h


# ---- synthetic clouds shared by CPU and GPU tests ---------------------------------------------
def uniform_cloud(n, seed=0):
    return np.random.default_rng(seed).random((n, 3), dtype=np.float32)


def brats_cloud(n, seed=0, grid=(240, 240, 155)):
    """Voxel-lattice cloud: n voxels sampled without replacement from an ellipsoid mask inside `grid`,
    xyz = ijk / grid as float32, shuffled (mirrors PointSegment/utils/dataPrepareBraTS.py:78-89).  Lattice
    coordinates make equal-distance ties ubiquitous -- the hard case for KNN parity."""
    rng = np.random.default_rng(seed)
    g = np.asarray(grid)
    # rejection-sample integer voxels inside the ellipsoid
    out = np.empty((0, 3), np.int64)
    while len(out) < n:
        c = rng.integers(0, g, size=(2 * n, 3))
        u = (c - g / 2.0) / (0.45 * g)
        c = c[(u * u).sum(1) < 1.0]
        out = np.unique(np.concatenate([out, c]), axis=0)
    out = out[rng.permutation(len(out))[:n]]
    return (out / g).astype(np.float32)
