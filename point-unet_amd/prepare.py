"""Dataset preparation on the device: MR volume -> full point cloud -> grid-subsampled cloud -> projection indices, the
pipeline of PointSegment/utils/dataPrepareBraTS.py (load_volume :31-72, convert_pc2ply :75-116) without the file I/O.

    volume_to_cloud      ps_volume_to_cloud   z-score per modality over its voxels > 0, non-zero voxels -> points (x-major)
    prepare_brats_volume                      + DataProcessing.grid_sub_sampling(sub_grid_size) and the 1-NN projection of the
                                              full cloud onto the sub-cloud (the reference asks sklearn's KDTree; here the
                                              exact HIP KNN -- equal distances, possibly a different index among exact ties)
"""
import ctypes

import numpy as np

from . import _lib, runtime
from .helper_tool import DataProcessing as DP


def volume_to_cloud(volumes, seg=None, device=0, ctx=None):
    """volumes: [4, X, Y, Z] raw intensities (any real dtype; taken as float32); seg: [X, Y, Z] integer labels or None.
    Returns xyz f32 [n,3], colors f32 [n,4], labels (uint8 [n], zeros without `seg`), xyz_origin int32 [n,3]."""
    vol = np.ascontiguousarray(volumes, dtype=np.float32)
    if vol.ndim != 4 or vol.shape[0] != 4:
        raise ValueError("volumes must have shape [4, X, Y, Z]")
    X, Y, Z = vol.shape[1:]
    s = None
    if seg is not None:
        s = np.ascontiguousarray(seg, dtype=np.int32)
        if s.shape != (X, Y, Z):
            raise ValueError("seg must have shape [X, Y, Z]")
    ctx = ctx or runtime.default_context(device)
    lib = _lib.lib()
    n = ctypes.c_int64(0)
    null = ctypes.c_void_p(0)
    _lib.check(lib.ps_volume_to_cloud(ctx.handle, runtime.ptr(vol), runtime.ptr(s), X, Y, Z, ctypes.byref(n), null, null, null, null))
    m = int(n.value)
    xyz = np.empty((m, 3), np.float32)
    colors = np.empty((m, 4), np.float32)
    labels = np.zeros(m, np.int32)
    origin = np.empty((m, 3), np.int32)
    if m:
        _lib.check(lib.ps_volume_to_cloud(ctx.handle, runtime.ptr(vol), runtime.ptr(s), X, Y, Z, ctypes.byref(n), runtime.ptr(xyz),
                                          runtime.ptr(colors), runtime.ptr(labels), runtime.ptr(origin)))
    return xyz, colors, labels.astype(np.uint8), origin


def prepare_brats_volume(volumes, seg=None, sub_grid_size=0.01, merge_label_4=True):
    """The arrays convert_pc2ply writes for one case: the full cloud, the sub-cloud and `proj_idx` (index of the nearest
    sub-cloud point for every point of the full cloud).  merge_label_4 applies load_volume's `img[img == 4] = 3`."""
    if seg is not None and merge_label_4:
        seg = np.where(np.asarray(seg) == 4, 3, seg)
    xyz, colors, labels, origin = volume_to_cloud(volumes, seg)
    sub_xyz, sub_colors, sub_labels = DP.grid_sub_sampling(xyz, colors, labels.astype(np.int32), sub_grid_size)
    proj = DP.knn_search(sub_xyz[None], xyz[None], 1)[0, :, 0].astype(np.int32)
    return dict(xyz=xyz, colors=colors, labels=labels, xyz_origin=origin, sub_xyz=sub_xyz, sub_colors=sub_colors,
                sub_labels=np.asarray(sub_labels).reshape(-1).astype(np.uint8), proj_idx=proj)
