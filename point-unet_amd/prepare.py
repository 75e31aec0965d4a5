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


def prepare_brats_volume(volumes, seg=None, sub_grid_size=0.01, merge_label_4=True, chained=True, device=0):
    """The arrays convert_pc2ply writes for one case: the full cloud, the sub-cloud and `proj_idx` (index of the nearest
    sub-cloud point for every point of the full cloud).  merge_label_4 applies load_volume's `img[img == 4] = 3`.
    chained (default): ONE pipeline on the device, as dataPrepareBraTS.py:75-116 is one pipeline on the host -- the volume goes up
    once, ps_volume_to_cloud_dev -> ps_grid_subsample_dev -> ps_knn_batch(device pointers) hand their rows on in HBM, the results come
    down once.  chained=False: the three ops through their host-pointer entry points (three round trips over PCIe; same results, bit for bit)."""
    if seg is not None and merge_label_4:
        seg = np.where(np.asarray(seg) == 4, 3, seg)
    if chained:
        return _prepare_chained(volumes, seg, sub_grid_size, device)
    xyz, colors, labels, origin = volume_to_cloud(volumes, seg)
    sub_xyz, sub_colors, sub_labels = DP.grid_sub_sampling(xyz, colors, labels.astype(np.int32), sub_grid_size)
    proj = DP.knn_search(sub_xyz[None], xyz[None], 1)[0, :, 0].astype(np.int32)
    return dict(xyz=xyz, colors=colors, labels=labels, xyz_origin=origin, sub_xyz=sub_xyz, sub_colors=sub_colors,
                sub_labels=np.asarray(sub_labels).reshape(-1).astype(np.uint8), proj_idx=proj)


def _prepare_chained(volumes, seg, sub_grid_size, device):
    import torch
    vol = np.ascontiguousarray(volumes, dtype=np.float32)
    if vol.ndim != 4 or vol.shape[0] != 4:
        raise ValueError("volumes must have shape [4, X, Y, Z]")
    X, Y, Z = vol.shape[1:]
    dev = torch.device("cuda", device)
    ctx = runtime.default_context(device)
    ctx.use_torch_stream()
    lib, h, p = _lib.lib(), ctx.handle, runtime.ptr
    d_vol = torch.from_numpy(vol).to(dev)
    d_seg = None
    if seg is not None:
        s = np.ascontiguousarray(seg, dtype=np.int32)
        if s.shape != (X, Y, Z):
            raise ValueError("seg must have shape [X, Y, Z]")
        d_seg = torch.from_numpy(s).to(dev)
    nvox = X * Y * Z
    xyz = torch.empty((nvox, 3), dtype=torch.float32, device=dev)
    colors = torch.empty((nvox, 4), dtype=torch.float32, device=dev)
    labels = torch.zeros(nvox, dtype=torch.int32, device=dev)
    origin = torch.empty((nvox, 3), dtype=torch.int32, device=dev)
    n = ctypes.c_int64(nvox)
    _lib.check(lib.ps_volume_to_cloud_dev(h, p(d_vol), p(d_seg), X, Y, Z, ctypes.byref(n), p(xyz), p(colors), p(labels), p(origin)))
    n = int(n.value)
    del d_vol
    if n == 0:
        raise RuntimeError("Error")  # (grid_sub_sampling's message for an empty result, wrapper.cpp:225-229)
    sub_xyz = torch.empty((n, 3), dtype=torch.float32, device=dev)
    sub_colors = torch.empty((n, 4), dtype=torch.float32, device=dev)
    sub_labels = torch.empty(n, dtype=torch.int32, device=dev)
    m = ctypes.c_int64(0)
    _lib.check(lib.ps_grid_subsample_dev(h, p(xyz), n, p(colors), 4, p(labels), 1, float(sub_grid_size), n, ctypes.byref(m), p(sub_xyz), p(sub_colors),
                                         p(sub_labels)))
    m = int(m.value)
    proj = torch.empty((n, 1), dtype=torch.int32, device=dev)
    _lib.check(lib.ps_knn_batch(h, p(sub_xyz), p(xyz), 1, m, n, 3, 1, p(proj), 1))
    torch.cuda.synchronize(dev)
    return dict(xyz=xyz[:n].cpu().numpy(), colors=colors[:n].cpu().numpy(), labels=labels[:n].cpu().numpy().astype(np.uint8),
                xyz_origin=origin[:n].cpu().numpy(), sub_xyz=sub_xyz[:m].cpu().numpy(), sub_colors=sub_colors[:m].cpu().numpy(),
                sub_labels=sub_labels[:m].cpu().numpy().reshape(-1).astype(np.uint8), proj_idx=proj[:, 0].cpu().numpy())
