"""Output side of the tester (the reference's PointSegment/testBraTS.py:83-101, 226-231 and testPancreas.py:71-85):
class probabilities of the sampled points scattered back into the image volume, on the device."""
import ctypes

import torch

from . import _lib, runtime


def point2prod(logits, p_idx, xyz_origin, volume_shape=(155, 240, 240)):
    """logits [n, C] (CUDA f32), p_idx [n] (int32 rows of the original cloud, or None), xyz_origin [total, 3] int32 voxel
    coordinates (x, y, z) of the original cloud.  Returns the volume f32 [Z, Y, X, C] -- what the reference saves after
    `np.moveaxis(volume, 1, 2)` (testBraTS.py:90), with softmax applied (testBraTS.py:183 `prob_logits`)."""
    logits = logits.contiguous()
    n, C = logits.shape
    xyz = xyz_origin.to(torch.int32).contiguous()
    total = xyz.shape[0]
    Z, X, Y = volume_shape
    vol = torch.empty((Z, Y, X, C), dtype=torch.float32, device=logits.device)
    scratch = torch.empty(total + Z * X * Y, dtype=torch.int32, device=logits.device)
    pi = p_idx.to(torch.int32).contiguous() if p_idx is not None else None
    ctx = runtime.default_context(logits.device.index)
    _lib.check(_lib.lib().ps_op_probs_to_volume(ctx.handle, runtime.ptr(logits), n, C, runtime.ptr(pi), runtime.ptr(xyz), total, Z, X, Y,
                                                runtime.ptr(vol), runtime.ptr(scratch)))
    return vol
