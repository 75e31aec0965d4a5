"""The per-cloud index pyramid: device counterpart of tf_map (PointSegment/runBraTS.py:140-161,
runPancreas.py:124-145).  One call builds, for every layer, the K-NN table, the prefix "random" subsample, the
pooling table and the 1-NN up-sampling table -- entirely on the GPU (kd-tree build, search, slicing)."""
import ctypes

import torch

from . import _lib, runtime


class Pyramid:
    """Holds the device tensors and the ps_pyramid struct that points at them."""

    def __init__(self, xyz, neigh_idx, sub_idx, interp_idx, K, order=None):
        self.xyz, self.neigh_idx, self.sub_idx, self.interp_idx = xyz, neigh_idx, sub_idx, interp_idx
        self.order = order  # optional per-layer int32 [B, n_l]: kd-tree leaf order of the layer's points (ps_pyramid.order)
        L = len(xyz)
        s = _lib.PsPyramid()
        s.num_layers = L
        s.K = K
        s.B = xyz[0].shape[0]
        for i in range(L):
            s.n[i] = xyz[i].shape[1]
            s.xyz[i] = xyz[i].data_ptr()
            s.neigh_idx[i] = neigh_idx[i].data_ptr()
            s.sub_idx[i] = sub_idx[i].data_ptr()
            s.interp_idx[i] = interp_idx[i].data_ptr()
            s.order[i] = order[i].data_ptr() if order is not None else None
        s.n[L] = sub_idx[L - 1].shape[1]
        self.struct = s

    def invalidate(self):
        """Drop the stamp ps_pyramid_build left (ps_pyramid.built): call after editing any of the exposed tensors in place -- the training step
        then re-checks the tables (prefix property) and stops trusting `order` as permutations instead of taking the build's word for them."""
        self.struct.built = 0

    def flat_inputs(self):
        """The first 4*num_layers entries of the reference's flat input list (RandLANet.py:33-36)."""
        return list(self.xyz) + list(self.neigh_idx) + list(self.sub_idx) + list(self.interp_idx)


def alloc_pyramid(B, n0, ratios, K, device):
    n = [int(n0)]
    for r in ratios:
        n.append(n[-1] // int(r))
    L = len(ratios)
    xyz = [torch.empty((B, n[i], 3), dtype=torch.float32, device=device) for i in range(L)]
    nbr = [torch.empty((B, n[i], K), dtype=torch.int32, device=device) for i in range(L)]
    sub = [torch.empty((B, n[i + 1], K), dtype=torch.int32, device=device) for i in range(L)]
    up = [torch.empty((B, n[i], 1), dtype=torch.int32, device=device) for i in range(L)]
    order = [torch.empty((B, n[i]), dtype=torch.int32, device=device) for i in range(L)]
    return Pyramid(xyz, nbr, sub, up, K, order)


def build_pyramid(batch_xyz, cfg, ctx=None, out=None):
    """batch_xyz: float32 CUDA tensor [B, N0, 3] (pre-shuffled clouds, as the reference's generator yields them,
    runBraTS.py:114).  Returns a Pyramid; pass a previous one as `out` to reuse its buffers."""
    assert batch_xyz.is_cuda and batch_xyz.dtype == torch.float32 and batch_xyz.dim() == 3 and batch_xyz.shape[2] == 3
    batch_xyz = batch_xyz.contiguous()
    B, n0 = batch_xyz.shape[0], batch_xyz.shape[1]
    ratios = list(cfg.sub_sampling_ratio)[:cfg.num_layers]
    ctx = ctx or runtime.default_context(batch_xyz.device.index)
    pyr = out or alloc_pyramid(B, n0, ratios, cfg.k_n, batch_xyz.device)
    r = (ctypes.c_int32 * len(ratios))(*ratios)
    _lib.check(_lib.lib().ps_pyramid_build(ctx.handle, runtime.ptr(batch_xyz), B, n0, len(ratios), r, cfg.k_n,
                                           ctypes.byref(pyr.struct)))
    return pyr
