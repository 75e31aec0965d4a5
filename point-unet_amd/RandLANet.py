"""Host-side mirror of PointSegment/RandLANet.py's Network for the forward path, executing on MI355X through
libpointseg_hip.so.

`Network.inference` is the fused production path (one C-ABI call, csrc/randla.hip).  The individual block
methods keep the reference's names, argument meaning and tensor shapes (with the dummy axis of the reference's
[B,N,1,C] layout preserved) so that call sites written against RandLANet.py:314-401 keep working; they run the
op-by-op kernels of csrc/ops.hip.  All tensors are float32 / int32 CUDA torch tensors.
"""
import ctypes

import numpy as np
import torch

from . import _lib, runtime, weights
from .pyramid import Pyramid


def _ctx(t):
    c = runtime.default_context(t.device.index)
    return c


class Network:
    def __init__(self, config, params=None, device=0, seed=0, ctx=None):
        """config: ConfigBraTS-like (k_n, num_layers, d_out, num_classes, in_channels).
        params: dict of TF-named arrays (weights.init_params / a converted checkpoint); random init if None."""
        self.config = config
        self.device = torch.device("cuda", device)
        self.ctx = ctx or runtime.default_context(device)
        self.params = params if params is not None else weights.init_params(config, seed=seed)
        cfg = _lib.PsRandlaConfig()
        cfg.num_layers = config.num_layers
        cfg.k_n = config.k_n
        cfg.num_classes = config.num_classes
        cfg.in_channels = config.in_channels
        for i in range(config.num_layers):
            cfg.d_out[i] = config.d_out[i]
        self._h = ctypes.c_void_p()
        _lib.check(_lib.lib().ps_randla_create(self.ctx.handle, ctypes.byref(cfg), ctypes.byref(self._h)))
        self.set_params(self.params)

    def close(self):
        """Frees the packed weights and chain caches on the device (ps_randla_destroy); the context stays open."""
        h, self._h = getattr(self, "_h", None), None
        if h:
            _lib.lib().ps_randla_destroy(h)

    def __del__(self):
        try:
            if getattr(self.ctx, "handle", None):  # a closed context has already released everything it owned
                self.close()
        except Exception:
            pass

    def set_params(self, params):
        self.params = params
        self.__dict__.pop("_folded", None)
        blob = weights.fold_to_blob(self.config, params)
        want = _lib.lib().ps_randla_weight_count(self._h)
        assert blob.size == want, (blob.size, want)
        _lib.check(_lib.lib().ps_randla_set_weights(self._h, runtime.ptr(blob), blob.size))

    # ------------------------------------------------------------------------------------------------
    def inference(self, inputs, is_training=False):
        """inputs: dict with 'features' [B,N0,Cin] and either 'pyramid' (a pyramid.Pyramid) or the reference's
        'xyz','neigh_idx','sub_idx','interp_idx' lists (RandLANet.py:33-36); optional 'out': the logits buffer to fill.
        Returns logits [B,N0,num_classes]."""
        if is_training:
            raise NotImplementedError("training-mode forward (batch-statistics BN, dropout) lives in point_unet_amd.train.Trainer")
        pyr = inputs.get("pyramid")
        if pyr is None:
            pyr = Pyramid([t.contiguous() for t in inputs["xyz"]], [t.contiguous() for t in inputs["neigh_idx"]],
                          [t.contiguous() for t in inputs["sub_idx"]], [t.contiguous() for t in inputs["interp_idx"]],
                          self.config.k_n)
        feats = inputs["features"].contiguous()
        if feats.dtype == torch.float16:  # half-precision feature files (BASELINE configs[4]): widened on the device
            wide = torch.empty(feats.shape, dtype=torch.float32, device=feats.device)
            _lib.check(_lib.lib().ps_op_half_to_float(self.ctx.handle, runtime.ptr(feats), feats.numel(), runtime.ptr(wide)))
            feats = wide
        B, n0 = feats.shape[0], feats.shape[1]
        logits = inputs.get("out")  # optional: a caller-owned float32 [B,N0,num_classes] buffer (ForwardPipeline's coalesced pairs)
        if logits is None:
            logits = torch.empty((B, n0, self.config.num_classes), dtype=torch.float32, device=feats.device)
        else:
            assert logits.is_contiguous() and logits.dtype == torch.float32 and tuple(logits.shape) == (B, n0, self.config.num_classes)
        _lib.check(_lib.lib().ps_randla_forward(self._h, ctypes.byref(pyr.struct), runtime.ptr(feats), runtime.ptr(logits)))
        return logits

    def keep_taps(self, on=True):
        """Forwards also store the activations only tap() reads (the last decoder step's rows); off by default."""
        _lib.check(_lib.lib().ps_randla_keep_taps(self._h, 1 if on else 0))

    def tap(self, which, shape):
        """Copy an internal activation of the last forward to the host (parity tests)."""
        out = np.empty(shape, np.float32)
        _lib.check(_lib.lib().ps_randla_tap(self._h, int(which), runtime.ptr(out), out.size))
        return out

    # ---- op-by-op surface (RandLANet.py:337-401) ---------------------------------------------------------
    @staticmethod
    def gather_neighbour(pc, neighbor_idx):
        """pc [B,N,d], neighbor_idx [B,N',K] -> [B,N',K,d]   (RandLANet.py:377-386)"""
        pc, idx = pc.contiguous(), neighbor_idx.contiguous()
        B, N, d = pc.shape
        M, K = idx.shape[1], idx.shape[2]
        out = torch.empty((B, M, K, d), dtype=torch.float32, device=pc.device)
        _lib.check(_lib.lib().ps_op_gather_neighbour(_ctx(pc).handle, runtime.ptr(pc), runtime.ptr(idx), B, N, M, K, d, runtime.ptr(out)))
        return out

    @staticmethod
    def relative_pos_encoding(xyz, neigh_idx):
        """xyz [B,N,3], neigh_idx [B,N,K] -> [B,N,K,10]   (RandLANet.py:337-343)"""
        xyz, idx = xyz.contiguous(), neigh_idx.contiguous()
        B, N, K = idx.shape
        out = torch.empty((B, N, K, 10), dtype=torch.float32, device=xyz.device)
        _lib.check(_lib.lib().ps_op_relative_pos_encoding(_ctx(xyz).handle, runtime.ptr(xyz), runtime.ptr(idx), B, N, K, runtime.ptr(out)))
        return out

    @staticmethod
    def random_sample(feature, pool_idx):
        """feature [B,N,1,d], pool_idx [B,N',K] -> [B,N',1,d]   (RandLANet.py:345-360)"""
        f = feature.squeeze(2).contiguous()
        idx = pool_idx.contiguous()
        B, N, d = f.shape
        M, K = idx.shape[1], idx.shape[2]
        out = torch.empty((B, M, d), dtype=torch.float32, device=f.device)
        _lib.check(_lib.lib().ps_op_random_sample(_ctx(f).handle, runtime.ptr(f), runtime.ptr(idx), B, N, M, K, d, runtime.ptr(out)))
        return out.unsqueeze(2)

    @staticmethod
    def nearest_interpolation(feature, interp_idx):
        """feature [B,N,1,d], interp_idx [B,up,1] -> [B,up,1,d]   (RandLANet.py:362-375)"""
        f = feature.squeeze(2).contiguous()
        idx = interp_idx.contiguous()
        B, N, d = f.shape
        M = idx.shape[1]
        out = torch.empty((B, M, d), dtype=torch.float32, device=f.device)
        _lib.check(_lib.lib().ps_op_nearest_interpolation(_ctx(f).handle, runtime.ptr(f), runtime.ptr(idx), B, N, M, d, runtime.ptr(out)))
        return out.unsqueeze(2)

    @staticmethod
    def conv2d(inputs, w, b, leaky=True):
        """helper_tf_util.conv2d with a 1x1 kernel and BN already folded into (w [Cin,Cout], b [Cout])."""
        x = inputs.contiguous()
        cin, cout = w.shape
        R = x.numel() // cin
        out = torch.empty(tuple(x.shape[:-1]) + (cout,), dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().ps_op_conv1x1(_ctx(x).handle, runtime.ptr(x), runtime.ptr(w.contiguous()), runtime.ptr(b.contiguous()),
                                            R, cin, cout, 1 if leaky else 0, runtime.ptr(out)))
        return out

    @staticmethod
    def att_pooling(feature_set, wfc, w_mlp, b_mlp):
        """feature_set [B,N,K,d]; wfc [d,d]; trailing conv2d (w_mlp [d,d_out], b_mlp) -> [B,N,1,d_out]
        (RandLANet.py:388-401)."""
        f = feature_set.contiguous()
        B, N, K, d = f.shape
        agg = torch.empty((B, N, d), dtype=torch.float32, device=f.device)
        _lib.check(_lib.lib().ps_op_att_pool(_ctx(f).handle, runtime.ptr(f), runtime.ptr(wfc.contiguous()), B * N, K, d, runtime.ptr(agg)))
        return Network.conv2d(agg.unsqueeze(2), w_mlp, b_mlp, leaky=True)

    # ---- the reference's block methods, composed from the ops above with this network's (BN-folded) parameters ----------
    def _layer(self, scope):
        """(w [Cin,Cout], b [Cout]) of a conv2d scope with its BatchNorm folded in (inference mode), cached on the device."""
        cache = self.__dict__.setdefault("_folded", {})
        if scope not in cache:
            p = self.params
            w, b = weights._fold(p[scope + "/weights"], p[scope + "/biases"], p, scope + "/batch_normalization")
            cache[scope] = (torch.from_numpy(w).to(self.device), torch.from_numpy(b).to(self.device))
        return cache[scope]

    def _att(self, feature_set, name):
        wfc = self.__dict__.setdefault("_folded", {}).get(name + "fc")
        if wfc is None:
            wfc = torch.from_numpy(np.ascontiguousarray(self.params[name + "fc/kernel"])).to(self.device)
            self._folded[name + "fc"] = wfc
        w, b = self._layer(name + "mlp")
        return Network.att_pooling(feature_set, wfc, w, b)

    def building_block(self, xyz, feature, neigh_idx, d_out, name, is_training=False):
        """Local feature aggregation: xyz [B,N,3], feature [B,N,1,d_out/2], neigh_idx [B,N,K] -> [B,N,1,d_out]
        (RandLANet.py:323-335); `name` is the scope prefix, e.g. 'Encoder_layer_0LFA'."""
        assert not is_training, "op-by-op blocks run in inference mode (training: point_unet_amd.train.Trainer)"
        f_xyz = Network.relative_pos_encoding(xyz, neigh_idx)
        f_xyz = Network.conv2d(f_xyz, *self._layer(name + "mlp1"))
        f_nb = Network.gather_neighbour(feature.squeeze(2), neigh_idx)
        f_agg = self._att(torch.cat([f_nb, f_xyz], dim=-1), name + "att_pooling_1")
        f_xyz = Network.conv2d(f_xyz, *self._layer(name + "mlp2"))
        f_nb = Network.gather_neighbour(f_agg.squeeze(2), neigh_idx)
        return self._att(torch.cat([f_nb, f_xyz], dim=-1), name + "att_pooling_2")

    def dilated_res_block(self, feature, xyz, neigh_idx, d_out, name, is_training=False):
        """feature [B,N,1,d_in] -> [B,N,1,2*d_out]: mlp1, building_block, mlp2 (no activation), + shortcut, LeakyReLU(0.2)
        (RandLANet.py:314-321); `name` e.g. 'Encoder_layer_0'."""
        assert not is_training, "op-by-op blocks run in inference mode (training: point_unet_amd.train.Trainer)"
        f_pc = Network.conv2d(feature, *self._layer(name + "mlp1"))
        f_pc = self.building_block(xyz, f_pc, neigh_idx, d_out, name + "LFA")
        f_pc = Network.conv2d(f_pc, *self._layer(name + "mlp2"), leaky=False)
        shortcut = Network.conv2d(feature, *self._layer(name + "shortcut"), leaky=False)
        out = torch.empty_like(f_pc)
        _lib.check(_lib.lib().ps_op_add_lrelu(_ctx(f_pc).handle, runtime.ptr(f_pc.contiguous()), runtime.ptr(shortcut.contiguous()), f_pc.numel(),
                                              runtime.ptr(out)))
        return out
