"""Wide-level fused attentive pooling: split-source forms against the materialised ones, per kernel, on a REAL neighbour table (a BraTS-shaped
cloud's level-2 / level-3 K-NN indices, 8 clouds).  usage (GPU box): python profiles/tools/exp_attg_split.py [bf16]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import bench
from point_unet_amd import _lib, runtime
from point_unet_amd.helper_tool import ConfigBraTS as cfg
from point_unet_amd.pyramid import build_pyramid
L, ctx = _lib.lib(), runtime.default_context(0)
h = ctx.handle
p = lambda t: ctypes.c_void_p(t.data_ptr())
K = 16
bf16 = len(sys.argv) > 1 and sys.argv[1] == "bf16"
_lib.check(L.ps_set_train_gemm_bf16(h, 1 if bf16 else 0))


def timed(fn, n=5):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


B = 8
xyz = np.stack([bench.brats_cloud(180000, 1000 + b) for b in range(B)])
pyr = build_pyramid(torch.from_numpy(xyz).cuda(), cfg)
for lvl, d in ((2, 128), (3, 256)):
    idx = pyr.neigh_idx[lvl]
    N = idx.shape[1]
    hh = d // 2
    g = torch.Generator().manual_seed(1)
    fsrc = torch.randn(B * N, hh, generator=g).cuda(); fx = torch.randn(B * N * K, hh, generator=g).cuda()
    W = (torch.randn(d, d, generator=g) / d ** 0.5).cuda(); dagg = torch.randn(B * N, d, generator=g).cuda()
    cat = torch.empty(B * N * K, d).cuda(); agg = torch.empty(B * N, d).cuda()
    dcat = torch.empty(B * N * K, d).cuda(); dS = torch.empty(B * N * K, d).cuda(); dW = torch.empty(d, d).cuda()
    rows = torch.empty(B * N * K, hh).cuda(); dfx = torch.zeros(B * N * K, hh).cuda()
    t = {}
    t["gather"] = timed(lambda: _lib.check(L.ps_op_gather_neighbour_ex(h, p(fsrc), p(idx), B, N, N, K, hh, p(cat), d)))
    cat[:, hh:] = fx
    t["fwd mat"] = timed(lambda: _lib.check(L.ps_op_att_pool_gemm_fwd(h, p(cat), d, p(W), B * N, K, d, p(agg))))
    t["fwd split"] = timed(lambda: _lib.check(L.ps_op_att_pool_gemm_fwd_split(h, p(fsrc), hh, p(idx), B, N, N, p(fx), hh, p(W), K, d, p(agg))))
    t["bwd mat"] = timed(lambda: _lib.check(L.ps_op_att_pool_gemm_bwd(h, p(cat), d, p(W), p(dagg), B * N, K, d, p(dcat), d, 0, p(dS), d)))
    t["bwd split"] = timed(lambda: _lib.check(L.ps_op_att_pool_gemm_bwd_split(h, p(fsrc), hh, p(idx), B, N, N, p(fx), hh, p(W), p(dagg), K, d, p(rows), hh, p(dfx), hh, 1, p(dS), d)))
    t["wgrad mat"] = timed(lambda: _lib.check(L.ps_op_linear_wgrad_ex(h, p(cat), d, p(dS), d, B * N * K, d, d, p(dW), None)))
    t["wgrad split"] = timed(lambda: _lib.check(L.ps_op_linear_wgrad_split(h, p(fsrc), hh, p(idx), B, N, N, K, p(fx), hh, p(dS), d, d, d, p(dW))))
    print(("bf16" if bf16 else "fp32") + " level %d d %d:" % (lvl, d), "  ".join("%s %.3f" % kv for kv in t.items()), flush=True)
    del cat, dcat, dS, rows, dfx, fx
_lib.check(L.ps_set_train_gemm_bf16(h, 0))
