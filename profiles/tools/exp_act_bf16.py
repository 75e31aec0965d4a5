"""bfloat16 storage of the LFA rows (ps_set_train_act_bf16): every storage-aware op at the level-0 / level-1 sizes of the batch-8 step, fp32 rows
against bfloat16 rows.  usage (GPU box): python profiles/tools/exp_act_bf16.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from point_unet_amd import _lib, runtime
L, ctx = _lib.lib(), runtime.default_context(0)
hd = ctx.handle
p = lambda t: ctypes.c_void_p(t.data_ptr())
_lib.check(L.ps_set_train_gemm_bf16(hd, 1))


def timed(fn, n=5):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


K = 16
for B, N, h in ((8, 180000, 8), (8, 45000, 32), (8, 11250, 64)):
    R, d = B * N * K, 2 * h
    g = torch.Generator().manual_seed(1)
    xyz = torch.rand(B * N, 3, generator=g).cuda()
    idx = torch.randint(0, N, (B, N, K), generator=g, dtype=torch.int32).cuda()
    W1 = (torch.randn(10, h, generator=g) * 0.5).cuda(); b1 = torch.zeros(h).cuda()
    mean, scale, beta, inv = torch.zeros(h).cuda(), torch.ones(h).cuda(), torch.zeros(h).cuda(), torch.ones(h).cuda()
    W2 = (torch.randn(h, h, generator=g) / h ** 0.5).cuda(); b2 = torch.zeros(h).cuda()
    fsrc = torch.randn(B * N, h, generator=g).cuda(); Wfc = (torch.randn(d, d, generator=g) / d ** 0.5).cuda(); dagg = torch.randn(B * N, d, generator=g).cuda()
    dz = torch.randn(R, h, generator=g).cuda(); dx = torch.empty(R, h).cuda(); rows = torch.empty(R, h).cuda(); dfr = torch.zeros(R, h).cuda()
    agg = torch.empty(B * N, d).cuda(); dW = torch.empty(d, d).cuda(); dw = torch.empty(h, h).cuda(); db = torch.empty(h).cuda()
    s64 = torch.zeros(3 * max(h, 16), dtype=torch.float64).cuda(); s12 = torch.zeros(3 * h).cuda()
    res = {}
    for on in (0, 1):
        _lib.check(L.ps_set_train_act_bf16(hd, on))
        x = torch.empty(R, h, dtype=torch.bfloat16 if on else torch.float32).cuda()
        z = torch.empty(R, h, dtype=torch.bfloat16 if on else torch.float32).cuda()
        t = {}
        t["locse_apply"] = timed(lambda: _lib.check(L.ps_op_locse_train_apply(hd, p(xyz), p(idx), B, N, K, p(W1), p(b1), h, p(mean), p(scale), p(beta), p(x), h)))
        t["cb_sums"] = timed(lambda: _lib.check(L.ps_op_conv_bn_train_sums(hd, p(x), h, p(W2), p(b2), R, h, p(s64))))
        t["cb_apply"] = timed(lambda: _lib.check(L.ps_op_conv_bn_train_apply(hd, p(x), h, p(W2), p(b2), R, h, p(mean), p(scale), p(beta), p(z), h)))
        t["cb_bwd_sums"] = timed(lambda: _lib.check(L.ps_op_conv_bn_train_bwd_sums2(hd, p(x), h, p(W2), p(b2), R, h, p(mean), p(inv), p(scale), p(beta), p(dz), h, p(s12))))
        t["cb_bwd_apply"] = timed(lambda: _lib.check(L.ps_op_conv_bn_train_bwd_apply_w(hd, p(x), h, p(W2), p(b2), R, h, p(mean), p(inv), p(scale), p(beta), p(s12), 1.0 / R, p(dz), h, 0, p(dx), h, p(dw), p(db))))
        t["att_fwd"] = timed(lambda: _lib.check(L.ps_op_att_pool_train_fwd_split(hd, p(fsrc), h, p(idx), B, N, N, p(x), h, p(Wfc), K, d, p(agg))))
        t["att_bwd"] = timed(lambda: _lib.check(L.ps_op_att_pool_train_bwd_split_rows(hd, p(fsrc), h, p(idx), B, N, N, p(x), h, p(Wfc), p(dagg), K, d, p(rows), h, p(dfr), h, p(dW))))
        res[on] = t
        del x, z
    print("h %2d: " % h + "  ".join("%s %.3f->%.3f" % (k, res[0][k], res[1][k]) for k in res[0]), flush=True)
_lib.check(L.ps_set_train_act_bf16(hd, 0)); _lib.check(L.ps_set_train_gemm_bf16(hd, 0))
