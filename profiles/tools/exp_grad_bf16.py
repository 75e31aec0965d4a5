"""The recompute-form convolution's backward (ps_op_conv_bn_train_bwd_sums2 / _bwd_apply_w) and the LocSE backward with fp32 and with
bfloat16 rows (x, dz, dx: ps_set_train_act_bf16), plain and accumulating, at the row counts of a batch-8 step.  usage (GPU box):
python profiles/tools/exp_grad_bf16.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from point_unet_amd import _lib, runtime
L, ctx = _lib.lib(), runtime.default_context(0)
hd = ctx.handle
p = lambda t: ctypes.c_void_p(t.data_ptr())


def timed(fn, n=5):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


_lib.check(L.ps_set_train_gemm_bf16(hd, 1))
for R, h in [(23040000, 8), (5760000, 32), (1440000, 64)]:
    g = torch.Generator().manual_seed(1)
    W = (torch.randn(h, h, generator=g) / h ** 0.5).cuda(); b = (0.1 * torch.randn(h, generator=g)).cuda()
    m, inv = (0.1 * torch.randn(h, generator=g)).cuda(), (1 + 0.2 * torch.rand(h, generator=g)).cuda()
    sc, be = (inv * 1.1).contiguous(), (0.1 * torch.randn(h, generator=g)).cuda()
    x32 = torch.randn(R, h, device="cuda"); dz32 = torch.randn(R, h, device="cuda")
    for on in (0, 1):
        _lib.check(L.ps_set_train_act_bf16(hd, on))
        x = x32.bfloat16() if on else x32
        dz = dz32.bfloat16() if on else dz32
        dx = torch.zeros(R, h, device="cuda", dtype=torch.bfloat16 if on else torch.float32)
        s12 = torch.zeros(3 * h, device="cuda"); dw = torch.empty(h, h, device="cuda"); db = torch.empty(h, device="cuda")
        t_s = timed(lambda: _lib.check(L.ps_op_conv_bn_train_bwd_sums2(hd, p(x), h, p(W), p(b), R, h, p(m), p(inv), p(sc), p(be), p(dz), h, p(s12))))
        ts = []
        for acc in (0, 1):
            ts.append(timed(lambda: _lib.check(L.ps_op_conv_bn_train_bwd_apply_w(hd, p(x), h, p(W), p(b), R, h, p(m), p(inv), p(sc), p(be), p(s12), 1.0 / R, p(dz), h, acc,
                                                                                 p(dx), h, p(dw), p(db)))))
        print("rows %9d h %2d %s: bwd sums %.3f ms, bwd apply %.3f ms, accumulating %.3f ms" % (R, h, "bf16 rows" if on else "fp32 rows", t_s, ts[0], ts[1]), flush=True)
    del x32, dz32
_lib.check(L.ps_set_train_act_bf16(hd, 0))
_lib.check(L.ps_set_train_gemm_bf16(hd, 0))
