"""How fast are the BatchNorm passes of the training step on their own?  Times ps_op_bn_train_fwd_mov (statistics + normalise) and
ps_op_bn_train_bwd_ex (sums + apply) on the [rows, C] shapes of a batch of 8 x 180 000 points, against torch elementwise kernels of the
same traffic (x * a + b -> y: 1 read + 1 write; x * y + z -> out: 2 reads + 1 write) on the same buffers.  hipEvent timing, 20 repeats,
buffers rotated over 4 copies so that a 184 MB tensor is not simply served from the 256 MB Infinity Cache.
usage (GPU box): python profiles/tools/bn_bw.py"""
import ctypes, os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import torch
from point_unet_amd import runtime, _lib

ctx = runtime.default_context(0)
L = _lib.lib()
vp = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731


def timeit(fn, n=20):
    for _ in range(3):
        fn(0)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n):
        fn(i)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3  # us


for R, C in ((1440000, 32), (1440000, 16), (360000, 128), (360000, 64), (90000, 256), (22496, 512)):
    xs = [torch.randn(R, C, device="cuda") for _ in range(4)]
    dys = [torch.randn(R, C, device="cuda") for _ in range(4)]
    outs = [torch.empty(R, C, device="cuda") for _ in range(4)]
    gamma, beta = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    mean, invstd, var, scr = (torch.zeros(k * C, device="cuda") for k in (1, 1, 1, 2))
    mm, mv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    ctx.set_stream(torch.cuda.current_stream())
    mb = R * C * 4 / 1e6
    t_f = timeit(lambda i: L.ps_op_bn_train_fwd_mov(ctx.handle, vp(xs[i % 4]), vp(gamma), vp(beta), R, C, ctypes.c_float(1e-5), 1, vp(outs[i % 4]), C,
                                                     vp(mean), vp(invstd), vp(var), vp(scr), vp(mm), vp(mv), ctypes.c_float(0.99)))
    t_b = timeit(lambda i: L.ps_op_bn_train_bwd_ex(ctx.handle, vp(dys[i % 4]), C, vp(xs[i % 4]), vp(gamma), vp(beta), vp(mean), vp(invstd), R, C, 1,
                                                    vp(outs[i % 4]), vp(dg), vp(db)))
    t_1 = timeit(lambda i: torch.add(xs[i % 4], 1.0, out=outs[i % 4]))
    t_2 = timeit(lambda i: torch.addcmul(xs[i % 4], xs[(i + 1) % 4], dys[i % 4], out=outs[i % 4]))
    t_r = timeit(lambda i: xs[i % 4].sum())
    print("[%8d, %3d] %6.1f MB  fwd (3 passes) %7.1f us = %5.2f TB/s | bwd (5 passes) %7.1f us = %5.2f TB/s | torch 1R1W %6.1f us = %5.2f TB/s, 3R1W %6.1f us = %5.2f TB/s, "
          "sum 1R %6.1f us = %5.2f TB/s" % (R, C, mb, t_f, 3 * mb / t_f, t_b, 5 * mb / t_b, t_1, 2 * mb / t_1, t_2, 4 * mb / t_2, t_r, mb / t_r), flush=True)
