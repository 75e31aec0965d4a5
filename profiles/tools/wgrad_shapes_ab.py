"""ps_op_linear_wgrad_ex per shape, split-bf16 MFMA (gemm_b3.hip: wgrad_b3_kernel) on / off.  usage (GPU box): python profiles/tools/wgrad_shapes_ab.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from point_unet_amd import _lib, runtime
L, ctx = _lib.lib(), runtime.default_context(0)
h = ctx.handle
p = lambda t: ctypes.c_void_p(t.data_ptr())
for R, K, N in [(1440000, 128, 128), (359936, 128, 256), (359936, 256, 256), (89984, 256, 512), (90000, 128, 256), (22496, 256, 512), (22496, 768, 256), (90000, 384, 128)]:
    x = torch.randn(R, K, device="cuda"); dy = torch.randn(R, N, device="cuda"); gW = torch.zeros(K, N, device="cuda"); gb = torch.zeros(N, device="cuda")
    out = []
    for on in (1, 0):
        _lib.check(L.ps_set_train_gemm_b3(h, on))
        for _ in range(2):
            _lib.check(L.ps_op_linear_wgrad_ex(h, p(x), K, p(dy), N, R, K, N, p(gW), p(gb)))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            _lib.check(L.ps_op_linear_wgrad_ex(h, p(x), K, p(dy), N, R, K, N, p(gW), p(gb)))
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 5)
    gf, gb_ = 2e-9 * R * K * N, 4e-9 * R * (K + N)
    print("%9d x %4d x %4d   b3 %.3f ms  fp32 %.3f ms   %6.1f GFLOP %5.2f GB -> %6.1f TF/s %5.0f GB/s (incl. the slab reduction)" % (R, K, N, out[0], out[1], gf, gb_, gf / out[0], gb_ / out[0] * 1e3))
    del x, dy
_lib.check(L.ps_set_train_gemm_b3(h, 1))
