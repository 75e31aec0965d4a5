"""Every GEMM shape of the batch-8 training step through ps_op_conv1x1_ex, split-bf16 MFMA (gemm_b3.hip) on / off: ms, TFLOP/s and
GB/s of the rows moved.  usage (GPU box): python profiles/tools/gemm_shapes_ab.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from point_unet_amd import _lib, runtime
L, ctx = _lib.lib(), runtime.default_context(0)
h = ctx.handle
p = lambda t: ctypes.c_void_p(t.data_ptr())
shapes = [(23040000, 8, 8), (5760000, 32, 32), (1440000, 128, 128), (1440000, 64, 64), (1440000, 32, 64), (1440000, 64, 32), (1440000, 8, 32),
          (359936, 128, 256), (359936, 256, 128), (359936, 128, 128), (360000, 160, 32), (360000, 32, 160), (360000, 128, 64), (360000, 256, 128),
          (89984, 256, 512), (89984, 512, 256), (90000, 384, 128), (90000, 128, 384), (90000, 256, 512), (22496, 768, 256), (22496, 256, 768),
          (22496, 512, 1024), (22496, 256, 512), (5624, 1536, 512), (5624, 512, 1536), (5624, 512, 256), (5624, 512, 1024), (5624, 1024, 512),
          (2808, 1024, 1024)]
for R, K, N in shapes:
    x = torch.randn(R, K, device="cuda"); W = torch.randn(K, N, device="cuda") / K ** 0.5; y = torch.zeros(R, N, device="cuda")
    out = []
    for on in (1, 0):
        _lib.check(L.ps_set_train_gemm_b3(h, on))
        for _ in range(2):
            _lib.check(L.ps_op_conv1x1_ex(h, p(x), K, p(W), None, R, K, N, 0, 0, p(y), N))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            _lib.check(L.ps_op_conv1x1_ex(h, p(x), K, p(W), None, R, K, N, 0, 0, p(y), N))
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 5)
    gf, gb = 2e-9 * R * K * N, 4e-9 * R * (K + N)
    print("%9d x %4d x %4d   b3-allowed %.3f ms  fp32-only %.3f ms   %7.1f GFLOP %6.2f GB -> %6.1f TF/s %6.0f GB/s" % (
        R, K, N, out[0], out[1], gf, gb, gf / min(out) , gb / min(out) * 1e3))
    del x, W, y
_lib.check(L.ps_set_train_gemm_b3(h, 1))
