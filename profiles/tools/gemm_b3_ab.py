"""ps_op_conv1x1_ex: split-bf16 MFMA (gemm_b3.hip) against the fp32 MFMA (rowgemm.hip) per shape.  usage: PS_B3_MINK=64 python profiles/tools/gemm_b3_ab.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from point_unet_amd import _lib, runtime
L, ctx = _lib.lib(), runtime.default_context(0)
h = ctx.handle
p = lambda t: ctypes.c_void_p(t.data_ptr())
for R, K, N in [(1440000, 128, 128), (359936, 256, 256), (359936, 128, 256), (359936, 256, 128), (89984, 512, 512), (89984, 256, 512), (89984, 512, 256)]:
    x = torch.randn(R, K, device="cuda"); W = torch.randn(K, N, device="cuda") / K ** 0.5; y = torch.zeros(R, N, device="cuda")
    out = []
    for acc in (0, 1):
        for on in (1, 0):
            _lib.check(L.ps_set_train_gemm_b3(h, on))
            for _ in range(2):
                _lib.check(L.ps_op_conv1x1_ex(h, p(x), K, p(W), None, R, K, N, 0, acc, p(y), N))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                _lib.check(L.ps_op_conv1x1_ex(h, p(x), K, p(W), None, R, K, N, 0, acc, p(y), N))
            e1.record(); torch.cuda.synchronize()
            out.append(e0.elapsed_time(e1) / 5)
    print("%8d x %4d x %4d   b3 %.3f  fp32 %.3f ms | accumulate: b3 %.3f  fp32 %.3f ms   (%.1f GFLOP)" % (R, K, N, out[0], out[1], out[2], out[3], 2e-9 * R * K * N))
_lib.check(L.ps_set_train_gemm_b3(h, 1))
