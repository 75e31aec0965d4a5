"""One GEMM shape through ps_op_conv1x1_ex, N times (for rocprofv3 --pmc passes).  usage: python3 profiles/tools/gemm_one.py R K N [b3 0|1] [reps]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from point_unet_amd import _lib, runtime
L, ctx = _lib.lib(), runtime.default_context(0)
h = ctx.handle
R, K, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
b3 = int(sys.argv[4]) if len(sys.argv) > 4 else 1
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
p = lambda t: ctypes.c_void_p(t.data_ptr())
x = torch.randn(R, K, device="cuda"); W = torch.randn(K, N, device="cuda") / K ** 0.5; y = torch.zeros(R, N, device="cuda")
_lib.check(L.ps_set_train_gemm_b3(h, b3))
for _ in range(reps):
    _lib.check(L.ps_op_conv1x1_ex(h, p(x), K, p(W), None, R, K, N, 0, 0, p(y), N))
torch.cuda.synchronize()
