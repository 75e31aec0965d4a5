"""Level-1 attentive pooling of the training step (d = 64, split-source form, 8 x 45 000 points): attpool_gemm.hip's attg64_kernel against
attpool_train.hip's per-point kernels.  Run once per setting: PS_ATT64_GEMM=0|1, PS_ATT64_OCC=1|2 (read once per process).
usage (GPU box): PS_ATT64_GEMM=1 python profiles/tools/exp_att64.py [bf16]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from point_unet_amd import _lib, runtime
L, ctx = _lib.lib(), runtime.default_context(0)
h = ctx.handle
p = lambda t: ctypes.c_void_p(t.data_ptr())
K, d, hh = 16, 64, 32
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


for B, N in [(8, 45000), (1, 45000)]:
    g = torch.Generator().manual_seed(1)
    fl = torch.randn(B * N, hh, generator=g).cuda(); fr = torch.randn(B * N * K, hh, generator=g).cuda()
    idx = torch.randint(0, N, (B, N, K), generator=g, dtype=torch.int32).cuda()
    W = (torch.randn(d, d, generator=g) / 8).cuda(); dagg = torch.randn(B * N, d, generator=g).cuda()
    agg = torch.empty(B * N, d).cuda(); rows = torch.empty(B * N * K, hh).cuda(); dfr = torch.zeros(B * N * K, hh).cuda(); dW = torch.empty(d, d).cuda()
    tf = timed(lambda: _lib.check(L.ps_op_att_pool_train_fwd_split(h, p(fl), hh, p(idx), B, N, N, p(fr), hh, p(W), K, d, p(agg))))
    tb = timed(lambda: _lib.check(L.ps_op_att_pool_train_bwd_split_rows(h, p(fl), hh, p(idx), B, N, N, p(fr), hh, p(W), p(dagg), K, d, p(rows), hh, p(dfr), hh, p(dW))))
    gf = 2e-9 * B * N * K * d * d
    print("%s GEMM=%s OCC=%s  B %d: fwd %.3f ms (%.0f TF/s)  bwd %.3f ms (%.0f TF/s)" % ("bf16" if bf16 else "fp32", os.environ.get("PS_ATT64_GEMM", "1"),
          os.environ.get("PS_ATT64_OCC", "2"), B, tf, gf / tf, tb, 3 * gf / tb), flush=True)
_lib.check(L.ps_set_train_gemm_bf16(h, 0))
