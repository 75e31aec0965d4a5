"""attpool_gemm.hip (scores in registers) against the op-by-op path it replaces, at the wide levels of the batch-8 / batch-1 training
step: forward (score GEMM + softmax-pool  vs  ps_op_att_pool_gemm_fwd) and backward (softmax-pool backward + input-gradient GEMM + weight
gradient  vs  ps_op_att_pool_gemm_bwd + weight gradient), fp32 and bf16-MLP mode.  usage (GPU box): python profiles/tools/exp_attg.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from point_unet_amd import _lib, runtime
L, ctx = _lib.lib(), runtime.default_context(0)
h = ctx.handle
p = lambda t: ctypes.c_void_p(t.data_ptr())
K = 16


def timed(fn, n=5):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for mode in (0, 1):
    _lib.check(L.ps_set_train_gemm_bf16(h, mode))
    for R, d in [(90000, 128), (22496, 256), (11250, 128), (2812, 256), (5624, 512)]:
        RK = R * K
        F = torch.randn(RK, d, device="cuda"); W = torch.randn(d, d, device="cuda") / d ** 0.5
        S = torch.empty(RK, d, device="cuda"); agg = torch.empty(R, d, device="cuda"); dagg = torch.randn(R, d, device="cuda")
        dF = torch.empty(RK, d, device="cuda"); dS = torch.empty(RK, d, device="cuda"); dW = torch.empty(d, d, device="cuda")

        def old_fwd():
            _lib.check(L.ps_op_conv1x1_ex(h, p(F), d, p(W), None, RK, d, d, 0, 0, p(S), d))
            _lib.check(L.ps_op_softmax_pool_fwd(h, p(F), p(S), R, K, d, None, p(agg)))

        def old_bwd():
            _lib.check(L.ps_op_softmax_pool_bwd_scores(h, p(dagg), p(F), p(S), R, K, d, p(dF), p(dS)))
            _lib.check(L.ps_op_conv1x1_ex(h, p(dS), d, p(W.t().contiguous()), None, RK, d, d, 0, 1, p(dF), d))

        def wgrad():
            _lib.check(L.ps_op_linear_wgrad_ex(h, p(F), d, p(dS), d, RK, d, d, p(dW), None))

        def new_fwd():
            _lib.check(L.ps_op_att_pool_gemm_fwd(h, p(F), d, p(W), R, K, d, p(agg)))

        def new_bwd():
            _lib.check(L.ps_op_att_pool_gemm_bwd(h, p(F), d, p(W), p(dagg), R, K, d, p(dF), d, 0, p(dS), d))

        t_of, t_ob, t_w = timed(old_fwd), timed(old_bwd), timed(wgrad)
        t_nf = timed(new_fwd)
        t_nb = timed(new_bwd) if L.ps_op_att_pool_gemm_supported(K, d) else float("nan")
        gf = 2e-9 * RK * d * d
        print("%s R %6d d %3d: fwd old %.3f new %.3f ms (%.0f TF/s) | bwd old %.3f new %.3f ms (%.0f TF/s) | wgrad %.3f ms" % (
            "bf16" if mode else "fp32", R, d, t_of, t_nf, gf / t_nf, t_ob, t_nb, 2 * gf / t_nb, t_w), flush=True)
        del F, S, dF, dS
_lib.check(L.ps_set_train_gemm_bf16(h, 0))
