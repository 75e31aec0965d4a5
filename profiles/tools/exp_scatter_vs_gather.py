"""Level-0 gradient rows ([23 M, 8] fp32 = 32-byte rows): what a permuted WRITE costs against a permuted READ (the fixed-order
gather-reduction reads them in source order today).  torch index ops as the stand-in.  usage (GPU box): python profiles/tools/exp_scatter_vs_gather.py"""
import torch
n = 8 * 180000 * 16
for w in (8, 32):
    rows = n if w == 8 else n // 4
    x = torch.randn(rows, w, device="cuda")
    out = torch.empty_like(x)
    g = torch.Generator(device="cuda").manual_seed(0)
    perm = torch.randperm(rows, device="cuda", generator=g)

    def timed(fn, k=5):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(k):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / k
    t_copy = timed(lambda: out.copy_(x))
    t_gather = timed(lambda: torch.index_select(x, 0, perm, out=out))
    t_scatter = timed(lambda: out.index_copy_(0, perm, x))
    gb = x.numel() * 4 / 1e9
    print("rows of %d floats (%.2f GB): copy %.3f ms, permuted read %.3f ms, permuted write %.3f ms" % (w, gb, t_copy, t_gather, t_scatter), flush=True)
