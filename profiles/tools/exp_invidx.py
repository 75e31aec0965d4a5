"""ps_op_inverse_index on the tables of a batch-8 training step (neigh levels 0-4, interp 0-4): per-table device time of the bucket form
against the radix-sort / fill-and-sort forms (PS_INV_BUCKET=0 in a child process).  usage (GPU box): python profiles/tools/exp_invidx.py"""
import ctypes, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    import torch
    from point_unet_amd import _lib, runtime
    L, ctx = _lib.lib(), runtime.default_context(0)
    h = ctx.handle
    p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    g = torch.Generator().manual_seed(1)
    B, K = 8, 16
    ns = [180000, 45000, 11250, 2812, 703, 351]
    tables = [("neigh%d" % i, ns[i], ns[i] * K) for i in range(5)] + [("interp%d" % i, ns[i + 1], ns[i]) for i in range(5)]
    tot = 0.0
    for name, N, rpc in tables:
        idx = torch.randint(0, N, (B, rpc), generator=g).int().cuda()
        off = torch.empty(B * N + 1, dtype=torch.int32, device="cuda")
        src = torch.empty(B * rpc, dtype=torch.int32, device="cuda")
        ws = torch.empty(int(L.ps_op_inverse_index_workspace(B * N, B * rpc)), dtype=torch.int32, device="cuda")
        run = lambda: _lib.check(L.ps_op_inverse_index(h, p(idx), B, N, rpc, p(off), p(src), p(ws)))  # noqa: E731
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        tot += ms
        print("  %-8s N %7d rows/cloud %8d: %.4f ms" % (name, N, rpc, ms), flush=True)
    print("  total %.4f ms per step" % tot, flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        main()
    else:
        for env in ({"PS_INV_BUCKET": "1"}, {"PS_INV_BUCKET": "1", "PS_INV_TILE": "6144"}, {"PS_INV_BUCKET": "1", "PS_INV_TILE": "4096"}, {"PS_INV_BUCKET": "0"}):
            print(env, flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, **env), check=True)
