"""Why does the PCIe-inclusive service rate lose the pipeline's overlap on some boxes (1.661 ms against 1.607 serial)?
Times, over the pipeline's lanes: resident inputs; H2D only; D2H only; both (hipMemcpyAsync on the lane's stream = bench.py's
pcie_step), each after 20 untimed steps of the same kind, then the same WITHOUT a warm-up (the first transfers through a freshly
pinned buffer are slow), and the per-lane event timeline (H2D, compute, D2H) of serialised steps.
Round-3 findings on the builder's box: 0.903 resident / 1.031 H2D only / 0.926 D2H only / 1.015 both; serial timeline H2D 0.154 ms
(7.2 MB = 47 GB/s), compute 1.406 ms, D2H 0.065 ms (2.9 MB).  The same transfers as ordinary copy kernels reading / writing the
pinned host memory (16 / 64 / 256 workgroups on the lane's stream) were no better: 1.045 / 1.055 / 1.084 ms -- not kept.
usage (GPU box): python profiles/tools/exp_pcie.py [lanes] [steps]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "6")
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import numpy as np
import torch
import bench
from point_unet_amd import weights
from point_unet_amd.helper_tool import ConfigBraTS as cfg
from point_unet_amd.pipeline import ForwardPipeline

lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
n0 = 180000
host, dev = [], []
for i in range(8):
    x = bench.brats_cloud(n0, 17 * i)[None]
    f = np.concatenate([x, np.random.default_rng(i).standard_normal((1, n0, cfg.in_channels - 3)).astype(np.float32)], -1)
    host.append((torch.from_numpy(x).pin_memory(), torch.from_numpy(f).pin_memory()))
    dev.append((torch.from_numpy(x).cuda(), torch.from_numpy(f).cuda()))
params = weights.init_params(cfg, seed=2, randomize_bn=True)
pipe = ForwardPipeline(cfg, params=params, device=0, lanes=lanes)
pipe.prime(*dev[0])
d_in = [(torch.empty_like(dev[0][0]), torch.empty_like(dev[0][1])) for _ in range(lanes)]
h_out = [torch.empty((1, n0, cfg.num_classes), dtype=torch.float32).pin_memory() for _ in range(lanes)]


def one(i, h2d, d2h):
    k = pipe._i % lanes
    ln = pipe.lanes[k]
    hx, hf = host[i % 8]
    dx, df = d_in[k] if h2d else dev[i % 8]
    with torch.cuda.stream(ln.stream):
        if h2d:
            dx.copy_(hx, non_blocking=True)
            df.copy_(hf, non_blocking=True)
        out = pipe.submit(dx, df)
        if d2h:
            h_out[k].copy_(out, non_blocking=True)
    return out


def run(name, warm=20, n=steps, **kw):
    for i in range(warm):
        one(i, **kw)
    pipe.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        one(i, **kw)
    pipe.synchronize(); torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    print("%-60s lanes %d: %.4f ms/step" % (name, lanes, ms), flush=True)
    return ms


# first: the un-warmed service path at the driver's step count (what bench.py's include_pcie sub-result used to time)
run("H2D + D2H, NO warm-up, 20 steps (fresh pinned buffers)", warm=0, n=20, h2d=True, d2h=True)
run("H2D + D2H, 20 steps after that", warm=0, n=20, h2d=True, d2h=True)
run("resident inputs", h2d=False, d2h=False)
run("H2D only (hipMemcpyAsync on the lane)", h2d=True, d2h=False)
run("D2H only (hipMemcpyAsync on the lane)", h2d=False, d2h=True)
run("H2D + D2H (hipMemcpyAsync on the lane)", h2d=True, d2h=True)
run("resident inputs (again)", h2d=False, d2h=False)

# per-lane timeline of serialised steps: how long the three phases take on their own
ln = pipe.lanes[0]
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
acc = np.zeros(3)
for i in range(30):
    hx, hf = host[i % 8]
    dx, df = d_in[0]
    pipe.synchronize(); torch.cuda.synchronize()
    with torch.cuda.stream(ln.stream):
        ev[0].record(ln.stream)
        dx.copy_(hx, non_blocking=True); df.copy_(hf, non_blocking=True)
        ev[1].record(ln.stream)
        pipe._i = 0
        out = pipe.submit(dx, df)
        ev[2].record(ln.stream)
        h_out[0].copy_(out, non_blocking=True)
        ev[3].record(ln.stream)
    torch.cuda.synchronize()
    if i >= 5:
        acc += [ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2]), ev[2].elapsed_time(ev[3])]
acc /= 25
print("serial timeline: H2D %.3f ms (7.2 MB), compute %.3f ms, D2H %.3f ms (2.9 MB)" % tuple(acc), flush=True)
