"""Where does the pipelined step go?  Times the pyramid builds alone, the network alone and both, over the pipeline's lanes.
usage (GPU box): python profiles/tools/exp_overlap.py [lanes] [steps]"""
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
from point_unet_amd.pyramid import build_pyramid

lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
n0 = 180000
clouds = []
for i in range(8):
    x = bench.brats_cloud(n0, 17 * i)[None]
    f = np.concatenate([x, np.random.default_rng(i).standard_normal((1, n0, cfg.in_channels - 3)).astype(np.float32)], -1)
    clouds.append((torch.from_numpy(x).cuda(), torch.from_numpy(f).cuda()))
params = weights.init_params(cfg, seed=2, randomize_bn=True)
pipe = ForwardPipeline(cfg, params=params, device=0, lanes=lanes)
pipe.prime(*clouds[0])


def run(kind):
    def one(i):
        x, f = clouds[i % 8]
        ln = pipe.lanes[i % lanes]
        with torch.cuda.stream(ln.stream):
            if kind in ("both", "pyramid"):
                build_pyramid(x, cfg, ctx=ln.ctx, out=ln.pyramid)
            if kind in ("both", "network"):
                ln.net.inference({"pyramid": ln.pyramid, "features": f})
    for i in range(20):
        one(i)
    pipe.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        one(i)
    t_sub = time.perf_counter() - t0  # host time to enqueue everything (the GPU may still be far behind)
    pipe.synchronize(); torch.cuda.synchronize()
    run.submit_ms = t_sub / steps * 1e3
    return (time.perf_counter() - t0) / steps * 1e3


for kind in ("both", "pyramid", "network", "both"):
    ms = run(kind)
    print("%-8s lanes %d: %.4f ms/step (host enqueue %.4f ms/step)" % (kind, lanes, ms, run.submit_ms), flush=True)

# pure host cost of the two calls: the GPU is idle when each one is issued (synchronised before), the call itself is timed
ln = pipe.lanes[0]
tb = tn = 0.0
for i in range(50):
    x, f = clouds[i % 8]
    with torch.cuda.stream(ln.stream):
        torch.cuda.synchronize(); pipe.synchronize()
        t0 = time.perf_counter(); build_pyramid(x, cfg, ctx=ln.ctx, out=ln.pyramid); t1 = time.perf_counter()
        torch.cuda.synchronize(); pipe.synchronize()
        t2 = time.perf_counter(); ln.net.inference({"pyramid": ln.pyramid, "features": f}); t3 = time.perf_counter()
    if i >= 10:
        tb += t1 - t0; tn += t3 - t2
pipe.synchronize()
print("host cost per call with an idle GPU: build_pyramid %.3f ms, inference %.3f ms" % (tb / 40 * 1e3, tn / 40 * 1e3))
