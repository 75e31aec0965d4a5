"""Is the four-lane forward bound by the rate at which kernels can be dispatched?  Every cloud's 84 launches (37 of them the kd-tree build) go
through the command processor; this adds k EMPTY launches per cloud on the lane's own stream, in the middle of the cloud's work, and times the pipelined step.
usage (GPU box): python profiles/tools/exp_launch_rate.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import bench
from point_unet_amd import weights
from point_unet_amd.helper_tool import ConfigBraTS as cfg
from point_unet_amd.pipeline import ForwardPipeline

n0, n_clouds = 180000, 8
clouds = []
for i in range(n_clouds):
    x = bench.brats_cloud(n0, 17 * i)[None]
    f = np.concatenate([x, np.random.default_rng(i).standard_normal((1, n0, 4)).astype(np.float32)], -1)
    clouds.append((torch.from_numpy(x).cuda(), torch.from_numpy(f).cuda()))
pipe = ForwardPipeline(cfg, params=weights.init_params(cfg, seed=2, randomize_bn=True), device=0, lanes=4)
pipe.prime(*clouds[0])
tiny = [torch.zeros(64, device="cuda") for _ in pipe.lanes]
k_i = [0]


def step(extra):
    i = k_i[0]
    k_i[0] += 1
    x, f = clouds[i % n_clouds]
    out = pipe.submit(x, f)
    ln = pipe.lanes[i % len(pipe.lanes)]
    with torch.cuda.stream(ln.stream):
        for _ in range(extra):
            tiny[i % len(pipe.lanes)].add_(1.0)  # one tiny kernel each
    return out


def run(extra, steps=300):
    for _ in range(40):
        step(extra)
    pipe.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(extra)
    pipe.synchronize(); torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps


run(0)
for extra in (0, 8, 16, 32, 64, 0):
    print("extra launches per cloud %2d: %.4f ms per cloud" % (extra, run(extra)), flush=True)
