"""What the four lanes hide: the pipelined forward timed (a) whole, (b) network only (every lane re-runs the forward on a pyramid it built before the
timed region), (c) pyramid only.  usage (GPU box): python profiles/tools/exp_pipeline_parts.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import bench
from point_unet_amd import weights
from point_unet_amd.helper_tool import ConfigBraTS as cfg
from point_unet_amd.pipeline import ForwardPipeline
from point_unet_amd.pyramid import build_pyramid

n0, n_clouds = 180000, 8
clouds = []
for i in range(n_clouds):
    x = bench.brats_cloud(n0, 17 * i)[None]
    f = np.concatenate([x, np.random.default_rng(i).standard_normal((1, n0, 4)).astype(np.float32)], -1)
    clouds.append((torch.from_numpy(x).cuda(), torch.from_numpy(f).cuda()))
for lanes in (4, 2, 1):
    pipe = ForwardPipeline(cfg, params=weights.init_params(cfg, seed=2, randomize_bn=True), device=0, lanes=lanes)
    pipe.prime(*clouds[0])
    k = [0]

    def step(mode):
        i = k[0]
        k[0] += 1
        x, f = clouds[i % n_clouds]
        if mode == "whole":
            return pipe.submit(x, f)
        ln = pipe.lanes[i % len(pipe.lanes)]
        with torch.cuda.stream(ln.stream):
            if mode == "pyramid":
                build_pyramid(x, cfg, ctx=ln.ctx, out=ln.pyramid)
                return None
            return ln.net.inference({"pyramid": ln.pyramid, "features": f})

    def run(mode, steps=300):
        for _ in range(40):
            step(mode)
        pipe.synchronize(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(mode)
        pipe.synchronize(); torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / steps

    run("whole")
    print("lanes %d: whole %.4f  network only %.4f  pyramid only %.4f ms per cloud" % (lanes, run("whole"), run("network"), run("pyramid")), flush=True)
    pipe.close()
