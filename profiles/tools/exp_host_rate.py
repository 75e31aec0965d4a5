"""Is the four-lane pipeline fed fast enough by ONE host thread?  Measures (a) the host time to ENQUEUE a cloud (pyramid + forward) when the
queues are empty -- a burst of `lanes` clouds after a device sync, no waiting --, (b) the steady pipelined time per cloud, (c) the same steady
state with the lanes driven by TWO / FOUR host threads (ctypes releases the GIL inside the library; every thread owns its lanes).
usage (GPU box): python profiles/tools/exp_host_rate.py"""
import os, sys, time, threading
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
pipe = ForwardPipeline(cfg, params=weights.init_params(cfg, seed=2, randomize_bn=True), device=0, lanes=4)
pipe.prime(*clouds[0])


def lane_step(ln, i):
    x, f = clouds[i % n_clouds]
    with torch.cuda.stream(ln.stream):
        build_pyramid(x, cfg, ctx=ln.ctx, out=ln.pyramid)
        return ln.net.inference({"pyramid": ln.pyramid, "features": f})


def sync():
    pipe.synchronize(); torch.cuda.synchronize()


for _ in range(40):
    pipe.submit(*clouds[0])
sync()
# (a) enqueue cost with empty queues
costs = []
for rep in range(20):
    sync()
    t0 = time.perf_counter()
    for k in range(4):
        lane_step(pipe.lanes[k], k)
    costs.append((time.perf_counter() - t0) / 4)
    sync()
print("host time to enqueue one cloud (pyramid + forward, empty queues): median %.3f ms, min %.3f ms" % (1e3 * np.median(costs), 1e3 * min(costs)), flush=True)
# the same for the two halves
for what in ("pyramid", "network"):
    costs = []
    for rep in range(20):
        sync()
        t0 = time.perf_counter()
        for k in range(4):
            ln = pipe.lanes[k]
            x, f = clouds[k]
            with torch.cuda.stream(ln.stream):
                if what == "pyramid":
                    build_pyramid(x, cfg, ctx=ln.ctx, out=ln.pyramid)
                else:
                    ln.net.inference({"pyramid": ln.pyramid, "features": f})
        costs.append((time.perf_counter() - t0) / 4)
    sync()
    print("   %s alone: median %.3f ms" % (what, 1e3 * np.median(costs)), flush=True)


def run_threads(n_threads, steps=400):
    per = steps // n_threads
    lanes_of = [pipe.lanes[t::n_threads] for t in range(n_threads)]

    def worker(t):
        for i in range(per):
            lane_step(lanes_of[t][i % len(lanes_of[t])], i * n_threads + t)

    sync()
    t0 = time.perf_counter()
    ths = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
    for th in ths: th.start()
    for th in ths: th.join()
    sync()
    return 1e3 * (time.perf_counter() - t0) / (per * n_threads)


for rep in range(2):
    for nt in (1, 2, 4):
        print("steady state, %d host thread(s): %.4f ms per cloud" % (nt, run_threads(nt)), flush=True)
pipe.close()
