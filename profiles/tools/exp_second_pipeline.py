"""Why is a SECOND ForwardPipeline in one process slower than the first (bench.py's config5 sub-result: 2.31 / 3.93 ms pipelined / serial
against 1.94 / 2.64 in a process of its own)?  A = config2 pipeline run first or not; B = config5 pipeline, fresh or on A's streams."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "6")
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import numpy as np
import torch
import bench
from point_unet_amd import weights
from point_unet_amd.helper_tool import ConfigBraTS
from point_unet_amd.pipeline import ForwardPipeline

mode = sys.argv[1] if len(sys.argv) > 1 else "both"


class cfg5(ConfigBraTS):
    k_n, num_classes, in_channels = 32, 2, 4


def clouds_for(cfg, n0, n, half):
    out = []
    for i in range(n):
        x = bench.brats_cloud(n0, 5000 + 17 * i)[None]
        f = np.concatenate([x, np.random.default_rng(i).standard_normal((1, n0, cfg.in_channels - 3)).astype(np.float32)], -1)
        out.append((torch.from_numpy(x).cuda(), torch.from_numpy(f.astype(np.float16) if half else f).cuda()))
    return out


def measure(pipe, clouds, steps, tag):
    k = [0]

    def step(overlap=True):
        x, f = clouds[k[0] % len(clouds)]
        k[0] += 1
        return pipe.submit(x, f, overlap=overlap)

    def sync():
        pipe.synchronize(); torch.cuda.synchronize()

    for _ in range(8):
        step()
    t, _ = bench.timed_region(step, steps, sync, None)
    ts, _ = bench.timed_region(lambda: step(overlap=False), 10, sync, None)
    ts2, _ = bench.timed_region(lambda: step(overlap=False), 20, sync, None)
    print("%-40s pipelined %.3f ms, serial %.3f ms (10 steps), %.3f ms (next 20)" % (tag, 1e3 * t / steps, 1e3 * ts / 10, 1e3 * ts2 / 20), flush=True)


A = None
if mode in ("both", "a_then_b", "reuse"):
    cA = clouds_for(ConfigBraTS, 180000, 8, False)
    A = ForwardPipeline(ConfigBraTS, params=weights.init_params(ConfigBraTS, seed=2, randomize_bn=True), lanes=4)
    A.prime(*cA[0])
    measure(A, cA, 40, "A: config2, first pipeline")
    if mode != "reuse":
        A.close()
        A = None
    del cA
    torch.cuda.empty_cache()
if mode == "dummy_streams":   # no pipeline A at all: just take four streams out of torch's pool first (and use them once)
    dummies = [torch.cuda.Stream() for _ in range(4)]
    for d in dummies:
        with torch.cuda.stream(d):
            torch.zeros(16, device="cuda").add_(1)
    torch.cuda.synchronize()
if mode == "dummy_contexts":  # four idle library contexts (own streams, workspaces) and nothing else
    from point_unet_amd import runtime
    keep = [runtime.Context(0) for _ in range(4)]
cB = clouds_for(cfg5, 262144, 4, True)
B = ForwardPipeline(cfg5, params=weights.init_params(cfg5, seed=2, randomize_bn=True), lanes=4, reuse=A)
B.prime(*cB[0])
measure(B, cB, 40, "B: config5, %s" % {"b_only": "only pipeline", "reuse": "on A's streams and contexts"}.get(mode, "second pipeline"))
measure(B, cB, 40, "B again")
