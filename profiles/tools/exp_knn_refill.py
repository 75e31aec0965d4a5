"""A/B of the K-NN search kernel: one query per lane (PS_KNN_REFILL=0) against the persistent form that refills finished lanes
(PS_KNN_REFILL=1; PS_KNN_Q queries per wave, refill once PS_KNN_REFILL_MIN lanes are idle).  Checks that every table of the pyramid is identical, times the
`knn_search` stage alone, and the 4-lane pipeline (whole / pyramid only).  The refill kernel only exists in the experiment flavour of the library:
    sh profiles/tools/build_variant.sh refill "-DPS_KNN_REFILL_EXP" knn.hip        (here)
    PS_LIB_VARIANT=refill python profiles/tools/exp_knn_refill.py [quick|stage]    (GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import bench
from point_unet_amd import _lib as _plib
if os.environ.get("PS_LIB_VARIANT"):
    _plib.LIB_PATH = os.path.join(os.path.dirname(_plib.LIB_PATH), "csrc", "build", "variants", "libps_%s.so" % os.environ["PS_LIB_VARIANT"])
    print("library:", _plib.LIB_PATH, flush=True)
from point_unet_amd import weights, runtime
from point_unet_amd.helper_tool import ConfigBraTS, ConfigPancreas
from point_unet_amd.pipeline import ForwardPipeline
from point_unet_amd.pyramid import build_pyramid

quick = len(sys.argv) > 1 and sys.argv[1] in ("quick", "stage", "base")
base_only = len(sys.argv) > 1 and sys.argv[1] == "base"  # one-query-per-lane kernel of whatever library is loaded: stage time + pipeline
stage_only = len(sys.argv) > 1 and sys.argv[1] == "stage"


def tables(p):
    out = []
    for grp in (p.neigh_idx, p.sub_idx, p.interp_idx, p.order):
        out += [t.cpu().numpy().copy() for t in grp]
    return out


def setenv(refill, q=None, rmin=None):
    os.environ["PS_KNN_REFILL"] = str(refill)
    if q: os.environ["PS_KNN_Q"] = str(q)
    if rmin: os.environ["PS_KNN_REFILL_MIN"] = str(rmin)


def stage_ms(ctx, x, cfg, reps=30):
    pyr = build_pyramid(x, cfg, ctx=ctx)
    for _ in range(5):
        build_pyramid(x, cfg, ctx=ctx, out=pyr)
    torch.cuda.synchronize()
    ctx.timing_begin(only="knn_search")
    for _ in range(reps):
        build_pyramid(x, cfg, ctx=ctx, out=pyr)
    rows = ctx.timing_end()
    ms = [m for name, m, _ in rows if name == "knn_search"][0] / reps
    prof = None
    import ctypes
    from point_unet_amd import _lib
    L = _lib.lib()
    if hasattr(L, "ps_debug_knn_prof"):
        buf = (ctypes.c_ulonglong * 16)()
        L.ps_debug_knn_prof(buf, 1)
        build_pyramid(x, cfg, ctx=ctx, out=pyr)
        torch.cuda.synchronize()
        L.ps_debug_knn_prof(buf, 1)
        w = max(1, buf[0])
        prof = "waves %d cyc/wave %.0f refill %.0f desc %.0f leaf %.0f pop %.0f | iters %.1f events %.1f lanes/iter %.1f steps %.1f" % (
            buf[0], buf[1] / w, buf[2] / w, buf[3] / w, buf[4] / w, buf[5] / w, buf[6] / w, buf[7] / w, buf[8] / max(1, buf[6]), buf[9] / w)
    return ms, tables(pyr), prof


ctx = runtime.default_context(0)
cases = [("brats180k/K16", ConfigBraTS, torch.from_numpy(bench.brats_cloud(180000, 3)[None]).cuda()),
         ("uniform180k/K16", ConfigBraTS, torch.rand((1, 180000, 3), generator=torch.Generator().manual_seed(5)).cuda())]
if not quick:
    cases.append(("uniform262k/K32", ConfigPancreas, torch.rand((1, 262144, 3), generator=torch.Generator().manual_seed(6)).cuda()))
    cases.append(("brats2x45000/K16", ConfigBraTS, torch.from_numpy(np.stack([bench.brats_cloud(45000, 8), bench.brats_cloud(45000, 9)])).cuda()))
variants = [(0, None, None), (1, 64, 16), (1, 128, 16), (1, 256, 16), (1, 256, 24), (1, 192, 16), (1, 512, 16), (1, 256, 8), (1, 256, 32)]
if quick:
    variants = variants[:5]
if stage_only:
    variants = variants[:3]
if base_only:
    variants = variants[:1]
for name, cfg, x in cases:
    ref = None
    for refill, q, rmin in variants:
        setenv(refill, q, rmin)
        ms, tb, prof = stage_ms(ctx, x, cfg)
        if ref is None:
            ref = tb
        same = all(np.array_equal(a, b) for a, b in zip(ref, tb))
        print("%-18s refill=%d Q=%s min=%s  knn_search %.4f ms  identical=%s" % (name, refill, q, rmin, ms, same), flush=True)
        if prof and refill: print("      ", prof, flush=True)

if stage_only:
    sys.exit(0)
# the 4-lane pipeline, whole and pyramid only
cfg = ConfigBraTS
n0, n_clouds = 180000, 8
clouds = []
for i in range(n_clouds):
    x = bench.brats_cloud(n0, 17 * i)[None]
    f = np.concatenate([x, np.random.default_rng(i).standard_normal((1, n0, 4)).astype(np.float32)], -1)
    clouds.append((torch.from_numpy(x).cuda(), torch.from_numpy(f).cuda()))
pipe = ForwardPipeline(cfg, params=weights.init_params(cfg, seed=2, randomize_bn=True), device=0, lanes=4)
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
        build_pyramid(x, cfg, ctx=ln.ctx, out=ln.pyramid)


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
for rep in range(2):
    for refill, q, rmin in variants[:1 if base_only else 6]:
        setenv(refill, q, rmin)
        print("pipeline x4: refill=%d Q=%s min=%s  whole %.4f  pyramid only %.4f ms per cloud" % (refill, q, rmin, run("whole"), run("pyramid")), flush=True)
pipe.close()
