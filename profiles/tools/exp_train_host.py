"""Is the batch-1 training step (the per-rank work of BASELINE configs[3]) bound by the host's launch rate or by the device?
Times the host side of ps_pyramid_build + ps_randla_train_step (the calls return as soon as everything is enqueued) against the
synchronised step time.  usage: python profiles/tools/exp_train_host.py [batch]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402
from point_unet_amd import runtime, weights  # noqa: E402
from point_unet_amd.helper_tool import ConfigBraTS as cfg  # noqa: E402
from point_unet_amd.pyramid import alloc_pyramid, build_pyramid  # noqa: E402
from point_unet_amd.train import Trainer  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n0 = 180000
xyz = np.stack([bench.brats_cloud(n0, 1000 + b) for b in range(B)])
rng = np.random.default_rng(7)
feats = np.concatenate([xyz, rng.standard_normal((B, n0, 4)).astype(np.float32)], -1)
labels = rng.integers(0, 4, (B, n0)).astype(np.int32)
ctx = runtime.default_context(0)
ctx.use_torch_stream()
ctx.set_deferred_checks(True)
tr = Trainer(cfg, params=weights.init_params(cfg, seed=2), ctx=ctx, keep_prob=0.5)
d_xyz, d_f, d_l = torch.from_numpy(xyz).cuda(), torch.from_numpy(feats).cuda(), torch.from_numpy(labels).cuda()
pyr = alloc_pyramid(B, n0, cfg.sub_sampling_ratio[:cfg.num_layers], cfg.k_n, d_xyz.device)
for _ in range(30):
    build_pyramid(d_xyz, cfg, ctx=ctx, out=pyr)
    tr.train_step(pyr, d_f, d_l)
torch.cuda.synchronize()
host_p, host_t, total = [], [], []
for _ in range(30):
    t0 = time.perf_counter()
    build_pyramid(d_xyz, cfg, ctx=ctx, out=pyr)
    t1 = time.perf_counter()
    tr.train_step(pyr, d_f, d_l)
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    host_p.append(t1 - t0), host_t.append(t2 - t1), total.append(t3 - t0)
print("batch %d: host pyramid %.3f ms, host train_step %.3f ms, synchronised step %.3f ms (medians of 30)" % (
    B, 1e3 * np.median(host_p), 1e3 * np.median(host_t), 1e3 * np.median(total)))
# back-to-back (the host runs ahead): the steady-state step time
t0 = time.perf_counter()
for _ in range(30):
    build_pyramid(d_xyz, cfg, ctx=ctx, out=pyr)
    tr.train_step(pyr, d_f, d_l)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("30 steps back to back: host enqueue %.3f ms/step, device-inclusive %.3f ms/step" % (1e3 * (t1 - t0) / 30, 1e3 * (t2 - t0) / 30))
