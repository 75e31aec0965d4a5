"""Can the index pyramid of the NEXT batch be hidden under the training step of this one if the two get DISJOINT sets of CUs?
(PyramidPrefetcher on a plain second stream: same step time -- the K-NN search and the training kernels fight for the same CUs.)  Streams
with a CU mask (hipExtStreamCreateWithCUMask): the pyramid on `p` CUs, the training step on the other 256 - p.
usage (GPU box): python profiles/tools/exp_cu_mask.py"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import bench
from point_unet_amd import runtime, weights
from point_unet_amd.helper_tool import ConfigBraTS as cfg
from point_unet_amd.pyramid import alloc_pyramid, build_pyramid
from point_unet_amd.train import Trainer

hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(bits):
    """bits: iterable of CU indices (0..255) the stream may use"""
    words = (ctypes.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


B, n0 = 8, 180000
bf16 = "--bf16" in sys.argv
xyz = np.stack([bench.brats_cloud(n0, b) for b in range(B)])
rng = np.random.default_rng(7)
feats = np.concatenate([xyz, rng.standard_normal((B, n0, cfg.in_channels - 3)).astype(np.float32)], -1)
labels = rng.integers(0, cfg.num_classes, (B, n0)).astype(np.int32)
d_xyz, d_feats, d_lab = torch.from_numpy(xyz).cuda(), torch.from_numpy(feats).cuda(), torch.from_numpy(labels).cuda()
ratios = cfg.sub_sampling_ratio[:cfg.num_layers]


def run(p_cus, layout, steps=12):
    """layout: 'low' = the pyramid takes mask bits [0, p); 'spread' = every (256 / p)-th bit"""
    if p_cus == 0:
        sp = st = None
    else:
        pb = list(range(p_cus)) if layout == "low" else [i * (256 // p_cus) for i in range(p_cus)]
        sp, st = masked_stream(pb), masked_stream([i for i in range(256) if i not in set(pb)])
    ctx_t, ctx_p = runtime.Context(0), runtime.Context(0)
    if st is not None:
        ctx_t.set_stream(st); ctx_p.set_stream(sp)
    else:
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        ctx_t.set_stream(s1); ctx_p.set_stream(s1)  # (no overlap: one stream, the pyramid in front of its step)
        st = sp = s1
    ctx_p.set_deferred_checks(True)
    tr = Trainer(cfg, params=weights.init_params(cfg, seed=2), device=0, ctx=ctx_t, keep_prob=0.5, mlp_dtype="bf16" if bf16 else "fp32")
    pyrs = [alloc_pyramid(B, n0, ratios, cfg.k_n, d_xyz.device) for _ in range(2)]
    ready = [None, None]
    torch.cuda.synchronize()

    def build(k):
        with torch.cuda.stream(sp):
            build_pyramid(d_xyz, cfg, ctx=ctx_p, out=pyrs[k])
            e = torch.cuda.Event(); e.record(sp); ready[k] = e

    def step(i):
        build((i + 1) % 2)                    # the next batch's pyramid
        st.wait_event(ready[i % 2])
        with torch.cuda.stream(st):
            loss = tr.train_step(pyrs[i % 2], d_feats, d_lab)
            done = torch.cuda.Event(); done.record(st)
        sp.wait_event(done)                   # (slot i % 2 is rebuilt two steps later; this orders it conservatively)
        return loss

    build(0)
    for i in range(4):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(4, 4 + steps):
        loss = step(i)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    tr.close(); ctx_p.close(); ctx_t.close()
    return ms, float(loss)


for p_cus, layout in [(0, "-"), (8, "low"), (16, "low"), (32, "low"), (16, "spread"), (32, "spread"), (64, "spread")]:
    ms, loss = run(p_cus, layout)
    print("%s pyramid on %3d CUs (%s): %.2f ms per step (loss %.4f)" % ("bf16" if bf16 else "fp32", p_cus, layout, ms, loss), flush=True)
