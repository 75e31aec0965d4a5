"""Which shapes the training step's GEMM time goes to: wraps the C-ABI calls of one step with CUDA events, grouped by (op, R, cin, cout).
usage (GPU box): python profiles/tools/train_shapes.py [batch]"""
import collections, os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import numpy as np
import torch
import bench
from point_unet_amd import _lib, weights
from point_unet_amd.helper_tool import ConfigBraTS as cfg
from point_unet_amd.pyramid import build_pyramid
from point_unet_amd.train import Trainer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n0 = 180000
xyz = np.stack([bench.brats_cloud(n0, 17 * b) for b in range(B)])
feats = np.concatenate([xyz, np.random.default_rng(0).standard_normal((B, n0, cfg.in_channels - 3)).astype(np.float32)], -1)
labels = np.random.default_rng(1).integers(0, cfg.num_classes, (B, n0)).astype(np.int32)
params = weights.init_params(cfg, seed=2)
tr = Trainer(cfg, params, engine="python")  # the wrappers below hook the ctypes entry points the host-side tape calls
if os.environ.get("PS_NO_B3"):
    _lib.check(_lib.lib().ps_set_train_gemm_b3(tr.ctx.handle, 0))
dx, df, dl = torch.from_numpy(xyz).cuda(), torch.from_numpy(feats).cuda(), torch.from_numpy(labels).cuda()
pyr = build_pyramid(dx, cfg)
tr.train_step(pyr, df, dl)
torch.cuda.synchronize()

L = _lib.lib()
rec = []
def wrap(name, keyf):
    fn = getattr(L, name)
    def w(*a):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = fn(*a); e1.record()
        rec.append((name, keyf(a), e0, e1))
        return r
    setattr(L, name, w)
val = lambda v: v.value if hasattr(v, "value") else v
wrap("ps_op_conv1x1_ex", lambda a: (val(a[5]), val(a[6]), val(a[7]), "acc" if val(a[9]) else ""))
wrap("ps_op_linear_wgrad_ex", lambda a: (val(a[5]), val(a[6]), val(a[7])))
wrap("ps_op_bn_train_fwd_ex", lambda a: (val(a[4]), val(a[5])))
wrap("ps_op_bn_train_bwd_ex", lambda a: (val(a[8]), val(a[9])))
tr.train_step(pyr, df, dl)
torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for name, key, e0, e1 in rec:
    k = (name, key)
    agg[k][0] += 1
    agg[k][1] += e0.elapsed_time(e1)
tot = collections.defaultdict(float)
for (name, key), (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%-26s %-34s x%-3d %8.3f ms" % (name, key, n, ms))
for (name, key), (n, ms) in agg.items():
    tot[name] += ms
print(dict(tot))
