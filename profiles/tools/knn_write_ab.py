"""K-NN write side A/B for rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE passes: N pyramids of the 180 000-point BraTS-shaped cloud with the
one-query-per-lane kernel (knn_pair_kernel: every lane stores its 64-byte row as four int4 stores), then N with the experiment kernel at 64
queries per wave (knn_pair_refill_kernel: the rows staged in LDS, four adjacent lanes write one row's 64 contiguous bytes in ONE store
instruction) -- the two kernels have different names in the counter CSV.  Needs the refill flavour:
    PS_LIB_VARIANT=refill python3 profiles/tools/knn_write_ab.py [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from point_unet_amd import _lib as _plib
_plib.LIB_PATH = os.path.join(os.path.dirname(_plib.LIB_PATH), "csrc", "build", "variants", "libps_%s.so" % os.environ.get("PS_LIB_VARIANT", "refill"))
import numpy as np
import torch
import bench
from point_unet_amd import runtime
from point_unet_amd.helper_tool import ConfigBraTS as cfg
from point_unet_amd.pyramid import build_pyramid
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
ctx = runtime.default_context(0)
x = torch.from_numpy(bench.brats_cloud(180000, 3)[None]).cuda()
os.environ["PS_KNN_Q"] = "64"
os.environ["PS_KNN_REFILL_MIN"] = "16"
ref = None
for refill in (0, 1):
    os.environ["PS_KNN_REFILL"] = str(refill)
    pyr = build_pyramid(x, cfg, ctx=ctx)
    for _ in range(n):
        build_pyramid(x, cfg, ctx=ctx, out=pyr)
    torch.cuda.synchronize()
    tb = [t.cpu().numpy().copy() for t in pyr.neigh_idx + pyr.sub_idx + pyr.interp_idx]
    if ref is None:
        ref = tb
    print("refill=%d identical=%s" % (refill, all(np.array_equal(a, b) for a, b in zip(ref, tb))), flush=True)
