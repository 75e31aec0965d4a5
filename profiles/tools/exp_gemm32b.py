"""The deep dense layers one by one through the test door ps_debug_gemm32 (gemm32b.hip: split-bf16 MFMA), each REPS times, error against
float64 once per shape.  Meant to run under `rocprofv3 --kernel-trace`: profiles/tools/trace_by_dispatch.py then prints the duration of
every shape from the per-dispatch trace.  usage (GPU box): python3 profiles/tools/exp_gemm32b.py [reps]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from point_unet_amd import _lib, runtime
if os.environ.get("PS_LIB_VARIANT"):
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "csrc", "build", "variants", "libps_%s.so" % os.environ["PS_LIB_VARIANT"])
L = _lib.lib()
dbg = ctypes.CDLL(os.path.join(os.path.dirname(_lib.LIB_PATH) if not os.environ.get("PS_LIB_VARIANT") else os.path.join(os.path.dirname(_lib.LIB_PATH), "..", "..", ".."), "libpointseg_debug.so"))
c_vp, i64 = ctypes.c_void_p, ctypes.c_int64
dbg.ps_debug_gemm32.restype = ctypes.c_int
dbg.ps_debug_gemm32.argtypes = [c_vp, ctypes.c_int, c_vp, ctypes.c_int, ctypes.c_int, c_vp, c_vp, ctypes.c_int, ctypes.c_int, c_vp, ctypes.c_int, ctypes.c_int,
                                c_vp, c_vp, i64, ctypes.c_int, ctypes.c_int, c_vp, ctypes.c_int]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
ctx = runtime.default_context(0)
# (name, R, c1, c2, cout, gather rows of x2 or 0)
SHAPES = [("enc2 [mlp2;sc]", 11250, 128, 128, 256, 0), ("enc3 [mlp2;sc]", 2812, 256, 256, 512, 0), ("enc4 [mlp2;sc]", 703, 512, 512, 1024, 0),
          ("enc3 att2-mlp", 2812, 256, 0, 256, 0), ("enc4 att2-mlp", 703, 512, 0, 512, 0), ("decoder_0", 703, 1024, 0, 1024, 0),
          ("dec0", 2812, 512, 1024, 512, 703), ("dec1", 11250, 256, 512, 256, 2812), ("enc4 mlp1", 703, 512, 0, 256, 0)]
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
for name, R, c1, c2, cout, gn in SHAPES:
    rng = np.random.default_rng(R + cout)
    x1 = rng.standard_normal((R, c1)).astype(np.float32) * rng.uniform(0.01, 4.0, (1, c1)).astype(np.float32)
    x2 = rng.standard_normal((gn if gn else R, max(c2, 1))).astype(np.float32)
    g2 = rng.integers(0, gn, R).astype(np.int32) if gn else None
    W = (rng.standard_normal((c1 + c2, cout)) / np.sqrt(c1 + c2)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    d_x1, d_x2 = torch.from_numpy(x1).cuda(), torch.from_numpy(x2).cuda()
    d_g2 = torch.from_numpy(g2).cuda() if gn else None
    y = torch.full((R, cout), float("nan"), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    for _ in range(reps):
        rc = dbg.ps_debug_gemm32(ctx.handle, 1, p(d_x1), c1, c1, None, p(d_x2) if c2 else None, max(c2, 1), c2, p(d_g2), 0, 0,
                                 W.ctypes.data_as(c_vp), b.ctypes.data_as(c_vp), R, cout, 1, p(y), cout)
        assert rc == 0, L.ps_last_error()
    torch.cuda.synchronize()
    X = x1.astype(np.float64) if not c2 else np.concatenate([x1.astype(np.float64), (x2[g2] if gn else x2).astype(np.float64)], 1)
    want = X @ W.astype(np.float64) + b.astype(np.float64)
    scale = np.abs(X) @ np.abs(W.astype(np.float64)) + np.abs(b)
    want = np.where(want >= 0, want, 0.2 * want)
    got = y.cpu().numpy().astype(np.float64)
    rel = float((np.abs(got - want) / scale).max()) if np.isfinite(got).all() else float("nan")
    print("SHAPE %-16s R %6d K %5d N %5d  %.3f GFLOP  err %.2e %s" % (name, R, c1 + c2, cout, 2e-9 * R * (c1 + c2) * cout, rel, "ok" if rel <= 2e-6 else "BAD"), flush=True)
