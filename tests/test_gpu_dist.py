"""The multi-GPU code path of bench.py pushed through RCCL on ONE GPU (world size 1): no scaling can be measured on a 1-GPU box,
but everything an 8-GPU run executes besides having peers does run -- `init_process_group("nccl")`, the `ranks_seen` all-reduce,
the barriers and the MAX-reduce around the timed region, and in training mode `allreduce_mean_` on the 19.97 MB flat gradient
buffer plus the BatchNorm-statistics all-reduces (Trainer(sync_bn=True) whenever a process group exists).  Each run is a fresh
child process launched the way `torch.distributed.run` would (RANK / WORLD_SIZE / MASTER_* in the environment, 127.0.0.1)."""
import json

import numpy as np
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _bench(args, with_group):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    if with_group:
        env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", LOCAL_WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.lstrip().startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_forward_bench_through_rccl_world_size_one():
    """Forward mode: the process group is created AFTER the lanes' streams (bench.py: RCCL's internal streams, created first, used to
    push two lanes onto one hardware queue: 1.72 vs 1.31 ms).  With that order the line under a one-rank RCCL group must stay within
    5 % of the line without any process group (best of two runs each: the box's run-to-run spread is 2-3 %)."""
    common = ["--gpus", "1", "--steps", "100", "--warmup", "10", "--no-cpu-baseline", "--no-sub-results", "--no-stage-timing"]
    with_pg = [_bench(common + ["--dist-backend", "nccl"], True) for _ in range(2)]
    without = [_bench(common, False) for _ in range(2)]
    for line in with_pg:
        assert line["n_gpus"] == 1 and line["ranks_seen"] == 1 and line["metric"] == "points_per_sec_forward"
    a = min(x["ms_per_step"] for x in with_pg)
    b = min(x["ms_per_step"] for x in without)
    print("forward ms/step: RCCL world-1 %.4f, no process group %.4f" % (a, b))
    assert a <= 1.05 * b, (a, b)


def test_train_bench_through_rccl_world_size_one():
    """Training mode, batch 1 (BASELINE configs[3]'s per-GPU work): the gradient all-reduce of the flat buffer and the SyncBN all-reduces
    really execute through RCCL -- a callback given with world size 1 is still called (ps_trainer_set_collective), and the line reports how
    many calls a step makes (1 flat gradient buffer + 2 per BatchNorm layer) and what they cost.  The shared-statistics form sums in another
    order than the one-rank kernels, and the bench line's loss is the one after some forty optimisation steps (pre-heat + warm-up + timed):
    the two trajectories agree to a few per cent, not to rounding -- the numerics of SyncBN are test_gpu_train.py's business
    (two ranks on this GPU = one rank with batch 2)."""
    common = ["--gpus", "1", "--mode", "train", "--batch", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"]
    with_pg = _bench(common + ["--dist-backend", "nccl"], True)
    without = _bench(common + ["--no-stage-timing"], False)
    assert with_pg["ranks_seen"] == 1 and with_pg["metric"] == "points_per_sec_train_step"
    assert with_pg["config"]["parameters"] == 4992852
    assert with_pg["collectives_per_step"] is not None and with_pg["collectives_per_step"] >= 50, with_pg["collectives_per_step"]
    assert with_pg["collective_bytes_per_step"] >= 4 * 4992852
    assert np.isfinite(with_pg["loss"]) and abs(with_pg["loss"] - without["loss"]) <= 5e-2 * abs(without["loss"]), (with_pg["loss"], without["loss"])
    print("train ms/step: RCCL world-1 %.3f (%d collectives per step, %.3f ms between their event pairs, %.3f ms of host time), no process group %.3f" % (
        with_pg["ms_per_step"], with_pg["collectives_per_step"], with_pg["collective_ms"], with_pg["collective_host_ms"], without["ms_per_step"]))
