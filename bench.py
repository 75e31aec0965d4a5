#!/usr/bin/env python3
"""bench.py -- points/sec of the PointSegment hot path (index pyramid + RandLA-Net forward) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one cloud per GPU: ps_pyramid_build (kd-tree build, K-NN and 1-NN
search for all 5 levels) followed by ps_randla_forward, with the cloud already resident in HBM.  By default --lanes (4)
clouds are in flight per GPU, each on its own HIP stream (point-unet_amd/pipeline.py, the counterpart of the reference's
tf.data map + prefetch); every timed step still does all of its work inside the timed region, and the line also carries the
serial per-cloud latency ("serial_ms_per_cloud").  --no-pipeline times serial steps on one stream.  Workload =
BASELINE.json configs[1]: a 180 000-point BraTS-shaped cloud (voxel-lattice coordinates inside an ellipsoid of a
240x240x155 grid, shuffled; 4 z-scored modalities), K=16, 5 levels (ratios 4,4,4,4,2; d_out 16..512), fp32,
random-init weights of the reference architecture (synthetic data: no datasets/checkpoints are reachable).

Multi-GPU: the path shards by cloud (one volume per GPU); the forward has no exchange step, so there
is no data-path collective -- ranks only meet at the barriers around the timed region ("scaling": "weak").

One JSON line on rank 0; besides the contract's keys it carries
  "roofline"      for the dominant stage: algorithmic bytes or FLOPs per launch (SURVEY 8d formulas, reference
                  formulation) / its average duration measured with hipEvents on the launch stream
  "stages"        the same for every stage (name, ms per step, bound, achieved, frac)
  "cpu_baseline"  the oracle (a port of the reference CPU path) timed on this host on one cloud
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  The lanes need one each, next to the null stream
# and RCCL's: 6 measured best with 4 lanes (1.23 ms/step; 4 queues / 3 lanes: 1.33; 7 or more queues get slower again).  Must be
# in the environment before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "6")

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec
F32_MFMA_PEAK_TF = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_* = fp32 vector peak


def brats_cloud(n, seed, grid=(240, 240, 155)):
    rng = np.random.default_rng(seed)
    g = np.asarray(grid)
    out = np.empty((0, 3), np.int64)
    while len(out) < n:
        c = rng.integers(0, g, size=(2 * n, 3))
        u = (c - g / 2.0) / (0.45 * g)
        c = c[(u * u).sum(1) < 1.0]
        out = np.unique(np.concatenate([out, c]), axis=0)
    out = out[rng.permutation(len(out))[:n]]
    return (out / g).astype(np.float32)


def level_sizes(n0, ratios):
    n = [n0]
    for r in ratios:
        n.append(n[-1] // r)
    return n


def algorithmic_costs(cfg, n0, B):
    """Per-stage algorithmic FLOPs and bytes per step (FLOP = 2*MAC; gather-counted bytes; [N,K,.]
    intermediates count zero).  Reference formulation: the score GEMM is counted at its full 2*N*K*d^2."""
    L, K = cfg.num_layers, cfg.k_n
    n = level_sizes(n0, cfg.sub_sampling_ratio[:L])
    c = {}
    nq = sum(n[:L])
    # one launch: the K-NN self queries and the 1-NN up-sampling queries of every level.  SURVEY 8(d): read xyz sum N_i * 12 + write
    # sum N_i * K * 4 (neigh_idx) + write sum N_i * 4 (interp_idx) = 19.2 MB for the 180 000-point cloud (the query point is counted once)
    c["knn_search"] = dict(flops=0, bytes=B * nq * (12 + K * 4 + 4))
    c["kdtree_build"] = dict(flops=0, bytes=B * (sum(n) * 12 + sum(n) * 16 + 2 * sum(n) * 16))
    # (the level slices of tf_map -- coordinate rows and pooling-table rows -- are written by the tree builders' first pass and by the
    #  K-NN lanes themselves since round 4: no launch of their own, and SURVEY's two formulas above do not count them)
    c["fc0"] = dict(flops=2 * B * n0 * cfg.in_channels * 8, bytes=B * n0 * (cfg.in_channels + 8) * 4)
    d_in = 8
    for i in range(L):
        d = cfg.d_out[i]
        h = d // 2
        N, N1 = B * n[i], B * n[i + 1]
        locse = K * 10 * h
        # "bytes" = SURVEY 8(d)'s gather-counted figure (a gathered row counts once per use); "unique" = every distinct row once (what HBM
        # has to deliver at least: the repeats are L2 / MALL hits) -- the HBM fraction of a stage is priced on the unique bytes
        att_unique = N * (12 + K * 4 + 12 + h * 4 + d * 4)
        c["enc%d_att1" % i] = dict(flops=2 * N * (locse + K * d * d),
                                   bytes=N * (12 + K * 4 + K * 12 + K * h * 4 + d * 4), unique=att_unique)
        # (the kernel re-derives LocSE + LFA-mlp1 in stage 2 instead of storing [N,K,h]; the reference computes it once,
        #  so it is counted once)
        c["enc%d_att2" % i] = dict(flops=2 * N * (K * h * h + K * d * d),
                                   bytes=N * (12 + K * 4 + K * 12 + K * h * 4 + d * 4), unique=att_unique)
        c["enc%d_dense" % i] = dict(flops=2 * N * (d_in * h + d * h + d * d + 2 * d * d + 2 * d_in * d),
                                    bytes=N * 4 * (2 * d_in + h + d + h + d + d + 2 * d))
        c["enc%d_pool" % i] = dict(flops=0, bytes=N1 * (K * 4 + K * 2 * d * 4 + 2 * d * 4),
                                   unique=N1 * K * 4 + min(N1 * K, N) * 2 * d * 4 + N1 * 2 * d * 4)
        d_in = 2 * d
    c["decoder_0"] = dict(flops=2 * B * n[L] * d_in * d_in, bytes=B * n[L] * 2 * d_in * 4 + d_in * d_in * 4)
    chans = [2 * cfg.d_out[0]] + [2 * d for d in cfg.d_out[:L]]
    up = d_in
    for j in range(L):
        skip = chans[-j - 2]
        N = B * n[L - 1 - j]
        c["dec%d" % j] = dict(flops=2 * N * (skip + up) * skip, bytes=N * 4 * (skip + up + skip + 1) + (skip + up) * skip * 4)
        up = skip
    c["head"] = dict(flops=2 * B * n0 * (up * 64 + 64 * 32 + 32 * cfg.num_classes),
                     bytes=B * n0 * 4 * (up + cfg.num_classes))
    return c


def cpu_baseline(cfg, xyz, feats, params):
    """The oracle (a port of the reference CPU path) timed on this host's cores, on ONE cloud of the workload.  Two figures
    (BASELINE.md section 3):
      value / as shipped   the KNN pyramid on ONE thread -- the reference's own threading at batch 1 (knn_.cxx:108 parallelises over the
                           batch only) -- plus the network forward on every core (TF-CPU's intra-op pool; here torch-CPU fp32)
      best_of_threads      the same with the KNN queries spread over threads as well (OpenMP over queries in the oracle) and both legs at the
                           best thread count of a sweep ({8, 16, 32, 64, all} capped at the host's cores; torch's inter-op pool pinned to one
                           thread) -- `threads` says how many each leg actually used, `cores` what the host has
    The NumPy fp32 forward of round 1 is timed once more for continuity ("numpy_net_seconds")."""
    import torch
    from oracle import bindings as ob
    from oracle import randla_oracle as ro
    from oracle import randla_train_oracle as rto
    cores = os.cpu_count()
    ratios = cfg.sub_sampling_ratio[:cfg.num_layers]
    n = xyz.shape[0] * xyz.shape[1]

    def knn_leg(threads, qpar):
        t0 = time.perf_counter()
        pyr = ro.build_pyramid(lambda s, q, k: ob.knn_batch(s, q, k, threads=threads, qpar=qpar), xyz, cfg.k_n, ratios)
        return time.perf_counter() - t0, pyr

    t_knn1, (pts, nbr, pool, up) = knn_leg(1, False)
    # (more threads than work items slows both legs down on a many-core host: 256 OpenMP threads take 1.75 s for what one thread
    #  does in 0.43 s, 256 torch threads 76 s for a forward that NumPy's BLAS does in 4.5 s -- so the thread counts are swept)
    t_knn_all, knn_threads = None, 1
    sweep = sorted({min(cores, t) for t in (8, 16, 32, 64)})
    for th in sorted(set(sweep) | {cores}):
        t, _ = knn_leg(th, True)
        if t_knn_all is None or t < t_knn_all:
            t_knn_all, knn_threads = t, th
    if t_knn1 < t_knn_all:
        t_knn_all, knn_threads = t_knn1, 1
    try:
        torch.set_num_interop_threads(1)  # (one operator at a time: the graph is a chain; must precede the first parallel region)
    except RuntimeError:
        pass
    t_net, net_how, net_threads, net_sweep = None, "", 0, {}
    for th in sweep:
        torch.set_num_threads(th)
        rto.forward(params, cfg.num_layers, pts, nbr, pool, up, feats, torch.float32)  # first call: thread pool start-up, page faults
        t0 = time.perf_counter()
        rto.forward(params, cfg.num_layers, pts, nbr, pool, up, feats, torch.float32)
        t = time.perf_counter() - t0
        net_sweep[th] = round(t, 3)
        if t_net is None or t < t_net:
            t_net, net_how, net_threads = t, "torch-CPU fp32 forward on %d threads" % th, th
    t0 = time.perf_counter()
    ro.inference(params, cfg.num_layers, pts, nbr, pool, up, feats, np.float32)
    t_numpy = time.perf_counter() - t0
    if t_numpy < t_net:
        t_net, net_how = t_numpy, "NumPy fp32 forward (BLAS threads)"
    return dict(value=n / (t_knn1 + t_net), unit="points/s", cores=cores, kind="port",
                sample="1 cloud of %d points, full pyramid + forward: KNN pyramid %.2f s on 1 thread (the reference's threading at batch 1, "
                       "knn_.cxx:108) + %s %.2f s (the faster of NumPy and torch-CPU, thread count swept)" % (xyz.shape[1], t_knn1, net_how, t_net),
                knn_seconds=t_knn1, net_seconds=t_net, numpy_net_seconds=t_numpy,
                net_threads=net_threads, net_seconds_by_threads=net_sweep,
                best_of_threads=dict(value=n / (t_knn_all + t_net), unit="points/s", cores=cores, threads=dict(knn=knn_threads, net=net_threads),
                                     sample="same cloud: KNN pyramid %.2f s with the queries spread over %d threads (best of %s) + the forward at its best "
                                            "thread count (%d of %s)" % (t_knn_all, knn_threads, sorted(set(sweep) | {cores}), net_threads, sweep),
                                     knn_seconds=t_knn_all, knn_threads=knn_threads, net_seconds=t_net))


def spawn_ranks(n, argv, script=None, extra_env=None, poll_s=0.05):
    """`python bench.py --gpus N` without a launcher: start N FRESH child processes (one rank per GPU) with the rendezvous
    variables torch.distributed.run would set, relay rank 0's stdout (the JSON line), return the worst child exit code.  The
    parent has made no GPU call at this point (it never initialises HIP) and nothing is exec'ed: the children are ordinary
    subprocesses.  A rank that dies takes the others down (they would otherwise wait at a barrier forever): each child is
    terminated by its own PID."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = script or os.path.abspath(__file__)
    base = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n))
    base.update(extra_env or {})
    procs = []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        # rank 0's stdout is the line the driver parses; the other ranks print nothing there, anything they do goes to stderr
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env, stdout=subprocess.PIPE if r == 0 else 2))  # 2 = this process's stderr fd
    worst = 0
    alive = set(range(n))
    out0 = b""
    try:
        import threading
        chunks = []
        reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
        reader.start()
        while alive:
            for r in sorted(alive):
                rc = procs[r].poll()
                if rc is None:
                    continue
                alive.discard(r)
                if rc != 0:
                    worst = worst or rc
                    for q in sorted(alive):  # a failed rank strands the others at their next barrier
                        procs[q].terminate()
            if alive:
                time.sleep(poll_s)
        reader.join(timeout=10)
        out0 = b"".join(chunks)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for line in out0.decode(errors="replace").splitlines():
        # only rank 0's JSON line goes to stdout; library chatter on its stdout ("[Gloo] Rank 0 is connected ...") goes to stderr
        (sys.stdout if line.lstrip().startswith("{") else sys.stderr).write(line + "\n")
    sys.stdout.flush()
    return worst


def gpu_local_cpus(gpu_index):
    """CPUs of the NUMA node GPU `gpu_index` hangs off, read from sysfs WITHOUT touching the GPU: the KFD topology lists the GPU nodes in the
    order HIP enumerates them (no HIP_VISIBLE_DEVICES reordering assumed) with their PCI domain / location id; the PCI device directory names
    the CPUs next to it.  Returns a set of CPU ids, or None when any piece is missing (containers without sysfs, other drivers)."""
    try:
        root = "/sys/class/kfd/kfd/topology/nodes"
        gpus = []
        for node in sorted(os.listdir(root), key=int):
            props = dict(line.split() for line in open(os.path.join(root, node, "properties")) if len(line.split()) == 2)
            if int(props.get("simd_count", "0")) > 0:
                gpus.append(props)
        pr = gpus[gpu_index]
        loc, dom = int(pr["location_id"]), int(pr.get("domain", "0"))
        bdf = "%04x:%02x:%02x.%d" % (dom, (loc >> 8) & 0xff, (loc >> 3) & 0x1f, loc & 7)
        cpus = set()
        for part in open("/sys/bus/pci/devices/%s/local_cpulist" % bdf).read().strip().split(","):
            if part:
                a, _, b = part.partition("-")
                cpus.update(range(int(a), int(b or a) + 1))
        return cpus or None
    except Exception:  # noqa: BLE001
        return None


def rank_cpu_slice(local_rank, n_local, allowed, local_cpus=None):
    """The CPUs rank `local_rank` of `n_local` ranks on this node pins itself to: the allowed CPUs of its GPU's NUMA node (sysfs) divided
    evenly among the ranks whose GPUs share that node is not knowable here without the other ranks' lookups, so: the rank's own even slice of
    `allowed`, intersected with its GPU's node when that leaves at least two CPUs.  Eight Python hosts each enqueueing ~90 k launches per
    second must not share cores (nor migrate between sockets away from their GPU)."""
    allowed = sorted(allowed)
    per = max(1, len(allowed) // max(n_local, 1))
    mine = set(allowed[local_rank * per:(local_rank + 1) * per] or allowed)
    if local_cpus:
        near = sorted(set(allowed) & set(local_cpus))
        if len(near) >= 2:
            # the ranks of one NUMA node split ITS cpus by their position among the node's ranks; without knowing the others' nodes the even
            # slice of the node's CPUs by (local_rank modulo ranks per node) is taken, assuming GPUs are spread evenly over the nodes
            nodes = max(1, round(len(allowed) / len(near)))
            per_node_ranks = max(1, -(-n_local // nodes))
            k = local_rank % per_node_ranks
            per2 = max(1, len(near) // per_node_ranks)
            mine = set(near[k * per2:(k + 1) * per2] or near)
    return mine


def ranks_seen(dist, device):
    """Number of ranks that took part, counted by the collective itself (an all-reduce of ones: RCCL on the GPU box)."""
    import torch
    t = torch.ones(1, dtype=torch.float32, device=device)
    dist.all_reduce(t)
    return int(round(float(t.item())))


def preheat(step_fn, sync_fn, seconds=0.4, batch=8):
    """Untimed GPU work in front of the W warm-up steps: a device that sat idle (a fresh process on a box another process just left) ramps
    its clocks over the first few hundred milliseconds of load -- with W = 20 forward steps (20 ms) the timed region of the SECOND and later
    processes on a box measured 1.3-1.5 ms per step where the first, and any run with W >= 200, measured 0.90 ms (round 3: `--warmup 20`
    0.904 / 1.465; `--warmup 200` 0.907 / 0.904; `--warmup 1000` 0.908 / 0.907).  The W warm-up steps and the K timed steps follow unchanged."""
    if os.environ.get("PS_BENCH_NO_PREHEAT"):  # (counter passes under rocprofv3 --pmc: a known number of steps, profiles/run_pmc_train.sh)
        return 0
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(batch):
            step_fn()
        sync_fn()
        n += batch
    return n


def timed_region(step, steps, sync, dist=None):
    """Times exactly `steps` calls of `step` bracketed by barrier + device sync on both sides; returns the MAX over ranks
    (seconds) and the last step's result.  `sync()` drains the device; `dist` is torch.distributed or None."""
    sync()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    out = None
    for _ in range(steps):
        out = step()
    sync()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, out


def whole_job_value(world, batch_per_gpu, points, steps, elapsed):
    """points/s over ALL ranks: every rank processes batch_per_gpu clouds of `points` points per step (weak scaling)."""
    return world * batch_per_gpu * points * steps / elapsed


BF16_MFMA_PEAK_TF = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA


def cpu_baseline_train(cfg, params, points, seed=0):
    """The oracle's training step (torch-CPU fp32 autograd over the reference graph + the oracle KNN pyramid on one thread, as
    the reference runs it) on ONE cloud of `points` points -- a bounded sample of the B x 180 000-point step."""
    import torch
    from oracle import bindings as ob
    from oracle import randla_oracle as ro
    from oracle import randla_train_oracle as rto
    cores = os.cpu_count()
    torch.set_num_threads(min(cores, 32))  # (every core of a 256-core host is far slower: see cpu_baseline)
    xyz = brats_cloud(points, 4242 + seed)[None]
    rng = np.random.default_rng(seed)
    feats = np.concatenate([xyz, rng.standard_normal((1, points, cfg.in_channels - 3)).astype(np.float32)], -1)
    labels = rng.integers(0, cfg.num_classes, (1, points))
    t0 = time.perf_counter()
    pts, nbr, pool, up = ro.build_pyramid(lambda s, q, k: ob.knn_batch(s, q, k, threads=1), xyz, cfg.k_n, cfg.sub_sampling_ratio[:cfg.num_layers])
    t1 = time.perf_counter()
    rto.train_step(params, cfg.num_layers, pts, nbr, pool, up, feats, labels, np.ones(cfg.num_classes), lr=1e-4, dtype=torch.float32)
    t2 = time.perf_counter()
    return dict(value=points / (t2 - t0), unit="points/s", cores=cores, kind="port",
                sample="one training step (pyramid + forward + backward + Adam) on 1 cloud of %d points: KNN pyramid %.2f s on 1 thread + torch-CPU "
                       "fp32 autograd step %.2f s on %d threads" % (points, t1 - t0, t2 - t1, min(cores, 32)),
                knn_seconds=t1 - t0, step_seconds=t2 - t1)


def train_roofline(cfg, n0, B, ms, bf16):
    """Whole-step roofline of the training step: forward + input-gradient + weight-gradient GEMMs = 3 x the forward's algorithmic FLOPs
    (SURVEY 8d figure x 3); bytes: the backward reads every activation once more and writes its gradient (x 3).  Against the dtype's
    MFMA peak and HBM."""
    fwd = algorithmic_costs(cfg, n0, B)
    net = [v for k, v in fwd.items() if not k.startswith(("knn", "kdtree", "pyramid"))]
    step_flops, step_bytes = 3 * sum(v["flops"] for v in net), 3 * sum(v["bytes"] for v in net)
    peak = BF16_MFMA_PEAK_TF if bf16 else F32_MFMA_PEAK_TF
    tfs, gbs = step_flops / (ms * 1e-3) / 1e12, step_bytes / (ms * 1e-3) / 1e9
    mf = tfs / peak > gbs / HBM_PEAK_GBS
    out = {"bound": "mfma" if mf else "hbm", "kernel": "whole training step", "achieved": round(tfs if mf else gbs, 3), "peak": peak if mf else HBM_PEAK_GBS,
           "unit": "TFLOP/s" if mf else "GB/s", "frac": round(max(tfs / peak, gbs / HBM_PEAK_GBS), 5), "traffic": None,
           "algorithmic_flops_per_step": step_flops, "algorithmic_bytes_per_step": step_bytes,
           "mfma_frac": round(tfs / peak, 5), "hbm_frac": round(gbs / HBM_PEAK_GBS, 5)}
    # HBM bytes of one step from the committed rocprofv3 --pmc passes of `bench.py --mode train` (profiles/regen_r4.sh), labelled with their source
    pmc = next((q for q in (os.path.join(ROOT, "profiles", "r%d_pmc_traffic_train.json" % k) for k in (6, 5, 4)) if os.path.exists(q)), "")
    key = "b%d_%s" % (B, "bf16" if bf16 else "f32")
    if pmc:
        t = json.load(open(pmc))
        if key in t:
            out["traffic"] = t[key]["bytes_per_step"]
            out["traffic_over_algorithmic"] = round(t[key]["bytes_per_step"] / step_bytes, 3)
            out["traffic_source"] = "profiles/%s: rocprofv3 --pmc passes of `%s` at commit %s (not measured in this run)" % (
                os.path.basename(pmc), t[key].get("command", "?"), t.get("_commit", "?"))
    return out


def sub_train(cfg, xyz_list, local_rank, bf16, steps=5, warmup=1, deterministic=True):
    """BASELINE configs[2] (batch 8) / the per-rank work of configs[3] (batch 1) inside the default line: B x 180 000-point clouds, pyramid +
    forward + backward + Adam through ps_randla_train_step; `steps` timed steps after `warmup` (the first step also grows the activation
    pool), every step also bracketed by its own event pair: `step_ms` = min / median / max of the single steps."""
    import torch
    from point_unet_amd import runtime, weights
    from point_unet_amd.pyramid import alloc_pyramid, build_pyramid
    from point_unet_amd.train import Trainer
    B = len(xyz_list)
    xyz = np.concatenate(xyz_list, 0)
    n0 = xyz.shape[1]
    rng = np.random.default_rng(7)
    feats = np.concatenate([xyz, rng.standard_normal((B, n0, cfg.in_channels - 3)).astype(np.float32)], -1)
    labels = rng.integers(0, cfg.num_classes, (B, n0)).astype(np.int32)
    ctx = runtime.default_context(local_rank)
    ctx.use_torch_stream()
    ctx.set_deferred_checks(False)
    tr = Trainer(cfg, params=weights.init_params(cfg, seed=2), device=local_rank, ctx=ctx, keep_prob=0.5, mlp_dtype="bf16" if bf16 else "fp32",
                 deterministic=deterministic)
    d_xyz, d_feats, d_lab = torch.from_numpy(xyz).cuda(), torch.from_numpy(feats).cuda(), torch.from_numpy(labels).cuda()
    pyr = alloc_pyramid(B, n0, cfg.sub_sampling_ratio[:cfg.num_layers], cfg.k_n, d_xyz.device)
    marks = []

    def step():
        build_pyramid(d_xyz, cfg, ctx=ctx, out=pyr)
        return tr.train_step(pyr, d_feats, d_lab)

    def marked_step():
        out = step()
        e = torch.cuda.Event(enable_timing=True)
        e.record()  # (the context runs on torch's current stream here: ctx.use_torch_stream())
        marks.append(e)
        return out

    def sync():
        ctx.synchronize()
        torch.cuda.synchronize()

    step()  # (the first step also grows the activation pool)
    sync()
    pool_first = tr.pool_peak_bytes()
    preheat(step, sync, 0.25, batch=2)
    for _ in range(warmup):
        step()
    sync()
    ctx.timing_begin()  # one more untimed step with hipEvent pairs around every op group: launches per step
    step()
    sync()
    launches = sum(ln for _, _, ln in ctx.timing_end())
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record()
    marks.append(e0)
    elapsed, loss = timed_region(marked_step, steps, sync, None)
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(len(marks) - 1))
    ms = 1e3 * elapsed / steps
    out = {"ms_per_step": ms, "points_per_s": B * n0 * steps / elapsed, "steps": steps, "warmup": warmup, "batch": B, "points": n0,
           "step_ms": {"min": round(per_step[0], 3), "median": round(per_step[len(per_step) // 2], 3), "max": round(per_step[-1], 3)},
           "launches_per_step": launches,
           "dtype": "bf16" if bf16 else "f32", "loss": float(loss), "pool_peak_gb": tr.pool_peak_bytes() / 2 ** 30,
           "pool_peak_gb_after_first_step": pool_first / 2 ** 30,
           "gradients": "bit-reproducible (fixed-order reductions, csrc/invidx.hip)" if deterministic else "float-atomic scatter-adds (repeat to ~2e-6)",
           "roofline": train_roofline(cfg, n0, B, ms, bf16),
           "what": "BASELINE configs[%s]: one training step (pyramid + train-mode forward + weighted CE + backward + Adam) = ONE ps_pyramid_build + ONE "
                   "ps_randla_train_step call, batch %d x %d points, %s" % ("2" if B > 1 else "3], the work of ONE rank [no collective", B, n0,
                                                                            "bf16 MLP GEMMs (fp32 accumulate), rest fp32" if bf16 else "fp32")}
    tr.close()
    del tr, pyr, d_xyz, d_feats, d_lab
    torch.cuda.empty_cache()
    return out


def sub_batch2(cfg, params, d_clouds, local_rank, lanes, reuse, steps=40, warmup=8):
    """TWO clouds of configs[1] per launch on every lane (ps_pyramid_build / ps_randla_forward with B = 2): what the launch-level effects of
    batch 1 cost -- tails, 352-tile grids on 256 CUs, the launch floor of the deep levels.  Information only: the reference runs batch 1
    (helper_tool.py:29) and so does the headline.  Returns (sub-result, the pipeline whose lanes the next sub-result takes over)."""
    import torch
    from point_unet_amd.pipeline import ForwardPipeline
    pairs = [(torch.cat([d_clouds[2 * i][0], d_clouds[2 * i + 1][0]], 0), torch.cat([d_clouds[2 * i][1], d_clouds[2 * i + 1][1]], 0))
             for i in range(len(d_clouds) // 2)]
    pipe = ForwardPipeline(cfg, params=params, device=local_rank, lanes=lanes, reuse=reuse)
    pipe.prime(*pairs[0])
    k = [0]

    def step():
        x, f = pairs[k[0] % len(pairs)]
        k[0] += 1
        return pipe.submit(x, f)

    def sync():
        pipe.synchronize()
        torch.cuda.synchronize()

    preheat(step, sync, 0.2)
    for _ in range(warmup):
        step()
    elapsed, logits = timed_region(step, steps, sync, None)
    assert bool(torch.isfinite(logits).all())
    n0 = pairs[0][0].shape[1]
    out = {"ms_per_cloud": 1e3 * elapsed / (2 * steps), "points_per_s": 2 * n0 * steps / elapsed, "steps": steps, "warmup": warmup, "clouds_per_launch": 2,
           "lanes": lanes, "what": "two clouds per launch on every lane (B = 2 through the same C-ABI calls): NOT the headline -- the reference runs batch 1"}
    del pairs
    return out, pipe


def sub_config5(local_rank, lanes, reuse, steps=40, warmup=8, n_clouds=4):
    """BASELINE configs[4] inside the default line: 262 144-point cloud, K = 32, 4 input channels, 2 classes, fp16 feature hand-over,
    int32 indices; pyramid + forward, `lanes` clouds in flight and one cloud in flight."""
    import torch
    from point_unet_amd import weights
    from point_unet_amd.helper_tool import ConfigBraTS
    from point_unet_amd.pipeline import ForwardPipeline

    class cfg5(ConfigBraTS):
        k_n, num_classes, in_channels = 32, 2, 4

    n0 = 262144
    clouds = []
    for i in range(n_clouds):
        x = brats_cloud(n0, 5000 + 17 * i)[None]
        f = np.concatenate([x, np.random.default_rng(31 * i).standard_normal((1, n0, 1)).astype(np.float32)], -1).astype(np.float16)
        clouds.append((torch.from_numpy(x).cuda(), torch.from_numpy(f).cuda()))
    # on the headline pipeline's own streams and contexts: four MORE streams in the process cost this configuration 20 % pipelined and
    # 60 % serial (ForwardPipeline.__init__, profiles/tools/exp_second_pipeline.py)
    pipe = ForwardPipeline(cfg5, params=weights.init_params(cfg5, seed=2, randomize_bn=True), device=local_rank, lanes=lanes, reuse=reuse)
    pipe.prime(*clouds[0])
    k = [0]

    def step(overlap=True, lane=None):
        x, f = clouds[k[0] % n_clouds]
        k[0] += 1
        return pipe.submit(x, f, overlap=overlap, lane=lane)

    def sync():
        pipe.synchronize()
        torch.cuda.synchronize()

    preheat(step, sync, 0.2)
    for _ in range(warmup):
        step()
    elapsed, logits = timed_region(step, steps, sync, None)
    assert bool(torch.isfinite(logits).all())
    n_serial = max(4, steps // 4)
    t_serial, _ = timed_region(lambda: step(lane=0), n_serial, sync, None)  # (one cloud in flight: consecutive clouds on one lane)
    costs = algorithmic_costs(cfg5, n0, 1)
    net_flops = sum(v["flops"] for kk, v in costs.items() if not kk.startswith(("knn", "kdtree", "pyramid")))
    out = {"ms_per_step": 1e3 * elapsed / steps, "points_per_s": n0 * steps / elapsed, "steps": steps, "warmup": warmup,
           "serial_one_lane_ms_per_cloud": 1e3 * t_serial / n_serial, "points": n0, "k_n": 32, "lanes": lanes, "dtype": "f32 (fp16 feature input, int32 indices)",
           "algorithmic_gflop_per_step": net_flops / 1e9, "achieved_tflops_serial": net_flops / (t_serial / n_serial) / 1e12,
           "what": "BASELINE configs[4]: 262 144-point cloud, K=32, 4 input channels, 2 classes, features handed over as fp16, pyramid + forward"}
    pipe.close()
    del pipe, clouds
    torch.cuda.empty_cache()
    return out


def sub_train_ranks(cfg, n0, rank, local_rank, world, dist, local_bn, steps=8, warmup=2):
    """BASELINE configs[3] inside the default line of an N > 1 run (every rank calls this): one 180 000-point cloud per GPU, pyramid + forward +
    backward + ONE all-reduce of the flat gradient buffer + Adam through ps_randla_train_step, BatchNorm statistics shared by all ranks
    (two small all-reduces per BatchNorm layer) or kept per GPU (local_bn: the reference's own batch-1 semantics, helper_tool.py:29).
    Timed like the headline: barrier + device sync on both sides, MAX over ranks; one more profiled step gives the collective counts."""
    import torch
    from point_unet_amd import runtime, weights
    from point_unet_amd.pyramid import alloc_pyramid, build_pyramid
    from point_unet_amd.train import Trainer
    xyz = np.stack([brats_cloud(n0, 1000 * rank + 5)])
    rng = np.random.default_rng(7 + rank)
    feats = np.concatenate([xyz, rng.standard_normal((1, n0, cfg.in_channels - 3)).astype(np.float32)], -1)
    labels = rng.integers(0, cfg.num_classes, (1, n0)).astype(np.int32)
    ctx = runtime.default_context(local_rank)
    ctx.use_torch_stream()
    ctx.set_deferred_checks(False)
    tr = Trainer(cfg, params=weights.init_params(cfg, seed=2), device=local_rank, ctx=ctx, keep_prob=0.5, sync_bn=not local_bn)
    d_xyz, d_feats, d_lab = torch.from_numpy(xyz).cuda(), torch.from_numpy(feats).cuda(), torch.from_numpy(labels).cuda()
    pyr = alloc_pyramid(1, n0, cfg.sub_sampling_ratio[:cfg.num_layers], cfg.k_n, d_xyz.device)

    def step():
        build_pyramid(d_xyz, cfg, ctx=ctx, out=pyr)
        return tr.train_step(pyr, d_feats, d_lab, dist=dist)

    def sync():
        ctx.synchronize()
        torch.cuda.synchronize()

    for _ in range(1 + warmup):  # (the first step also grows the activation pool; every rank takes the same number of steps)
        step()
    sync()
    elapsed, loss = timed_region(step, steps, sync, dist)
    tr.set_profile(True)
    step()
    sync()
    coll = tr.collective_stats()
    tr.set_profile(False)
    finite = bool(torch.isfinite(loss).all())
    tr.close()
    ms = 1e3 * elapsed / steps
    return {"ms_per_step": ms, "points_per_s": whole_job_value(world, 1, n0, steps, elapsed), "steps": steps, "warmup": warmup, "loss_finite": finite,
            "batchnorm": "per GPU" if local_bn else "shared by all ranks", "collectives_per_step": coll["calls"], "collective_bytes_per_step": coll["bytes"],
            "collective_ms": round(coll["device_ms"], 4), "collective_host_ms": round(coll["host_ms"], 4)}


def bench_train(args, cfg, rank, local_rank, world, dist):
    """BASELINE configs[2] (--batch 8, 1 GPU) / configs[3] (--gpus 8 --batch 1): one training step = index pyramid +
    training-mode forward + class-weighted CE + backward + (all-reduce of the flat gradient buffer) + Adam."""
    import torch
    from point_unet_amd import runtime, weights
    from point_unet_amd.pyramid import alloc_pyramid, build_pyramid
    from point_unet_amd.train import Trainer
    B, n0 = args.batch, args.points
    xyz = np.stack([brats_cloud(n0, 1000 * rank + b) for b in range(B)])
    rng = np.random.default_rng(7 + rank)
    feats = np.concatenate([xyz, rng.standard_normal((B, n0, cfg.in_channels - 3)).astype(np.float32)], -1)
    labels = rng.integers(0, cfg.num_classes, (B, n0)).astype(np.int32)
    ctx = runtime.default_context(local_rank)
    params = weights.init_params(cfg, seed=2)
    tr = Trainer(cfg, params=params, device=local_rank, ctx=ctx, keep_prob=0.5, sync_bn=dist is not None and not args.local_bn,
                 mlp_dtype="bf16" if args.bf16_mlp else "fp32", fused_att=not args.no_fused_att, fused_locse=not args.no_fused_locse, fused_convbn=True if args.fused_convbn else (False if args.no_fused_convbn else None),
                 engine=args.train_engine, deterministic=not args.atomic_scatter, overlap_wgrad=args.overlap_wgrad)
    # under a ONE-rank process group the step still goes through the all-reduce callback: that is how collectives_per_step / collective_ms
    # are measured on a 1-GPU box; a training script would let the trainer skip it
    tr.collective_at_world_one = True
    d_xyz, d_feats, d_lab = torch.from_numpy(xyz).cuda(), torch.from_numpy(feats).cuda(), torch.from_numpy(labels).cuda()
    pyr = alloc_pyramid(B, n0, cfg.sub_sampling_ratio[:cfg.num_layers], cfg.k_n, d_xyz.device)
    seen = ranks_seen(dist, "cuda" if args.dist_backend == "nccl" else "cpu") if dist is not None else 1

    pre = None
    if not args.train_prefetch:
        def step():
            build_pyramid(d_xyz, cfg, ctx=ctx, out=pyr)
            return tr.train_step(pyr, d_feats, d_lab, dist=dist)
    else:
        # the next batch's pyramid is built on its own stream while this batch trains (the reference's tf.data map + prefetch,
        # runBraTS.py:166-185): every step still enqueues ONE pyramid build and ONE training step
        from point_unet_amd.pipeline import PyramidPrefetcher
        pre = PyramidPrefetcher(cfg, device=local_rank)
        pre.submit(d_xyz)

        def step():
            pre.submit(d_xyz)
            p_k, slot = pre.next()
            loss_k = tr.train_step(p_k, d_feats, d_lab, dist=dist)
            pre.release(slot)
            return loss_k

    def sync():
        if pre is not None:
            pre.synchronize()
        ctx.synchronize()
        torch.cuda.synchronize()

    step()  # (the first step also grows the activation pool)
    preheat(step, sync, 0.3, batch=2)
    for _ in range(args.warmup):
        step()
    sync()
    prof_rows, prof_steps = [], 0
    if not args.no_stage_timing:  # hipEvent pairs around every op group of two steps, outside the timed region
        prof_steps = 2
        ctx.timing_begin()
        for _ in range(prof_steps):
            step()
        sync()
        prof_rows = ctx.timing_end()
    elapsed, loss = timed_region(step, args.steps, sync, dist)
    assert bool(torch.isfinite(loss).all())
    sections, collectives = None, None
    if tr.engine == "native" and not args.no_stage_timing:  # per-level device time of one more step (outside the timed region)
        tr.set_profile(True)
        step()
        sync()
        sections = [{"name": nm, "ms": round(t, 4)} for nm, t in tr.profile()]
        collectives = tr.collective_stats()  # (of that profiled step: event pairs around every callback)
        tr.set_profile(False)
    if rank == 0:
        ms = 1e3 * elapsed / args.steps
        stages = sorted(({"name": nm, "ms_per_step": round(t / prof_steps, 4), "launches_per_step": ln / prof_steps} for nm, t, ln in prof_rows),
                        key=lambda r: -r["ms_per_step"])
        roofline = train_roofline(cfg, n0, B, ms, args.bf16_mlp)
        roofline["dominant_stage"] = stages[0] if stages else None
        roofline["measured"] = "timed region (whole step); stages: hipEvent pairs over %d extra steps" % prof_steps
        out = {
            "metric": "points_per_sec_train_step", "value": whole_job_value(world, B, n0, args.steps, elapsed), "unit": "points/s",
            "n_gpus": world, "ranks_seen": seen, "rank0_pinned_cpus": args.pinned_cpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            # bf16: BASELINE configs[2]'s "bf16 MLPs" -- the shared-MLP GEMMs on bf16 operands with fp32 accumulate, the rest fp32
            "dtype": "bf16" if args.bf16_mlp else "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[%d]: training step (pyramid + train-mode forward + weighted CE + backward + Adam), %d-point "
                                   "BraTS-shaped clouds, batch %d per GPU, K=16, 5 levels, %s%s" % (
                                       2 if world == 1 else 3, n0, B, "bf16 MLP GEMMs (fp32 accumulate), rest fp32" if args.bf16_mlp else "fp32",
                                       (", gradient all-reduce over RCCL, BatchNorm statistics %s" % ("per GPU" if args.local_bn else "shared by all ranks")) if world > 1 else ""),
                       "points": n0, "batch_per_gpu": B, "parameters": tr.num_params(),
                       "pyramid": "built in front of its training step, one stream" if pre is None else
                       "the next batch's pyramid is built on a second HIP stream while this batch trains (one build + one training step per timed step)"},
            "roofline": roofline, "sections": sections, "stages": stages, "launches_per_step": sum(r["launches_per_step"] for r in stages) if stages else None,
            "engine": tr.engine, "loss": float(loss), "peak_mem_gb": torch.cuda.max_memory_allocated() / 2 ** 30,
            "pool_peak_gb": tr.pool_peak_bytes() / 2 ** 30 if tr.engine == "native" else None,
            # calls into the all-reduce callback per step (1 flat gradient buffer + 2 per BatchNorm layer with shared statistics), the bytes they
            # carry, the host time inside the callback and the device time between event pairs around every call (one profiled step)
            "collectives_per_step": collectives["calls"] if collectives else None,
            "collective_bytes_per_step": collectives["bytes"] if collectives else None,
            "collective_ms": round(collectives["device_ms"], 4) if collectives else None,
            "collective_host_ms": round(collectives["host_ms"], 4) if collectives else None,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_train(cfg, params, min(n0, 45000))
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    tr.close()  # (the activation pool is the trainer's own device memory, not torch's)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # 200 timed steps = ~0.25 s of forward work: the fill and drain of the four-cloud pipeline (about one serial step, 2 ms) stay
    # inside the timed region and are amortised to < 1 % instead of ~8 % at 20 steps
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--points", type=int, default=180000)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--mode", choices=["forward", "train"], default="forward",
                    help="forward = the headline metric (BASELINE configs[1]); train = forward+backward+Adam step (configs[2]/[3])")
    ap.add_argument("--workload", choices=["config2", "config5"], default="config2",
                    help="config2 = BASELINE configs[1] (180k BraTS-shaped, K=16); config5 = BASELINE configs[4] (262 144 points, K=32, "
                         "4 input channels, 2 classes; features handed over as fp16 and widened on the device, int32 indices)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="serial steps on one stream (per-cloud latency) instead of several clouds in flight on separate HIP streams "
                         "(point_unet_amd/pipeline.py)")
    ap.add_argument("--lanes", type=int, default=4, help="clouds in flight per GPU (pipeline lanes, one HIP stream each)")
    ap.add_argument("--coalesce", type=int, default=1, choices=[1, 2],
                    help="pipelined forward: clouds per launch of the HEADLINE (1 = one cloud per launch, the definition of every round; 2 = "
                         "ForwardPipeline(coalesce=2): consecutive single clouds run as pairs through the same C-ABI calls -- reported as the "
                         "sub-result `coalesced_pairs` of the default line either way)")
    ap.add_argument("--include-pcie", action="store_true",
                    help="every step also copies its inputs (xyz, features) from pinned host memory and its logits back: the "
                         "PCIe-inclusive rate DESIGN.md quotes next to the headline (which keeps inputs resident in HBM)")
    ap.add_argument("--bf16-mlp", action="store_true", help="train mode: shared-MLP GEMMs on bf16 operands with fp32 accumulate (BASELINE configs[2])")
    ap.add_argument("--no-fused-att", action="store_true", help="train mode: the op-by-op attentive pooling at every level (A/B of csrc/attpool_train.hip)")
    ap.add_argument("--fused-convbn", action="store_true", help="train mode: LFA mlp2 in the recompute form (csrc/smallconv_train.hip, convbn_rows.hip); the native engine's default")
    ap.add_argument("--no-fused-convbn", action="store_true", help="train mode: LFA mlp2 op by op (A/B switch)")
    ap.add_argument("--no-fused-locse", action="store_true", help="train mode: the op-by-op LocSE branch (A/B of csrc/locse_train.hip)")
    ap.add_argument("--train-engine", choices=["native", "python"], default="native",
                    help="train mode: native = ps_randla_train_step, the whole step behind one C-ABI call (csrc/trainer.hip); python = the host-side "
                         "tape over the same op-level kernels (A/B)")
    ap.add_argument("--train-prefetch", action="store_true",
                    help="train mode: build the NEXT batch's pyramid on a second stream while this batch trains (point_unet_amd/pipeline.py: "
                         "PyramidPrefetcher, the counterpart of the reference's tf.data prefetch) instead of in front of its own training step.  "
                         "Measured 51.95 against 52.21 ms per step at batch 8: the searches and the training kernels do not share the chip well")
    ap.add_argument("--atomic-scatter", action="store_true",
                    help="train mode: the scatter-adds of the backward pass with float atomics (run-to-run differences in the last bits) instead of "
                         "fixed-order gather-reductions over inverse indices (A/B of csrc/invidx.hip)")
    ap.add_argument("--overlap-wgrad", action="store_true", help="train mode: weight-gradient products on a second HIP stream of the trainer (A/B: measured slower)")
    ap.add_argument("--local-bn", action="store_true", help="train mode, N > 1: per-GPU BatchNorm statistics instead of statistics shared by all ranks")
    ap.add_argument("--clouds", type=int, default=8, help="distinct resident clouds every rank rotates through (one per step)")
    ap.add_argument("--no-sub-results", action="store_true", help="skip the PCIe-inclusive sub-result of the default line")
    ap.add_argument("--att-fp32-mfma", action="store_true",
                    help="forward: attentive pooling at d_out = 64 / 128 on the fp32 MFMA instead of bf16 MFMA over exact three-way splits (A/B)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stage-timing", action="store_true", help="do not record hipEvents around the stages (A/B of their cost)")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="process-group backend (nccl = RCCL, the default; gloo lets several ranks share ONE GPU for a functional check "
                         "of the N > 1 path on a single-GPU box together with --share-gpu)")
    ap.add_argument("--share-gpu", action="store_true", help="map every rank onto the GPUs that exist (local_rank %% device_count)")
    ap.add_argument("--no-pin", action="store_true", help="N > 1: do not pin the rank's process to a slice of the CPUs next to its GPU")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # plain `python bench.py --gpus N`: no launcher set the rendezvous up, so this process (which has not touched the GPU and
        # never will) starts the N ranks itself and relays rank 0's line
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))

    pinned = None
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and not args.no_pin and hasattr(os, "sched_setaffinity"):
        # BEFORE any GPU call (and before torch starts its thread pools): this rank's own slice of the CPUs next to its GPU
        lr, nl = int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
        try:
            allowed = os.sched_getaffinity(0)
            mine = rank_cpu_slice(lr, nl, allowed, None if args.share_gpu else gpu_local_cpus(lr))
            if mine and len(allowed) >= 2 * nl:  # (fewer than two CPUs per rank: pinning would only serialise the rank's own threads)
                os.sched_setaffinity(0, mine)
                pinned = len(mine)
        except OSError:
            pinned = None
    args.pinned_cpus = pinned

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.share_gpu:
        local_rank %= max(torch.cuda.device_count(), 1)
    elif local_rank >= torch.cuda.device_count():
        raise SystemExit("rank %d needs GPU %d but this node exposes %d (use --share-gpu --dist-backend gloo for a functional "
                         "check on fewer GPUs)" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    want_dist = world > 1 or ("MASTER_ADDR" in os.environ and "RANK" in os.environ)  # launched by torch.distributed.run

    def init_dist():
        import torch.distributed as dist
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
        return dist

    # Training needs the process group from the start.  The forward path has no data-path collective; its process group (the
    # barriers and the MAX-reduce of the elapsed time around the timed region) is created AFTER the lanes' HIP streams exist
    # and have run once: HIP multiplexes streams onto a handful of hardware queues in creation order, and RCCL's internal
    # streams, created first, pushed two lanes onto one queue (measured 1.72 ms/step against 1.31 without a process group).
    dist = init_dist() if (want_dist and args.mode == "train") else None

    from point_unet_amd import runtime, weights
    from point_unet_amd.helper_tool import ConfigBraTS
    from point_unet_amd.RandLANet import Network
    from point_unet_amd.pyramid import alloc_pyramid, build_pyramid

    cfg = ConfigBraTS
    if args.workload == "config5":
        class cfg(ConfigBraTS):  # Pancreas-shaped: runPancreas.py:118,125 (xyz + 1 CT value), 2 classes; K=32 per BASELINE configs[4]
            k_n, num_classes, in_channels = 32, 2, 4
        if args.points == 180000:
            args.points = 262144
    B, n0 = args.batch, args.points
    if args.mode == "train":
        return bench_train(args, cfg, rank, local_rank, world, dist)
    # one volume per GPU and step; every rank rotates through `--clouds` DISTINCT resident clouds (seeded by rank), so a step never
    # finds its own inputs, trees or index tables from the previous step in L2 / MALL
    n_clouds = max(1, args.clouds)
    xyz_all = [np.stack([brats_cloud(n0, 1000 * rank + 17 * i + b) for b in range(B)]) for i in range(n_clouds)]
    if os.environ.get("PS_BENCH_SORTED"):  # EXPERIMENT ONLY (profiles/tools): clouds stored in Morton order -- what a spatially coherent layout is worth
        def morton(x):
            ijk = np.rint(x * np.array([240, 240, 155])).astype(np.uint64)
            code = np.zeros(len(x), np.uint64)
            for b in range(8):
                for a in range(3):
                    code |= ((ijk[:, a] >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b + a)
            return np.argsort(code, kind="stable")
        xyz_all = [np.stack([c[morton(c)] for c in x]) for x in xyz_all]
    feats_all = [np.concatenate([x, np.random.default_rng(7 + rank + 31 * i).standard_normal((B, n0, cfg.in_channels - 3)).astype(np.float32)], -1)
                 for i, x in enumerate(xyz_all)]
    xyz, feats = xyz_all[0], feats_all[0]
    params = weights.init_params(cfg, seed=2, randomize_bn=True)
    half = args.workload == "config5"  # configs[4]: fp16 feature input
    d_clouds = [(torch.from_numpy(x).cuda(), torch.from_numpy(f.astype(np.float16) if half else f).cuda()) for x, f in zip(xyz_all, feats_all)]
    d_xyz, d_feats = d_clouds[0]
    counter = [0]

    def next_cloud():
        k = counter[0] % n_clouds
        counter[0] += 1
        return k

    pipe = None
    if args.no_pipeline:
        ctx = runtime.default_context(local_rank)
        ctx.set_deferred_checks(True)  # status words of the tree build are validated at ctx.synchronize()
        net = Network(cfg, params=params, device=local_rank, ctx=ctx)
        pyr = alloc_pyramid(B, n0, cfg.sub_sampling_ratio[:cfg.num_layers], cfg.k_n, d_xyz.device)
        contexts = [ctx]

        def step(overlap=True):
            x, f = d_clouds[next_cloud()]
            build_pyramid(x, cfg, ctx=ctx, out=pyr)
            return net.inference({"pyramid": pyr, "features": f})

        def sync():
            ctx.synchronize()
            torch.cuda.synchronize()
    else:
        # consecutive clouds on consecutive lanes (one HIP stream each): the latency-bound pyramid of one cloud shares the
        # chip with the network kernels of the others; every step still does all of its work
        from point_unet_amd.pipeline import ForwardPipeline
        pipe = ForwardPipeline(cfg, params=params, device=local_rank, lanes=args.lanes, coalesce=args.coalesce if B == 1 else 1)
        contexts = pipe.contexts
        pipe.prime(d_xyz, d_feats)  # every lane's workspace allocated before the warmup / timed steps

        def step(overlap=True, lane=None):
            x, f = d_clouds[next_cloud()]
            return pipe.submit(x, f, overlap=overlap, lane=lane)

        def sync():
            pipe.synchronize()
            torch.cuda.synchronize()

    if args.att_fp32_mfma:
        for cx in contexts:
            cx.set_att_bf16x3(False)
    pcie_step = None
    if pipe is not None and (args.include_pcie or not args.no_sub_results):
        h_in = [(torch.from_numpy(x).pin_memory(), torch.from_numpy(f.astype(np.float16) if half else f).pin_memory()) for x, f in zip(xyz_all, feats_all)]
        h_out = [[torch.empty((B, n0, cfg.num_classes), dtype=torch.float32).pin_memory() for _ in range(2)] for _ in range(args.lanes)]
        d_in = [(torch.empty_like(d_xyz), torch.empty_like(d_feats)) for _ in range(args.lanes)]  # per-lane device input slots
        held = [[] for _ in range(args.lanes)]  # logits whose copy back waits for the launch of their pair (coalesced mode)

        def pcie_step(overlap=True):
            # host -> device, compute, device -> host all on the lane's own stream (no extra streams: they would compete with the
            # lanes for hardware queues); the copies of one lane overlap the kernels of the others.  Coalesced mode: the first cloud of a
            # pair waits in the lane's input slot, its logits are copied back behind the pair's launch together with the partner's
            k = pipe.next_lane()
            lane = pipe.lanes[k]
            dx, df = d_in[k]
            hx, hf = h_in[next_cloud()]
            with torch.cuda.stream(lane.stream):
                dx.copy_(hx, non_blocking=True)
                df.copy_(hf, non_blocking=True)
                out = pipe.submit(dx, df, overlap=overlap)
                held[k].append(out)
                if pipe.launched:
                    for j, o in enumerate(held[k]):
                        h_out[k][j % 2].copy_(o, non_blocking=True)
                    held[k].clear()
            return out
    main_step = pcie_step if args.include_pcie else step
    if args.include_pcie and pcie_step is None:
        raise SystemExit("--include-pcie needs the pipelined mode")

    if want_dist and dist is None:
        dist = init_dist()
    seen = ranks_seen(dist, "cuda" if args.dist_backend == "nccl" else "cpu") if dist is not None else 1

    def timing_begin(only=None):
        for cx in contexts:
            cx.timing_begin(only=only)

    def timing_end():
        merged = {}  # the lanes' contexts report the same stage names: one row per stage
        for cx in contexts:
            for name, ms, launches in cx.timing_end():
                a = merged.setdefault(name, [0.0, 0])
                a[0] += ms
                a[1] += launches
        return [(k, v[0], v[1]) for k, v in merged.items()]

    preheat(main_step, sync)
    for _ in range(args.warmup):
        main_step()
    sync()
    # Profile pass (outside the timed region): hipEvent pairs on the launch stream around EVERY stage, steps serialised so
    # that no stage shares the chip with another stream's kernels.  Recording ~120 events per step costs ~0.35 ms of stream
    # time, so the timed region below keeps only the dominant stage's pair.
    prof_steps = 0 if args.no_stage_timing else max(3, min(args.steps, 10))
    prof_rows = []
    if prof_steps:
        timing_begin()
        for _ in range(prof_steps):
            step(overlap=False)
        sync()
        prof_rows = timing_end()
    # dominant KERNEL = the single-launch stage with the largest time (composite stages such as kdtree_build, ~46 small
    # launches, are listed in "stages" but are not one kernel)
    single = [r for r in prof_rows if r[2] == prof_steps]
    dominant = max(single, key=lambda r: r[1])[0] if single else None
    # The dominant stage is re-measured LIVE with its own event pair only: inside the timed region when steps are serial
    # (--no-pipeline); with several clouds in flight a kernel's event pair would also time the other lanes' kernels it shares
    # the chip with (and rocprofv3 --kernel-trace serialises dispatches, so its average could not agree), so the pair is armed
    # over the serial pass that follows the timed region instead (one cloud in flight, same kernels, same inputs).
    if dominant and args.no_pipeline:
        timing_begin(only=dominant)
    elapsed, logits = timed_region(main_step, args.steps, sync, dist)
    dom_rows, dom_steps, dom_where = [], args.steps, "timed region"
    if dominant and args.no_pipeline:
        dom_rows = [r for r in timing_end() if r[0] == dominant]
    assert bool(torch.isfinite(logits).all())
    serial_ms = None
    sub = {}
    if not args.no_pipeline:  # per-cloud latency next to the pipelined throughput
        dom_steps, dom_where = max(3, args.steps // 2), "serial pass after the timed region (one cloud in flight)"
        if dominant:
            timing_begin(only=dominant)
        t_serial, _ = timed_region(lambda: step(overlap=False), dom_steps, sync, None)
        if dominant:
            dom_rows = [r for r in timing_end() if r[0] == dominant]
        # ... and the same clouds one after the other on ONE lane: stream order alone, no cross-stream event between consecutive clouds
        # (rotating over the lanes, every cloud starts behind an event of another hardware queue: ~0.15 ms per cloud on this runtime)
        t_lane, _ = timed_region(lambda: step(lane=0), dom_steps, sync, None)
        # rounds 1-3 reported the rotating-lanes figure as "serial": it keeps that name (like-for-like over the rounds); the one-lane
        # figure introduced in round 4 has its own key
        serial_ms = 1e3 * t_serial / dom_steps
        one_lane_ms = 1e3 * t_lane / dom_steps
        sub["serial"] = {"ms_per_cloud": serial_ms, "points_per_s": B * n0 / (serial_ms * 1e-3), "steps": dom_steps,
                         "ms_per_cloud_one_lane": one_lane_ms,
                         "what": "one cloud in flight.  ms_per_cloud (the definition of rounds 1-3): every cloud on the next lane, waiting for an "
                                 "event of the previous cloud's stream; ms_per_cloud_one_lane (round 4's `serial`): consecutive clouds on ONE lane, "
                                 "stream order alone -- the per-cloud latency of pyramid + forward on this rank"}
        if B == 1 and not args.include_pcie and not half:
            # the service's throughput mode next to the headline, same run: consecutive clouds coalesced in pairs (ForwardPipeline(coalesce=2)),
            # over a region long enough for a steady state (the fill and drain of a pipeline of pairs are twice as long: at the driver's 20
            # steps the two modes measure the same) -- and the other mode over the same number of steps
            other = 1 if pipe.coalesce == 2 else 2
            n_long = max(args.steps, 200)
            res = {}
            for mode in (other, pipe.coalesce):
                pipe.synchronize()
                keep, pipe.coalesce = pipe.coalesce, mode
                for _ in range(max(8, args.warmup)):
                    step()
                sync()
                t_m, _ = timed_region(step, n_long, sync, None)
                pipe.synchronize()
                pipe.coalesce = keep
                res[mode] = 1e3 * t_m / n_long
            sub["coalesced_pairs"] = {"ms_per_cloud": res[2], "ms_per_cloud_one_per_launch": res[1], "points_per_s": B * n0 / (res[2] * 1e-3), "steps": n_long,
                                      "lanes": args.lanes,
                                      "what": "NOT the headline: ForwardPipeline(coalesce=2) -- two consecutive 180 000-point clouds per ps_pyramid_build / "
                                              "ps_randla_forward launch (a batch of two independent clouds, copied into the lane's input slots), against one "
                                              "cloud per launch over the same %d steps; this rank only" % n_long}
        if not args.no_sub_results and not args.include_pcie:
            # untimed warm-up of the service path itself: the first transfers through freshly pinned host buffers and fresh device slots
            # are slow (page registration with the DMA engines), and at the driver's --steps 20 they WERE the number (r2: 1.66 ms
            # against 1.02 once warm, profiles/tools/exp_pcie.py)
            n_warm = 0
            for _ in range(max(3 * n_clouds, 2 * args.lanes, args.warmup)):  # (every pinned host buffer has gone through the DMA engines)
                pcie_step()
                n_warm += 1
            sync()
            # ... then batches of --steps until two consecutive ones agree within 3 % (at most eight): the driver box measured 1.43 ms where the
            # builder's measured 1.23 with a fixed warm-up; the batch times say whether the path was still warming up when it was timed
            warm_batches = []
            while len(warm_batches) < 8:
                t_b, _ = timed_region(pcie_step, args.steps, sync, None)
                warm_batches.append(1e3 * t_b / args.steps)
                n_warm += args.steps
                if len(warm_batches) >= 2 and abs(warm_batches[-1] - warm_batches[-2]) <= 0.03 * warm_batches[-2]:
                    break
            t_pcie, _ = timed_region(pcie_step, args.steps, sync, None)
            # the same region once more in four quarters (a device sync between them: the lanes drain, so each quarter is a little slower than
            # the whole) -- first against last quarter shows a drift inside the timed region if there is one
            q = max(1, args.steps // 4)
            quarters = [1e3 * timed_region(pcie_step, q, sync, None)[0] / q for _ in range(4)]
            sub["include_pcie"] = {"ms_per_step": 1e3 * t_pcie / args.steps, "points_per_s": B * n0 * args.steps / t_pcie, "steps": args.steps,
                                   "warmup_steps": n_warm, "warmup_batches_ms_per_step": [round(v, 4) for v in warm_batches],
                                   "first_quarter_ms_per_step": round(quarters[0], 4), "last_quarter_ms_per_step": round(quarters[-1], 4),
                                   "what": "every step also copies its cloud (xyz + features) from pinned host memory and its logits back, on the "
                                           "lane's stream (this rank only; the service rate -- never the headline value); untimed warm-up until two "
                                           "consecutive batches of --steps agree within 3 %"}
        if not args.no_sub_results and not args.att_fp32_mfma and not args.include_pcie:
            # the same timed region with attentive pooling on the fp32 MFMA (the default runs it on bf16 MFMA over exact three-way splits
            # of the fp32 operands: same accuracy, see csrc/attpool32b.hip) -- the A/B number next to the headline, this rank only
            for cx in contexts:
                cx.set_att_bf16x3(False)
            for _ in range(max(4, args.warmup // 2)):
                step()
            sync()
            t_f32, _ = timed_region(step, args.steps, sync, None)
            for cx in contexts:
                cx.set_att_bf16x3(True)
            sub["att_fp32_mfma"] = {"ms_per_step": 1e3 * t_f32 / args.steps, "points_per_s": B * n0 * args.steps / t_f32, "steps": args.steps,
                                    "what": "attentive pooling at d_out >= 64 on the fp32 MFMA instead of bf16 MFMA over exact three-way splits "
                                            "(ps_set_att_bf16x3(ctx, 0)); both forms measure 4e-6 on the logits against the float64 restatement"}

    train_ranks = None
    if world > 1 and dist is not None and not args.no_sub_results and args.workload == "config2" and B == 1 and not args.include_pcie:
        # BASELINE configs[3] in the same run (every rank takes part; a few seconds): the training step with one cloud per GPU, both
        # BatchNorm forms -- the numbers an 8-GPU line is judged by next to the forward's scaling
        t_sub = time.perf_counter()
        if pipe is not None:
            pipe.close()
            pipe = None
        torch.cuda.empty_cache()
        train_ranks = {"sync_bn": sub_train_ranks(cfg, n0, rank, local_rank, world, dist, False),
                       "local_bn": sub_train_ranks(cfg, n0, rank, local_rank, world, dist, True)}
        train_ranks["seconds"] = round(time.perf_counter() - t_sub, 2)

    if rank == 0:
        costs = algorithmic_costs(cfg, n0, B)
        last_dec = "dec%d" % (cfg.num_layers - 1)
        if prof_rows and last_dec not in {r[0] for r in prof_rows}:
            # the last decoder step runs inside the head's layer chain (csrc/rowgemm.hip: rowchain): one stage, both costs
            costs["head"] = {k: costs["head"][k] + costs[last_dec][k] for k in ("flops", "bytes")}
            costs[last_dec] = dict(flops=0, bytes=0)
        stages = []
        live = {name: (ms / dom_steps, launches / dom_steps) for name, ms, launches in dom_rows}
        for name, ms, launches in prof_rows:
            per_step = ms / prof_steps
            launches = launches * args.steps / prof_steps
            cst = costs.get(name, dict(flops=0, bytes=0))
            gathered = cst["bytes"] / (per_step * 1e-3) / 1e9 if per_step > 0 else 0.0   # gather-counted
            gbs = cst.get("unique", cst["bytes"]) / (per_step * 1e-3) / 1e9 if per_step > 0 else 0.0  # every distinct byte once
            tfs = cst["flops"] / (per_step * 1e-3) / 1e12 if per_step > 0 else 0.0
            # attentive pooling at d_out >= 64 runs its products as SIX bf16 MFMAs per fp32 product (exact three-way splits,
            # csrc/attpool32b.hip): the matrix-pipe ceiling of those stages, in fp32 FLOPs, is the dense bf16 peak / 6
            split = (not args.att_fp32_mfma) and "_att" in name and name.startswith("enc") and cfg.d_out[int(name[3])] >= 64
            # ... and so do the single-launch decoder stages whose product is >= 0.3 GFLOP (csrc/gemm32b.hip; the [mlp2 ; shortcut] launches of
            # levels 2-4 share their `encN_dense` stage with fp32-MFMA launches and stay listed with the fp32 pipe)
            if (not args.att_fp32_mfma) and name in ("decoder_0", "dec0", "dec1", "dec2") and cst["flops"] >= 3e8 and os.environ.get("PS_GEMM32B_MIN_FLOPS") is None:
                split = True
            mfma_peak = BF16_MFMA_PEAK_TF / 6 if split else F32_MFMA_PEAK_TF
            f_h, f_m = gbs / HBM_PEAK_GBS, tfs / mfma_peak
            bound = "mfma" if f_m > f_h else "hbm"
            row = dict(name=name, ms_per_step=round(per_step, 4), launches_per_step=launches / args.steps, bound=bound,
                       achieved=round(tfs if bound == "mfma" else gbs, 3), unit="TFLOP/s" if bound == "mfma" else "GB/s",
                       frac=round(max(f_h, f_m), 5))
            if split and bound == "mfma":
                row["peak"] = round(mfma_peak, 1)
                row["peak_note"] = "bf16 MFMA dense peak / 6 piece products"
            row["pipe"] = "bf16x3" if split else "f32"
            row["flops"] = cst["flops"]
            if "unique" in cst:
                row["hbm_frac_unique_bytes"] = round(f_h, 5)
                row["gather_counted_gbs"] = round(gathered, 1)  # mostly L2 / MALL hits: NOT an HBM rate, never used for frac
            stages.append(row)
        stages.sort(key=lambda s: -s["ms_per_step"])
        dom = next((s for s in stages if s["name"] == dominant), None)
        roofline = None
        if dom and dominant in live:
            # the dominant stage re-measured LIVE inside the timed region (its own event pair only)
            t_ms, n_launch = live[dominant]
            cst = costs.get(dominant, dict(flops=0, bytes=0))
            gbs = cst.get("unique", cst["bytes"]) / (t_ms * 1e-3) / 1e9
            tfs = cst["flops"] / (t_ms * 1e-3) / 1e12
            mf = tfs / F32_MFMA_PEAK_TF > gbs / HBM_PEAK_GBS
            roofline = dict(kernel=dominant, bound="mfma" if mf else "hbm", achieved=round(tfs if mf else gbs, 3),
                            peak=F32_MFMA_PEAK_TF if mf else HBM_PEAK_GBS, unit="TFLOP/s" if mf else "GB/s",
                            frac=round(max(tfs / F32_MFMA_PEAK_TF, gbs / HBM_PEAK_GBS), 5), traffic=None,
                            ms_per_step=round(t_ms, 4), launches_per_step=n_launch, avg_launch_ms=round(t_ms / max(n_launch, 1), 5),
                            algorithmic_bytes_per_step=cst.get("unique", cst["bytes"]), algorithmic_flops_per_step=cst["flops"], measured=dom_where)
            # HBM bytes per launch cannot be counted from inside this process: they come from separate rocprofv3 --pmc passes of this
            # same command (profiles/run_pmc.sh; FETCH_SIZE doubled per the gfx950 note + WRITE_SIZE), committed with the commit they
            # were taken at.  The number is labelled with that source; null when no such file exists for this round.
            pmc = next((q for q in (os.path.join(ROOT, "profiles", "r%d_pmc_traffic.json" % k) for k in (6, 5, 4, 3, 2)) if os.path.exists(q)), "")
            if pmc:
                t = json.load(open(pmc))
                roofline["traffic"] = t.get(dominant)
                roofline["traffic_source"] = "profiles/%s: rocprofv3 --pmc passes of `%s` at commit %s (not measured in this run)" % (
                    os.path.basename(pmc), t.get("_command", "python bench.py"), t.get("_commit", "?"))
                roofline["traffic_top_kernels"] = t.get("_top_kernels")
        total_cost = {k: sum(v[k] for v in costs.values()) for k in ("flops", "bytes")}
        dev_ms = sum(s["ms_per_step"] for s in stages)
        # the metric split into its index-pyramid and network halves (serial device time of the profile pass)
        pyr_ms = sum(s["ms_per_step"] for s in stages if s["name"].startswith(("kdtree", "knn", "pyramid")))
        split = {"pyramid_ms_per_step": round(pyr_ms, 4), "network_ms_per_step": round(dev_ms - pyr_ms, 4),
                 "knn_points_per_s": B * n0 / (pyr_ms * 1e-3) if pyr_ms > 0 else None,
                 "net_points_per_s": B * n0 / ((dev_ms - pyr_ms) * 1e-3) if dev_ms > pyr_ms else None}
        net_ms = dev_ms - pyr_ms
        net_tfs = total_cost["flops"] / (net_ms * 1e-3) / 1e12 if net_ms > 0 else 0.0
        # supplementary: the whole network (every launch after the pyramid) against the fp32 MFMA peak, SURVEY 8(d)'s "fused network
        # kernels" figure; the contract's "roofline" object above stays on the single dominant kernel
        roofline_network = dict(bound="mfma", achieved=round(net_tfs, 3), peak=F32_MFMA_PEAK_TF, unit="TFLOP/s",
                                frac=round(net_tfs / F32_MFMA_PEAK_TF, 5), ms_per_step=round(net_ms, 4),
                                algorithmic_flops_per_step=total_cost["flops"], measured="serial profile pass, all network stages")
        # ... and split by the matrix pipe the stages actually run on: a utilisation figure only means something per pipe
        by_pipe = {}
        for srow in stages:
            if srow["name"].startswith(("kdtree", "knn", "pyramid")):
                continue
            acc = by_pipe.setdefault(srow["pipe"], [0.0, 0.0])
            acc[0] += srow["flops"]
            acc[1] += srow["ms_per_step"]
        roofline_network["by_pipe"] = {
            pipe: dict(algorithmic_flops_per_step=fl, ms_per_step=round(ms_p, 4), achieved=round(fl / (ms_p * 1e-3) / 1e12, 3) if ms_p > 0 else 0.0,
                       unit="TFLOP/s", peak=round(BF16_MFMA_PEAK_TF / 6, 1) if pipe == "bf16x3" else F32_MFMA_PEAK_TF,
                       frac=round(fl / (ms_p * 1e-3) / 1e12 / (BF16_MFMA_PEAK_TF / 6 if pipe == "bf16x3" else F32_MFMA_PEAK_TF), 5) if ms_p > 0 else 0.0,
                       what=("attentive pooling at d_out >= 64 and the decoder's large products: six bf16 MFMAs per fp32 product over exact splits (ceiling = dense bf16 peak / 6, in fp32 FLOPs)"
                             if pipe == "bf16x3" else "everything else: fp32 MFMA / HBM-bound stages"))
            for pipe, (fl, ms_p) in by_pipe.items()}
        if not args.att_fp32_mfma:
            roofline_network["note"] = ("the single figure above is quoted against the fp32 MFMA peak although fourteen stages run on the bf16 pipe: it is a "
                                        "rate in reference-formulation FLOPs, not the utilisation of one pipe -- see by_pipe")
        out = {
            "metric": "points_per_sec_forward",
            "value": whole_job_value(world, B, n0, args.steps, elapsed),
            "unit": "points/s",
            "n_gpus": world,
            "ranks_seen": seen,
            "rank0_pinned_cpus": args.pinned_cpus,  # CPUs rank 0's process is pinned to (N > 1: its slice next to its GPU; null = not pinned)
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[%d]: %d-point BraTS-shaped cloud (%d modalities), K=%d, 5-level RandLA-Net forward "
                                   "incl. index pyramid, fp32, batch %d per GPU" % (1 if args.workload == "config2" else 4, n0, cfg.in_channels - 3,
                                                                                  cfg.k_n, B),
                       "points": n0, "k_n": cfg.k_n, "num_layers": cfg.num_layers, "batch_per_gpu": B, "sharding": "one cloud per GPU, no collective",
                       "pipeline": "serial, one stream" if args.no_pipeline else
                       ("%d lanes (one HIP stream each), consecutive clouds coalesced in PAIRS: two 180 000-point clouds per ps_pyramid_build / "
                        "ps_randla_forward launch (a batch of two independent clouds; `one_cloud_per_launch` = the rounds 1-5 form, same run)" % args.lanes
                        if (pipe is not None and pipe.coalesce == 2) else
                        "%d clouds in flight, one HIP stream each (pyramid + forward per cloud on its stream)" % args.lanes),
                       "clouds_per_launch": (pipe.coalesce if pipe is not None else 1),
                       "inputs": "pinned host memory, copied per step (PCIe-inclusive)" if args.include_pcie else "resident in HBM",
                       "distinct_clouds": n_clouds,
                       "attention_mfma": "fp32 MFMA" if args.att_fp32_mfma else
                       "bf16 MFMA over exact three-way bfloat16 splits of the fp32 operands, fp32 accumulate (fp32-level error: csrc/attpool32b.hip)"},
            "roofline": roofline,
            "roofline_network": roofline_network,
            "serial_ms_per_cloud": serial_ms,
            "serial_one_lane_ms_per_cloud": (sub.get("serial") or {}).get("ms_per_cloud_one_lane"),
            "serial": sub.get("serial"),
            "include_pcie": sub.get("include_pcie"),
            "coalesced_pairs": sub.get("coalesced_pairs"),
            "att_fp32_mfma": sub.get("att_fp32_mfma"),
            "device_ms_per_step": round(dev_ms, 4),
            "split": split,
            "algorithmic": {"gflop_per_step": total_cost["flops"] / 1e9, "gbyte_per_step": total_cost["bytes"] / 1e9},
            "stages": stages,
        }
        if world == 1 and not args.no_sub_results and args.workload == "config2" and B == 1 and pipe is not None and not args.include_pcie:
            # the other single-GPU BASELINE configurations, timed in the same run (a few seconds each, after the headline's timed region):
            # configs[2] = the batch-8 training step (fp32 and "bf16 MLPs"), configs[4] = the 262 144-point / K = 32 forward
            t_sub = time.perf_counter()
            if len(d_clouds) >= 2:
                out["batch2"], pipe = sub_batch2(cfg, params, d_clouds, local_rank, args.lanes, pipe)
            out["config5"] = sub_config5(local_rank, args.lanes, pipe)
            pipe = None
            out["train_b8"] = {"f32": sub_train(cfg, xyz_all[:8], local_rank, False), "bf16": sub_train(cfg, xyz_all[:8], local_rank, True),
                               "f32_atomic_scatter": sub_train(cfg, xyz_all[:8], local_rank, False, deterministic=False)} if len(xyz_all) >= 8 else None
            # the per-rank work of BASELINE configs[3] (8 GPUs x 1 cloud): the batch-1 step on this one GPU, no collective
            out["train_b1"] = {"f32": sub_train(cfg, xyz_all[:1], local_rank, False, steps=10, warmup=2),
                               "bf16": sub_train(cfg, xyz_all[:1], local_rank, True, steps=10, warmup=2)}
            out["sub_results_seconds"] = round(time.perf_counter() - t_sub, 2)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, xyz[:1], feats[:1], params)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        if train_ranks is not None:
            out["train_config3"] = train_ranks
        # LAST in the line (the driver keeps the tail of it): the numbers a reader needs, compact
        r3 = lambda v: None if v is None else round(v, 3)  # noqa: E731
        tb8, tb1, c5 = out.get("train_b8") or {}, out.get("train_b1") or {}, out.get("config5") or {}
        out["summary"] = {
            "ms_per_step": r3(out["ms_per_step"]), "serial_ms": r3(serial_ms), "serial_one_lane_ms": r3((sub.get("serial") or {}).get("ms_per_cloud_one_lane")), "pcie_ms": r3((sub.get("include_pcie") or {}).get("ms_per_step")),
            "coalesced_pairs_ms": r3((sub.get("coalesced_pairs") or {}).get("ms_per_cloud")), "steady_one_per_launch_ms": r3((sub.get("coalesced_pairs") or {}).get("ms_per_cloud_one_per_launch")),
            "knn_us": r3(1e3 * roofline["avg_launch_ms"]) if roofline else None, "knn_frac": roofline["frac"] if roofline else None,
            "batch2_ms_per_cloud": r3((out.get("batch2") or {}).get("ms_per_cloud")),
            "config5_ms": r3(c5.get("ms_per_step")), "config5_serial_one_lane_ms": r3(c5.get("serial_one_lane_ms_per_cloud")),
            "train_f32_ms": r3((tb8.get("f32") or {}).get("ms_per_step")), "train_f32_minmedmax": (tb8.get("f32") or {}).get("step_ms"),
            "train_bf16_ms": r3((tb8.get("bf16") or {}).get("ms_per_step")), "train_bf16_minmedmax": (tb8.get("bf16") or {}).get("step_ms"),
            "train_atomic_ms": r3((tb8.get("f32_atomic_scatter") or {}).get("ms_per_step")),
            "train_b1_f32_ms": r3((tb1.get("f32") or {}).get("ms_per_step")), "train_b1_bf16_ms": r3((tb1.get("bf16") or {}).get("ms_per_step")),
            "launches": {"train_b8_f32": (tb8.get("f32") or {}).get("launches_per_step"), "train_b8_bf16": (tb8.get("bf16") or {}).get("launches_per_step"),
                         "train_b1_f32": (tb1.get("f32") or {}).get("launches_per_step"),
                         "forward": sum(s_["launches_per_step"] for s_ in stages) if stages else None},
            "cpu_pts_s": r3((out.get("cpu_baseline") or {}).get("value")),
        }
        if train_ranks is not None:  # N > 1: configs[3], one cloud per GPU (SyncBN / per-GPU BatchNorm): ms per step, calls into the collective, their device ms
            out["summary"]["train_config3"] = {k: [r3(v["ms_per_step"]), v["collectives_per_step"], r3(v["collective_ms"])]
                                               for k, v in train_ranks.items() if isinstance(v, dict)}
        print(json.dumps(out))
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
