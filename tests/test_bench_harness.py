"""The N>1 path of bench.py on CPU: two processes on the gloo backend run the timed-region helper with a stand-in
step of rank-dependent duration; both must agree on the MAX-over-ranks time and rank 0's whole-job value must count
both ranks' clouds (weak scaling, no data-path collective)."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import json, os, sys, time
    sys.path.insert(0, %r)
    import torch.distributed as dist
    import bench
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    def step():
        time.sleep(0.02 * (rank + 1))       # rank 1 is the slow one
        return rank
    elapsed, last = bench.timed_region(step, 5, lambda: None, dist)
    value = bench.whole_job_value(world, 1, 1000, 5, elapsed)
    print(json.dumps({"rank": rank, "world": world, "elapsed": elapsed, "value": value, "last": last}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
""") % ROOT


def test_two_rank_timed_region_takes_the_max_over_ranks(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, text=True)
             for r in range(2)]
    outs = [json.loads(p.communicate(timeout=120)[0].strip().splitlines()[-1]) for p in procs]
    assert all(p.returncode == 0 for p in procs)
    outs.sort(key=lambda o: o["rank"])
    assert outs[0]["world"] == 2
    assert abs(outs[0]["elapsed"] - outs[1]["elapsed"]) < 1e-9          # both hold the MAX
    assert outs[0]["elapsed"] >= 5 * 0.04 * 0.95                          # the slow rank's time
    assert abs(outs[0]["value"] - 2 * 1000 * 5 / outs[0]["elapsed"]) < 1e-6


def test_single_process_region_without_dist():
    sys.path.insert(0, ROOT)
    import bench
    n = []
    elapsed, last = bench.timed_region(lambda: n.append(1) or len(n), 7, lambda: None, None)
    assert len(n) == 7 and last == 7 and elapsed > 0
    assert bench.whole_job_value(1, 2, 10, 7, 2.0) == 70.0


def test_algorithmic_costs_match_the_survey_totals():
    """73.3 GFLOP per 180 000-point cloud for the network (reference formulation)."""
    sys.path.insert(0, ROOT)
    import bench
    from point_unet_amd.helper_tool import ConfigBraTS
    c = bench.algorithmic_costs(ConfigBraTS, 180000, 1)
    net = sum(v["flops"] for k, v in c.items() if not k.startswith(("knn", "kdtree", "pyramid")))
    assert abs(net / 1e9 - 73.3) < 0.8


GRAD_WORKER = textwrap.dedent("""
    import json, sys
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    from point_unet_amd.train import allreduce_mean_
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    g = torch.arange(1000, dtype=torch.float32) * (rank + 1)      # rank 0: x, rank 1: 2x  -> mean 1.5x
    allreduce_mean_(g, dist)
    print(json.dumps({"rank": rank, "ok": bool(torch.allclose(g, torch.arange(1000, dtype=torch.float32) * 1.5))}), flush=True)
    dist.destroy_process_group()
""") % ROOT


def test_gradient_allreduce_mean_two_ranks(tmp_path):
    """Config 4's only collective: mean of the flat gradient buffer over ranks (gloo stand-in for RCCL)."""
    script = tmp_path / "gworker.py"
    script.write_text(GRAD_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29537", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, text=True)
             for r in range(2)]
    outs = [json.loads(p.communicate(timeout=120)[0].strip().splitlines()[-1]) for p in procs]
    assert all(p.returncode == 0 for p in procs) and all(o["ok"] for o in outs)


SPAWN_CHILD = textwrap.dedent("""
    import json, os, sys, time
    sys.path.insert(0, %r)
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if "--die" in sys.argv and rank == 1:
        sys.exit(3)                                    # a rank that fails before the rendezvous
    import torch.distributed as dist
    import bench
    dist.init_process_group("gloo")
    seen = bench.ranks_seen(dist, "cpu")
    print("rank %%d noise on stdout" %% rank if rank else json.dumps({"n_gpus": world, "ranks_seen": seen, "argv": sys.argv[1:]}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
""") % ROOT


def test_bench_spawns_its_own_ranks_and_relays_rank0(tmp_path, capsys):
    """`python bench.py --gpus N` without a launcher (the driver's form): the parent starts N fresh rank processes, only rank 0's
    stdout (the JSON line) reaches the parent's stdout, the exit code is the worst child's."""
    sys.path.insert(0, ROOT)
    import bench
    child = tmp_path / "child.py"
    child.write_text(SPAWN_CHILD)
    rc = bench.spawn_ranks(2, ["--gpus", "2", "--steps", "3"], script=str(child))
    out = capsys.readouterr().out.strip().splitlines()
    assert rc == 0 and len(out) == 1
    line = json.loads(out[0])
    assert line == {"n_gpus": 2, "ranks_seen": 2, "argv": ["--gpus", "2", "--steps", "3"]}


def test_bench_spawner_reports_a_dead_rank_and_stops_the_others(tmp_path, capsys):
    sys.path.insert(0, ROOT)
    import bench
    child = tmp_path / "child.py"
    child.write_text(SPAWN_CHILD)
    t0 = __import__("time").time()
    rc = bench.spawn_ranks(2, ["--die"], script=str(child))
    assert rc == 3 and __import__("time").time() - t0 < 60          # rank 0 was stopped, not left at the rendezvous
    capsys.readouterr()


def test_bench_parent_never_touches_the_gpu_before_spawning():
    """The spawning branch sits before `import torch` / any device call in main()."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = src[src.index("def main():"):]
    assert main.index("spawn_ranks(args.gpus") < main.index("import torch")
    assert "os.exec" not in src and "execv" not in src


def test_ranks_pin_themselves_to_disjoint_cpu_slices_before_any_gpu_call():
    """`bench.py --gpus N`: every rank pins its process (os.sched_setaffinity) to its own slice of the CPUs -- next to its GPU's NUMA node when
    sysfs says which -- before `import torch` / any GPU call; `--no-pin` opts out.  Eight Python hosts must not share cores."""
    sys.path.insert(0, ROOT)
    import bench
    allowed = set(range(128))
    slices = [bench.rank_cpu_slice(r, 8, allowed, None) for r in range(8)]
    assert all(len(s) == 16 for s in slices) and len(set().union(*slices)) == 128                      # disjoint, all CPUs used
    # two NUMA nodes with four GPUs each: ranks 0-3 of a node get disjoint slices of ITS cpus
    near = [set(range(0, 64)) if r < 4 else set(range(64, 128)) for r in range(8)]
    slices = [bench.rank_cpu_slice(r, 8, allowed, near[r]) for r in range(8)]
    for r in range(8):
        assert slices[r] <= near[r] and len(slices[r]) == 16
    assert len(set().union(*slices)) == 128
    assert bench.rank_cpu_slice(0, 1, {3, 5}, None) == {3, 5}
    assert bench.gpu_local_cpus(10 ** 6) is None                                                          # no such GPU / no sysfs: no pinning hint
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = src[src.index("def main():"):]
    assert main.index("os.sched_setaffinity(0, mine)") < main.index("import torch") and "--no-pin" in main
