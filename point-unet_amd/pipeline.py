"""Several clouds in flight on one GPU.

The reference feeds its network through tf.data: `tf_map` (the CPU KNN pyramid, runBraTS.py:140-161) runs in the
input pipeline's `.map(...)` workers and `.prefetch(...)` keeps the next clouds' pyramids ready while the GPU runs the
current one (runBraTS.py:166-185).  Here both halves are on the device, and the overlap is between LANES: every lane
owns a HIP stream, a context (workspace), a copy of the folded weights and a pyramid slot, and runs its cloud start to
finish -- ps_pyramid_build then ps_randla_forward -- on its stream; consecutive clouds go to consecutive lanes.  The
pyramid of one cloud (latency-bound tree build and searches that fill a fraction of the chip) and the deep, few-point
levels of its network then share the chip with the wide MFMA / HBM-bound kernels of the clouds on the other lanes.
Nothing is copied between lanes and the host never blocks.

Hardware queues: HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) in creation order, and two lanes
on one queue run one after the other.  With the default, three lanes plus the null stream fit (1.33 ms per 180 000-point cloud);
GPU_MAX_HW_QUEUES=6 in the environment (before the HIP runtime starts) and four lanes measured 1.23 ms, seven or more queues
get slower again.  Streams created BEFORE the lanes by somebody else (RCCL's internal streams after
`init_process_group("nccl")`) shift the assignment and can put two lanes on one queue (measured 1.72 ms): create and prime()
the pipeline first, the process group afterwards (bench.py does both).  Results are identical to the serial path (same kernels,
same order per cloud) -- tests/test_gpu_network.py::test_pipeline_matches_serial.
"""
import os
import warnings

import torch

from . import _lib, runtime
from .pyramid import alloc_pyramid, build_pyramid
from .RandLANet import Network


def hw_queues():
    """GPU_MAX_HW_QUEUES as the HIP runtime sees it (the package sets 6 at import when the user set nothing; HIP's own default is 4)."""
    try:
        return int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))
    except ValueError:
        return 4


def _check_environment(lanes, reusing):
    """The two traps of the lane pipeline, checked where the lanes are created (module docstring): too few hardware queues puts two
    lanes on one queue (they then run one after the other: measured 1.33 against 1.23 ms per cloud with 3 effective lanes), and HIP
    streams created by RCCL before the lanes shift the stream -> queue assignment (measured 1.72 against 1.31 ms)."""
    q = hw_queues()
    if q < lanes + 2:
        warnings.warn("ForwardPipeline: %d lanes want GPU_MAX_HW_QUEUES >= %d (lanes + the null stream + RCCL's) but the HIP runtime has %d: lanes "
                      "that share a hardware queue run one after the other (measured 1.33 vs 1.23 ms per 180 000-point cloud).  Set the variable "
                      "before the first HIP call -- `import point_unet_amd` does when it is unset." % (lanes, lanes + 2, q), RuntimeWarning, stacklevel=3)
    if reusing:
        return  # (the streams exist already: created before whatever came later)
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        try:
            backend = str(dist.get_backend())
        except Exception:  # noqa: BLE001
            backend = ""
        if "nccl" in backend:
            # throughput only, never correctness: init_process_group first is the normal torchrun order, so this is a warning; a launcher that
            # wants the fast order enforced sets PS_PIPELINE_STRICT_ORDER=1 (bench.py creates its pipeline first either way)
            msg = ("ForwardPipeline created AFTER torch.distributed.init_process_group('nccl'): RCCL's internal streams take hardware queues the "
                   "lanes would use (measured 1.72 vs 1.31 ms per cloud).  Create (and prime()) the pipeline before the process group for the "
                   "faster assignment.")
            if os.environ.get("PS_PIPELINE_STRICT_ORDER", "0") == "1":
                raise RuntimeError(msg)
            warnings.warn(msg, RuntimeWarning, stacklevel=3)


class _Lane:
    def __init__(self, config, params, device, seed, stream=None, ctx=None):
        self.stream = stream if stream is not None else torch.cuda.Stream(torch.device("cuda", device))
        self.ctx = ctx if ctx is not None else runtime.Context(device)
        self.ctx.set_stream(self.stream)
        self.ctx.set_deferred_checks(True)  # tree-build status words are validated at synchronize()
        self.net = Network(config, params=params, device=device, seed=seed, ctx=self.ctx)
        self.pyramids = {}  # batch size -> Pyramid (buffers reused from cloud to cloud)
        self.done = None
        self.submissions = []  # global submission index of every pyramid build this lane's context has run, in order
        # coalesced mode: the clouds waiting for their partner, copied into the lane's own [2, N, .] input slots
        self.staged = 0
        self.xyz2 = self.feat2 = self.logits2 = None

    @property
    def pyramid(self):  # (the batch-1 pyramid: what the measurement tools of profiles/tools reach for)
        return self.pyramids.get(1)

    @pyramid.setter
    def pyramid(self, value):
        self.pyramids[1] = value


class ForwardPipeline:
    def __init__(self, config, params=None, device=0, seed=0, lanes=4, reuse=None, coalesce=1):
        """reuse: a ForwardPipeline that is done (another network / configuration): its lanes' HIP streams and contexts (workspaces) are
        taken over instead of creating new ones and it must not be used afterwards.  Every extra stream a process has touched costs:
        measured with this very class, a second pipeline on four NEW streams runs 2.33 ms pipelined / 4.3 ms serial per 262 144-point
        cloud against 1.93 / 2.63 ms for the same pipeline alone in a process (profiles/tools/exp_second_pipeline.py: four idle streams
        taken from torch's pool beforehand are enough; GPU_MAX_HW_QUEUES=10 does the same to the FIRST pipeline) -- dependent kernels
        on streams spread over more hardware queues are scheduled later.

        coalesce = 2 (opt-in; default 1 = every cloud its own launch): single clouds (B = 1) submitted one after the other are run TWO PER
        LAUNCH -- the first waits in the lane's input slot for its partner, the pair then goes through ps_pyramid_build /
        ps_randla_forward as a batch of two (the same C-ABI calls; clouds of a batch are independent rows of every kernel).  A
        service's throughput mode: measured on MI355X with four lanes over 200 clouds 0.798 ms per 180 000-point cloud against 0.835 one
        cloud per launch (bench.py `coalesced_pairs`; contiguous batches of two: 0.769, of four: 0.766 -- the ~120 launches of a cloud,
        most of them at the 5-25 us floor of a launch on the deep levels, are paid once per pair) -- for one more cloud of latency: a
        cloud's logits are complete when its pair is (synchronize() launches a cloud still waiting alone), and no gain over a run as short
        as the driver's 20 steps (the pipeline's fill and drain are twice as long).  Per cloud the result is that of the batch-1 call up
        to summation order in the split-K dense layers (their K split depends on the row count)."""
        self.cfg = config
        self.device = torch.device("cuda", device)
        self.coalesce = 2 if int(coalesce) >= 2 else 1
        _check_environment(int(lanes), reuse is not None)
        if params is None:
            from . import weights
            params = weights.init_params(config, seed=seed)
        if reuse is not None:
            reuse.synchronize()
            old = reuse.lanes
            reuse.lanes = []
            for ln in old:
                ln.net.close()
            self.lanes = [_Lane(config, params, device, seed, stream=ln.stream, ctx=ln.ctx) for ln in old[:int(lanes)]]
            for ln in old[int(lanes):]:
                ln.ctx.close()
            while len(self.lanes) < int(lanes):
                self.lanes.append(_Lane(config, params, device, seed))
        else:
            self.lanes = [_Lane(config, params, device, seed) for _ in range(int(lanes))]
        self._n0 = None
        self._i = 0        # clouds submitted so far
        self._launch = 0   # launches so far: picks the lane (one launch = one cloud, or one coalesced pair)
        self.last_done = None
        self.launched = True  # did the last submit() enqueue its cloud's work (False: the cloud waits for its partner)

    @property
    def contexts(self):
        return [ln.ctx for ln in self.lanes]

    def next_lane(self):
        """Index of the lane the next submit() (default arguments) will use."""
        return self._launch % len(self.lanes)

    def _pyramid(self, ln, B, n0, device):
        pyr = ln.pyramids.get(B)
        if pyr is None or pyr.xyz[0].shape[1] != n0:
            ratios = list(self.cfg.sub_sampling_ratio)[:self.cfg.num_layers]
            pyr = ln.pyramids[B] = alloc_pyramid(B, n0, ratios, self.cfg.k_n, device)
        return pyr

    def _run(self, ln, xyz, features, out=None):
        """pyramid + forward of one batch on the lane's stream (which is current)."""
        B, n0 = xyz.shape[0], xyz.shape[1]
        pyr = self._pyramid(ln, B, n0, xyz.device)
        ln.submissions.append(self._i - 1)
        build_pyramid(xyz, self.cfg, ctx=ln.ctx, out=pyr)
        inputs = {"pyramid": pyr, "features": features}
        if out is not None:
            inputs["out"] = out
        logits = ln.net.inference(inputs)
        ln.done = torch.cuda.Event()
        ln.done.record(ln.stream)
        self.last_done = ln.done
        self._launch += 1
        return logits

    def _flush(self, ln):
        """launches what waits in the lane's input slots (a pair, or one cloud whose partner never came)"""
        if ln.staged == 0:
            return
        k = ln.staged
        ln.staged = 0
        with torch.cuda.stream(ln.stream):
            self._run(ln, ln.xyz2[:k], ln.feat2[:k], out=ln.logits2[:k])

    def submit(self, xyz, features, overlap=True, lane=None):
        """xyz [B,N0,3], features [B,N0,Cin]: float32 CUDA tensors (ready on torch's current stream).  Enqueues the
        pyramid build and the forward on the next lane; returns the logits tensor [B,N0,classes], complete after
        synchronize() (or after waiting on `last_done`).  overlap=False makes the lane wait for the previously submitted
        cloud first: bench.py's per-stage profile pass uses it to time kernels without a neighbour on the chip.  lane=k pins the cloud
        to lane k instead of the next one in turn (consecutive clouds on ONE lane run one after the other by stream order alone: the
        per-cloud latency without any cross-stream event in the chain).  With coalesce = 2 a single cloud submitted with the default
        arguments may wait for its partner (`launched` says whether this call enqueued the work)."""
        B, n0 = xyz.shape[0], xyz.shape[1]
        if self._n0 != n0:
            self.synchronize()
            for ln in self.lanes:
                ln.pyramids = {}
                ln.xyz2 = ln.feat2 = ln.logits2 = None
            self._n0 = n0
        self._i += 1
        cur = torch.cuda.current_stream(self.device)
        if self.coalesce == 2 and B == 1 and overlap and lane is None and features.dtype == torch.float32:
            ln = self.lanes[self._launch % len(self.lanes)]
            ln.stream.wait_stream(cur)  # the inputs were produced on the caller's stream
            slot = ln.staged
            with torch.cuda.stream(ln.stream):
                if ln.xyz2 is None or ln.feat2.shape[2] != features.shape[2]:
                    ln.xyz2 = torch.empty((2, n0, 3), dtype=torch.float32, device=xyz.device)
                    ln.feat2 = torch.empty((2, n0, features.shape[2]), dtype=torch.float32, device=xyz.device)
                if slot == 0:  # (the pair's logits belong to the caller: a fresh buffer per pair)
                    ln.logits2 = torch.empty((2, n0, self.cfg.num_classes), dtype=torch.float32, device=xyz.device)
                ln.xyz2[slot].copy_(xyz[0], non_blocking=True)
                ln.feat2[slot].copy_(features[0], non_blocking=True)
                out = ln.logits2[slot:slot + 1]
                ln.staged = slot + 1
                self.launched = ln.staged == 2
                if self.launched:
                    self._flush(ln)
            xyz.record_stream(ln.stream)
            features.record_stream(ln.stream)
            return out
        # one launch for this submission (batches, pinned lanes, serialised passes, half-precision features)
        for other in self.lanes:
            self._flush(other)
        ln = self.lanes[(self._launch if lane is None else int(lane)) % len(self.lanes)]
        ln.stream.wait_stream(cur)
        if not overlap and self.last_done is not None:
            ln.stream.wait_event(self.last_done)
        with torch.cuda.stream(ln.stream):  # the logits are allocated on (and belong to) the lane's stream
            logits = self._run(ln, xyz, features)
        xyz.record_stream(ln.stream)
        features.record_stream(ln.stream)
        self.launched = True
        return logits

    def prime(self, xyz, features):
        """Runs one cloud through EVERY lane and drains: workspaces grow to their high-water mark (the only hipMalloc calls
        of the library) before anything is timed or latency-sensitive.  Coalesced mode: a pair per lane as well (the batch-2 pyramid and
        workspaces), and one cloud launched alone."""
        for _ in self.lanes:
            self.submit(xyz, features, overlap=False)
        self.synchronize()
        if self.coalesce == 2 and xyz.shape[0] == 1 and features.dtype == torch.float32:
            for _ in range(2 * len(self.lanes)):
                self.submit(xyz, features)
            self.synchronize()

    def synchronize(self):
        """Drains every lane and validates the status words of the tree builds.  The build itself always completes on the device
        (very unbalanced clouds included: csrc/kdtree_build.hip, straggler kernel), so nothing needs to be re-run; what can still
        be reported are the two degenerate cases (builder queue overflow, tree deeper than the traversal stack) -- the error then
        names the submission (0-based index over submit() calls, prime() included) whose logits are affected; every other
        submission's results are valid."""
        import re
        first = None
        for ln in self.lanes:
            self._flush(ln)  # (a cloud whose partner never came is launched alone)
        for k, ln in enumerate(self.lanes):
            try:
                ln.ctx.synchronize()
            except _lib.PointSegError as e:
                m = re.search(r"pyramid build #(\d+)", str(e))
                which = ln.submissions[int(m.group(1))] if m and int(m.group(1)) < len(ln.submissions) else None
                first = first or _lib.PointSegError("%s [ForwardPipeline: lane %d, submission %s]" % (e, k, which))
        if first is not None:
            raise first

    def close(self):
        self.synchronize()
        for ln in self.lanes:
            ln.net.close()
            ln.ctx.close()


class PyramidPrefetcher:
    """The training-side counterpart of ForwardPipeline: the index pyramid of the NEXT batch is built on its own HIP stream and context
    while the current batch trains -- the reference's `tf.data` `.map(tf_map).prefetch()` (runBraTS.py:166-185), whose CPU workers
    build pyramids underneath the GPU's training step.  `depth` pyramid slots (double buffered by default); stream-ordered with events,
    the host never blocks:

        pre.submit(xyz_next)            # enqueue batch k+1's pyramid on the prefetch stream
        pyr, slot = pre.next()          # batch k's pyramid (submitted a step earlier); the caller's stream waits for its build
        loss = trainer.train_step(pyr, features, labels)
        pre.release(slot)               # the slot may be rebuilt once the work enqueued so far has run
    """

    def __init__(self, config, device=0, depth=2):
        self.cfg = config
        self.device = torch.device("cuda", device)
        self.stream = torch.cuda.Stream(self.device)
        self.ctx = runtime.Context(device)
        self.ctx.set_stream(self.stream)
        self.ctx.set_deferred_checks(True)  # status words validated at synchronize()
        self.depth = int(depth)
        self.slots = [None] * self.depth
        self.ready = [None] * self.depth
        self.free = [None] * self.depth
        self._in = self._out = 0

    def submit(self, xyz):
        k = self._in % self.depth
        assert self._in - self._out < self.depth, "PyramidPrefetcher: every slot holds a pyramid nobody has taken yet"
        self._in += 1
        B, n0 = xyz.shape[0], xyz.shape[1]
        slot = self.slots[k]
        if slot is None or slot.xyz[0].shape[:2] != (B, n0):
            ratios = list(self.cfg.sub_sampling_ratio)[:self.cfg.num_layers]
            slot = self.slots[k] = alloc_pyramid(B, n0, ratios, self.cfg.k_n, xyz.device)
        self.stream.wait_stream(torch.cuda.current_stream(self.device))  # xyz was produced on the caller's stream
        if self.free[k] is not None:
            self.stream.wait_event(self.free[k])                        # the consumer of this slot's previous pyramid has run
        with torch.cuda.stream(self.stream):
            build_pyramid(xyz, self.cfg, ctx=self.ctx, out=slot)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        xyz.record_stream(self.stream)
        self.ready[k] = ev

    def next(self):
        assert self._out < self._in, "PyramidPrefetcher.next() without a submitted pyramid"
        k = self._out % self.depth
        self._out += 1
        torch.cuda.current_stream(self.device).wait_event(self.ready[k])
        return self.slots[k], k

    def release(self, k):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self.free[k] = ev

    def synchronize(self):
        self.ctx.synchronize()

    def close(self):
        self.synchronize()
        self.ctx.close()
