"""Multi-GPU plumbing: one process per GPU, RCCL over xGMI (`torch.distributed` backend "nccl"), gloo in the
CPU tests.

* Inference / PGD evaluation (BASELINE configs 2-4): REPLICAS ONLY -- each rank runs the hot path on its own shard of the
  batch, no data-path collective (every op is per-sample, and the glue's batch-global min/max is defined per rank, exactly
  what wrapping the reference in DDP would do).  The only collectives are the timing barrier and the max-over-ranks.
* Adversarial-training step (configs[4]): `GradAllReduce` -- the one real exchange step of the path: a bucketed all-reduce
  (average) of the flat gradient arena, launched bucket by bucket on a side stream while the reverse pass is still running.
"""
import os

import torch


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_range(rank, world, per_rank):
    """Samples [start, start + per_rank) of rank `rank` (weak scaling: per-rank batch fixed)."""
    return rank * per_rank, per_rank


def max_over_ranks(seconds, dist=None, device="cpu"):
    """The bench contract's elapsed time: MAX over ranks."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def global_minmax(mn, mx, dist=None):
    """`global_minmax=True` mode of the composite models: min/max of the glue's batch-global
    normalisation across ALL ranks -- one all-reduce of 2 floats (MIN over [mn, -mx]) -- which reproduces what ONE process
    running the whole batch would compute (core/model_fusion_auto.py:721-723).  Inference only."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return mn, mx
    t = torch.stack([mn.reshape(()), -mx.reshape(())])
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return t[0], -t[1]


class GradAllReduce:
    """Bucketed gradient all-reduce over a ParamArena's flat gradient buffer (paif_amd.utils.optimizer).

    The arena is laid out in backward-completion order (assign_grad_order), so a bucket = one contiguous slice, and buckets
    become ready front to back.  The reverse pass calls ops.grads_ready(module) whenever a milestone module's parameter
    gradients are final (wgrad passes only -- the attacks' input-gradient passes never do); between `begin()` and `finish()`
    `mark_ready` then launches -- on a side stream, behind an event recorded on the compute stream --
    every bucket that is now complete.  `finish()` (after loss.backward()) launches what is left and makes the compute
    stream wait for all of them; gradients end up AVERAGED over the ranks.  Parameters flagged `_paif_never_grad` sit in
    the arena's tail and are never communicated (the DDP analogue would need find_unused_parameters).

    xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring all-reduce of the full 179.5 MB is per-link bound at
    2*(7/8)*179.5 MB / 153 GB/s = 2.05 ms; 25 MB buckets keep each call in RCCL's bandwidth regime while leaving ~7 launch
    points to overlap with the MiT and fusion-network reverse passes (hundreds of ms at 8 pairs per GPU).
    """

    def __init__(self, arena, model=None, process_group=None, bucket_mb=25.0, milestones=None):
        import torch.distributed as dist
        self.dist = dist
        self.arena = arena
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.cuda = arena.grad.is_cuda
        # communicated region: everything before the never-grad tail
        live = [(o, e) for (p, _, o, n), e in ((ent, arena.range_of[id(ent[0])][1]) for ent in arena.entries)
                if not getattr(p, "_paif_never_grad", False)]
        self.live_end = max(e for _, e in live) if live else 0
        bucket_floats = max(1, int(bucket_mb * 1024 * 1024 / 4))
        # buckets are cut at parameter boundaries (a parameter is never split), in arena order
        self.buckets, start = [], 0
        for (p, _, o, n) in arena.entries:
            end = arena.range_of[id(p)][1]
            if getattr(p, "_paif_never_grad", False) or end > self.live_end:
                continue
            if end - start >= bucket_floats:
                self.buckets.append((start, end))
                start = end
        if start < self.live_end:
            self.buckets.append((start, self.live_end))
        # milestone module -> end of its last parameter in the arena (ready prefix grows monotonically when milestones
        # arrive in backward order; out-of-order arrivals are handled by per-parameter bookkeeping)
        if milestones is None and model is not None:
            from .core.model_fusion_auto import grad_milestones
            milestones = grad_milestones(model)
        self.module_params = {id(m): [p for p in m.parameters() if id(p) in arena.range_of] for m in (milestones or [])}
        self.side = torch.cuda.Stream(device=arena.grad.device) if self.cuda else None
        self.armed = False
        self.reset()

    def reset(self):
        self.pending = {id(p) for (p, _, _, _) in self.arena.entries if not getattr(p, "_paif_never_grad", False)}
        self.next_bucket = 0
        self.works = []
        self.launched = []          # (start, end) in launch order -- inspected by the tests

    def begin(self):
        """Arm the reducer for ONE training backward: call right before `loss.backward()`.  Milestones that arrive while
        it is not armed (an attack's input-gradient-only reverse pass, a stray call between steps) are ignored, so a bucket
        can never leave before the wgrad kernels of THIS backward were queued."""
        self.reset()
        self.armed = True
        return self

    def install(self):
        """Hook into the reverse pass (ops.grads_ready)."""
        from . import ops
        ops.GRAD_READY[0] = self.mark_ready
        return self

    def uninstall(self):
        from . import ops
        if ops.GRAD_READY[0] == self.mark_ready:
            ops.GRAD_READY[0] = None

    def _bucket_ready(self, b):
        s, e = self.buckets[b]
        return not any(id(p) in self.pending and s <= self.arena.range_of[id(p)][0] < e for (p, _, _, _) in self._bucket_params[b])

    @property
    def _bucket_params(self):
        if not hasattr(self, "_bp"):
            self._bp = [[ent for ent in self.arena.entries if s <= ent[2] < e] for (s, e) in self.buckets]
        return self._bp

    def _launch(self, b):
        s, e = self.buckets[b]
        buf = self.arena.grad[s:e]
        self.launched.append((s, e))
        if self.world == 1 and not (self.dist.is_available() and self.dist.is_initialized()):
            return
        dist = self.dist
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(buf.device))
            with torch.cuda.stream(self.side):
                self.side.wait_event(ev)          # the bucket's wgrad kernels were queued before this point
                work = dist.all_reduce(buf, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
            self.works.append((work, None))
        else:                                     # gloo (CPU tests): no AVG -> SUM, scaled at finish()
            work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self.works.append((work, buf))

    def mark_ready(self, module):
        """The parameter gradients of `module` are final for this backward pass."""
        if not self.armed:
            return
        for p in self.module_params.get(id(module), ()):
            self.pending.discard(id(p))
        while self.next_bucket < len(self.buckets) and self._bucket_ready(self.next_bucket):
            self._launch(self.next_bucket)
            self.next_bucket += 1

    def finish(self):
        """After loss.backward(): communicate every bucket not yet launched, then make the compute stream wait for all."""
        self.pending.clear()
        while self.next_bucket < len(self.buckets):
            self._launch(self.next_bucket)
            self.next_bucket += 1
        for work, buf in self.works:
            work.wait()                           # NCCL: the current stream waits for the collective; gloo: blocks
            if buf is not None:
                buf.mul_(1.0 / self.world)
        covered = sum(e - s for s, e in self.launched)
        assert covered == self.live_end and len(self.launched) == len(self.buckets), "a gradient bucket was skipped or sent twice"
        self.armed = False
        self.reset()
