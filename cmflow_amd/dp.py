"""Data parallelism for the training step: one process per GPU, one RCCL all-reduce per step.

The reference's only multi-GPU mode is single-process ``nn.DataParallel``
(models/model.py:40-42): per step it broadcasts all parameters, scatters the batch, gathers
outputs to GPU 0 and reduces gradients there; BatchNorm statistics are per replica.  The
MI355X-native equivalent keeps the same semantics (per-rank BN statistics, mean gradient)
with ``torch.distributed`` (backend "nccl" = RCCL over xGMI): gradients of all parameters are
views into ONE flat fp32 bucket (16.9 MB for CMFlow, 18.5 MB for CMFlow-T), reduced with a
single ``all_reduce(SUM)`` per optimizer step and scaled by 1/world.  Nothing is exchanged in
the forward pass.  Parameters that never receive a gradient (the 12 unused WeightNet-BN
tensors) stay outside the bucket with grad=None, so Adam skips them as it does in the reference.
"""
import torch
import torch.distributed as dist


class FlatGradBucket:
    def __init__(self, module: torch.nn.Module):
        self.params = [p for p in module.parameters() if p.requires_grad and not getattr(p, '_cmf_unused', False)]
        n = sum(p.numel() for p in self.params)
        p0 = self.params[0]
        self.flat = torch.zeros(n, dtype=p0.dtype, device=p0.device)
        off = 0
        self.views = []
        for p in self.params:
            k = p.numel()
            p.grad = self.flat[off:off + k].view_as(p)        # gradients accumulate in place into the bucket
            self.views.append(p.grad)
            off += k
        self.numel = n
        for p in self.params:                                 # = fused_blocks.enable_grad_sinks: the backward kernels
            p._cmf_sink = True                                # accumulate straight into the bucket

    def zero(self):
        """optimizer.zero_grad() equivalent that keeps the views (set_to_none would break them).  The check is an
        identity comparison per parameter (this runs on the host between forward and backward, every step)."""
        self.flat.zero_()
        for p, v in zip(self.params, self.views):
            if p.grad is not v:
                raise RuntimeError("a gradient left the flat bucket (zero_grad(set_to_none=True)?)")

    def all_reduce_mean(self, group=None, force=False):
        if dist.is_available() and dist.is_initialized() and (force or dist.get_world_size(group) > 1):
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
            self.flat.mul_(1.0 / dist.get_world_size(group))

    # ---- segments: all-reduce overlapped with the backward pass ---------------------------------------------------
    def segment_of(self, modules):
        """(offset, length) of the contiguous stretch of the bucket that holds the parameters of `modules` (they must be
        adjacent in module.parameters() order, which is how the bucket is laid out)."""
        ids = {id(p) for m in modules for p in m.parameters()}
        off, lo, hi = 0, None, None
        for p in self.params:
            if id(p) in ids:
                lo = off if lo is None else lo
                if hi is not None and hi != off:
                    raise ValueError("parameters of the given modules are not contiguous in the bucket")
                hi = off + p.numel()
            off += p.numel()
        if lo is None:
            raise ValueError("no bucket parameter belongs to the given modules")
        return lo, hi - lo


class SegmentedReducer:
    """The gradient all-reduce of a step, cut into segments that are launched as soon as their part of the backward pass
    has been enqueued -- the counterpart of nn.DataParallel's reduction running inside backward (models/model.py:40-42).

    segments: [(offset, length)] in the order their gradients become complete.  launch(i) is called from a tensor hook on
    the autograd thread (after the chains that write the segment have been joined into the current stream); the
    collective runs on the backend's own stream behind an event of the current stream, so it overlaps the rest of the
    backward pass.  finish() makes the current stream wait for every launched segment, reduces the ones that were never
    launched (hooks that did not fire) and applies the 1/world mean.  RCCL: one ncclAvg all-reduce per segment (no scaling
    kernel); gloo (CPU tests, the 1-GPU two-rank tests): SUM followed by a multiply."""

    def __init__(self, bucket, segments, group=None):
        self.bucket, self.segments, self.group = bucket, list(segments), group
        covered = sorted(self.segments)
        pos = 0
        for off, n in covered:
            if off != pos:
                raise ValueError("segments must tile the bucket")
            pos = off + n
        if pos != bucket.numel:
            raise ValueError("segments must tile the bucket")
        self.views = [bucket.flat[off:off + n] for off, n in self.segments]
        self.work = [None] * len(self.segments)
        self.active = False
        self.early = 0

    def begin(self, force=False):
        """Arm the reducer for one backward pass; False when there is nothing to reduce (single rank, not forced)."""
        self.active = dist.is_available() and dist.is_initialized() and (force or dist.get_world_size(self.group) > 1)
        self.work = [None] * len(self.segments)
        if self.active:
            self.avg = dist.get_backend(self.group) == "nccl"
        return self.active

    def launch(self, i):
        if not self.active or self.work[i] is not None:
            return
        op = dist.ReduceOp.AVG if self.avg else dist.ReduceOp.SUM
        self.work[i] = dist.all_reduce(self.views[i], op=op, group=self.group, async_op=True)

    def finish(self):
        if not self.active:
            return
        self.early = sum(w is not None for w in self.work)      # segments launched from inside backward (diagnostics / tests)
        for i in range(len(self.segments)):
            self.launch(i)
        for w in self.work:
            w.wait()                                    # the current stream waits; the host does not block (NCCL semantics)
        if not self.avg:
            self.bucket.flat.mul_(1.0 / dist.get_world_size(self.group))
        self.active = False


def pin_rank_to_cores(local_rank=None, local_world=None):
    """Restrict this rank (and every thread it creates later: the library's chain workers, the autograd thread) to its own
    block of host cores: with 8 ranks x 5 enqueueing threads on one host the launch path otherwise migrates across
    sockets.  Cores are dealt in contiguous blocks by LOCAL_RANK (on the usual two-socket hosts the lower ranks' GPUs
    hang off the first socket).  CMF_NO_AFFINITY=1 opts out.  Returns the core list or None."""
    import os
    if os.environ.get("CMF_NO_AFFINITY") == "1" or not hasattr(os, "sched_setaffinity"):
        return None
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if local_rank is None else local_rank
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))) if local_world is None else local_world
    if local_world <= 1:
        return None
    avail = sorted(os.sched_getaffinity(0))
    per = len(avail) // local_world
    if per < 2:                                          # fewer than two cores per rank: leave the scheduler alone
        return None
    cores = avail[local_rank * per:(local_rank + 1) * per]
    os.sched_setaffinity(0, cores)
    return cores


def broadcast_module(module: torch.nn.Module, src=0, group=None):
    """One-time parameter/buffer broadcast from rank `src` (replicas start identical)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)


def shard_batch(batch: dict, rank: int, world: int) -> dict:
    """Rank r's contiguous slice of a global batch (dim 0), as DataParallel's scatter does."""
    out = {}
    for k, v in batch.items():
        B = v.shape[0]
        assert B % world == 0, "global batch must divide evenly over ranks"
        per = B // world
        out[k] = v[rank * per:(rank + 1) * per].contiguous()
    return out
