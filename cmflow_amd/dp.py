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
