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
        allp = list(module.parameters())
        keep = [i for i, p in enumerate(allp) if p.requires_grad and not getattr(p, '_cmf_unused', False)]
        self.params = [allp[i] for i in keep]
        # position of every bucket parameter in module.parameters() order, and that list's length: an optimizer checkpoint saved over
        # ALL parameters of the module (the reference's torch.optim.Adam(net.parameters()), main.py:107) is indexed that way
        self.module_index, self.module_params = keep, len(allp)
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
            if dist.get_backend(group) == "nccl":
                # RCCL averages inside the collective (ncclAvg, as SegmentedReducer does): no scaling kernel behind it
                dist.all_reduce(self.flat, op=dist.ReduceOp.AVG, group=group)
            else:                                             # gloo (CPU tests, the one-GPU two-rank tests) has no AVG
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


class FlatAdam(torch.optim.Optimizer):
    """torch.optim.Adam (L2 weight decay, no amsgrad: the reference's optimizer, main.py:107) over a FlatGradBucket as ONE kernel
    launch (cmf_adam_step): the gradients already lie contiguously in the bucket, the moments are two flat arrays in the same order, and
    the parameters stay where the module holds them (a device table of their addresses).  torch's fused Adam needs six multi-tensor
    launches for the model's 182 tensors at the very end of a step.  A torch Optimizer all the same: lr schedulers act on
    param_groups[0]['lr'] (the reference steps a StepLR per epoch, main.py:108,151)."""

    def __init__(self, bucket, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(bucket.params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        if len(self.param_groups) != 1:
            raise ValueError("FlatAdam: one parameter group (the bucket's parameters) -- per-group options are not supported")
        self.bucket = bucket
        self.exp_avg = torch.zeros_like(bucket.flat)
        self.exp_avg_sq = torch.zeros_like(bucket.flat)
        self.steps = 0
        self._ptrs = None
        self._tables()

    def _tables(self):
        ps = self.bucket.params
        ptrs = tuple(p.data_ptr() for p in ps)
        if ptrs == self._ptrs:
            return
        for p in ps:
            if not (p.is_contiguous() and p.dtype == torch.float32 and p.is_cuda):
                raise RuntimeError("FlatAdam: parameters must be contiguous fp32 device tensors")
        offs, o = [], 0
        for p in ps:
            offs.append(o)
            o += p.numel()
        offs.append(o)
        dev = self.bucket.flat.device
        self._offs = torch.tensor(offs, dtype=torch.int64, device=dev)
        self._pt = torch.tensor(ptrs, dtype=torch.int64, device=dev)
        self._ptrs = ptrs

    @torch.no_grad()
    def step(self, closure=None):
        from . import _lib
        loss = closure() if closure is not None else None
        b = self.bucket
        for p, v in zip(b.params, b.views):
            if p.grad is not v:
                raise RuntimeError("a gradient left the flat bucket (zero_grad(set_to_none=True)?)")
        self._tables()                                       # a parameter whose storage was replaced gets its new address
        if len(self.param_groups) != 1:
            raise RuntimeError("FlatAdam: add_param_group is not supported (one group: the bucket's parameters)")
        g = self.param_groups[0]
        self.steps += 1
        _lib.check(_lib.lib().cmf_adam_step(len(b.params), self._offs.data_ptr(), self._pt.data_ptr(), b.numel, b.flat.data_ptr(),
                                            self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), float(g["lr"]), float(g["betas"][0]),
                                            float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]), self.steps,
                                            _lib.stream_ptr()), "cmf_adam_step")
        # the kernel writes the parameters through raw pointers: bump their autograd version counters like an in-place torch op would,
        # so that a backward pass through a graph recorded BEFORE this step raises instead of silently using the updated weights
        bump = getattr(torch.autograd.graph, "increment_version", None)
        if bump is not None:
            bump(b.params)
        return loss

    def zero_grad(self, set_to_none=False):
        self.bucket.zero()

    def state_dict(self):
        d = super().state_dict()
        d["flat"] = dict(exp_avg=self.exp_avg, exp_avg_sq=self.exp_avg_sq, steps=self.steps)
        return d

    def load_state_dict(self, d):
        """Accepts this class's own format (`flat`) and a torch.optim.Adam checkpoint -- per-parameter `state` with exp_avg /
        exp_avg_sq / step -- saved over the bucket's parameters in bucket order (what the previous TrainStep saved) OR over all of
        the module's parameters (the reference's `torch.optim.Adam(net.parameters())`, main.py:107, or an oracle / CPU run): its
        indices are then positions in module.parameters(); the parameters outside the bucket (never used, frozen) have no state in
        such a checkpoint either and are skipped.  The moments are scattered into the flat arrays.  Anything else raises:
        restarting the moments silently would change the training run."""
        d = dict(d)
        flat = d.pop("flat", None)
        per_param = d.get("state") or {}
        ps = self.bucket.params
        groups = d.get("param_groups")
        if groups is None or len(groups) != 1:
            raise ValueError("FlatAdam: the checkpoint has %s parameter groups, one is supported" % (len(groups) if groups is not None else "no"))
        saved = list(groups[0]["params"])
        if len(saved) == len(ps):
            key_of = list(saved)                              # bucket order
        elif len(saved) == self.bucket.module_params:
            key_of = [saved[i] for i in self.bucket.module_index]
            extra = set(per_param.keys()) - set(key_of)
            if extra:
                raise ValueError("FlatAdam: the checkpoint holds optimizer state for %d parameters outside the gradient bucket" % len(extra))
        else:
            raise ValueError("FlatAdam: the checkpoint covers %d parameters; the bucket holds %d of the module's %d"
                             % (len(saved), len(ps), self.bucket.module_params))
        group = dict(groups[0], params=list(range(len(ps))))
        super().load_state_dict(dict(d, state={}, param_groups=[group]))
        if flat is not None:
            self.exp_avg.copy_(flat["exp_avg"]); self.exp_avg_sq.copy_(flat["exp_avg_sq"]); self.steps = int(flat["steps"])
            return
        if not per_param:
            return                                           # a fresh optimizer's checkpoint: nothing to restore
        if not all(k in per_param for k in key_of):
            raise ValueError("FlatAdam: the checkpoint's per-parameter state does not cover the bucket's %d parameters" % len(ps))
        steps = set()
        off = 0
        for i, p in enumerate(ps):
            st = per_param[key_of[i]]
            n = p.numel()
            if st["exp_avg"].numel() != n:
                raise ValueError("FlatAdam: state %d has %d elements, the parameter %d" % (i, st["exp_avg"].numel(), n))
            self.exp_avg[off:off + n].copy_(st["exp_avg"].reshape(-1)); self.exp_avg_sq[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
            steps.add(int(st["step"]))
            off += n
        if len(steps) != 1:
            raise ValueError("FlatAdam: the checkpoint's parameters are at different step counts %s" % sorted(steps))
        self.steps = steps.pop()


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


def _gpu_local_cpus(index):
    """Host CPUs local to GPU `index` (its PCI device's NUMA node): /sys/bus/pci/devices/<domain:bus:device.0>/local_cpulist,
    or None when the property or the file is missing."""
    import os
    try:
        pr = torch.cuda.get_device_properties(index)
        bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        with open(os.path.join("/sys/bus/pci/devices", bdf, "local_cpulist")) as f:
            text = f.read().strip()
    except Exception:
        return None
    cpus = []
    for part in text.split(","):
        if part:
            lo, _, hi = part.partition("-")
            cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus or None


def pin_rank_to_cores(local_rank=None, local_world=None):
    """OPT-IN (CMF_PIN_CORES=1; unmeasured -- no multi-GPU box so far): restrict this rank (and every thread it creates later:
    the library's chain workers, the autograd thread) to host cores of ITS GPU's NUMA node.  The cores local to the GPU come
    from sysfs (logical CPU ids are not laid out socket by socket: on SMT hosts the second half of the ids are the siblings of
    the first); the ranks whose GPUs share a node split that node's cores evenly, in local-rank order.  local_world is the
    number of ranks on THIS host (LOCAL_WORLD_SIZE), not the job's world size.  Returns the core list or None (not enabled,
    no locality information, or fewer than two cores per rank: the scheduler is left alone)."""
    import os
    if os.environ.get("CMF_PIN_CORES") != "1" or not hasattr(os, "sched_setaffinity"):
        return None
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if local_rank is None else local_rank
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", "1")) if local_world is None else local_world
    if local_world <= 1:
        return None
    avail = set(os.sched_getaffinity(0))
    mine = _gpu_local_cpus(local_rank)
    if not mine:
        return None
    key = tuple(mine)
    sharing = [r for r in range(local_world) if tuple(_gpu_local_cpus(r) or ()) == key]     # ranks whose GPUs hang off the same node
    cores = sorted(c for c in mine if c in avail)
    per = len(cores) // max(len(sharing), 1)
    if per < 2 or local_rank not in sharing:
        return None
    i = sharing.index(local_rank)
    cores = cores[i * per:(i + 1) * per]
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
