"""GPU: stream-dependency stress test of the multi-stream training step, fp32 (the product's arithmetic).

The step forks independent chains onto a pool of three side streams at five kinds of places (the block calls of the
encoders incl. the two-cloud node, the two heads, the cost-volume branches, the final join before the gradient bucket
is read) and the library issues each encoder chain from its own host thread.  The kernels are deterministic, so a step
whose chains are shifted against each other in time must be BIT-identical to the same step with every chain on the
caller's stream -- outputs, loss, the flat gradient bucket and every BN buffer.  A missing dependency, an arena shared
by two chains, or memory handed back to the caching allocator while a side stream still uses it shows up as a
difference under some timing; `fused_blocks.stress_seed(seed)` makes every fork point enqueue a random 0-300 us delay
(cmf_debug_spin) in front of the chains it starts (forward and backward), and 50 seeds are run at B = 8 and at the
benchmark's B = 64.  (Round 2 saw a run-to-run difference in an experimental bf16 arithmetic mode under side streams;
this is the test that would find the same fault in the fp32 product.)
"""
import os

import pytest
import torch

from cmflow_amd import synth

pytestmark = pytest.mark.gpu
SEEDS = int(os.environ.get("CMF_STRESS_SEEDS", "50"))


def _snapshot(step, net, batch):
    from cmflow_amd.fused_blocks import join_side_streams
    loss, _, outs, _ = step.forward_loss(batch)
    step.bucket.zero()
    loss.backward()
    join_side_streams()
    torch.cuda.synchronize()
    bufs = {k: v.detach().clone() for k, v in net.state_dict().items() if "running_" in k or "num_batches" in k}
    return [o.detach().clone() for o in outs[:3]], loss.detach().clone(), step.bucket.flat.detach().clone(), bufs


def _same(a, b, what):
    for i, (x, y) in enumerate(zip(a[0], b[0])):
        assert torch.equal(x, y), "%s: output %d differs (max %.3g)" % (what, i, float((x - y).abs().max()))
    assert torch.equal(a[1], b[1]), "%s: loss %r vs %r" % (what, a[1].item(), b[1].item())
    if not torch.equal(a[2], b[2]):
        d = (a[2] - b[2]).abs()
        raise AssertionError("%s: gradient bucket differs in %d of %d floats (max %.3g)" % (what, int((d > 0).sum()), d.numel(), float(d.max())))
    for k, v in b[3].items():
        assert torch.equal(a[3][k], v), "%s: BN buffer %s" % (what, k)


@pytest.mark.parametrize("model", ["cmflow", "cmflow_t"])
@pytest.mark.parametrize("B", [8, 64])
def test_fork_point_delays_do_not_change_results(B, model, monkeypatch):
    import bench
    from cmflow_amd import fused_blocks as FB
    from cmflow_amd.cmflow import CMFlow, CMFlow_T
    from cmflow_amd.train import TrainStep
    dev = torch.device("cuda:0")
    torch.backends.cuda.matmul.allow_tf32 = False
    cls = CMFlow if model == "cmflow" else CMFlow_T
    sd = bench.load_weights(model)
    batch = {k: v.to(dev) for k, v in synth.make_batch(B, seed=1234, train_extras=True).items()}

    def build():
        net = cls(bench.Args())
        net.load_state_dict(sd)
        net = net.to(dev).train()
        return net, TrainStep(net, vr_thres=bench.Args.vr_thres)

    # reference: the SAME code path (block calls, two-cloud node, gradient sinks) with every "side stream" being the
    # caller's stream -- all chains serialised, nothing to get wrong
    FB.stress_seed(None)
    real = FB.side_stream
    monkeypatch.setattr(FB, "side_stream", lambda slot, device=None: torch.cuda.current_stream())
    net0, step0 = build()
    want = _snapshot(step0, net0, batch)
    monkeypatch.setattr(FB, "side_stream", real)
    assert float(want[2].abs().sum()) > 0 and torch.isfinite(want[2]).all()

    net, step = build()
    got = _snapshot(step, net, batch)                        # the product's deal, no perturbation, cold plans
    _same(got, want, "side streams, no delays")
    try:
        for seed in range(SEEDS):
            net.load_state_dict(sd)                          # BN buffers back to their start (parameters never moved)
            if model == "cmflow_t":
                step.reset_clip()
            FB.stress_seed(seed)
            got = _snapshot(step, net, batch)
            _same(got, want, "B=%d %s seed %d" % (B, model, seed))
    finally:
        FB.stress_seed(None)


@pytest.mark.parametrize("model", ["cmflow"])
def test_delays_do_not_change_a_full_step_with_overlapped_rccl_all_reduce(model, monkeypatch):
    """The whole optimizer step (TrainStep.__call__) with the gradient bucket reduced in three segments launched from tensor hooks
    DURING backward on RCCL's stream (ReduceOp.AVG, async_op; world size 1 on this box: the collective is real, its result
    the identity) and random delays at every fork point, chain end and tail: bucket after the reduction and parameters
    after Adam must be bit-identical to the undelayed single-stream step without a collective."""
    import torch.distributed as dist
    import bench
    from cmflow_amd import fused_blocks as FB
    from cmflow_amd.cmflow import CMFlow
    from cmflow_amd.train import TrainStep
    dev = torch.device("cuda:0")
    torch.backends.cuda.matmul.allow_tf32 = False
    sd = bench.load_weights(model)
    batch = {k: v.to(dev) for k, v in synth.make_batch(8, seed=1234, train_extras=True).items()}

    def run(overlap, force, seed):
        net = CMFlow(bench.Args())
        net.load_state_dict(sd)
        net = net.to(dev).train()
        step = TrainStep(net, vr_thres=bench.Args.vr_thres)
        step.overlap_allreduce, step.force_allreduce = overlap, force
        FB.stress_seed(seed)
        try:
            loss, _, _, _ = step(batch)
        finally:
            FB.stress_seed(None)
        FB.join_side_streams()
        torch.cuda.synchronize()
        early = step.reducer.early if (overlap and step.reducer is not None) else None
        return loss.clone(), step.bucket.flat.clone(), torch.cat([p.detach().reshape(-1) for p in net.parameters()]).clone(), early

    real = FB.side_stream
    monkeypatch.setattr(FB, "side_stream", lambda slot, device=None: torch.cuda.current_stream())
    want = run(False, False, None)
    monkeypatch.setattr(FB, "side_stream", real)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        for seed in [None] + list(range(min(SEEDS, 20))):
            got = run(True, True, seed)
            assert got[3] == 2, "segments launched from inside backward: %r" % (got[3],)
            for i, what in enumerate(("loss", "gradient bucket", "parameters after Adam")):
                assert torch.equal(got[i], want[i]), "seed %r: %s differs (max %.3g)" % (seed, what, float((got[i] - want[i]).abs().max()))
    finally:
        dist.destroy_process_group()


def test_spin_kernel_delays_only_its_stream():
    """cmf_debug_spin occupies the stream it is given for about the requested time and nothing else."""
    from cmflow_amd import _lib
    from cmflow_amd import fused_blocks as FB
    L = _lib.lib()
    side = FB.side_stream(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(side):
        e0.record()
        _lib.check(L.cmf_debug_spin(2000.0, side.cuda_stream), "spin")
        e1.record()
    m0, m1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    m0.record()
    x = torch.ones(1 << 20, device="cuda").sum()
    m1.record()
    torch.cuda.synchronize()
    assert 1.5 <= e0.elapsed_time(e1) <= 10.0, e0.elapsed_time(e1)
    assert m0.elapsed_time(m1) < 1.5 and x.item() == float(1 << 20)
