"""CPU: the N>1 path with world_size 2 over gloo -- flat-bucket gradient all-reduce (mean), parameter
broadcast, batch sharding (rank r's shard == a single-GPU run on that shard)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cmflow_amd.dp import FlatGradBucket, broadcast_module, shard_batch
    torch.manual_seed(100 + rank)                              # replicas start different ...
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.BatchNorm1d(5), torch.nn.Linear(5, 2))
    broadcast_module(net)                                      # ... and are made identical
    bucket = FlatGradBucket(net)
    g = torch.Generator().manual_seed(7)
    full = {"x": torch.randn(8, 6, generator=g), "y": torch.randn(8, 2, generator=g)}
    mine = shard_batch(full, rank, world)
    bucket.zero()
    loss = ((net(mine["x"]) - mine["y"]) ** 2).mean()
    loss.backward()
    local = bucket.flat.clone()
    bucket.all_reduce_mean()
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    ok = torch.allclose(bucket.flat, sum(gathered) / world, atol=1e-7)
    params = torch.cat([p.detach().flatten() for p in net.parameters()])
    plist = [torch.zeros_like(params) for _ in range(world)]
    dist.all_gather(plist, params)
    same = all(torch.equal(plist[0], q) for q in plist)
    # rank r's shard of the global batch == the r-th contiguous slice
    slice_ok = torch.equal(mine["x"], full["x"][rank * 4:(rank + 1) * 4])
    if rank == 0:
        out.put((ok, same, slice_ok, float((gathered[0] - gathered[1]).abs().max())))
    dist.destroy_process_group()


def test_flat_bucket_allreduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok, same, slice_ok, diff = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok and same and slice_ok
    assert diff > 0           # the two ranks really had different local gradients before the reduce


def _seg_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cmflow_amd.dp import FlatGradBucket, SegmentedReducer
    torch.manual_seed(5)
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 4), torch.nn.Linear(4, 2))
    bucket = FlatGradBucket(net)
    segs = [bucket.segment_of([net[2]]), bucket.segment_of([net[1]]), bucket.segment_of([net[0]])]      # backward order
    red = SegmentedReducer(bucket, segs)
    g = torch.Generator().manual_seed(20 + rank)
    x, y = torch.randn(4, 6, generator=g), torch.randn(4, 2, generator=g)
    # reference: whole-bucket all-reduce after backward
    bucket.zero()
    ((net(x) - y) ** 2).mean().backward()
    local = bucket.flat.clone()
    bucket.all_reduce_mean()
    want = bucket.flat.clone()
    # segments launched from tensor hooks while backward is still running (the last one is left to finish())
    bucket.zero()
    assert red.begin()
    h1 = net[0](x)
    h2 = net[1](h1)
    h2.register_hook(lambda gr: red.launch(0))          # gradient of the last layer's input: its parameters are done
    h1.register_hook(lambda gr: red.launch(1))
    ((net[2](h2) - y) ** 2).mean().backward()
    launched = [w is not None for w in red.work]
    red.finish()
    ok = torch.equal(bucket.flat, want) and launched == [True, True, False]
    bad_cover = False
    try:
        SegmentedReducer(bucket, segs[:2])
    except ValueError:
        bad_cover = True
    if rank == 0:
        out.put((ok, bad_cover, float((local - want).abs().max())))
    dist.destroy_process_group()


def test_segmented_reducer_matches_whole_bucket_allreduce_world2():
    """The overlapped form of the gradient all-reduce (segments launched from tensor hooks during backward, the rest in
    finish()) leaves the bucket bit-identical to one all-reduce after backward; segments must tile the bucket."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_seg_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok, bad_cover, diff = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok and bad_cover and diff > 0


def test_pin_rank_to_cores_is_opt_in_and_follows_gpu_locality(monkeypatch):
    """dp.pin_rank_to_cores: off unless CMF_PIN_CORES=1; the cores come from the GPU's NUMA node (sysfs), ranks that share a
    node split it evenly, a rank whose GPU has no locality information is left alone.  (sysfs is faked: no GPU here.)"""
    from cmflow_amd import dp
    if not hasattr(os, "sched_getaffinity"):
        return
    before = os.sched_getaffinity(0)
    cpus = sorted(before)
    try:
        monkeypatch.delenv("CMF_PIN_CORES", raising=False)
        assert dp.pin_rank_to_cores(0, 2) is None                          # not enabled
        monkeypatch.setenv("CMF_PIN_CORES", "1")
        monkeypatch.setattr(dp, "_gpu_local_cpus", lambda i: None)
        assert dp.pin_rank_to_cores(0, 2) is None                          # no locality information
        if len(cpus) < 4:
            return
        half = len(cpus) // 2
        # four ranks: GPUs 0, 1 on the first node, 2, 3 on the second
        monkeypatch.setattr(dp, "_gpu_local_cpus", lambda i: cpus[:half] if i < 2 else cpus[half:2 * half])
        got = []
        for r in range(4):
            os.sched_setaffinity(0, before)
            got.append(dp.pin_rank_to_cores(r, 4))
        if half >= 4:
            assert all(got) and len({tuple(g) for g in got}) == 4
            assert set(got[0]) | set(got[1]) <= set(cpus[:half]) and set(got[2]) | set(got[3]) <= set(cpus[half:2 * half])
            assert not (set(got[0]) & set(got[1])) and not (set(got[2]) & set(got[3]))
        else:
            assert got == [None] * 4                                       # fewer than two cores per rank: scheduler left alone
        os.sched_setaffinity(0, before)
        assert dp.pin_rank_to_cores(0, 1) is None                          # one rank on the host
    finally:
        os.sched_setaffinity(0, before)
