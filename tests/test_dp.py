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
