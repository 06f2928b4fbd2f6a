"""Worker of tests/test_gpu_model.py::test_two_rank_cmflow_step_matches_single_rank_shards -- one rank of a 2-rank
data-parallel CMFlow training step (both ranks on cuda:0 over gloo: the GPU box has one GPU; the collective semantics
are backend-independent, RCCL itself is exercised with world size 1 by test_single_rank_rccl_all_reduce).

    python -m torch.distributed.run --nproc-per-node 2 ... tests/dp_worker.py OUT_DIR

Each rank takes its shard of the global batch (dp.shard_batch, as nn.DataParallel's scatter: models/model.py:40-42),
runs forward / loss / backward, keeps a copy of its LOCAL gradient bucket, all-reduces (mean), steps Adam, and saves
what the test compares.
"""
import json
import os
import sys

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    out_dir = sys.argv[1]
    global_b = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    model = sys.argv[3] if len(sys.argv) > 3 else "cmflow"
    n_frames = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    import bench
    from cmflow_amd import synth
    from cmflow_amd.cmflow import CMFlow, CMFlow_T
    from cmflow_amd.dp import broadcast_module, shard_batch
    from cmflow_amd.fused_blocks import join_side_streams
    from cmflow_amd.train import TrainStep

    net = {"cmflow": CMFlow, "cmflow_t": CMFlow_T}[model](bench.Args())
    net.load_state_dict(bench.load_weights(model))
    net = net.to(dev).train()
    if rank != 0:                                   # replicas must come out of the broadcast identical, whatever they held
        with torch.no_grad():
            for p in net.parameters():
                p.add_(0.01)
    broadcast_module(net)
    step = TrainStep(net, vr_thres=bench.Args.vr_thres)
    step.overlap_allreduce = True                   # the segmented form (opt-in in the product): launched from hooks during backward
    frames = []
    # CMFlow-T: the frames of one mini-clip (clip_util.py:34-62) -- the GRU state is carried per rank, detached, and the
    # optimizer steps after every frame on the all-reduced gradient
    for f in range(n_frames):
        gb = synth.make_batch(global_b, seed=777 + f, train_extras=True)
        b = {k: v.to(dev) for k, v in shard_batch(gb, rank, world).items()}
        # the product's own step: the gradient all-reduce is cut into segments launched from tensor hooks during backward
        # (train.TrainStep / dp.SegmentedReducer).  The LOCAL gradient of a segment is copied right before its collective.
        local = torch.empty_like(step.bucket.flat)
        red = step.reducer
        real_launch = red.launch

        def launch(i, red=red, real=real_launch, local=local):
            if red.active and red.work[i] is None:
                off, n = red.segments[i]
                local[off:off + n].copy_(step.bucket.flat[off:off + n])
            real(i)
        red.launch = launch
        opt_step = step.opt.step
        averaged = []
        step.opt.step = lambda: (averaged.append(step.bucket.flat.detach().clone()), opt_step())[1]
        loss, items, outs, _ = step(b)
        step.opt.step = opt_step
        red.launch = real_launch
        averaged = averaged[0]
        torch.cuda.synchronize()
        frames.append({"local": local.cpu(), "averaged": averaged.cpu(), "loss": loss.detach().cpu(), "early": red.early,
                       "outs": [o.detach().cpu() for o in outs[:3]],
                       "gfeat": step.gfeat.detach().cpu() if step.gfeat is not None else None})
    sd = net.state_dict()
    rec = dict(frames[0])
    rec.update({"frames": frames,
                "params": {k: v.detach().cpu() for k, v in net.named_parameters()},
                "buffers": {k: v.detach().cpu() for k, v in sd.items() if "running_" in k or "num_batches" in k}})
    torch.save(rec, os.path.join(out_dir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        json.dump({"world": world, "global_b": global_b, "model": model, "frames": n_frames}, open(os.path.join(out_dir, "done.json"), "w"))


if __name__ == "__main__":
    main()
