"""Is the training step host-bound or GPU-bound?  Times each step() call on the host (no synchronisation) and the final
drain: if the calls return faster than the GPU executes, the last synchronize() waits for the backlog.

    python tools/host_time_probe.py [steps]
"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cmflow_amd import synth
from cmflow_amd.cmflow import CMFlow
from cmflow_amd.train import TrainStep

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
net = CMFlow(bench.Args()); net.load_state_dict(bench.load_weights("cmflow")); net = net.to(dev).train()
batch = {k: v.to(dev) for k, v in synth.make_batch(64, seed=1234, train_extras=True).items()}
step = TrainStep(net, vr_thres=bench.Args.vr_thres)
for _ in range(5):
    step(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
calls = []
phases = {"fwd": 0.0, "bwd": 0.0, "opt": 0.0}
for _ in range(n):
    a = time.perf_counter()
    loss, items, outs, labels = step.forward_loss(batch)
    b = time.perf_counter()
    step.bucket.zero(); loss.backward()
    c = time.perf_counter()
    from cmflow_amd.fused_blocks import join_side_streams
    join_side_streams(); step.bucket.all_reduce_mean(); step.opt.step()
    d = time.perf_counter()
    calls.append(d - a)
    phases["fwd"] += b - a; phases["bwd"] += c - b; phases["opt"] += d - c
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("steps %d: host time in the calls %.2f ms/step (fwd %.2f, bwd %.2f, opt %.2f), drain after the last call %.2f ms, wall %.2f ms/step"
      % (n, (t1 - t0) / n * 1e3, phases["fwd"] / n * 1e3, phases["bwd"] / n * 1e3, phases["opt"] / n * 1e3, (t2 - t1) * 1e3, (t2 - t0) / n * 1e3))
print("per-call host ms:", " ".join("%.1f" % (c * 1e3) for c in calls))
