"""Which aten ops (and shapes) a training step still issues around the HIP kernels -- torch.profiler, CPU-side op table."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cmflow_amd import synth
from cmflow_amd.cmflow import CMFlow
from cmflow_amd.train import TrainStep
dev = torch.device("cuda:0")
net = CMFlow(bench.Args()); net.load_state_dict(bench.load_weights("cmflow")); net = net.to(dev).train()
step = TrainStep(net)
batch = {k: v.to(dev) for k, v in synth.make_batch(64, seed=1, train_extras=True).items()}
for _ in range(3):
    step(batch)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(batch)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if e.key.startswith("aten::") and e.device_time_total > 0:
        rows.append((e.self_device_time_total, e.count, e.key, str(e.input_shapes)[:110]))
rows.sort(reverse=True)
for t, c, k, s in rows[:70]:
    print("%8.1f us  n=%3d  %-28s %s" % (t, c, k, s))
