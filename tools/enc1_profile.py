"""Kernel-time breakdown of the first encoder alone (both clouds, forward + backward, train-mode BN) at B=64, N=256.
ENC1_SERIAL=1: every chain on one stream (isolated kernel durations)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cmflow_amd import synth
from cmflow_amd.cmflow import CMFlow
from cmflow_amd.train import TrainStep
dev = torch.device("cuda:0")
net = CMFlow(bench.Args()); net.load_state_dict(bench.load_weights("cmflow")); net = net.to(dev).train()
step = TrainStep(net)                     # gradient sinks in place, as in training
if os.environ.get("ENC1_SERIAL") == "1":
    from cmflow_amd import fused_blocks as _FB
    _FB.set_serial(net, True)
b = {k: v.to(dev) for k, v in synth.make_batch(64, seed=1).items()}
x1, x2 = b["pc1"].transpose(1, 2).contiguous(), b["pc2"].transpose(1, 2).contiguous()
a1 = torch.nn.functional.pad(b["ft1"].transpose(1, 2).contiguous(), (0, 1))
a2 = torch.nn.functional.pad(b["ft2"].transpose(1, 2).contiguous(), (0, 1))


def one():
    f1, f2 = net.mse_layer.forward_pm_pair(x1, a1, x2, a2)
    g = torch.ones_like(f1)
    torch.autograd.backward([f1, f2], [g, g])


for _ in range(3):
    one()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    one()
e1.record(); torch.cuda.synchronize()
print("first encoder, both clouds, fwd+bwd: %.2f ms" % (e0.elapsed_time(e1) / 10))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(5):
        one()
    torch.cuda.synchronize()
rows = [(e.device_time_total / 5, e.count / 5, e.key) for e in prof.key_averages() if e.device_time_total > 0]
rows.sort(reverse=True)
print("sum of kernel time per call: %.2f ms" % (sum(r[0] for r in rows) / 1e3))
for t, c, k in rows[:28]:
    print("%8.1f us  n=%5.1f  avg %6.1f us  %s" % (t, c, t / c, k[:90]))
