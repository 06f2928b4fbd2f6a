"""Kernel-time breakdown of the cost volume alone (fc_layer.forward_pm, forward + backward, train-mode BN) at B=64, N=256.
FC_SERIAL=1: every chain on one stream (isolated kernel durations)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cmflow_amd import synth
from cmflow_amd.cmflow import CMFlow
from cmflow_amd.train import TrainStep
dev = torch.device("cuda:0")
net = CMFlow(bench.Args()); net.load_state_dict(bench.load_weights("cmflow")); net = net.to(dev).train()
step = TrainStep(net)                     # gradient sinks in place, as in training
if os.environ.get("FC_SERIAL") == "1":
    from cmflow_amd import fused_blocks as _FB
    _FB.set_serial(net, True)
b = {k: v.to(dev) for k, v in synth.make_batch(64, seed=1).items()}
x1, x2 = b["pc1"].transpose(1, 2).contiguous(), b["pc2"].transpose(1, 2).contiguous()
torch.manual_seed(0)
f1 = torch.randn(64, 256, 512, device=dev, requires_grad=True); f2 = torch.randn(64, 256, 512, device=dev, requires_grad=True)
g = torch.randn(64, 256, 512, device=dev)


def one():
    cor = net.fc_layer.forward_pm(x1, x2, f1, f2)
    cor.backward(g)
    f1.grad = None; f2.grad = None


for _ in range(3):
    one()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    one()
e1.record(); torch.cuda.synchronize()
print("cost volume, fwd+bwd: %.2f ms" % (e0.elapsed_time(e1) / 10))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(5):
        one()
    torch.cuda.synchronize()
rows = [(e.device_time_total / 5, e.count / 5, e.key) for e in prof.key_averages() if e.device_time_total > 0]
rows.sort(reverse=True)
print("sum of kernel time per call: %.2f ms" % (sum(r[0] for r in rows) / 1e3))
for t, c, k in rows[:40]:
    print("%8.1f us  n=%5.1f  avg %6.1f us  %s" % (t, c, t / c, k[:100]))
