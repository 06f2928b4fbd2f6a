"""GPU time per phase of one inference step (events on the main stream)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cmflow_amd import synth
from cmflow_amd.cmflow import CMFlow
dev = torch.device("cuda:0")
net = CMFlow(bench.Args()); net.load_state_dict(bench.load_weights("cmflow")); net = net.to(dev).eval()
b = {k: v.to(dev) for k, v in synth.make_batch(64, seed=1234).items()}
marks = []
def mark(name):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((name, e))
for name, attr in (("mse_layer", "forward_pm"), ("fc_layer", "forward_pm"), ("mse_layer2", "forward_pm"), ("fp", "forward_pm"), ("mp", "forward_pm")):
    m = getattr(net, name)
    orig = getattr(m, attr)
    def wrapped(*a, _o=orig, _n=name, **k):
        mark(_n + ":begin"); r = _o(*a, **k); mark(_n + ":end"); return r
    setattr(m, attr, wrapped)
net.head_streams = False
tot = {}
with torch.no_grad():
    for it in range(8):
        marks.clear()
        mark("step:begin")
        net(b["pc1"], b["pc2"], b["ft1"], b["ft2"], None, "test")
        mark("step:end")
        torch.cuda.synchronize()
        if it < 3:
            continue
        prev = marks[0]
        for name, e in marks[1:]:
            key = prev[0] + " -> " + name
            tot[key] = tot.get(key, 0.0) + prev[1].elapsed_time(e)
            prev = (name, e)
for k, v in tot.items():
    print("%-40s %.3f ms" % (k, v / 5))
