"""GPU time per phase of one training step (events on the main stream; side streams are joined at phase ends)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cmflow_amd import synth
from cmflow_amd.cmflow import CMFlow
from cmflow_amd.train import TrainStep
dev = torch.device("cuda:0")
net = CMFlow(bench.Args()); net.load_state_dict(bench.load_weights("cmflow")); net = net.to(dev).train()
step = TrainStep(net)
batch = {k: v.to(dev) for k, v in synth.make_batch(64, seed=1234, train_extras=True).items()}
for _ in range(3): step(batch)
marks = []
def mark(name):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((name, e))
def hook(mod, name):
    mod.register_forward_pre_hook(lambda m, i: mark(name + ":begin"))
    mod.register_forward_hook(lambda m, i, o: mark(name + ":end"))
# point-major path calls forward_pm, not forward: wrap them
import types
def tag_bwd(r, n):                       # the gradient of a module's output arrives right before its backward runs
    for t in (r if isinstance(r, tuple) else (r,)):
        if torch.is_tensor(t) and t.requires_grad:
            t.register_hook(lambda g, _n=n: mark(_n + ":grad_in"))
            break
for name, attr in (("mse_layer", "forward_pm_pair"), ("fc_layer", "forward_pm"), ("mse_layer2", "forward_pm"), ("fp", "forward_pm"),
                   ("mp", "forward_pm")):
    m = getattr(net, name)
    orig = getattr(m, attr)
    def wrapped(*a, _o=orig, _n=name, **k):
        mark(_n + ":begin"); r = _o(*a, **k); mark(_n + ":end"); tag_bwd(r, _n); return r
    setattr(m, attr, wrapped)
tot = {}
for it in range(5):
    marks.clear()
    mark("step:begin")
    loss, items, outs, labels = step.forward_loss(batch)
    mark("loss:end")
    step.bucket.zero(); loss.backward(); mark("backward:end")
    step.bucket.all_reduce_mean(); step.opt.step(); mark("opt:end")
    torch.cuda.synchronize()
    prev = marks[0]
    for name, e in marks[1:]:
        key = prev[0] + " -> " + name
        tot[key] = tot.get(key, 0.0) + prev[1].elapsed_time(e)
        prev = (name, e)
for k, v in tot.items():
    print("%-40s %.2f ms" % (k, v / 5))
