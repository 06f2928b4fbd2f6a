"""Which aten kernels a training step launches, and from where: torch.profiler with stacks over 4 steps; per (op, first frame under
cmflow_amd/) the calls per step.  The step's own kernels go through ctypes and do not appear here."""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cmflow_amd import synth
from cmflow_amd.cmflow import CMFlow
from cmflow_amd.train import TrainStep
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda:0")
net = CMFlow(bench.Args()); net.load_state_dict(bench.load_weights("cmflow")); net = net.to(dev).train()
step = TrainStep(net, vr_thres=bench.Args.vr_thres)
batch = {k: v.to(dev) for k, v in synth.make_batch(64, seed=1234, train_extras=True).items()}
for _ in range(4):
    step(batch)
torch.cuda.synchronize()
STEPS = 4
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(STEPS):
        step(batch)
    torch.cuda.synchronize()
kern = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
names = collections.Counter(e.name for e in kern)
print("GPU kernels + copies + memsets per step (all streams): %.1f" % (len(kern) / STEPS))
print("   of which memcpy/memset: %.1f" % (sum(v for k, v in names.items() if k.startswith("Mem")) / STEPS))
rows = []
for e in prof.key_averages(group_by_stack_n=25):
    if not e.key.startswith("aten::") or e.device_time_total <= 0 or e.self_device_time_total <= 0:
        continue
    site = [fr.split("cmflow_amd/")[-1] for fr in e.stack if "cmflow_amd/" in fr and "site-packages" not in fr]
    rows.append((e.count / STEPS, e.self_device_time_total / STEPS, e.key, " <- ".join(s_[:70] for s_ in site[:2]) or (e.stack[0][:100] if e.stack else "?")))
rows.sort(key=lambda r: -r[1])
print("aten ops with device time, per step: %.1f calls, %.1f us" % (sum(r[0] for r in rows), sum(r[1] for r in rows)))
for n, t, k, site in rows[:70]:
    print("%5.1f /step %7.1f us  %-22s %s" % (n, t, k, site))


# Call sites: one more step under a dispatch mode that records, for every aten op dispatched from Python (forward code and the
# backward methods of the blocks' autograd Functions), the innermost frame under cmflow_amd/.
import traceback
from torch.utils._python_dispatch import TorchDispatchMode
sites = collections.Counter()


class Census(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func.overloadpacket).replace("aten.", "aten::")
        if name.split("::")[-1] in ("view", "as_strided", "t", "transpose", "slice", "select", "unsqueeze", "squeeze", "expand", "detach", "alias",
                                    "_unsafe_view", "permute", "reshape", "empty", "empty_like", "empty_strided", "split", "unbind", "narrow",
                                    "new_empty", "new_empty_strided", "_local_scalar_dense", "is_same_size", "stride", "size", "sym_size", "lift_fresh"):
            return out
        site = "?"
        for fr in reversed(traceback.extract_stack()):
            if "cmflow_amd/" in fr.filename and "site-packages" not in fr.filename:
                site = "%s:%d %s" % (fr.filename.split("cmflow_amd/")[-1], fr.lineno, (fr.line or "")[:90])
                break
        sites[(name, site)] += 1
        return out


with Census():
    step(batch)
torch.cuda.synchronize()
print("\naten ops dispatched from Python in one step (views and allocations excluded): %d" % sum(sites.values()))
for (name, site), n in sorted(sites.items(), key=lambda kv: (-kv[1], kv[0])):
    print("%3d  %-24s %s" % (n, name, site))
