"""Which Python lines launch the torch (aten) kernels of a training step?

    python tools/aten_sources.py [cmflow|cmflow_t|raflow]

Runs a few warm steps, then one step under a TorchDispatchMode that records, for every aten call that may launch device
work (views and allocations excluded), the innermost Python frame inside cmflow_amd/ -- the list of what is left to fold
into the library's own kernels.  (torch.profiler's with_stack yields no stacks on this ROCm build.)
"""
import collections
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
from cmflow_amd import synth  # noqa: E402
from cmflow_amd.cmflow import CMFlow, CMFlow_T  # noqa: E402
from cmflow_amd.raflow import RaFlow  # noqa: E402
from cmflow_amd.train import TrainStep  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "cmflow"
dev = torch.device("cuda:0")
net = {"cmflow": CMFlow, "cmflow_t": CMFlow_T, "raflow": RaFlow}[model](bench.Args())
net.load_state_dict(bench.load_weights(model))
net = net.to(dev).train()
batch = {k: v.to(dev) for k, v in synth.make_batch(64, seed=1234, train_extras=True).items()}
step = TrainStep(net, vr_thres=bench.Args.vr_thres, lr=1e-6 if model == "raflow" else 0.001)
for _ in range(4):
    step(batch)
torch.cuda.synchronize()

from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

NO_LAUNCH = {"empty", "empty_like", "empty_strided", "new_empty", "new_empty_strided", "view", "_unsafe_view", "reshape",
             "as_strided", "slice", "select", "transpose", "t", "expand", "detach", "alias", "unsqueeze", "squeeze", "permute",
             "split", "split_with_sizes", "unbind", "narrow", "_local_scalar_dense", "record_stream", "is_pinned", "lift_fresh",
             "_reshape_alias", "unfold", "chunk", "view_as", "expand_as", "is_same_size", "stride", "size", "sym_size"}


class Sites(TorchDispatchMode):
    """Every aten call that reaches the dispatcher, keyed by the innermost frame inside cmflow_amd/ (forward and backward:
    the autograd engine carries the mode to its worker thread)."""

    def __init__(self):
        super().__init__()
        self.count = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if name not in NO_LAUNCH:
            f = sys._getframe(0).f_back
            site = "?"
            while f is not None:
                fn = f.f_code.co_filename
                if "cmflow_amd/" in fn:
                    site = "%s:%d %s" % (fn[fn.index("cmflow_amd/"):], f.f_lineno, f.f_code.co_name)
                    break
                f = f.f_back
            self.count[(name, site)] += 1
        return func(*args, **(kwargs or {}))


with Sites() as sites:
    step(batch)
    torch.cuda.synchronize()
print("aten calls that may launch device work, one training step: %d" % sum(sites.count.values()))
print("%5s  %-22s %s" % ("n", "operator", "innermost frame in cmflow_amd/"))
for (name, site), n in sorted(sites.count.items(), key=lambda kv: (-kv[1], kv[0])):
    print("%5d  %-22s %s" % (n, name, site))
