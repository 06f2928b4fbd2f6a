"""Peak device memory of the training step (B = 64): torch.cuda.max_memory_allocated over 3 steps behind 2 warm-up steps.  Run from the
root of the tree to measure (argv[1] = that root; default: this repo): tools/session.sh peak_mem compares the current tree with a worktree of
an older commit (the arenas sized for the materialised first-layer tensors)."""
import os, sys
root = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
os.chdir(root)
import torch
import bench
from cmflow_amd import synth
from cmflow_amd.cmflow import CMFlow
from cmflow_amd.train import TrainStep
dev = torch.device("cuda:0")
net = CMFlow(bench.Args()); net.load_state_dict(bench.load_weights("cmflow")); net = net.to(dev).train()
b = {k: v.to(dev) for k, v in synth.make_batch(64, seed=1234, train_extras=True).items()}
step = TrainStep(net, vr_thres=0.3)
for _ in range(2):
    step(b)
torch.cuda.synchronize(); torch.cuda.reset_peak_memory_stats()
for _ in range(3):
    step(b)
torch.cuda.synchronize()
print("%s: max_memory_allocated %.3f GB, max_memory_reserved %.3f GB" % (root, torch.cuda.max_memory_allocated() / 1e9, torch.cuda.max_memory_reserved() / 1e9))
