"""Test helper (run as a child process: the library reads its environment switches once): one training step of bench.py's weights on a
seeded batch of B samples -- loss, every parameter gradient and every BN buffer after the step are saved to the given file.
Third argument `eval`: BatchNorm in eval mode (the regime of every epoch after the first in the reference's training loop)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import bench
from cmflow_amd import synth
from cmflow_amd.cmflow import CMFlow
from cmflow_amd.train import TrainStep

out, B = sys.argv[1], int(sys.argv[2])
dev = torch.device("cuda:0")
net = CMFlow(bench.Args())
net.load_state_dict(bench.load_weights("cmflow"))
net = net.to(dev).train()
if len(sys.argv) > 3 and sys.argv[3] == "eval":
    net.eval()
b = {k: v.to(dev) for k, v in synth.make_batch(B, seed=1234, train_extras=True).items()}
step = TrainStep(net, vr_thres=0.3)
loss, items, outs, _ = step.forward_loss(b)
step.bucket.zero()
loss.backward()
torch.cuda.synchronize()
res = {"loss": loss.detach().cpu()}
for k, p in net.named_parameters():
    if p.grad is not None:
        res["g." + k] = p.grad.detach().cpu().clone()
for k, v in net.named_buffers():
    res["b." + k] = v.detach().cpu().clone()
torch.save(res, out)
