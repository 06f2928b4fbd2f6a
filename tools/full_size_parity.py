"""BASELINE config (B=64, N=256), eval mode: HIP path vs the CPU oracle on the same batch -- EPE, max errors, mask equality."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cmflow_amd import synth
from cmflow_amd.cmflow import CMFlow
from oracle import cmflow_oracle as O
torch.set_num_threads(min(16, os.cpu_count() or 1))
dev = torch.device("cuda:0")
sd = bench.load_weights("cmflow")
ref = O.CMFlow(bench.Args()); ref.load_state_dict(sd); ref.eval()
net = CMFlow(bench.Args()); net.load_state_dict(sd); net = net.to(dev).eval()
b = synth.make_batch(64, seed=2024)
t0 = time.perf_counter()
with torch.no_grad():
    want = ref(b["pc1"], b["pc2"], b["ft1"], b["ft2"], None, "test")
    t1 = time.perf_counter()
    got = net(*(b[k].to(dev) for k in ("pc1", "pc2", "ft1", "ft2")), None, "test")
flips = (got[3].cpu() != want[3])
epe = (got[0].cpu() - want[0]).norm(dim=1)
print("oracle forward %.1f s; mask flips %d of %d; EPE mean %.3g max %.3g (non-flipped max %.3g); stat_cls max err %.3g; pre_trans max err %.3g"
      % (t1 - t0, int(flips.sum()), flips.numel(), epe.mean().item(), epe.max().item(), epe[~flips].max().item(),
         (got[1].cpu() - want[1]).abs().max().item(), (got[2].cpu() - want[2]).abs().max().item()))
