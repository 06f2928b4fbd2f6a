"""Config 5 shape through the whole model (N=4096 LiDAR-like clouds, eval mode): HIP path vs the CPU oracle on the same
batch -- EPE, max errors, mask equality -- and the step time at B=32."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cmflow_amd import synth
from cmflow_amd.cmflow import CMFlow
from oracle import cmflow_oracle as O
torch.set_num_threads(min(16, os.cpu_count() or 1))
dev = torch.device("cuda:0")


class Args(bench.Args):
    num_points = 4096


sd = bench.load_weights("cmflow")
ref = O.CMFlow(Args()); ref.load_state_dict(sd); ref.eval()
net = CMFlow(Args()); net.load_state_dict(sd); net = net.to(dev).eval()
b = synth.make_batch(2, N=4096, seed=2025, lidar=True)
t0 = time.perf_counter()
with torch.no_grad():
    want = ref(b["pc1"], b["pc2"], b["ft1"], b["ft2"], None, "test")
    t1 = time.perf_counter()
    got = net(*(b[k].to(dev) for k in ("pc1", "pc2", "ft1", "ft2")), None, "test")
flips = (got[3].cpu() != want[3])
epe = (got[0].cpu() - want[0]).norm(dim=1)
print("N=4096 B=2: oracle forward %.1f s; mask flips %d of %d; EPE mean %.3g max %.3g (non-flipped max %.3g); stat_cls max err %.3g; "
      "pre_trans max err %.3g" % (t1 - t0, int(flips.sum()), flips.numel(), epe.mean().item(), epe.max().item(),
                                  epe[~flips].max().item() if (~flips).any() else float("nan"),
                                  (got[1].cpu() - want[1]).abs().max().item(), (got[2].cpu() - want[2]).abs().max().item()))
big = {k: v.to(dev) for k, v in synth.make_batch(32, N=4096, seed=7, lidar=True).items()}
with torch.no_grad():
    for _ in range(2):
        net(big["pc1"], big["pc2"], big["ft1"], big["ft2"], None, "test")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        net(big["pc1"], big["pc2"], big["ft1"], big["ft2"], None, "test")
    e1.record()
    torch.cuda.synchronize()
print("N=4096 B=32 forward: %.1f ms per step (%.0f frame-pairs/s), peak memory %.1f GB" % (e0.elapsed_time(e1) / 5, 32 / (e0.elapsed_time(e1) / 5e3),
                                                                                  torch.cuda.max_memory_allocated() / 2**30))
