"""Does a sample's result depend on the batch it sits in, under the bf16x3 GEMM mode?  (It must not.)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cmflow_amd import _lib, synth
from cmflow_amd.cmflow import CMFlow
from cmflow_amd.fused import gemm
dev = torch.device("cuda:0")
for mode in ("fp32", "bf16x3"):
    _lib.set_gemm_mode(mode)
    torch.manual_seed(0)
    for (M, N, K) in ((16384, 512, 512), (16384, 256, 512), (524288, 256, 512), (65536, 512, 1040)):
        A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev)
        full = gemm(A, W)
        part = gemm(A[: M // 8].contiguous(), W)
        again = gemm(A, W)
        print(mode, (M, N, K), "rows of a sub-batch equal:", torch.equal(full[: M // 8], part), " repeat equal:", torch.equal(full, again),
              " max diff %.3g" % (full[: M // 8] - part).abs().max().item())
    net = CMFlow(bench.Args()); net.load_state_dict(bench.load_weights("cmflow")); net = net.to(dev).eval()
    b = {k: v.to(dev) for k, v in synth.make_batch(64, seed=77).items()}
    with torch.no_grad():
        full = net(b["pc1"], b["pc2"], b["ft1"], b["ft2"], None, "test"); lf = {k: v.clone() for k, v in net.last.items()}
        part = net(*(b[k][:8] for k in ("pc1", "pc2", "ft1", "ft2")), None, "test"); lp = {k: v.clone() for k, v in net.last.items()}
    for k in lf:
        print(mode, k, "max diff %.3g (scale %.3g)" % ((lf[k][:8] - lp[k]).abs().max().item(), lf[k].abs().max().item()))
    for j, name in enumerate(("sf_agg", "stat_cls", "pre_trans")):
        print(mode, name, "max diff %.3g" % (full[j][:8] - part[j]).abs().max().item())
    print(mode, "mask flips", int((full[3][:8] != part[3]).sum()))
