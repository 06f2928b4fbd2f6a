#!/usr/bin/env python3
"""Golden vectors for the PointNet++ modules: the REFERENCE's lib/pointnet2_modules.py
(PointnetSAModuleMSG, PointnetSAModule incl. the GroupAll form, PointnetFPModule) run on torch-CPU behind the
shims of make_golden.py (its extension replaced by oracle/ops.py).  Build container only.

    python tests/golden/make_golden_modules.py   ->  tests/golden/pointnet2_modules_kat.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import install_shims, synth  # noqa: E402


def randomise_bn(mod, g):
    for m in mod.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data = torch.rand(m.weight.shape, generator=g) * 0.8 + 0.6
            m.bias.data = torch.randn(m.bias.shape, generator=g) * 0.2
            m.running_mean.data = torch.randn(m.running_mean.shape, generator=g) * 0.3
            m.running_var.data = torch.rand(m.running_var.shape, generator=g) * 1.5 + 0.5


def main():
    install_shims()
    from lib import pointnet2_modules as M                       # the reference
    g = torch.Generator().manual_seed(77)
    torch.manual_seed(77)
    b = synth.make_batch(3, 256, seed=77)
    xyz = b["pc1"].permute(0, 2, 1).contiguous()                 # (B,N,3)
    feats = b["ft1"].contiguous()                                # (B,3,N)
    out = {"xyz": xyz.numpy(), "feats": feats.numpy()}

    def save_state(prefix, mod):
        for k, v in mod.state_dict().items():
            out["%s/state/%s" % (prefix, k)] = v.detach().clone().numpy()

    msg = M.PointnetSAModuleMSG(npoint=64, radii=[2.0, 6.0], nsamples=[8, 16], mlps=[[3, 16, 32], [3, 16, 48]])
    sa = M.PointnetSAModule(mlp=[80, 64, 64], npoint=16, radius=12.0, nsample=8)
    ga = M.PointnetSAModule(mlp=[64, 96])                        # npoint None -> GroupAll
    fp = M.PointnetFPModule(mlp=[64 + 80, 64, 32])
    for m in (msg, sa, ga, fp):
        randomise_bn(m, g)
    for mode in ("eval", "train"):
        for m in (msg, sa, ga, fp):
            m.train(mode == "train")
            if mode == "train":
                save_state("%s_before" % {id(msg): "msg", id(sa): "sa", id(ga): "ga", id(fp): "fp"}[id(m)], m)
        with torch.no_grad():
            xyz1, f1 = msg(xyz, feats)                           # (B,64,3), (B,80,64)
            xyz2, f2 = sa(xyz1, f1)                              # (B,16,3), (B,64,16)
            _, f3 = ga(xyz2, f2)                                 # (B,96,1)
            up = fp(xyz1, xyz2, f1, f2)                          # (B,32,64)
        for k, v in dict(xyz1=xyz1, f1=f1, xyz2=xyz2, f2=f2, f3=f3, up=up).items():
            out["%s/%s" % (mode, k)] = v.detach().clone().numpy()
    for name, m in (("msg", msg), ("sa", sa), ("ga", ga), ("fp", fp)):
        save_state("%s_after" % name, m)
    np.savez_compressed(os.path.join(HERE, "pointnet2_modules_kat.npz"), **out)
    print({k: v.shape for k, v in out.items() if "/state/" not in k})


if __name__ == "__main__":
    main()
