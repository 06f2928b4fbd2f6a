"""GPU: the PointNet++ modules (cmflow_amd.pointnet2_modules) against outputs of the reference's
own lib/pointnet2_modules.py (tests/golden/make_golden_modules.py), state_dict keys included."""
import os

import numpy as np
import pytest
import torch

from cmflow_amd import pointnet2_modules as M

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _build(dev):
    mods = dict(msg=M.PointnetSAModuleMSG(npoint=64, radii=[2.0, 6.0], nsamples=[8, 16], mlps=[[3, 16, 32], [3, 16, 48]]),
                sa=M.PointnetSAModule(mlp=[80, 64, 64], npoint=16, radius=12.0, nsample=8),
                ga=M.PointnetSAModule(mlp=[64, 96]),
                fp=M.PointnetFPModule(mlp=[64 + 80, 64, 32]))
    return {k: v.to(dev) for k, v in mods.items()}


def _state(g, prefix):
    pre = prefix + "/state/"
    return {k[len(pre):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(pre)}


def test_modules_match_reference_goldens(dev, golden_dir):
    g = np.load(os.path.join(golden_dir, "pointnet2_modules_kat.npz"))
    mods = _build(dev)
    for name, m in mods.items():
        sd = _state(g, name + "_before")
        assert set(sd) == set(m.state_dict()), (name, set(sd) ^ set(m.state_dict()))     # same checkpoint layout
        m.load_state_dict(sd, strict=True)
    xyz, feats = torch.from_numpy(g["xyz"]).to(dev), torch.from_numpy(g["feats"]).to(dev)
    for mode in ("eval", "train"):
        for m in mods.values():
            m.train(mode == "train")
        with torch.no_grad():
            xyz1, f1 = mods["msg"](xyz, feats)
            xyz2, f2 = mods["sa"](xyz1, f1)
            none_xyz, f3 = mods["ga"](xyz2, f2)
            up = mods["fp"](xyz1, xyz2, f1, f2)
        assert none_xyz is None
        assert torch.equal(xyz1.cpu(), torch.from_numpy(g[mode + "/xyz1"]))               # FPS + gather: bit-exact
        assert torch.equal(xyz2.cpu(), torch.from_numpy(g[mode + "/xyz2"]))
        for k, v in dict(f1=f1, f2=f2, f3=f3, up=up).items():
            ref = g["%s/%s" % (mode, k)]
            np.testing.assert_allclose(v.cpu().numpy(), ref, rtol=1e-4, atol=1e-4 * float(np.abs(ref).max()), err_msg=mode + k)
    for name, m in mods.items():                                                          # running statistics after train
        for k, v in _state(g, name + "_after").items():
            got = m.state_dict()[k].cpu().numpy()
            np.testing.assert_allclose(got, v.numpy(), rtol=1e-4, atol=1e-5, err_msg=name + k)


def test_modules_backward_runs_through_hip_ops(dev):
    mods = _build(dev)
    g = torch.Generator().manual_seed(0)
    xyz = (torch.rand(2, 128, 3, generator=g) * 20).to(dev)
    feats = torch.randn(2, 3, 128, generator=g).to(dev).requires_grad_(True)
    xyz1, f1 = mods["msg"](xyz, feats)
    xyz2, f2 = mods["sa"](xyz1, f1)
    up = mods["fp"](xyz1, xyz2, f1, f2)
    up.square().mean().backward()
    assert feats.grad is not None and torch.isfinite(feats.grad).all() and feats.grad.abs().sum() > 0
    assert all(p.grad is not None for p in mods["fp"].parameters())
