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


@pytest.mark.parametrize("spec,shape,bn", [([6, 16, 32], (2, 6, 64, 8), True), ([83, 64, 64], (2, 83, 16, 8), True),
                                           ([144, 64, 32], (3, 144, 40, 1), True), ([8, 16, 32, 64, 64, 128], (2, 8, 32, 4), True),
                                           ([12, 32, 16], (2, 12, 24, 4), False)])
def test_shared_mlp_native_path_matches_torch_modules(dev, spec, shape, bn):
    """pytorch_utils.SharedMLP on device tensors runs cmf_mlp_forward / _backward (BatchNorm layers, up to four per call) or cmf_gemm
    with the bias / ReLU epilogue (no BatchNorm) instead of nn.Conv2d / nn.BatchNorm2d: outputs, input gradient, every parameter
    gradient and the BatchNorm running statistics against the same module run through torch (its own nn.Sequential.forward), train and
    eval mode."""
    from cmflow_amd import pytorch_utils as P
    torch.manual_seed(5)
    m = P.SharedMLP(list(spec), bn=bn).to(dev)
    assert m._native
    for p in m.parameters():
        p.data.add_(0.1 * torch.randn_like(p))
    x0 = torch.randn(*shape, device=dev)
    res = {}
    for native in (True, False):
        mm = P.SharedMLP(list(spec), bn=bn).to(dev)
        mm.load_state_dict(m.state_dict())
        mm._native = native
        out = {}
        for mode in ("train", "eval"):
            mm.train(mode == "train")
            x = x0.clone().requires_grad_(True)
            y = mm(x)
            (y * torch.linspace(0.5, 1.5, y.numel(), device=dev).view_as(y)).sum().backward()
            out[mode] = (y.detach(), x.grad.detach(), [p.grad.detach().clone() for p in mm.parameters()])
            mm.zero_grad()
        out["state"] = {k: v.clone() for k, v in mm.state_dict().items()}
        res[native] = out
    for mode in ("train", "eval"):
        (ya, ga, pa), (yb, gb, pb) = res[True][mode], res[False][mode]
        assert ya.shape == yb.shape
        tol = lambda r: dict(rtol=2e-4, atol=2e-5 * float(r.abs().max()) + 1e-7)
        assert torch.allclose(ya, yb, **tol(yb)), (mode, float((ya - yb).abs().max()))
        assert torch.allclose(ga, gb, **tol(gb)), (mode, float((ga - gb).abs().max()))
        for u, v in zip(pa, pb):
            assert torch.allclose(u, v, rtol=5e-4, atol=5e-5 * float(v.abs().max()) + 1e-6), (mode, float((u - v).abs().max()), float(v.abs().max()))
    for k, v in res[False]["state"].items():
        assert torch.allclose(res[True]["state"][k].float(), v.float(), rtol=1e-4, atol=1e-6), k


from test_host_logic import PYTORCH_UTILS_CASES, _pytorch_utils_case


@pytest.mark.parametrize("name", PYTORCH_UTILS_CASES)
def test_pytorch_utils_classes_match_reference_goldens_on_the_gpu(dev, golden_dir, name, monkeypatch):
    """The same goldens through the device path: every 1x1 conv / linear layer on cmf_gemm (cmf_mlp_forward / _backward for the
    conv + BN + ReLU stacks), outputs 1e-4, gradients 1e-3 of their largest entry, BN buffers 1e-4 -- and the library's GEMM really ran
    (no silent torch path)."""
    from cmflow_amd import fused_blocks as FB
    calls = []
    for fn in ("linear", "mlp_chain_w"):
        real = getattr(FB, fn)
        monkeypatch.setattr(FB, fn, (lambda real: lambda *a, **k: (calls.append(1), real(*a, **k))[1])(real))
    g, res, m = _pytorch_utils_case(golden_dir, name, dev)
    n_layers = sum(1 for x in m.modules() if isinstance(x, (torch.nn.Conv1d, torch.nn.Conv2d, torch.nn.Linear)))
    assert len(calls) >= 2 and len(calls) <= 2 * n_layers, (name, len(calls))          # eval + train call: every conv through the library
    for k, v in res.items():
        want = g[name + "/" + k]
        scale = max(1.0, float(np.abs(want).max()))
        tol = 1e-4 if (k in ("eval", "train") or k.startswith("buf/")) else 1e-3
        np.testing.assert_allclose(v.detach().cpu().numpy(), want, rtol=tol, atol=tol * scale, err_msg=name + "/" + k)
