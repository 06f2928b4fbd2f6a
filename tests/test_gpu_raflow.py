"""GPU: RaFlow (cmflow_amd.raflow) against the goldens of the reference's models/raflow.py --
evaluation (strict and loose inlier threshold: both branches of the SFR module) and one self-supervised train
step -- on the fused point-major path ("pm") and on the reference's own op sequence over the drop-in kernels ("ref":
the oracle's RaFlow on the GPU with its native ops bound to libcmflow_hip.so, tests/hip_ops.py)."""
import json
import os

import numpy as np
import pytest
import torch

from cmflow_amd import synth
from cmflow_amd.raflow import RaFlow
from cmflow_amd.train import TrainStep

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


class A:
    num_points = 256
    rigid_thres = 0.15


def _net(golden_dir, dev, path, monkeypatch, thres=0.15):
    man = json.load(open(os.path.join(golden_dir, "state_manifest_raflow.json")))
    a = A()
    a.rigid_thres = thres
    sd = synth.synth_state_dict(man, seed=1234, calib=os.path.join(golden_dir, "bn_calib_raflow.npz"))
    if path == "ref":
        import hip_ops
        from oracle import cmflow_oracle as O
        return hip_ops.reference_layout_net(O.RaFlow, a, sd, dev, monkeypatch)
    net = RaFlow(a)
    net.path = path
    net.load_state_dict(sd)
    return net.to(dev)


@pytest.mark.parametrize("path", ["pm", "ref"])
@pytest.mark.parametrize("tag,thres", [("raflow_eval_synth_b2", 0.15), ("raflow_eval_synth_b4_loose", 2.0)])
def test_raflow_eval_matches_reference(dev, golden_dir, path, tag, thres, monkeypatch):
    g = np.load(os.path.join(golden_dir, tag + ".npz"))
    net = _net(golden_dir, dev, path, monkeypatch, thres).eval()
    with torch.no_grad():
        out, sf, trans, mask_s = net(*(torch.from_numpy(g[k]).to(dev) for k in ("pc1", "pc2", "ft1", "ft2", "interval")))
    flips = (mask_s.cpu().numpy() != g["mask_s"])
    assert flips.mean() <= 2e-3, flips.mean()                       # |residual / v_r| against a threshold: near-ties only
    np.testing.assert_allclose(out.cpu().numpy(), g["output"], rtol=0, atol=2e-4)
    ok = ~flips[:, None, :].repeat(3, axis=1)
    np.testing.assert_allclose(sf.cpu().numpy()[ok], g["sf_agg"][ok], rtol=0, atol=3e-4)
    if not flips.any():
        np.testing.assert_allclose(trans.cpu().numpy(), g["pre_trans"], rtol=3e-5, atol=2e-4)


@pytest.mark.parametrize("path", ["pm", "ref"])
def test_raflow_train_step_matches_reference(dev, golden_dir, path, monkeypatch):
    g = np.load(os.path.join(golden_dir, "raflow_train_synth_b4.npz"))
    net = _net(golden_dir, dev, path, monkeypatch).train()
    step = TrainStep(net)
    assert step.self_supervised
    batch = {k: torch.from_numpy(g[k]).to(dev) for k in ("pc1", "pc2", "ft1", "ft2", "interval")}
    loss, items, outs, _ = step.forward_loss(batch)
    assert set(items) == {"Loss", "smoothnessLoss", "chamferLoss", "veloLoss"}
    step.bucket.zero()
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 2e-4
    for k, v in items.items():
        assert abs(v.item() - float(g["item_" + k])) < 2e-4, k
    np.testing.assert_allclose(outs[0].detach().cpu().numpy(), g["sf_agg"], rtol=0, atol=3e-4)
    params = dict(net.named_parameters())
    for name, ref in zip(g["grad_names"], g["grad_norms"]):
        got = float(params[str(name)].grad.norm())
        assert abs(got - ref) <= 1e-2 * max(ref, 1e-3), (name, got, ref)
    step.opt.step()
    for k in g.files:
        if k.startswith("after::"):
            np.testing.assert_allclose(params[k[7:]].detach().reshape(-1)[:64].cpu().numpy(), g[k], rtol=1e-4, atol=1e-5, err_msg=k)
