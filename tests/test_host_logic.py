"""CPU: host-side logic that needs no kernel launch -- module API / checkpoint layout of the product
modules, the gradient bucket, deterministic synthetic data."""
import json
import os

import torch

from cmflow_amd import synth


def _man(golden_dir, name):
    return json.load(open(os.path.join(golden_dir, "state_manifest_%s.json" % name)))


def test_product_state_dict_layout_matches_reference(golden_dir, args):
    """Row a17: same keys, shapes, dtypes and ORDER as models/cmflow.py / cmflow_t.py, so reference
    checkpoints load unchanged."""
    from cmflow_amd.cmflow import CMFlow, CMFlow_T
    for cls, name, n in ((CMFlow, "cmflow", 374), (CMFlow_T, "cmflow_t", 378)):
        sd = cls(args).state_dict()
        assert [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in sd.items()] == _man(golden_dir, name)
        assert len(sd) == n
    net = CMFlow(args)
    assert sum(p.numel() for p in net.parameters()) == 4230672
    missing, unexpected = net.load_state_dict(synth.synth_state_dict(_man(golden_dir, "cmflow")), strict=False)
    assert not missing and not unexpected


def test_forward_signature_follows_reference(args):
    import inspect
    from cmflow_amd.cmflow import CMFlow, CMFlow_T
    assert list(inspect.signature(CMFlow.forward).parameters) == ["self", "pc1", "pc2", "feature1", "feature2", "label_m", "mode"]
    assert list(inspect.signature(CMFlow_T.forward).parameters)[-1] == "gfeat"
    net = CMFlow(args)
    assert net.stat_thres == args.stat_thres and CMFlow_T(args).stat_thres == 0.5      # cmflow_t.py:18


def test_flat_grad_bucket_views_and_unused_params(args):
    from cmflow_amd.cmflow import CMFlow
    from cmflow_amd.dp import FlatGradBucket
    net = CMFlow(args)
    bucket = FlatGradBucket(net)
    unused = [n for n, p in net.named_parameters() if getattr(p, "_cmf_unused", False)]
    assert len(unused) == 12 and all("weightnet" in n and "mlp_bns" in n for n in unused)
    assert bucket.numel == 4230672 - 2 * 2 * (8 + 8 + 512)
    p0 = bucket.params[0]
    p0.grad.add_(1.0)
    assert float(bucket.flat[:p0.numel()].sum()) == p0.numel()          # grads are views into the bucket
    bucket.zero()
    assert float(bucket.flat.abs().sum()) == 0.0
    assert all(getattr(p, "grad", None) is None for n, p in net.named_parameters() if n in unused)


def test_synthetic_data_is_deterministic_and_vod_shaped():
    a, b = synth.make_batch(3, seed=5, train_extras=True), synth.make_batch(3, seed=5, train_extras=True)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    assert a["pc1"].shape == (3, 3, 256) and a["ft1"].shape == (3, 3, 256) and a["pc1"].is_contiguous()
    assert torch.equal(a["ft1"][:, 1], a["ft1"][:, 2])                  # [v_r, RCS, RCS] (dataset/vod.py:62)
    x = a["pc1"][:, 0]
    assert float(x.min()) >= 2.0 and float(x.max()) <= 90.0
    # mean occupancy of the query balls is in the range of real VoD clouds
    from oracle import ops
    xyz = a["pc1"].permute(0, 2, 1).contiguous()
    d = ops.square_distance(xyz, xyz)
    occ = [float((d < r * r).float().sum(-1).mean()) for r in (2.0, 4.0, 8.0, 16.0)]
    assert 4 < occ[0] < 11 and 10 < occ[1] < 25 and 28 < occ[2] < 60 and 60 < occ[3] < 120, occ


def test_label_prep_matches_reference_golden(golden_dir):
    """main_util.py:209-265 label prep (the torch-op form, device-agnostic) against the reference's output; the
    HIP form (make_labels -> cmf_pseudo_labels) is checked against the same golden in tests/test_gpu_eval.py."""
    import numpy as np
    from cmflow_amd.losses import make_labels_torch as make_labels
    with np.load(os.path.join(golden_dir, "cmflow_train_synth_b4.npz")) as z:
        g = {k: z[k] for k in z.files}
    batch = {k: torch.from_numpy(g[k]) for k in ("pc1", "ft1", "gt_trans", "flow_label", "fg_mask", "interval")}
    dyn, mseg = make_labels(batch, 0.3)
    assert np.array_equal(dyn.numpy(), g["dyn_mask"]) and np.array_equal(mseg.numpy(), g["mseg_gt"])
