"""CPU: host-side logic that needs no kernel launch -- module API / checkpoint layout of the product
modules, the gradient bucket, deterministic synthetic data."""
import json
import os

import numpy as np
import pytest
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
    from loss_torch import make_labels_torch as make_labels                 # test fixture (tests/loss_torch.py)
    with np.load(os.path.join(golden_dir, "cmflow_train_synth_b4.npz")) as z:
        g = {k: z[k] for k in z.files}
    batch = {k: torch.from_numpy(g[k]) for k in ("pc1", "ft1", "gt_trans", "flow_label", "fg_mask", "interval")}
    dyn, mseg = make_labels(batch, 0.3)
    assert np.array_equal(dyn.numpy(), g["dyn_mask"]) and np.array_equal(mseg.numpy(), g["mseg_gt"])


def test_vod_dataset_matches_reference_loader(golden_dir, tmp_path):
    """cmflow_amd.dataset.vodDataset against the tuples the reference's dataset/vod.py produced
    from the same sample files (tests/golden/make_golden_raflow.py), incl. the seeded resampling to 256 points."""
    import json
    import numpy as np
    from cmflow_amd import dataset as D
    g = np.load(os.path.join(golden_dir, "vod_dataset_kat.npz"))
    files = [k[6:] for k in g.files if k.startswith("file::")]
    assert len(files) == 6
    for rel in files:
        p = tmp_path / rel
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_text(str(g["file::" + rel]))

    class A:
        num_points = 256
        eval = False

    for part, ev in (("train", False), ("test", True)):
        a = A()
        a.eval = ev
        ds = D.vodDataset(a, root=str(tmp_path) + "/", partition=part)
        assert len(ds) == int(g["%s/len" % part])
        np.testing.assert_array_equal(np.array([ds.res['r_res'], ds.res['theta_res'], ds.res['phi_res']]), g["%s/res" % part])
        np.testing.assert_allclose(ds.camera_projection_matrix, g["%s/camera_projection_matrix" % part], rtol=1e-6)
        np.testing.assert_allclose(ds.t_camera_radar, g["%s/t_camera_radar" % part], rtol=1e-6)
        if ev:
            assert ds.clips_info == json.loads(str(g["%s/clips_info" % part]))
        np.random.seed(11)
        for i in range(len(ds)):
            item = ds[i]
            assert len(item) == 11
            for j, v in enumerate(item):
                ref = g["%s/%d/%d" % (part, i, j)]
                v = np.asarray(v)
                assert v.shape == ref.shape and v.dtype == ref.dtype, (part, i, j, v.dtype, ref.dtype)
                np.testing.assert_array_equal(v, ref)
    # collate + extract_data_info (main_util.py:21-36): model layout, on CPU here
    a = A()
    ds = D.vodDataset(a, root=str(tmp_path) + "/", partition="train")
    batch = next(iter(torch.utils.data.DataLoader(ds, batch_size=2, shuffle=False)))
    info = D.extract_data_info(batch, device="cpu")
    assert info[0].shape == (2, 3, 256) and info[2].shape == (2, 3, 256) and info[4].shape == (2, 4, 4)
    assert info[5].shape == (2, 256, 3) and info[10].shape == (2, 256, 2) and all(t.dtype == torch.float32 for t in info)
    assert set(D.as_batch_dict(info)) >= {"pc1", "ft1", "gt_trans", "flow_label", "fg_mask", "interval", "opt_flow"}


def test_vod_clip_dataset_matches_reference_loader(golden_dir, tmp_path):
    """cmflow_amd.dataset.vodClipDataset against what the reference's dataset/vod_clip.py produced from the same
    sample files (tests/golden/make_golden_clip.py): mini-clip tuples of the training partition under the same numpy seed (same RNG
    call order), per-frame tuples and clips_info of the evaluation partition, the file lists, the textio lines."""
    import json
    import numpy as np
    from cmflow_amd import dataset as D
    g = np.load(os.path.join(golden_dir, "vod_clip_kat.npz"))
    files = [k[6:] for k in g.files if k.startswith("file::")]
    assert len(files) == 13
    for rel in files:
        p = tmp_path / rel
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_text(str(g["file::" + rel]))

    class A:
        num_points = 96
        eval = False
        mini_clip_len = 2
        update_len = 1

    class Textio:
        def __init__(self):
            self.lines = []

        def cprint(self, s):
            self.lines.append(s)

    for part, ev in (("train", False), ("test", True)):
        a = A()
        a.eval = ev
        tio = Textio()
        ds = D.vodClipDataset(a, root=str(tmp_path) + "/", partition=part, textio=tio)
        assert len(ds) == int(g["%s/len" % part])
        assert tio.lines == json.loads(str(g["%s/textio" % part]))
        assert [os.path.relpath(q, str(tmp_path)) for q in ds.samples] == json.loads(str(g["%s/samples" % part]))
        assert ds.mini_clip_len == 2 and ds.update_len == 1
        if ev:
            assert ds.clips_info == json.loads(str(g["%s/clips_info" % part]))
        else:
            assert [[os.path.relpath(q, str(tmp_path)) for q in m] for m in ds.mini_samples] == json.loads(str(g["%s/mini_samples" % part]))
        np.random.seed(23)
        for i in range(len(ds)):
            item = ds[i]
            assert len(item) == 11
            for j, v in enumerate(item):
                ref = g["%s/%d/%d" % (part, i, j)]
                v = np.asarray(v)
                assert v.shape == ref.shape and v.dtype == ref.dtype, (part, i, j, v.shape, ref.shape, v.dtype, ref.dtype)
                np.testing.assert_array_equal(v, ref)
    # a collated mini-clip batch, frame by frame (clip_util.py:81-96), on CPU here
    ds = D.vodClipDataset(A(), root=str(tmp_path) + "/", partition="train")
    batch = next(iter(torch.utils.data.DataLoader(ds, batch_size=2, shuffle=False)))
    assert batch[0].shape == (2, 2, 96, 3) and batch[4].shape == (2, 2, 4, 4) and batch[7].shape == (2, 2)
    for j in range(ds.mini_clip_len):
        info = D.extract_data_info_clip(batch, j, device="cpu")
        assert info[0].shape == (2, 3, 96) and info[3].shape == (2, 3, 96) and info[4].shape == (2, 4, 4) and info[7].shape == (2,)
        assert info[5].shape == (2, 96, 3) and info[10].shape == (2, 96, 2) and all(t.dtype == torch.float32 for t in info)
        assert torch.equal(info[0], batch[0][:, j].transpose(2, 1))


def test_raflow_checkpoint_layout(golden_dir):
    """RaFlow mirror: the reference's 355 state tensors in the reference's order."""
    import json
    from cmflow_amd.raflow import RaFlow

    class A:
        num_points = 256
        rigid_thres = 0.15

    man = json.load(open(os.path.join(golden_dir, "state_manifest_raflow.json")))
    sd = RaFlow(A).state_dict()
    assert [k for k, _, _ in man] == list(sd.keys())
    assert all(list(sd[k].shape) == list(s) for k, s, _ in man)


def _pytorch_utils_case(golden_dir, name, device):
    """Build CASES[name] from cmflow_amd.pytorch_utils, load the reference-built module's state (strict: same keys), run it like
    make_golden_pytorch_utils.run; -> (golden npz, results dict, module)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden_pytorch_utils", os.path.join(golden_dir, "make_golden_pytorch_utils.py"))
    G = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(G)
    from cmflow_amd import pytorch_utils as U
    g = np.load(os.path.join(golden_dir, "pytorch_utils_kat.npz"))
    cls, a, kw, shape = G.CASES[name]
    m = getattr(U, cls)(*a, **kw)
    sd = {k[len(name) + 7:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(name + "/state/")}
    m.load_state_dict(sd, strict=True)
    m = m.to(device)
    x, w = torch.from_numpy(g[name + "/x"]).to(device), torch.from_numpy(g[name + "/w"]).to(device)
    res = {}
    m.eval()
    with torch.no_grad():
        res["eval"] = m(x.clone())
    m.train()
    xi = x.clone().requires_grad_(True)
    y = m(xi)
    (y * w).sum().backward()
    res["train"], res["dx"] = y.detach(), xi.grad
    for k, p in m.named_parameters():
        res["grad/" + k] = p.grad
    for k, b in m.named_buffers():
        res["buf/" + k] = b
    return g, res, m


PYTORCH_UTILS_CASES = ["mlp_default", "mlp_preact_first", "mlp_preact", "mlp_nobn", "mlp_odd", "conv1d_bn", "conv1d_preact", "conv1d_plain",
                       "conv2d_bias", "fc_bn", "fc_bias", "fc_preact", "fc_noact"]


@pytest.mark.parametrize("name", PYTORCH_UTILS_CASES)
def test_pytorch_utils_classes_match_reference_goldens_on_cpu(golden_dir, name):
    """lib/pytorch_utils.py:5-236 (SharedMLP incl. preact / first, Conv1d, Conv2d, FC, BatchNorm1d): same state_dict keys (strict load
    of the reference-built module's state), and on CPU tensors -- the plain torch modules of the Sequential -- the outputs, gradients
    and BN buffers of the reference's own classes (tests/golden/pytorch_utils_kat.npz, made by make_golden_pytorch_utils.py)."""
    g, res, _ = _pytorch_utils_case(golden_dir, name, "cpu")
    for k, v in res.items():
        want = g[name + "/" + k]
        np.testing.assert_allclose(v.detach().numpy(), want, rtol=1e-5, atol=1e-6 * max(1.0, float(np.abs(want).max())), err_msg=name + "/" + k)
