"""GPU: module-level parity -- the product CMFlow / CMFlow_T (HIP path) against (i) golden
vectors produced by the reference's own Python modules and (ii) the CPU oracle on fresh inputs.

Tolerances (north_star): neighbour indices bit-exact, flow / seg within 1e-4 fp32.
"""
import os

import numpy as np
import pytest
import torch

from cmflow_amd import synth
from oracle import cmflow_oracle as O
from oracle import train_oracle as TO

pytestmark = pytest.mark.gpu
EVAL_CASES = ["cmflow_eval_synth_b2", "cmflow_eval_synth_b1", "cmflow_eval_real_b4"]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False
    return torch.device("cuda:0")


def _load(golden_dir, name):
    with np.load(os.path.join(golden_dir, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


def _weights(manifest, golden_dir, t=False):
    return synth.synth_state_dict(manifest, seed=1234,
                                  calib=os.path.join(golden_dir, "bn_calib_cmflow_t.npz" if t else "bn_calib_cmflow.npz"))


def _build(path, args, sd, dev, monkeypatch):
    """pm: the product (fused point-major path, one C-ABI call per set-conv block); pm_py: same kernels sequenced from
    Python; pm_torch: same layout, dense math through torch (a test fixture: tests/pm_torch.py installs torch bodies on the
    block modules of this instance); ref: the reference's own op sequence in its (B,C,N,ns) layout -- the oracle's modules on
    the GPU with their native ops bound to the DROP-IN kernels (tests/hip_ops.py)."""
    from cmflow_amd import fused_blocks as FB
    from cmflow_amd.cmflow import CMFlow
    if path == "ref":
        import hip_ops
        return hip_ops.reference_layout_net(O.CMFlow, args, sd, dev, monkeypatch)
    monkeypatch.setattr(FB, "USE_BLOCK_CALLS", path != "pm_py")
    net = CMFlow(args)
    net.load_state_dict(sd)
    net = net.to(dev)
    if path == "pm_torch":
        import pm_torch
        pm_torch.install(net)
    return net


def _epe(a, b):
    return float(np.linalg.norm(a - b, axis=1).mean())


@pytest.mark.parametrize("path", ["pm", "pm_py", "pm_torch", "ref"])
@pytest.mark.parametrize("case", EVAL_CASES)
def test_forward_matches_reference_golden(case, path, dev, manifest, golden_dir, args, monkeypatch):
    from cmflow_amd import pointnet2_utils as pu, radarflow_util as ru
    g = _load(golden_dir, case)
    net = _build(path, args, _weights(manifest, golden_dir), dev, monkeypatch).eval()
    # record what crosses the op boundary, in call order
    bq, knn = [], []
    bq0, knn0 = pu.ball_query, ru.knn_point
    pu.ball_query = lambda *a: (bq.append(bq0(*a)) or bq[-1])
    ru.knn_point = lambda *a, **k: (knn.append(knn0(*a, **k)) or knn[-1])
    from cmflow_amd import fused_blocks as FB
    FB.IDX_TAP = tap = []   # "pm" issues its ball queries from C++ (csrc/setconv_block.hip): the debug tap reads them back from the blocks' arenas
    try:
        t = lambda k: torch.from_numpy(g[k]).to(dev)
        with torch.no_grad():
            sf, cls, trans, mask = net(t("pc1"), t("pc2"), t("ft1"), t("ft2"), None, "test")
    finally:
        pu.ball_query, ru.knn_point = bq0, knn0
        FB.IDX_TAP = None
    keys = sorted(k for k in g if k.startswith("bq"))
    if path == "pm":
        # the tap lists, per encoder call, the four scales of every cloud it handled: first encoder (both clouds in ONE call of 2B samples
        # in eval mode, or a dual-cloud call), then the second encoder -- reorder to the reference's call order
        # (pc1 x 4 scales, pc2 x 4 scales, embedding x 4 scales) and compare bit for bit
        nb = g["pc1"].shape[0]
        idx = [x for _, _, x in tap]
        assert len(idx) in (8, 12), len(idx)
        if len(idx) == 8:
            idx = [x[:nb] for x in idx[:4]] + [x[nb:] for x in idx[:4]] + idx[4:]
        for k, x in zip(keys, idx):
            assert np.array_equal(g[k], x.cpu().numpy()), k                 # a1 bit-exact, as issued by the library
    if path != "pm":
        if len(bq) == 8:    # eval mode on the fused path: both clouds go through the first encoder in ONE call (2B samples)
            nb = g["pc1"].shape[0]
            bq = [x[:nb] for x in bq[:4]] + [x[nb:] for x in bq[:4]] + bq[4:]
        assert len(bq) == 12
        for k, idx in zip(keys, bq):
            assert np.array_equal(g[k], idx.cpu().numpy()), k                 # a1 bit-exact
    from test_oracle import same_neighbours
    assert same_neighbours(g["pc2"], knn[0].cpu().numpy(), g["knn_cross_sorted"])   # a7 sets
    assert same_neighbours(g["pc1"], knn[1].cpu().numpy(), g["knn_self_sorted"])
    for k in ("pc1_features", "pc2_features", "cor_features", "prop_features"):
        # fp32 summation-order noise scales with the tensor's magnitude (cor_features reaches ~1e3)
        np.testing.assert_allclose(net.last[k][0, ::4].cpu().numpy(), g[k], rtol=1e-4,
                                   atol=1e-5 * max(1.0, float(np.abs(g[k]).max())), err_msg=k)
    assert np.array_equal(mask.cpu().numpy(), g["mask"])
    np.testing.assert_allclose(cls.cpu().numpy(), g["stat_cls"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(trans.cpu().numpy(), g["pre_trans"], rtol=2e-5, atol=1e-4)
    assert _epe(sf.cpu().numpy(), g["sf_agg"]) < 1e-4


def test_query_and_group_slice(dev, golden_dir):
    """Row a4 against the reference's own QueryAndGroup output (bit-exact: gather + subtract)."""
    from cmflow_amd.pointnet2_utils import QueryAndGroup
    g = _load(golden_dir, "cmflow_eval_synth_b2")
    xyz_t = torch.from_numpy(g["pc1"]).permute(0, 2, 1).contiguous().to(dev)
    out = QueryAndGroup(4.0, 8)(xyz_t, xyz_t, torch.from_numpy(g["ft1"]).to(dev))
    assert np.array_equal(out[0].cpu().numpy(), g["qg_scale1_b0"])


def test_cmflow_t_matches_reference_golden(dev, manifest_t, golden_dir, args):
    from cmflow_amd.cmflow import CMFlow_T
    g = _load(golden_dir, "cmflow_t_eval_synth_b2")
    net = CMFlow_T(args)
    net.load_state_dict(_weights(manifest_t, golden_dir, t=True))
    net = net.to(dev).eval()
    t = lambda k: torch.from_numpy(g[k]).to(dev)
    with torch.no_grad():
        o1 = net(t("a_pc1"), t("a_pc2"), t("a_ft1"), t("a_ft2"), None, "test", None)
        o2 = net(t("b_pc1"), t("b_pc2"), t("b_ft1"), t("b_ft2"), None, "test", o1[4])
    for tag, o in (("a", o1), ("b", o2)):
        assert np.array_equal(o[3].cpu().numpy(), g[tag + "_mask"])
        assert _epe(o[0].cpu().numpy(), g[tag + "_sf_agg"]) < 1e-4
        np.testing.assert_allclose(o[1].cpu().numpy(), g[tag + "_stat_cls"], rtol=0, atol=1e-4)
        np.testing.assert_allclose(o[2].cpu().numpy(), g[tag + "_pre_trans"], rtol=2e-5, atol=1e-4)
        np.testing.assert_allclose(o[4].cpu().numpy(), g[tag + "_gfeat"], rtol=0, atol=1e-4)


def test_forward_matches_oracle_fresh_inputs(dev, manifest, golden_dir, args):
    """Same seeded inputs through the HIP path and the CPU oracle (B=8, not in the goldens)."""
    from cmflow_amd.cmflow import CMFlow
    sd = _weights(manifest, golden_dir)
    ref = O.CMFlow(args)
    ref.load_state_dict(sd)
    ref.eval()
    net = CMFlow(args)
    net.load_state_dict(sd)
    net = net.to(dev).eval()
    b = synth.make_batch(8, seed=2024)
    with torch.no_grad():
        r = ref(b["pc1"], b["pc2"], b["ft1"], b["ft2"], None, "test")
        o = net(*(b[k].to(dev) for k in ("pc1", "pc2", "ft1", "ft2")), None, "test")
    assert torch.equal(o[3].cpu(), r[3])
    assert _epe(o[0].cpu().numpy(), r[0].numpy()) < 1e-4
    np.testing.assert_allclose(o[1].cpu().numpy(), r[1].numpy(), rtol=0, atol=1e-4)
    np.testing.assert_allclose(o[2].cpu().numpy(), r[2].numpy(), rtol=2e-5, atol=1e-4)   # t reaches ~10 m: fp32 relative


@pytest.mark.parametrize("path", ["pm", "pm_py", "pm_torch", "ref"])
@pytest.mark.parametrize("case", ["cmflow_train_synth_b4", "cmflow_train_evalbn_synth_b4"])
def test_train_step_matches_reference_golden(case, path, dev, manifest, golden_dir, args, monkeypatch):
    """Rows a3 + a15('train') + losses + Adam against the reference's own train step
    (main_util.py:63-76 run behind the shims, tests/golden/make_golden.py).  `evalbn`: the same step with the network
    in eval mode -- the regime of every CMFlow epoch after the first (train_one_epoch never calls net.train(),
    main_util.py:39-76,96): BN normalises with its running statistics, gradients flow, buffers must not move."""
    from cmflow_amd.train import TrainStep
    g = _load(golden_dir, case)
    net = _build(path, args, _weights(manifest, golden_dir), dev, monkeypatch)
    net.eval() if "evalbn" in case else net.train()
    before = {k: v.clone() for k, v in net.state_dict().items() if "running_" in k or "num_batches" in k}
    batch = {k: torch.from_numpy(g[k]).to(dev) for k in ("pc1", "pc2", "ft1", "ft2", "gt_trans", "flow_label", "fg_mask",
                                                            "interval", "radar_u", "radar_v", "opt_flow")}
    step = TrainStep(net, vr_thres=args.vr_thres)
    loss, items, outs, (dyn, mseg) = step.forward_loss(batch)
    step.bucket.zero()
    loss.backward()
    assert np.array_equal(dyn.cpu().numpy(), g["dyn_mask"]) and np.array_equal(mseg.cpu().numpy(), g["mseg_gt"])
    assert abs(loss.item() - float(g["loss"])) < 2e-4 * max(1.0, abs(float(g["loss"])))
    for k, v in items.items():
        assert abs(v.item() - float(g["item_" + k])) < 2e-4 * max(1.0, abs(float(g["item_" + k]))), k
    assert _epe(outs[0].detach().cpu().numpy(), g["sf_agg"]) < 1e-4
    np.testing.assert_allclose(outs[2].detach().cpu().numpy(), g["pre_trans"], rtol=2e-5, atol=1e-4)
    params = dict(net.named_parameters())
    for name, ref in zip(g["grad_names"], g["grad_norms"]):
        p = params[str(name)]
        if ref < 0:
            assert p.grad is None, name
        else:   # fp32 gradient noise floor of this net is ~1e-3 relative
            assert abs(float(p.grad.norm()) - ref) <= 1e-2 * max(ref, 1e-3), (name, float(p.grad.norm()), ref)
    for k in g:
        if k.startswith("grad::"):
            got = params[k[6:]].grad.reshape(-1)[:64].cpu().numpy()
            np.testing.assert_allclose(got, g[k], rtol=1e-2, atol=1e-2 * np.abs(g[k]).max(), err_msg=k)
    step.opt.step()
    sd = net.state_dict()
    for k in g:
        if k.startswith("after::"):
            np.testing.assert_allclose(sd[k[7:]].reshape(-1)[:64].cpu().numpy(), g[k], rtol=1e-3, atol=1e-4, err_msg=k)
    if "evalbn" in case:
        assert all(torch.equal(sd[k], v) for k, v in before.items())            # eval-mode BN never moves its buffers


def test_full_size_properties(dev, manifest, golden_dir, args):
    """BASELINE.json sizes (B=64, N=256): size-independent properties instead of the (slow) oracle:
    per-sample independence (a batch of 64 equals 64/8 batches of 8 in eval mode), rigid part of the
    output is a proper rotation, indices within range."""
    from cmflow_amd.cmflow import CMFlow
    net = CMFlow(args)
    net.load_state_dict(_weights(manifest, golden_dir))
    net = net.to(dev).eval()
    b = {k: v.to(dev) for k, v in synth.make_batch(64, seed=77).items()}
    with torch.no_grad():
        full = net(b["pc1"], b["pc2"], b["ft1"], b["ft2"], None, "test")
        parts = [net(*(b[k][i:i + 8] for k in ("pc1", "pc2", "ft1", "ft2")), None, "test") for i in range(0, 64, 8)]
    for j in range(4):
        cat = torch.cat([p[j] for p in parts], dim=0)
        if cat.dtype == torch.bool:
            assert torch.equal(cat, full[j])
        else:
            np.testing.assert_allclose(cat.cpu().numpy(), full[j].cpu().numpy(), rtol=0, atol=1e-4)
    R = full[2][:, :3, :3].double()
    eye = torch.eye(3, dtype=torch.float64, device=dev).expand(64, 3, 3)
    assert torch.allclose(R @ R.transpose(1, 2), eye, atol=1e-5)
    assert torch.all(torch.abs(torch.linalg.det(R).abs() - 1) < 1e-5)


def _bench_setup(dev, train):
    """Exactly bench.py's rank-0 workload: its weights (load_weights), its batch (seed 1234, B=64, N=256), its Args."""
    import bench
    from cmflow_amd.cmflow import CMFlow
    torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
    sd = bench.load_weights("cmflow")
    ref = O.CMFlow(bench.Args())
    ref.load_state_dict(sd)
    net = CMFlow(bench.Args())
    net.load_state_dict(sd)
    net = net.to(dev)
    (ref.train(), net.train()) if train else (ref.eval(), net.eval())
    b = synth.make_batch(64, seed=1234, train_extras=True)
    return ref, net, b, {k: v.to(dev) for k, v in b.items()}


def _rot_angle(Ra, Rb):
    """Angle (rad) of Ra Rb^T, fp64, from the skew part (accurate for tiny angles)."""
    D = Ra.double() @ Rb.double().transpose(1, 2)
    skew = 0.5 * (D - D.transpose(1, 2))
    return torch.sqrt(skew[:, 2, 1] ** 2 + skew[:, 0, 2] ** 2 + skew[:, 1, 0] ** 2)


def _check_transform(got, want, pc1, weight):
    """SURVEY 8d asks pre_trans <= 1e-4.  R meets an absolute bound (rotation error <= 2e-6 rad, entries <= 1e-5).  t does
    NOT have an absolute fp32 bound at this size: t = cB - R cA (models/cmflow.py:163) with |cA| ~ 50 m, so a rotation
    difference dR between two correct fp32 evaluations moves t by up to |dR| |cA| (3e-6 rad * 50 m = 1.5e-4).  The
    bound that holds for every correct implementation is the conditioned one:
        |dt| <= 1e-4 + |dR|_2 * |cA|        
    weight: the normalised Kabsch weights (B,N) the reference uses for the centroid (:137-139)."""
    got, want = got.double().cpu(), want.double()
    ang = _rot_angle(got[:, :3, :3], want[:, :3, :3])
    assert float(ang.max()) <= 2e-6, float(ang.max())
    assert float((got[:, :3, :3] - want[:, :3, :3]).abs().max()) <= 1e-5
    cA = (pc1.double() * weight.double().unsqueeze(1)).sum(dim=2)                     # (B,3)
    dR = torch.linalg.matrix_norm(got[:, :3, :3] - want[:, :3, :3], ord=2)
    dt = (got[:, :3, 3] - want[:, :3, 3]).norm(dim=1)
    bound = 1e-4 + dR * cA.norm(dim=1)
    assert bool((dt <= bound).all()), (float(dt.max()), float(bound.min()))
    assert torch.equal(got[:, 3], want[:, 3])                                          # bottom row 0 0 0 1 exactly
    return float(ang.max()), float(dt.max())


# (relative error, 1 - cos) of the concatenated gradient.  Measured (round 6, against the oracle): CMFlow B = 64 8.3e-4 / 3.5e-7, the
# N = 4096 step 2.0e-3 / 1.9e-6, the CMFlow-T clip's later frames up to 3.7e-3 / 6.7e-6 -- bounds 1.6-3x above the worst of them
WHOLE_GRADIENT_BOUND = (6e-3, 2e-5)


def _check_gradients(net, gref, what, bounds=(1e-2, 2e-4, 6e-2), loose=None):
    """Every parameter gradient against the oracle's, three ways: the norm, the DIRECTION (1 - cosine: a permutation, a
    swapped column block or a missing term inside a tensor keeps the norm and moves this to 1e-2 ... 1) and the largest
    single element relative to the tensor's largest entry.  bounds = (norm, 1 - cos, element).  They are set from
    measurements, not wishes (tests/grad_noise_floor.py: the ORACLE's own fp32 evaluation against its fp64 evaluation at
    B = 64, 8 threads): CMFlow norm 5.9e-4 / 1 - cos 3.6e-6 / element 1.3e-2; CMFlow-T first frame 2.6e-3 / 2.5e-5 / 2.2e-2,
    second frame (weights after one Adam step: the motion head's BCE saturates) 1.2e-2 / 9.5e-4 / 0.115.  The worst
    tensors are column sums over 524288 rows with heavy cancellation (BN shifts of the widest scale), whose fp32 value
    depends on the summation order -- i.e. on the oracle's thread count: on the 32-thread GPU hosts the same GPU gradient
    (bit-identical statistics from two different epilogue implementations) sits 4.0e-5 from the oracle in direction where
    the 8-thread run had < 1e-5.  Defaults: 2-3x above those floors, still 100x below what a structural error produces.
    -> (count, worst norm error, worst 1 - cos, worst element error), each with the parameter's name."""
    wn, wc, we, n = ("", 0.0), ("", 0.0), ("", 0.0), 0
    over = []
    for k, p in net.named_parameters():
        if gref[k] is None:
            assert p.grad is None, k
            continue
        a, r = p.grad.detach().double().cpu().reshape(-1), gref[k].double().reshape(-1)
        na, nr = float(a.norm()), float(r.norm())
        en = abs(na - nr) / max(nr, 1e-3)
        ec = 1.0 - float(a @ r) / (na * nr) if nr > 1e-6 else 0.0          # direction of a numerically-zero gradient is undefined
        ee = float((a - r).abs().max()) / max(float(r.abs().max()), 1e-6)
        # `loose`: (name prefixes, bounds) for the tensors that are allowed more (named, with the reason, at the call site);
        # every other tensor is held to `bounds`
        bd = loose[1] if (loose is not None and k.startswith(tuple(loose[0]))) else bounds
        if en > bd[0] or ec > bd[1] or ee > bd[2]:
            over.append((k, en, ec, ee))
        wn = max(wn, (k, en), key=lambda t: t[1]); wc = max(wc, (k, ec), key=lambda t: t[1]); we = max(we, (k, ee), key=lambda t: t[1])
        n += 1
    assert not over, (what, "tensors over their bounds (name, norm, 1 - cos, element)", over)
    # ... and the gradient as ONE vector (all tensors concatenated): the per-tensor bounds above are set by the few cancellation-prone
    # tensors; the whole gradient -- what the optimizer step sees -- agrees far better
    ga = torch.cat([p.grad.detach().double().cpu().reshape(-1) for k, p in net.named_parameters() if gref[k] is not None])
    gr = torch.cat([gref[k].double().reshape(-1) for k, p in net.named_parameters() if gref[k] is not None])
    whole = float((ga - gr).norm() / gr.norm())
    wcos = 1.0 - float(ga @ gr) / float(ga.norm() * gr.norm())
    print("%s: whole gradient relative error %.3g, 1 - cos %.3g" % (what, whole, wcos))
    assert whole <= WHOLE_GRADIENT_BOUND[0] and wcos <= WHOLE_GRADIENT_BOUND[1], (what, whole, wcos)
    return n, wn, wc, we


def test_full_size_forward_matches_oracle(dev):
    """BASELINE config 2 at its own size -- bench.py's batch (B=64, N=256, seed 1234) and weights, eval mode -- HIP
    path vs the CPU oracle, with the bound of every quantity of SURVEY 8d written out:
      ball-query idx (4 radii x 2 clouds) and kNN idx/dist: bit-exact;  EPE mean <= 1e-4;  stat_cls max <= 1e-4;
      mask equal wherever the oracle's score is further than 1e-4 (the stat_cls bound) from the threshold;
      rotation <= 2e-6 rad;  translation in the conditioned form of _check_transform."""
    from cmflow_amd import pointnet2_utils as pu, radarflow_util as ru
    from oracle import ops
    ref, net, b, bd = _bench_setup(dev, train=False)
    for c in ("pc1", "pc2"):
        xyz = b[c].transpose(1, 2).contiguous()
        for r, ns in ((2.0, 4), (4.0, 8), (8.0, 16), (16.0, 32)):
            got = pu.ball_query(r, ns, xyz.to(dev), xyz.to(dev)).cpu()
            assert torch.equal(got, ops.ball_query(r, ns, xyz, xyz)), (c, r)
    x1, x2 = b["pc1"].transpose(1, 2).contiguous(), b["pc2"].transpose(1, 2).contiguous()
    for db, q in ((x2, x1), (x1, x1)):
        gi, gd = ru.knn_point(8, db.to(dev), q.to(dev), return_dist=True)
        wi, wd = ops.knn(8, db, q, return_dist=True)
        assert torch.equal(gi.cpu().int(), wi) and torch.equal(gd.cpu(), wd)
    with torch.no_grad():
        want = ref(b["pc1"], b["pc2"], b["ft1"], b["ft2"], None, "test")
        got = net(bd["pc1"], bd["pc2"], bd["ft1"], bd["ft2"], None, "test")
    sf, cls, trans, mask = (t.cpu() for t in got)
    assert float((cls - want[1]).abs().max()) <= 1e-4
    flips = mask != want[3]
    near = (want[1].squeeze(1) - 0.5).abs() <= 1e-4
    assert not bool((flips & ~near).any()), int(flips.sum())
    epe = (sf - want[0]).norm(dim=1)
    assert float(epe[~flips].mean()) <= 1e-4
    score = want[1].squeeze(1) + 1e-4                                                   # models/cmflow.py:105-107
    ang, dt = _check_transform(trans, want[2], b["pc1"], score / score.sum(dim=1, keepdim=True))
    # the per-point bound follows the transform's: a static point's flow is (T - I) p, |p| up to ~95 m
    assert float(epe[~flips].max()) <= 1e-4 + 2e-6 * 100.0 + dt
    print("full-size fwd: EPE mean %.3g max %.3g, stat_cls %.3g, flips %d, rot %.3g rad, dt %.3g m"
          % (float(epe.mean()), float(epe.max()), float((cls - want[1]).abs().max()), int(flips.sum()), ang, dt))


def test_inference_forward_equals_eval_forward_with_autograd(dev):
    """Under torch.no_grad() the second encoder's blocks skip the grouped first-layer tensor (cmf_gemm_gather_affine: the layer is
    formed in the A-operand path of the next GEMM) and the first encoder's blocks run layers 1-3 + the max over the ball as ONE
    register-chain kernel (csrc/setconv_chain.hip: no activation is written between the layers); with autograd recording (eval-mode
    BN, a backward pass may follow) both materialise their layers with the per-layer kernels.  Same operations in the same order:
    every output of the model must be equal bit for bit (B = 64, bench.py's batch)."""
    ref, net, b, bd = _bench_setup(dev, train=False)
    with torch.no_grad():
        a = net(bd["pc1"], bd["pc2"], bd["ft1"], bd["ft2"], None, "test")
    g = net(bd["pc1"], bd["pc2"], bd["ft1"], bd["ft2"], None, "test")
    assert g[0].requires_grad and not a[0].requires_grad
    for x, y in zip(a, g):
        assert torch.equal(x, y.detach())


def test_training_without_the_grouped_first_layer_tensor_is_bit_identical(dev, tmp_path):
    """Default (CMF_TRAIN_GATHER=0 switches it off): the second encoder's grouped first-layer tensor is never written in training either -- its statistics come
    from cmf_group_affine's statistics-only form and the three GEMMs that read it form it from the per-point rows.  Every one of
    them is bit-identical to its materialised counterpart in the non-persistent kernel, so against CMF_GEMM_PERSIST=0 the loss,
    every gradient and every BN buffer of a whole training step (B = 64) must be equal bit for bit.  (Child processes: the
    library reads its switches once.)"""
    import subprocess, sys
    for bn_mode in ("train", "eval"):                   # eval-mode BatchNorm with gradients: the same path with folded running statistics
        outs = []
        # (CMF_GEMM_TALL=0: both runs on the 128 x 128 tiles -- the gathering forward GEMM has no 256-row form, and a statistic's partial
        #  sums are bit-equal only between kernels of one tile shape)
        for i, env in enumerate((dict(CMF_GEMM_PERSIST="0", CMF_GEMM_TALL="0", CMF_TRAIN_GATHER="0"),
                                 dict(CMF_GEMM_PERSIST="0", CMF_GEMM_TALL="0", CMF_TRAIN_GATHER="1", CMF_TRAIN_GATHER_SUM="0"))):
            f = str(tmp_path / ("step_%s%d.pt" % (bn_mode, i)))
            r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "train_step_dump.py"), f, "64", bn_mode],
                               env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
            assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
            outs.append(torch.load(f))
        a, b = outs
        assert a.keys() == b.keys() and len(a) > 300
        bad = [k for k in a if not torch.equal(a[k], b[k])]
        assert not bad, (bn_mode, bad[:10])


def test_batched_first_encoder_bodies_are_bit_identical_to_the_per_block_chains(dev, tmp_path):
    """Default: the slot-level bodies of the first encoder's eight blocks (2 clouds x 4 scales) run in lock step as batched launches --
    one per stage for all blocks -- instead of one chain of kernels per block on the stream pool (CMF_BODY_BATCH=0).  The same kernels'
    device code on the same data: loss, every gradient and every BN buffer of a whole training step (B = 64) equal bit for bit."""
    import subprocess, sys
    outs = []
    # (third run: column blocks of the cost volume's first conv accumulate straight into the columns of its gradient buffer instead of
    #  going through autograd's zero-padded copies and adds, CMF_COL_SINKS -- disjoint columns, one contribution each: the same bits)
    for i, env in enumerate((dict(CMF_BODY_BATCH="0"), dict(CMF_BODY_BATCH="1"), dict(CMF_BODY_BATCH="1", CMF_COL_SINKS="0"))):
        f = str(tmp_path / ("body%d.pt" % i))
        r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "train_step_dump.py"), f, "64"],
                           env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        outs.append(torch.load(f))
    a = outs[0]
    for b in outs[1:]:
        assert a.keys() == b.keys() and len(a) > 300
        bad = [k for k in a if not torch.equal(a[k], b[k])]
        assert not bad, bad[:10]


def test_chain_training_matches_the_per_layer_kernels(dev, tmp_path):
    """CMF_CHAIN_TRAIN=1 (opt-in): the first encoder's blocks train through the register chain (csrc/setconv_chain.hip: slot-level
    activations never stored, every pass recomputes them; statistics rows per wave instead of per 128 rows).  Same terms as the
    per-layer kernels in another association -- but already in the FORWARD pass (the batch statistics are summed in another order, so
    every activation moves in its last bits and a few ReLU / arg-max decisions flip): the loss agrees to 1e-6 relative and the BN
    buffers to 1e-5, the gradients only to fp32's own noise floor at this size (tests/grad_noise_floor.py: 6e-4 of the norm for the
    whole model, up to 1e-2 for cancellation-prone BN biases) -- the bound of test_full_size_train_step_matches_oracle."""
    import subprocess, sys
    outs = []
    for i, env in enumerate((dict(CMF_CHAIN_TRAIN="0"), dict(CMF_CHAIN_TRAIN="1"))):
        f = str(tmp_path / ("chain%d.pt" % i))
        r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "train_step_dump.py"), f, "64"],
                           env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        outs.append(torch.load(f))
    a, b = outs
    assert a.keys() == b.keys()
    assert abs(float(a["loss"]) - float(b["loss"])) <= 1e-6 * abs(float(a["loss"]))
    worst = 0.0
    for k in a:
        if k.startswith("b.") and a[k].is_floating_point():
            assert torch.allclose(a[k], b[k], rtol=1e-5, atol=1e-6), k
        elif k.startswith("g."):
            rel = float((a[k].double() - b[k].double()).norm() / (a[k].double().norm() + 1e-30))
            worst = max(worst, rel)
            assert rel <= 2e-2, (k, rel)
    ga = torch.cat([a[k].flatten().double() for k in a if k.startswith("g.")]); gb = torch.cat([b[k].flatten().double() for k in a if k.startswith("g.")])
    whole = float((ga - gb).norm() / ga.norm())
    assert whole <= 2e-3, whole
    print("chain training: worst per-tensor relative gradient difference %.3g, whole gradient %.3g" % (worst, whole))


def test_summed_data_gradient_changes_nothing_but_the_association(dev, tmp_path):
    """Default (CMF_TRAIN_GATHER_SUM=0 switches it off): the data gradient into the second encoder's first layer is not stored either --
    the GEMM reduces it over runs of equal source points (cmf_gemm_dx_gather_sum).  Same terms, another association: against the
    stored form the loss and everything computed in the forward pass are bit-identical, every gradient within 2e-5 of its norm
    (the oracle comparison at this size, test_full_size_train_step_matches_oracle, runs the default)."""
    import subprocess, sys
    outs = []
    for i, env in enumerate((dict(CMF_TRAIN_GATHER_SUM="0"), dict(CMF_TRAIN_GATHER_SUM="1"))):
        f = str(tmp_path / ("step%d.pt" % i))
        r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "train_step_dump.py"), f, "64"],
                           env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        outs.append(torch.load(f))
    a, b = outs
    assert a.keys() == b.keys() and torch.equal(a["loss"], b["loss"])
    worst = 0.0
    for k in a:
        if k.startswith("b."):
            assert torch.equal(a[k], b[k]), k                          # BN buffers: forward only
        elif k.startswith("g."):
            rel = float((a[k].double() - b[k].double()).norm() / (a[k].double().norm() + 1e-30))
            worst = max(worst, rel)
            assert rel <= 2e-5, (k, rel)
    print("summed data gradient: worst relative gradient difference %.3g" % worst)


# Every run-time A/B switch the product keeps (README "switches") is forced here once -- or the path behind it goes.
# A training step (B = 4, bench.py's weights) under the switch against the default build of the step: `None` = every gradient, the loss
# and every BN buffer bit-identical (the same kernels' arithmetic, issued or scheduled differently); a number = the documented bound for
# "same terms, another association" (relative to each tensor's norm), the forward pass (loss, BN buffers) still bit-identical unless noted.
_AB_SWITCHES = [
    (dict(CMF_TAIL_BATCH="0"), None),                   # per-block tails instead of batched launches
    (dict(CMF_NESTED_QUERIES="0"), None),               # every block issues its own ball query
    (dict(CMF_BALL_QUERY_BALLOT="0"), None),            # scan kernels instead of the ballot kernel: same indices
    (dict(CMF_SIDE_STREAMS="2"), None),                 # another deal of the chains onto the stream pool
    (dict(CMF_GEMM_NO_DIRECT="1"), None),               # register-staged main loop everywhere: same MFMA sequence per element
    (dict(CMF_GEMM_WIDE="0"), None),                    # 128 x 128 tiles for the gathering forward GEMM
    (dict(CMF_THIN_GENERAL="1"), None),                 # the narrow forward layers' general body on full tiles
    (dict(CMF_FIN_WIDE="0"), None),                     # 4-column fold kernels for every partial matrix
    (dict(CMF_FIN_WIDE="1"), None),                     # 16-column fold kernels for every partial matrix
    (dict(CMF_THIN_FUSED="0"), 2e-5),                   # narrow backward layers as three kernels: other split-K slabs
    (dict(CMF_THIN_WIDE="0"), 2e-5),                    # 64 <- 256 backward layer as max-pool backward + BN backward + two tiled GEMMs
    # BN backward inside the NON-gathering weight-gradient GEMM's staging (round 3): only reachable with the materialised first layer
    (dict(CMF_TRAIN_GATHER="0", CMF_BNB_FUSED="1"), 5e-5),
    (dict(CMF_BNB_GATHER="0"), None),                   # stand-alone BN-backward pass in front of the gathering weight-gradient GEMM
    (dict(CMF_STREAM_PROBE="0"), None),                 # side-stream pool as the streams come, not picked by the queue probe
]


def _dump_step(path, env, mode="train"):
    import subprocess, sys
    args = [sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "train_step_dump.py"), path, "4"] + (["eval"] if mode == "eval" else [])
    r = subprocess.run(args, env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return torch.load(path)


@pytest.fixture(scope="module")
def default_step_b4(tmp_path_factory):
    d = tmp_path_factory.mktemp("ab")
    return {"train": _dump_step(str(d / "default.pt"), {}), "eval": _dump_step(str(d / "default_eval.pt"), {}, "eval")}


def _compare_steps(a, b, env, bound, fwd_exact=True):
    assert a.keys() == b.keys() and len(a) > 300
    differ = [k for k in a if not torch.equal(a[k], b[k])]
    worst = max([float((a[k].double() - b[k].double()).norm() / (a[k].double().norm() + 1e-30)) for k in differ if a[k].is_floating_point()] or [0.0])
    print("%s: %d of %d tensors differ, worst relative difference %.3g" % (env, len(differ), len(a), worst))
    if bound is None:
        assert not differ, (env, differ[:10])
    else:
        if fwd_exact:                                                 # the forward pass does not go through the switched kernels
            fwd = [k for k in differ if k.startswith("b.") or k == "loss"]
            assert not fwd, (env, fwd[:10])
        assert worst <= bound, (env, worst)


@pytest.mark.parametrize("env,bound", _AB_SWITCHES, ids=[",".join("%s=%s" % kv for kv in e.items()) for e, _ in _AB_SWITCHES])
def test_ab_switch_leaves_the_training_step_unchanged(dev, tmp_path, default_step_b4, env, bound):
    _compare_steps(default_step_b4["train"], _dump_step(str(tmp_path / "switched.pt"), env), env, bound)


def test_ab_switch_chain_off_in_eval_mode_bn_training(dev, tmp_path, default_step_b4):
    """CMF_CHAIN=0: the first encoder's blocks through the per-layer kernels where the default takes the register chain -- training with
    eval-mode BatchNorm (the reference's regime after its first epoch; the default there).  Every layer of the chain is bit-identical to
    the per-layer kernels in the forward pass; the backward passes sum their weight-gradient slabs and statistics rows per wave instead
    of per 128 rows: same terms, another association."""
    env = dict(CMF_CHAIN="0")
    _compare_steps(default_step_b4["eval"], _dump_step(str(tmp_path / "switched.pt"), env, "eval"), env, 5e-5, fwd_exact=False)


def test_full_size_train_step_matches_oracle(dev):
    """BASELINE config 3 (the headline) at its own size: one training step of bench.py's batch and weights, train-mode
    BN, 7 losses -- loss within 2e-4, every loss item within 2e-4, labels bit-equal, outputs as in the forward test,
    EVERY parameter's gradient norm within 1e-2 relative (the fp32 gradient noise floor of this net is ~1e-3, DESIGN.md
    section 4), never-used parameters without gradient, and all BN running statistics after the step."""
    from cmflow_amd.train import TrainStep
    ref, net, b, bd = _bench_setup(dev, train=True)
    P, Tcr = torch.tensor(synth.CAMERA_PROJECTION), torch.tensor(synth.T_CAMERA_RADAR)

    class NoStep:                                                   # keep the oracle's pre-step weights and its .grad
        def zero_grad(self): ref.zero_grad()
        def step(self): pass
    loss_ref, items_ref, out_ref, (dyn_ref, mseg_ref) = TO.train_step(ref, NoStep(), b, P, Tcr)
    step = TrainStep(net, vr_thres=0.3)
    loss, items, outs, (dyn, mseg) = step.forward_loss(bd)
    step.bucket.zero()
    loss.backward()
    assert torch.equal(dyn.cpu(), dyn_ref) and torch.equal(mseg.cpu(), mseg_ref)
    assert abs(loss.item() - loss_ref.item()) <= 2e-4 * max(1.0, abs(loss_ref.item())), (loss.item(), loss_ref.item())
    for k, v in items.items():
        assert abs(v.item() - items_ref[k]) <= 2e-4 * max(1.0, abs(items_ref[k])), (k, v.item(), items_ref[k])
    assert torch.equal(outs[3].cpu(), out_ref[3])                   # train mode: the mask comes from the labels
    assert float((outs[1].detach().cpu() - out_ref[1].detach()).abs().max()) <= 1e-4
    score = mseg_ref + 1e-4                                         # train mode: the labels are the Kabsch scores (:181-185)
    ang, dt = _check_transform(outs[2].detach(), out_ref[2].detach(), b["pc1"], score / score.sum(dim=1, keepdim=True))
    epe = (outs[0].detach().cpu() - out_ref[0].detach()).norm(dim=1)
    assert float(epe.mean()) <= 1e-4 and float(epe.max()) <= 1e-4 + 2e-6 * 100.0 + dt
    from cmflow_amd.fused_blocks import join_side_streams
    join_side_streams()
    gref = {k: p.grad for k, p in ref.named_parameters()}
    n, worst, wcos, welem = _check_gradients(net, gref, "B=64 CMFlow")
    assert n >= 180                                     # 182 parameter tensors receive a gradient
    want, have = ref.state_dict(), net.state_dict()
    for k, v in want.items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            np.testing.assert_allclose(have[k].cpu().numpy(), v.numpy(), rtol=2e-4, atol=2e-5 * float(v.abs().max()) + 1e-7, err_msg=k)
    print("full-size train: loss %.6f vs %.6f, worst grad-norm rel err %.3g (%s), worst 1-cos %.3g (%s), worst element %.3g (%s), "
          "rot %.3g rad, dt %.3g m" % (loss.item(), loss_ref.item(), worst[1], worst[0], wcos[1], wcos[0], welem[1], welem[0], ang, dt))


def test_full_size_cmflow_t_clip_matches_oracle(dev):
    """BASELINE config 4's per-rank workload at its own size: bench.py --model cmflow_t's clip -- B=64, five frames
    (seeds 1234 + 1000 f), weights of load_weights('cmflow_t') -- trained the way clip_util.py:34-62 does: gfeat = None at
    the first frame, gfeat.detach() handed to the next one, one optimizer step per frame.  HIP path vs the CPU oracle,
    per frame: labels bit-equal, loss and every loss item within 2e-4, flow EPE / stat_cls / transform as in the CMFlow
    test, the new GRU state within 1e-4, every parameter gradient by norm, direction and largest element
    (_check_gradients), BN running statistics.  Adam's first steps move a weight by ~lr*sign(g), so elements whose
    gradient is at rounding level go different ways in two correct implementations: the weights (and the carried state)
    are re-synchronised from the oracle before each frame, as in the B=4 test."""
    import bench
    from cmflow_amd.cmflow import CMFlow_T
    from cmflow_amd.fused_blocks import join_side_streams
    from cmflow_amd.train import TrainStep
    torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
    sd = bench.load_weights("cmflow_t")
    ref = O.CMFlow_T(bench.Args())
    ref.load_state_dict(sd)
    ref.train()
    net = CMFlow_T(bench.Args())
    net.load_state_dict(sd)
    net = net.to(dev).train()
    step = TrainStep(net, vr_thres=bench.Args.vr_thres)
    opt = torch.optim.Adam(ref.parameters(), lr=0.001, weight_decay=1e-4)
    P, Tcr = torch.tensor(synth.CAMERA_PROJECTION), torch.tensor(synth.T_CAMERA_RADAR)
    g_prev = None
    for f in range(5):
        net.load_state_dict(ref.state_dict())
        b = synth.make_batch(64, seed=1234 + 1000 * f, train_extras=True)
        dyn_ref, mseg_ref = TO.make_labels(b)
        out = ref(b["pc1"], b["pc2"], b["ft1"], b["ft2"], mseg_ref, "train", g_prev)
        loss_ref, items_ref = TO.radar_flow_loss(b, out[0], out[2], out[1], mseg_ref, dyn_ref, P, Tcr)
        opt.zero_grad()
        loss_ref.backward()
        if f == 0:
            step.reset_clip()
        else:
            step.gfeat = g_prev.to(dev)                               # the same carried state on both sides
        loss, items, outs, (dyn, mseg) = step.forward_loss({k: v.to(dev) for k, v in b.items()})
        step.bucket.zero()
        loss.backward()
        join_side_streams()
        assert torch.equal(dyn.cpu(), dyn_ref) and torch.equal(mseg.cpu(), mseg_ref), f
        assert abs(loss.item() - loss_ref.item()) <= 2e-4 * max(1.0, abs(loss_ref.item())), (f, loss.item(), loss_ref.item())
        for k, v in items.items():
            assert abs(v.item() - items_ref[k]) <= 2e-4 * max(1.0, abs(items_ref[k])), (f, k, v.item(), items_ref[k])
        assert torch.equal(outs[3].cpu(), out[3])
        assert float((outs[1].detach().cpu() - out[1].detach()).abs().max()) <= 1e-4
        score = mseg_ref                                              # cmflow_t.py:119: no 1e-4 on the scores
        ang, dt = _check_transform(outs[2].detach(), out[2].detach(), b["pc1"], score / score.sum(dim=1, keepdim=True))
        epe = (outs[0].detach().cpu() - out[0].detach()).norm(dim=1)
        assert float(epe.mean()) <= 1e-4 and float(epe.max()) <= 1e-4 + 2e-6 * 100.0 + dt, (f, float(epe.mean()), float(epe.max()))
        np.testing.assert_allclose(step.gfeat.detach().cpu().numpy(), out[4].detach().numpy(), rtol=0, atol=1e-4)
        gref = {k: p.grad for k, p in ref.named_parameters()}
        # later frames: weights after Adam steps, saturating motion-head BCE -- the oracle's own fp32 floor is 10-40x the first frame's
        # Frame 0 is held to the defaults.  Later frames (weights after Adam steps): every tensor to the ORACLE's own fp32-vs-fp64
        # floor of a second frame (tests/grad_noise_floor.py: 1.2e-2 / 9.5e-4 / 0.115) -- measured here 3.7e-3 / 1.2e-4 / 5.2e-2 --
        # except the motion head ("mp.": its BCE saturates after a step, single elements of its small tensors are pure
        # cancellation), which alone keeps the wide bounds.
        n, worst, wcos, welem = _check_gradients(net, gref, "CMFlow-T frame %d" % f, (1e-2, 2e-4, 6e-2) if f == 0 else (1.5e-2, 1e-3, 0.12),
                                                 loose=None if f == 0 else (("mp.",), (4e-2, 5e-3, 0.4)))
        assert n >= 184                                               # 182 + the four GRU tensors
        want, have = ref.state_dict(), net.state_dict()
        for k, v in want.items():
            if k.endswith("running_mean") or k.endswith("running_var"):
                np.testing.assert_allclose(have[k].cpu().numpy(), v.numpy(), rtol=2e-4, atol=2e-5 * float(v.abs().max()) + 1e-7, err_msg=k)
        print("CMFlow-T B=64 frame %d: loss %.6f vs %.6f, grads: norm %.3g (%s) 1-cos %.3g (%s) element %.3g (%s), rot %.3g rad, dt %.3g m"
              % (f, loss.item(), loss_ref.item(), worst[1], worst[0], wcos[1], wcos[0], welem[1], welem[0], ang, dt))
        g_prev = out[4].detach().clone()
        opt.step()


def test_cmflow_t_clip_training_matches_oracle(dev, manifest_t, golden_dir, args):
    """Row a16 in training: two consecutive frames of a mini-clip with the GRU state hand-off
    (clip_util.py:34-62: gfeat.detach(), optimizer step per frame), HIP path vs CPU oracle."""
    from cmflow_amd.cmflow import CMFlow_T
    from cmflow_amd.train import TrainStep
    sd = _weights(manifest_t, golden_dir, t=True)
    ref = O.CMFlow_T(args)
    ref.load_state_dict(sd)
    ref.train()
    net = CMFlow_T(args)
    net.load_state_dict(sd)
    net = net.to(dev).train()
    step = TrainStep(net, vr_thres=args.vr_thres)
    opt = torch.optim.Adam(ref.parameters(), lr=0.001, weight_decay=1e-4)
    P, Tcr = torch.tensor(synth.CAMERA_PROJECTION), torch.tensor(synth.T_CAMERA_RADAR)
    g_ref = None
    for frame, seed in enumerate((21, 22)):
        # Adam's first steps move every weight by ~lr*sign(g): elements whose gradient is at the fp32 noise
        # level get different signs in two correct implementations, so the weights are re-synchronised before
        # each frame and the comparison is per frame (forward with carried GRU state, loss, gradients)
        net.load_state_dict(ref.state_dict())
        b = synth.make_batch(4, seed=seed, train_extras=True)
        dyn, mseg = TO.make_labels(b)
        out = ref(b["pc1"], b["pc2"], b["ft1"], b["ft2"], mseg, "train", g_ref.detach() if g_ref is not None else None)
        ref_after = None
        g_ref = out[4]
        loss_ref, _ = TO.radar_flow_loss(b, out[0], out[2], out[1], mseg, dyn, P, Tcr)
        opt.zero_grad(); loss_ref.backward()
        if frame == 1:
            step.gfeat = g_prev.to(dev)                       # same carried state on both sides
        loss, items, outs, _ = step.forward_loss({k: v.to(dev) for k, v in b.items()})[:4]
        step.bucket.zero()
        loss.backward()
        assert abs(loss.item() - loss_ref.item()) < 2e-4 * max(1.0, abs(loss_ref.item())), (frame, loss.item(), loss_ref.item())
        assert _epe(outs[0].detach().cpu().numpy(), out[0].detach().numpy()) < 1e-4
        np.testing.assert_allclose(step.gfeat.detach().cpu().numpy(), g_ref.detach().numpy(), rtol=0, atol=1e-4)
        for name in ("gru.weight_hh_l0", "gru.weight_ih_l0", "fp.conv2.weight", "mse_layer2.ms_ls.2.mlp_convs.1.weight"):
            ga = dict(net.named_parameters())[name].grad.cpu().numpy()
            gb = dict(ref.named_parameters())[name].grad.numpy()
            assert abs(np.linalg.norm(ga) - np.linalg.norm(gb)) <= 1e-2 * max(np.linalg.norm(gb), 1e-6), (frame, name)
        g_prev = g_ref.detach().clone()
        opt.step()


def test_single_rank_rccl_all_reduce(dev, manifest, golden_dir, args):
    """The gradient bucket goes through RCCL (backend 'nccl') -- world size 1 on this 1-GPU box; the
    2-rank semantics are covered on CPU with gloo (tests/test_dp.py)."""
    import os
    import torch.distributed as dist
    from cmflow_amd.dp import FlatGradBucket
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        lin = torch.nn.Linear(8, 4).to(dev)
        bucket = FlatGradBucket(lin)
        bucket.zero()
        lin(torch.ones(2, 8, device=dev)).sum().backward()
        before = bucket.flat.clone()
        bucket.all_reduce_mean(force=True)
        torch.cuda.synchronize()
        assert torch.equal(before, bucket.flat) and float(before.abs().sum()) > 0
    finally:
        dist.destroy_process_group()


def test_rccl_step_three_ways_bit_identical(dev, manifest, golden_dir, args):
    """The exact path N > 1 GPUs would run, on the REAL model over backend 'nccl' (= RCCL; world size 1 on this box): one CMFlow
    B = 8 optimizer step (a) without a collective, (b) with one all-reduce of the whole bucket after backward, (c) with the
    three overlapped segments (ReduceOp.AVG, async_op=True, launched from tensor hooks inside backward).  Gradient bucket and
    parameters after Adam must be bit-identical in all three (models/model.py:40-42 is what this replaces)."""
    import os
    import torch.distributed as dist
    from cmflow_amd.cmflow import CMFlow
    from cmflow_amd.fused_blocks import join_side_streams
    from cmflow_amd.train import TrainStep
    sd = _weights(manifest, golden_dir)
    b = {k: v.to(dev) for k, v in synth.make_batch(8, seed=11, train_extras=True).items()}

    def run(overlap, force):
        net = CMFlow(args)
        net.load_state_dict(sd)
        net = net.to(dev).train()
        step = TrainStep(net, vr_thres=args.vr_thres)
        step.overlap_allreduce, step.force_allreduce = overlap, force
        loss, _, _, _ = step(b)
        join_side_streams()
        torch.cuda.synchronize()
        early = step.reducer.early if (overlap and step.reducer is not None) else None
        return loss.clone(), step.bucket.flat.clone(), torch.cat([p.detach().reshape(-1) for p in net.parameters()]).clone(), early

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29513")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        none = run(False, False)
        one = run(False, True)
        three = run(True, True)
    finally:
        dist.destroy_process_group()
    assert three[3] == 2                                 # two of the three segments were launched while backward was still running
    assert float(none[1].abs().sum()) > 0
    for i, what in enumerate(("loss", "gradient bucket", "parameters after Adam")):
        assert torch.equal(one[i], none[i]), "one bucket vs none: %s" % what
        assert torch.equal(three[i], none[i]), "three overlapped segments vs none: %s" % what


def test_encoder_plan_follows_parameter_reallocation_and_guards_grad_sinks(dev, manifest, golden_dir, args):
    """The cached call plan of an encoder (fused_blocks.EncoderPlan) holds raw parameter pointers: it must notice when
    the parameters are re-allocated, and the in-place gradient sinks must refuse to run once the .grad buffers they
    point at are gone."""
    from cmflow_amd.cmflow import CMFlow
    from cmflow_amd.train import TrainStep
    net = CMFlow(args)
    net.load_state_dict(_weights(manifest, golden_dir))
    net = net.to(dev).train()
    step = TrainStep(net, vr_thres=args.vr_thres)
    b = {k: v.to(dev) for k, v in synth.make_batch(4, seed=5, train_extras=True).items()}
    loss0, _, _, _ = step.forward_loss(b)
    assert net.mse_layer2._plans, "the fused path did not build a call plan"
    # re-allocate every parameter (what .to()/.half().float() do): same values, new storage
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    for p in net.parameters():
        p.data = p.data.clone()
    step2 = TrainStep(net, vr_thres=args.vr_thres)                 # fresh gradient bucket over the new storage
    net.load_state_dict(sd)
    loss1, _, _, _ = step2.forward_loss(b)
    # BN running statistics moved by the first forward, batch statistics did not: train-mode outputs are identical
    assert abs(loss0.item() - loss1.item()) < 1e-5 * max(1.0, abs(loss0.item()))
    # sinks: dropping the .grad buffers between forward and backward must raise, not write through stale pointers
    loss2, _, _, _ = step2.forward_loss(b)
    for p in net.parameters():
        p.grad = None
    with pytest.raises(RuntimeError):
        loss2.backward()


def test_concurrent_first_encoder_equals_sequential_calls(dev, manifest, golden_dir, args, monkeypatch):
    """The two calls of the weight-shared first encoder run concurrently as one node (DualCloudBlockFn: deferred BN
    running-statistics update, scratch gradients for the second call).  One optimizer step must leave the network in
    the same state as the two calls issued one after the other: every parameter, every BN buffer, the loss."""
    from cmflow_amd import fused_blocks as FB
    from cmflow_amd.cmflow import CMFlow
    from cmflow_amd.train import TrainStep
    b = {k: v.to(dev) for k, v in synth.make_batch(8, seed=31, train_extras=True).items()}
    states, losses, used = [], [], []
    for concurrent in (True, False):
        net = CMFlow(args)
        net.load_state_dict(_weights(manifest, golden_dir))
        net = net.to(dev).train()
        step = TrainStep(net, vr_thres=args.vr_thres)
        calls = []
        real = FB.dual_cloud_set_conv
        if concurrent:
            monkeypatch.setattr(FB, "dual_cloud_set_conv", lambda *a: (calls.append(1), real(*a))[1])
        else:
            monkeypatch.setattr(FB, "dual_cloud_set_conv", lambda *a: None)
        loss, _, _, _ = step(b)
        monkeypatch.setattr(FB, "dual_cloud_set_conv", real)
        used.append(len(calls))
        losses.append(loss.item())
        states.append({k: v.detach().clone() for k, v in net.state_dict().items()})
    assert used == [1, 0]
    assert abs(losses[0] - losses[1]) <= 1e-6 * max(1.0, abs(losses[1]))
    for k, v in states[1].items():
        a = states[0][k]
        if v.dtype.is_floating_point:
            # Adam's first step moves a weight by lr * sign(g): entries whose gradient is at rounding level may flip
            tol = 2.1e-3 if (k.endswith("weight") or k.endswith("bias")) else 0.0
            diff = (a - v).abs()
            bad = diff > 1e-5 + 1e-5 * v.abs()
            assert bad.float().mean().item() <= (0.02 if tol else 0.0) and diff.max().item() <= max(tol, 1e-5 + 1e-5 * v.abs().max().item()), k
        else:
            assert torch.equal(a, v), k                     # num_batches_tracked: +2 for the first encoder


def test_side_streams_do_not_change_results(dev, manifest, golden_dir, args):
    """The motion head and the cost volume's neighbourhood branch / second per-point GEMM run on the shared pool of side
    streams (fused_blocks.side_stream), next to the encoder scales.  The kernels are the same ones and deterministic,
    so outputs, loss and all gradients must be BIT-identical to the same step with those branches on the caller's
    stream -- a missing stream dependency shows up here as a difference (or as garbage)."""
    from cmflow_amd import fused_blocks as FB
    from cmflow_amd.cmflow import CMFlow
    from cmflow_amd.radarflow_util import FeatureCorrelator
    from cmflow_amd.train import TrainStep
    assert FB.side_stream(0) is FB.side_stream(FB.N_SIDE) and len({FB.side_stream(i) for i in range(FB.N_SIDE)}) == FB.N_SIDE
    assert [s is FB.side_stream(i) for s, i in zip(FB.scale_streams(4, 0), (0, 1, 1, 2))] == [True] * 4
    b = {k: v.to(dev) for k, v in synth.make_batch(8, seed=77, train_extras=True).items()}
    results = []
    for side in (True, False, True):
        net = CMFlow(args)
        net.load_state_dict(_weights(manifest, golden_dir))
        net = net.to(dev).train()
        net.head_streams = side
        for m in net.modules():
            if isinstance(m, FeatureCorrelator):
                m.side_streams = side
        step = TrainStep(net, vr_thres=args.vr_thres)
        for _ in range(2):                                  # second pass: warm plans, re-used streams and cached blocks
            loss, items, outs, labels = step.forward_loss(b)
            step.bucket.zero()
            loss.backward()
        torch.cuda.synchronize()
        results.append(([o.detach().clone() for o in outs[:3]], loss.detach().clone(), step.bucket.flat.detach().clone()))
    ref = results[1]
    for got in (results[0], results[2]):
        for a, r in zip(got[0], ref[0]):
            assert torch.equal(a, r)
        assert torch.equal(got[1], ref[1])
        assert torch.equal(got[2], ref[2])


def test_all_bn_buffers_after_one_step_match_oracle(dev, manifest, golden_dir, args):
    """Every BatchNorm buffer of the network (running_mean, running_var, num_batches_tracked: 156 tensors) after one
    training step, HIP path vs the CPU oracle -- the goldens only pin a handful of them."""
    from cmflow_amd.cmflow import CMFlow
    from cmflow_amd.train import TrainStep
    sd = _weights(manifest, golden_dir)
    ref = O.CMFlow(args)
    ref.load_state_dict(sd)
    ref.train()
    net = CMFlow(args)
    net.load_state_dict(sd)
    net = net.to(dev).train()
    b = synth.make_batch(4, seed=41, train_extras=True)
    P, Tcr = torch.tensor(synth.CAMERA_PROJECTION), torch.tensor(synth.T_CAMERA_RADAR)
    TO.train_step(ref, torch.optim.Adam(ref.parameters(), lr=0.001, weight_decay=1e-4), b, P, Tcr)
    TrainStep(net, vr_thres=args.vr_thres)({k: v.to(dev) for k, v in b.items()})
    want, got = ref.state_dict(), net.state_dict()
    n = 0
    for k, v in want.items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            g = got[k].cpu()
            assert torch.isfinite(g).all(), k
            np.testing.assert_allclose(g.numpy(), v.numpy(), rtol=2e-4, atol=2e-5 * float(v.abs().max()) + 1e-7, err_msg=k)
            n += 1
        elif k.endswith("num_batches_tracked"):
            assert int(got[k]) == int(v), k
            n += 1
    assert n >= 150


@pytest.mark.parametrize("B,N", [(1, 256), (3, 128), (2, 200), (5, 64)])
def test_odd_shapes_match_oracle(dev, manifest, golden_dir, args, B, N):
    """Batch sizes and cloud sizes other than the benchmark's (ragged tiles, N not a multiple of 64 or 128): forward in
    eval mode and one training step (loss + gradient norms) against the CPU oracle."""
    from cmflow_amd.cmflow import CMFlow
    from cmflow_amd.train import TrainStep
    sd = _weights(manifest, golden_dir)
    ref = O.CMFlow(args)
    ref.load_state_dict(sd)
    net = CMFlow(args)
    net.load_state_dict(sd)
    net = net.to(dev)
    b = synth.make_batch(B, N, seed=100 + B + N, train_extras=True)
    bd = {k: v.to(dev) for k, v in b.items()}
    ref.eval(); net.eval()
    with torch.no_grad():
        want = ref(b["pc1"], b["pc2"], b["ft1"], b["ft2"], None, "test")
        got = net(bd["pc1"], bd["pc2"], bd["ft1"], bd["ft2"], None, "test")
    flips = (got[3].cpu() != want[3])
    assert flips.float().mean().item() <= 0.01                       # stat_cls against the 0.5 threshold
    ok = ~flips.unsqueeze(1).expand(-1, 3, -1)
    scale = max(1.0, float(want[0].abs().max()))
    assert (got[0].cpu() - want[0])[ok].abs().max().item() <= 2e-4 * scale
    assert (got[1].cpu() - want[1]).abs().max().item() <= 2e-4
    if not flips.any():
        np.testing.assert_allclose(got[2].cpu().numpy(), want[2].numpy(), rtol=3e-5, atol=2e-4)
    ref.train(); net.train()
    P, Tcr = torch.tensor(synth.CAMERA_PROJECTION), torch.tensor(synth.T_CAMERA_RADAR)
    opt = torch.optim.Adam(ref.parameters(), lr=0.001, weight_decay=1e-4)
    loss_ref, _, _, _ = TO.train_step(ref, opt, b, P, Tcr)
    step = TrainStep(net, vr_thres=args.vr_thres)
    loss, _, _, _ = step.forward_loss(bd)
    step.bucket.zero()
    loss.backward()
    assert abs(loss.item() - loss_ref.item()) <= 3e-4 * max(1.0, abs(loss_ref.item()))
    gref = {k: p.grad for k, p in ref.named_parameters() if p.grad is not None}
    worst = 0.0
    for k, p in net.named_parameters():
        if k in gref:
            a, r = float(p.grad.norm()), float(gref[k].norm())
            worst = max(worst, abs(a - r) / max(r, 1e-3))
    assert worst <= 2e-2, worst


def test_dense_cloud_forward_matches_oracle(dev, manifest, golden_dir, args):
    """BASELINE config 5's cloud size through the WHOLE model (N = 4096 LiDAR-like points; the op-level roofline run of
    that config is tools/op_bench.py): forward in eval mode against the CPU oracle -- flow within 1e-4, identical
    static masks, transform within 1e-4.  Exercises the large-N paths (multi-wave ball query with spilled hit lists,
    count/fill inverse index, 16-row LDS tiles of the global max, kNN over 4096 candidates)."""
    from cmflow_amd.cmflow import CMFlow
    sd = _weights(manifest, golden_dir)
    ref = O.CMFlow(args)
    ref.load_state_dict(sd)
    ref.eval()
    net = CMFlow(args)
    net.load_state_dict(sd)
    net = net.to(dev).eval()
    b = synth.make_batch(1, N=4096, seed=2025, lidar=True)
    with torch.no_grad():
        want = ref(b["pc1"], b["pc2"], b["ft1"], b["ft2"], None, "test")
        got = net(*(b[k].to(dev) for k in ("pc1", "pc2", "ft1", "ft2")), None, "test")
    flips = got[3].cpu() != want[3]
    assert int(flips.sum()) <= 2                                  # a score within rounding of the threshold may flip
    epe = (got[0].cpu() - want[0]).norm(dim=1)
    assert float(epe[~flips].max()) < 1e-4
    assert float((got[1].cpu() - want[1]).abs().max()) < 1e-4
    assert float((got[2].cpu() - want[2]).abs().max()) < 1e-4


def test_dense_cloud_train_step_matches_oracle(dev, manifest, golden_dir, args):
    """A whole training step at BASELINE config 5's cloud size (N = 4096 LiDAR-like points, B = 2, train-mode BN, the seven
    losses): the fused loss takes the cloud through its tiled kernels (losses/radar_loss.py has no size limit), the backward
    pass runs the large-N forms of the inverse index, the scatter and the gathering GEMMs.  Against the CPU oracle: labels
    bit-equal, loss and items within 2e-4, every gradient tensor by norm / direction / largest element."""
    from cmflow_amd.cmflow import CMFlow
    from cmflow_amd.fused_blocks import join_side_streams
    from cmflow_amd.train import TrainStep
    torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
    sd = _weights(manifest, golden_dir)
    ref = O.CMFlow(args)
    ref.load_state_dict(sd)
    ref.train()
    net = CMFlow(args)
    net.load_state_dict(sd)
    net = net.to(dev).train()
    b = synth.make_batch(2, N=4096, seed=2026, lidar=True, train_extras=True)
    bd = {k: v.to(dev) for k, v in b.items()}
    P, Tcr = torch.tensor(synth.CAMERA_PROJECTION), torch.tensor(synth.T_CAMERA_RADAR)

    class NoStep:
        def zero_grad(self): ref.zero_grad()
        def step(self): pass
    loss_ref, items_ref, out_ref, (dyn_ref, mseg_ref) = TO.train_step(ref, NoStep(), b, P, Tcr)
    step = TrainStep(net, vr_thres=0.3)
    loss, items, outs, (dyn, mseg) = step.forward_loss(bd)
    step.bucket.zero()
    loss.backward()
    join_side_streams()
    assert torch.equal(dyn.cpu(), dyn_ref) and torch.equal(mseg.cpu(), mseg_ref)
    assert abs(loss.item() - loss_ref.item()) <= 2e-4 * max(1.0, abs(loss_ref.item())), (loss.item(), loss_ref.item())
    for k, v in items.items():
        assert abs(v.item() - items_ref[k]) <= 2e-4 * max(1.0, abs(items_ref[k])), (k, v.item(), items_ref[k])
    assert float((outs[1].detach().cpu() - out_ref[1].detach()).abs().max()) <= 1e-4
    gref = {k: p.grad for k, p in ref.named_parameters()}
    # (two samples instead of 64: fewer terms per sum, but the near-tie decisions of 8192 points' neighbour searches weigh more)
    n, worst, wcos, welem = _check_gradients(net, gref, "N=4096 CMFlow", bounds=(2e-2, 5e-4, 0.1))
    assert n >= 180
    print("dense train: loss %.6f vs %.6f, worst grad-norm rel err %.3g (%s), worst 1-cos %.3g (%s), worst element %.3g (%s)"
          % (loss.item(), loss_ref.item(), worst[1], worst[0], wcos[1], wcos[0], welem[1], welem[0]))


def test_bench_two_ranks_control_flow(dev):
    """bench.py under torch.distributed.run with two ranks (both on cuda:0 over gloo, CMF_BENCH_ONE_GPU=1): every
    collective of the script -- parameter broadcast, the gradient all-reduce inside every step INCLUDING the extra
    isolated-roofline steps after the timed region, the max-over-ranks of the time -- is entered by both ranks, rank 0
    prints exactly one JSON line, and both processes exit cleanly (a rank-0-only training step would hang here)."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, CMF_BENCH_ONE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["parallelism"] == "dp2" and rec["cpu_baseline"] is None
    assert rec["roofline"] is not None and rec["roofline_isolated"] is not None and rec["value"] > 0


def test_two_rank_cmflow_step_matches_single_rank_shards(dev, tmp_path):
    """SURVEY 8e's parity check on the REAL model: a 2-rank data-parallel training step (both ranks on cuda:0 over gloo,
    tests/dp_worker.py; global B=8 -> 4+4) against single-process runs on each shard.
      * rank r's forward outputs, loss and LOCAL gradient bucket == a single-process run on shard r (same kernels:
        <= 1e-6 relative; bit-equal in practice -- the kernels are deterministic);
      * the all-reduced bucket is identical on both ranks and == the mean of the two single-process buckets;
      * parameters after the Adam step are bit-identical across ranks;
      * BN running statistics are PER RANK (each rank == the single-process run on its shard), which is what
        nn.DataParallel's replicas do as well (models/model.py:40-42: only replica 0's buffers persist)."""
    import socket
    import subprocess
    import sys
    import bench
    from cmflow_amd.cmflow import CMFlow
    from cmflow_amd.dp import shard_batch
    from cmflow_amd.train import TrainStep
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "tests", "dp_worker.py"), str(tmp_path), "8"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root, env=dict(os.environ))
    assert out.returncode == 0, out.stderr[-3000:]
    ranks = [torch.load(os.path.join(tmp_path, "rank%d.pt" % r)) for r in range(2)]
    assert torch.equal(ranks[0]["averaged"], ranks[1]["averaged"])
    assert ranks[0]["early"] == 2 and ranks[1]["early"] == 2      # heads + second encoder and the cost volume were reduced DURING backward
    for k, v in ranks[0]["params"].items():
        assert torch.equal(v, ranks[1]["params"][k]), k

    def close(a, b, what):
        a, b = a.double(), b.double()
        assert float((a - b).abs().max()) <= 1e-6 * max(1e-30, float(b.abs().max())), what
        return bool(torch.equal(a, b))

    gb = synth.make_batch(8, seed=777, train_extras=True)
    singles, exact = [], True
    for r in range(2):
        net = CMFlow(bench.Args())
        net.load_state_dict(bench.load_weights("cmflow"))
        net = net.to(dev).train()
        step = TrainStep(net, vr_thres=bench.Args.vr_thres)
        loss, _, outs, _ = step.forward_loss({k: v.to(dev) for k, v in shard_batch(gb, r, 2).items()})
        step.bucket.zero()
        loss.backward()
        torch.cuda.synchronize()
        exact &= close(ranks[r]["loss"], loss.detach().cpu(), "loss")
        for a, b in zip(ranks[r]["outs"], outs[:3]):
            exact &= close(a, b.detach().cpu(), "outputs of rank %d" % r)
        exact &= close(ranks[r]["local"], step.bucket.flat.cpu(), "local gradient bucket of rank %d" % r)
        have = net.state_dict()
        for k, v in ranks[r]["buffers"].items():
            exact &= close(v, have[k].cpu(), k)
        singles.append(step.bucket.flat.cpu().double())
    mean = (singles[0] + singles[1]) / 2
    assert float((ranks[0]["averaged"].double() - mean).abs().max()) <= 1e-6 * float(mean.abs().max())
    assert float(ranks[0]["local"].abs().sum()) > 0 and not torch.equal(ranks[0]["local"], ranks[1]["local"])
    print("2-rank CMFlow step vs single-rank shards: bit-identical" if exact else "2-rank CMFlow step vs single-rank shards: within 1e-6")


def test_two_rank_cmflow_t_clip_matches_single_rank_shards(dev, tmp_path):
    """BASELINE config 4's semantics on the real model: a 2-rank data-parallel CMFlow-T mini-clip (two frames; both ranks on
    cuda:0 over gloo, tests/dp_worker.py) against single-process runs on each shard.  The GRU state is carried PER RANK
    (gfeat.detach(), clip_util.py:54) and the optimizer steps after every frame on the all-reduced gradient (:60-62), so
    the single-process twin of rank r steps with the 2-rank run's averaged bucket and must then reproduce rank r's second
    frame -- loss, outputs, the new state and the local gradient bucket -- which only works if state hand-off, bucket
    layout (incl. the four GRU tensors) and parameter update agree."""
    import socket
    import subprocess
    import sys
    import bench
    from cmflow_amd.cmflow import CMFlow_T
    from cmflow_amd.dp import shard_batch
    from cmflow_amd.fused_blocks import join_side_streams
    from cmflow_amd.train import TrainStep
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "tests", "dp_worker.py"), str(tmp_path), "8", "cmflow_t", "2"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    ranks = [torch.load(os.path.join(tmp_path, "rank%d.pt" % r)) for r in range(2)]
    for f in range(2):
        assert ranks[0]["frames"][f]["early"] == 2 and ranks[1]["frames"][f]["early"] == 2
        assert torch.equal(ranks[0]["frames"][f]["averaged"], ranks[1]["frames"][f]["averaged"]), f
        assert not torch.equal(ranks[0]["frames"][f]["gfeat"], ranks[1]["frames"][f]["gfeat"])     # per-rank state
    for k, v in ranks[0]["params"].items():
        assert torch.equal(v, ranks[1]["params"][k]), k

    def close(a, b, what):
        a, b = a.double(), b.double()
        assert float((a - b).abs().max()) <= 1e-6 * max(1e-30, float(b.abs().max())), what

    for r in range(2):
        net = CMFlow_T(bench.Args())
        net.load_state_dict(bench.load_weights("cmflow_t"))
        net = net.to(dev).train()
        step = TrainStep(net, vr_thres=bench.Args.vr_thres)
        locals_ = []
        for f in range(2):
            gb = synth.make_batch(8, seed=777 + f, train_extras=True)
            loss, _, outs, _ = step.forward_loss({k: v.to(dev) for k, v in shard_batch(gb, r, 2).items()})
            step.bucket.zero()
            loss.backward()
            join_side_streams()
            torch.cuda.synchronize()
            fr = ranks[r]["frames"][f]
            close(fr["loss"], loss.detach().cpu(), "loss, rank %d frame %d" % (r, f))
            for a, b in zip(fr["outs"], outs[:3]):
                close(a, b.detach().cpu(), "outputs, rank %d frame %d" % (r, f))
            close(fr["gfeat"], step.gfeat.detach().cpu(), "GRU state, rank %d frame %d" % (r, f))
            close(fr["local"], step.bucket.flat.cpu(), "local gradient bucket, rank %d frame %d" % (r, f))
            locals_.append(step.bucket.flat.detach().cpu().double())
            step.bucket.flat.copy_(fr["averaged"].to(dev))            # step on what the all-reduce delivered
            step.opt.step()
        have = dict(net.named_parameters())
        for k, v in ranks[r]["params"].items():
            close(v, have[k].detach().cpu(), k)
        ranks[r]["_locals"] = locals_
    for f in range(2):
        mean = (ranks[0]["_locals"][f] + ranks[1]["_locals"][f]) / 2
        assert float((ranks[0]["frames"][f]["averaged"].double() - mean).abs().max()) <= 1e-6 * float(mean.abs().max())


def test_flat_adam_matches_torch_adam(dev):
    """dp.FlatAdam (cmf_adam_step: one launch over the flat gradient bucket) against torch.optim.Adam with the reference's settings
    (lr 1e-3, L2 weight decay 1e-4, main.py:107) on a module with odd tensor sizes: parameters within 2e-6 relative after each of 5
    steps (same update rule; torch forms its bias corrections in double as well), a StepLR acts on it like on torch's, and a
    parameter whose storage is replaced is followed."""
    from cmflow_amd.dp import FlatAdam, FlatGradBucket
    torch.manual_seed(3)
    # (no bias in front of the BatchNorm: its true gradient is zero, and Adam normalises the rounding noise it gets instead into steps of lr)
    mk = lambda: torch.nn.Sequential(torch.nn.Linear(37, 129, bias=False), torch.nn.BatchNorm1d(129), torch.nn.ReLU(), torch.nn.Linear(129, 5)).to(dev)
    a, b = mk(), mk()
    b.load_state_dict(a.state_dict())
    bucket = FlatGradBucket(a)
    oa = FlatAdam(bucket, lr=1e-3, weight_decay=1e-4)
    ob = torch.optim.Adam(b.parameters(), lr=1e-3, weight_decay=1e-4)
    sa, sb = torch.optim.lr_scheduler.StepLR(oa, 2, gamma=0.5), torch.optim.lr_scheduler.StepLR(ob, 2, gamma=0.5)
    for it in range(5):
        x = torch.randn(64, 37, device=dev)
        for net, opt in ((a, oa), (b, ob)):
            opt.zero_grad() if net is b else bucket.zero()
            net(x).square().mean().backward()
            opt.step()
        sa.step(); sb.step()
        assert oa.param_groups[0]["lr"] == ob.param_groups[0]["lr"]
        for (k, p), q in zip(a.named_parameters(), b.parameters()):
            np.testing.assert_allclose(p.detach().cpu().numpy(), q.detach().cpu().numpy(), rtol=2e-6, atol=2e-7, err_msg="%s step %d" % (k, it))
        if it == 2:                                         # a replaced storage (same values) must be picked up
            p0 = next(a.parameters())
            p0.data = p0.data.clone()
    # a torch.optim.Adam checkpoint over the same parameters is accepted: its moments and step count land in the flat arrays, and the
    # next step from it equals torch's next step
    oa2 = FlatAdam(bucket, lr=1e-3, weight_decay=1e-4)
    oa2.load_state_dict(ob.state_dict())
    assert oa2.steps == 5 and float(oa2.exp_avg.abs().sum()) > 0
    off = 0
    for q in b.parameters():
        st = ob.state[q]
        assert torch.equal(oa2.exp_avg[off:off + q.numel()], st["exp_avg"].reshape(-1)) and torch.equal(oa2.exp_avg_sq[off:off + q.numel()], st["exp_avg_sq"].reshape(-1))
        off += q.numel()
    with pytest.raises(ValueError):
        bad = ob.state_dict(); bad["state"] = {k: v for k, v in bad["state"].items() if k != 0}
        FlatAdam(bucket).load_state_dict(bad)
    # a checkpoint of torch.optim.Adam(net.parameters()) -- the reference's optimizer, main.py:107 -- over a module that
    # also holds parameters OUTSIDE the bucket (never-used ones with grad None, frozen ones): indices are positions in
    # module.parameters(); the outsiders carry no state and are skipped
    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = torch.nn.Linear(7, 5)
            self.unused = torch.nn.BatchNorm1d(5)
            self.b = torch.nn.Linear(5, 3)
            self.frozen = torch.nn.Linear(3, 3)
            for q in self.unused.parameters():
                q._cmf_unused = True
            for q in self.frozen.parameters():
                q.requires_grad_(False)

        def forward(self, x):
            return self.frozen(self.b(torch.relu(self.a(x))))
    torch.manual_seed(5)
    m1, m2 = Net().to(dev), Net().to(dev)
    m2.load_state_dict(m1.state_dict())
    for q in m2.unused.parameters():
        q._cmf_unused = True
    ref_opt = torch.optim.Adam(m1.parameters(), lr=1e-3, weight_decay=1e-4)
    for _ in range(3):
        ref_opt.zero_grad()
        m1(torch.randn(16, 7, device=dev)).square().mean().backward()
        ref_opt.step()
    bk = FlatGradBucket(m2)
    assert bk.module_params == 8 and bk.module_index == [0, 1, 4, 5]
    fa = FlatAdam(bk, lr=1e-3, weight_decay=1e-4)
    fa.load_state_dict(ref_opt.state_dict())
    assert fa.steps == 3
    off = 0
    for q in (m1.a.weight, m1.a.bias, m1.b.weight, m1.b.bias):
        st = ref_opt.state[q]
        assert torch.equal(fa.exp_avg[off:off + q.numel()], st["exp_avg"].reshape(-1))
        assert torch.equal(fa.exp_avg_sq[off:off + q.numel()], st["exp_avg_sq"].reshape(-1))
        off += q.numel()
    with pytest.raises(ValueError):                           # a checkpoint over some other parameter list
        other = torch.optim.Adam(list(m1.parameters())[:3]).state_dict()
        FlatAdam(bk).load_state_dict(other)
    # the raw-pointer update bumps the parameters' version counters like an in-place torch op
    p0 = next(a.parameters())
    v0 = p0._version
    bucket.zero(); a(torch.randn(8, 37, device=dev)).square().mean().backward(); oa.step()
    assert p0._version > v0
