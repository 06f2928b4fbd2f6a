#!/usr/bin/env python3
"""Golden vectors for SURVEY 8f rank 4, produced by the REFERENCE's own code on torch-CPU behind the shims of
make_golden.py (build container only):

  * models/raflow.py RaFlow: state-dict manifest, BN calibration, an eval forward (B=2) and one self-supervised
    train step (main_util.py:57-60,74-76: forward -> RadarFlowLoss('raflow') -> Adam);
  * dataset/vod.py vodDataset: the JSON sample format read back through __getitem__ (train partition with
    resampling to 256 points under a fixed numpy seed, and test partition), from samples written by
    cmflow_amd.dataset.write_sample.

    python tests/golden/make_golden_raflow.py
"""
import json
import os
import shutil
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import install_shims, manifest_of, np_, synth  # noqa: E402


class Args:
    num_points = 256
    rigid_thres = 0.15          # configs.yaml:29
    model = "raflow"
    eval = False


class Textio:
    def cprint(self, s):
        pass


def main():
    tmp = tempfile.mkdtemp(prefix="cmf_vod_")
    from cmflow_amd import dataset as D                               # writer only; imported before the chdir
    names = D.write_synthetic_split(tmp, seed=3)
    install_shims()
    from models.raflow import RaFlow
    from losses import RadarFlowLoss
    torch.manual_seed(1234)
    torch.set_num_threads(8)
    args = Args()
    net = RaFlow(args)
    man = manifest_of(net)
    json.dump(man, open(os.path.join(HERE, "state_manifest_raflow.json"), "w"), indent=0)
    net.load_state_dict(synth.synth_state_dict(man, seed=1234))
    net.train()
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.momentum = 1.0
    cb = synth.make_batch(8, seed=555, train_extras=True)
    with torch.no_grad():
        net(cb["pc1"], cb["pc2"], cb["ft1"], cb["ft2"], cb["interval"])
    CAL = os.path.join(HERE, "bn_calib_raflow.npz")
    np.savez_compressed(CAL, **{k: np_(v).copy() for k, v in net.state_dict().items()
                                if k.endswith("running_mean") or k.endswith("running_var")})
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.momentum = 0.1

    # ---- eval forward ----
    net.load_state_dict(synth.synth_state_dict(man, seed=1234, calib=CAL))
    net.eval()
    b = synth.make_batch(2, seed=1234, train_extras=True)
    with torch.no_grad():
        out, sf, trans, mask_s = net(b["pc1"], b["pc2"], b["ft1"], b["ft2"], b["interval"])
    ev = {k: np_(b[k]).copy() for k in ("pc1", "pc2", "ft1", "ft2", "interval")}
    ev.update(output=np_(out).copy(), sf_agg=np_(sf).copy(), pre_trans=np_(trans).copy(), mask_s=np_(mask_s).copy())
    np.savez_compressed(os.path.join(HERE, "raflow_eval_synth_b2.npz"), **ev)
    print("eval: |output| %.4f  |sf_agg| %.4f  mask_s frac %.3f" % (np.abs(ev["output"]).mean(), np.abs(ev["sf_agg"]).mean(), ev["mask_s"].mean()))

    # a second evaluation with a loose inlier threshold so that the re-fit branch (raflow.py:107-113) is taken
    # torch.topk leaves the order among equal distances open (fp32 squared distances ~90 m from the sensor are
    # quantised to 2^-10 m^2): take the first seed whose 8-NN sets (cross and self, radarflow_util.py:207,228) have
    # no tie at the 8th/9th neighbour, so the golden does not depend on a tie-break
    from utils.model_utils.radarflow_util import square_distance

    def unambiguous(bt):
        x1, x2 = bt["pc1"].permute(0, 2, 1), bt["pc2"].permute(0, 2, 1)
        for d in (square_distance(x1, x2), square_distance(x1, x1)):
            sd = torch.sort(d, dim=-1)[0]
            if not bool((sd[:, :, 7] < sd[:, :, 8]).all()):
                return False
        return True

    loose_seed = next(sd_ for sd_ in range(77, 400) if unambiguous(synth.make_batch(4, seed=sd_, train_extras=True)))
    print("loose eval seed", loose_seed)
    b4 = synth.make_batch(4, seed=loose_seed, train_extras=True)
    for thres in (2.0,):
        net.rigid_thres = thres
        with torch.no_grad():
            out, sf, trans, mask_s = net(b4["pc1"], b4["pc2"], b4["ft1"], b4["ft2"], b4["interval"])
        ev = {k: np_(b4[k]).copy() for k in ("pc1", "pc2", "ft1", "ft2", "interval")}
        ev.update(output=np_(out).copy(), sf_agg=np_(sf).copy(), pre_trans=np_(trans).copy(), mask_s=np_(mask_s).copy(),
                  rigid_thres=np.float32(thres))
        np.savez_compressed(os.path.join(HERE, "raflow_eval_synth_b4_loose.npz"), **ev)
        print("loose eval: inlier fraction per sample", ev["mask_s"].mean(axis=1), " refit changed flow:",
              float(np.abs(ev["sf_agg"] - ev["output"]).max()))
    net.rigid_thres = args.rigid_thres

    # ---- one train step ----
    net.load_state_dict(synth.synth_state_dict(man, seed=1234, calib=CAL))
    net.train()
    b = synth.make_batch(4, seed=4321, train_extras=True)
    opt = torch.optim.Adam(net.parameters(), lr=0.001, weight_decay=1e-4)          # main.py:107
    _, pred_f, _, _ = net(b["pc1"], b["pc2"], b["ft1"], b["ft2"], b["interval"])
    loss, items = RadarFlowLoss()(args, b["pc1"], b["pc2"], pred_f, b["ft1"][:, 0])
    opt.zero_grad()
    loss.backward()
    tr = {k: np_(b[k]).copy() for k in ("pc1", "pc2", "ft1", "ft2", "interval")}
    tr["sf_agg"] = np_(pred_f).copy()
    tr["loss"] = np.float32(loss.item())
    for k, v in items.items():
        tr["item_" + k] = np.float32(v)
    names_g, norms = [], []
    for k, p in net.named_parameters():
        if p.grad is not None:
            names_g.append(k)
            norms.append(float(p.grad.norm()))
    tr["grad_names"], tr["grad_norms"] = np.array(names_g), np.array(norms, dtype=np.float64)
    opt.step()
    for k in ("fd_layer.fp.conv2.weight", "mse_layer.ms_ls.0.mlp_convs.0.weight"):
        tr["after::" + k] = np_(dict(net.named_parameters())[k]).reshape(-1)[:64].copy()
    np.savez_compressed(os.path.join(HERE, "raflow_train_synth_b4.npz"), **tr)
    print("train: loss %.5f" % tr["loss"], items)

    # ---- dataset / sample format ----
    from dataset.vod import vodDataset
    ds_out = {}
    for part, ev_flag in (("train", False), ("test", True)):
        a = Args()
        a.eval = ev_flag
        ds = vodDataset(a, root=tmp + "/", partition=part, textio=Textio())
        np.random.seed(11)
        ds_out["%s/len" % part] = np.int64(len(ds))
        for i in range(len(ds)):
            item = ds[i]
            for j, v in enumerate(item):
                ds_out["%s/%d/%d" % (part, i, j)] = np.asarray(v).copy()
        ds_out["%s/res" % part] = np.array([ds.res['r_res'], ds.res['theta_res'], ds.res['phi_res']])
        ds_out["%s/camera_projection_matrix" % part] = ds.camera_projection_matrix.copy()
        ds_out["%s/t_camera_radar" % part] = ds.t_camera_radar.copy()
        if ev_flag:
            ds_out["%s/clips_info" % part] = np.array(json.dumps(ds.clips_info))
    # the sample files themselves (small): the JSON text the reference read
    for rel in names:
        ds_out["file::" + rel] = np.array(open(os.path.join(tmp, rel)).read())
    np.savez_compressed(os.path.join(HERE, "vod_dataset_kat.npz"), **ds_out)
    print("dataset:", {k: int(ds_out[k]) for k in ds_out if k.endswith("/len")}, "files", len(names))
    shutil.rmtree(tmp)


if __name__ == "__main__":
    main()
