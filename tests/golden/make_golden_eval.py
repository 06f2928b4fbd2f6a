#!/usr/bin/env python3
"""Golden vectors for the evaluation metrics, produced by the REFERENCE's own
utils/eval_util.py (eval_scene_flow :42-86, eval_trans_RPE :89-102, eval_motion_seg :104-118; with
utils/odometry_util.py) on inputs taken from the committed training golden.  Build container only.

    python tests/golden/make_golden_eval.py      ->  tests/golden/eval_metrics_kat.npz
"""
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import install_shims  # noqa: E402


class Args:
    radar_res = {'r_res': 0.2, 'theta_res': 1.5 * np.pi / 180, 'phi_res': 1.5 * np.pi / 180}   # dataset/vod.py:21-23


def main():
    install_shims()
    from utils import eval_util as E                       # the reference
    g = np.load(os.path.join(HERE, "cmflow_train_synth_b4.npz"))
    rng = np.random.default_rng(5)
    pc1 = g["pc1"].astype(np.float32)                      # (B,3,N) as main_util.py:175 passes it
    gt = g["flow_label"].astype(np.float32)                # (B,N,3)
    cases = {}

    def add(name, pc, pred, lab, mask, pred_m, trans, pred_t):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            sf = E.eval_scene_flow(torch.from_numpy(pc), torch.from_numpy(pred), torch.from_numpy(lab),
                                   torch.from_numpy(mask), Args)
            seg = E.eval_motion_seg(torch.from_numpy(pred_m), torch.from_numpy(mask))
            pose = E.eval_trans_RPE(torch.from_numpy(trans), torch.from_numpy(pred_t))
        for k, v in dict(pc=pc, pred=pred, labels=lab, mask=mask, pred_m=pred_m, trans=trans, pred_t=pred_t).items():
            cases["%s/in/%s" % (name, k)] = v
        for k, v in {**sf, **seg, **pose}.items():
            cases["%s/out/%s" % (name, k)] = np.float64(v)

    pred = np.ascontiguousarray(g["sf_agg"].transpose(0, 2, 1)).astype(np.float32)     # (B,N,3)
    mask = g["fg_mask"].astype(np.float32)
    pred_m = g["mask"].astype(np.float32)
    add("train_b4", pc1, pred, gt, mask, pred_m, g["gt_trans"].astype(np.float32), g["pre_trans"].astype(np.float32))
    for b in range(2):
        add("single_%d" % b, pc1[b:b + 1], pred[b:b + 1], gt[b:b + 1], mask[b:b + 1], pred_m[b:b + 1],
            g["gt_trans"][b:b + 1].astype(np.float32), g["pre_trans"][b:b + 1].astype(np.float32))
    noisy = (gt + rng.normal(0, 0.08, gt.shape)).astype(np.float32)
    rmask = (rng.random(mask.shape) < 0.7).astype(np.float32)
    rpm = (rng.random(mask.shape) < 0.6).astype(np.float32)
    # a larger ego-motion error: yaw 3 deg, pitch -1 deg, translation offset
    def rot(yaw, pitch):
        cy, sy, cp, sp = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch)
        return np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]]) @ np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    pt = g["gt_trans"].astype(np.float64).copy()
    for b in range(pt.shape[0]):
        d = np.eye(4); d[:3, :3] = rot(np.deg2rad(3.0 * (b + 1) / 4), np.deg2rad(-1.0)); d[:3, 3] = [0.3, -0.1 * b, 0.02]
        pt[b] = pt[b] @ d
    add("noisy", pc1, noisy, gt, rmask, rpm, g["gt_trans"].astype(np.float32), pt.astype(np.float32))
    add("all_static", pc1, noisy, gt, np.ones_like(mask), rpm, g["gt_trans"].astype(np.float32), pt.astype(np.float32))
    add("exact", pc1, gt.copy(), gt, rmask, rmask.copy(), g["gt_trans"].astype(np.float32), g["gt_trans"].astype(np.float32))
    np.savez_compressed(os.path.join(HERE, "eval_metrics_kat.npz"), **cases)
    for k in sorted(cases):
        if "/out/" in k:
            print(k, cases[k])


if __name__ == "__main__":
    main()
