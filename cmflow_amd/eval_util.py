"""Evaluation metrics on the device -- mirror of ``utils/eval_util.py`` (eval_scene_flow :42-86,
eval_trans_RPE :89-102, eval_motion_seg :104-118; same names, arguments and metric keys).

The reference moves every tensor to the host and evaluates with numpy/scipy; here ``cmf_eval_metrics``
(csrc/eval.hip) reduces a batch in two launches and the results stay on the device as 0-d float64 tensors
(``float(x)`` / ``.item()`` when a number is needed; accumulating ``batch_size * metric`` like
main_util.py:176-192 works on the tensors directly).  ``eval_batch`` returns all three groups from ONE call.
"""
import numpy as np
import torch

from . import _lib

SF_KEYS = ('rne', '50-50 rne', 'mov_rne', 'stat_rne', 'sas', 'ras', 'epe', 'accs', 'accr')
SEG_KEYS = ('acc', 'miou', 'sen')
POSE_KEYS = ('RTE', 'RAE')
VOD_RADAR_RES = {'r_res': 0.2, 'theta_res': 1.5 * np.pi / 180, 'phi_res': 1.5 * np.pi / 180}     # dataset/vod.py:21-23


def _f32(t):
    return None if t is None else t.detach().to(torch.float32).contiguous()


def _metrics(B, N, dev, pc=None, pred=None, labels=None, mask=None, pred_m=None, gt_trans=None, pred_trans=None, res=None):
    f32 = torch.float32
    pc, pred, labels, mask, pred_m, gt_trans, pred_trans = map(_f32, (pc, pred, labels, mask, pred_m, gt_trans, pred_trans))
    out = torch.empty(14, dtype=torch.float64, device=dev)
    ws = torch.empty(16 * B, dtype=torch.float64, device=dev)
    res = res or VOD_RADAR_RES
    ptr = lambda t: _lib.dev_ptr(t, f32)
    _lib.check(_lib.lib().cmf_eval_metrics(B, N, ptr(pc), ptr(pred), ptr(labels), ptr(mask), ptr(pred_m), ptr(gt_trans),
                                           ptr(pred_trans), res['r_res'], res['theta_res'], res['phi_res'],
                                           out.data_ptr(), ws.data_ptr(), _lib.stream_ptr()), "cmf_eval_metrics")
    return out


def eval_scene_flow(pc, pred, labels, mask, args):
    """pc (B,3,N) (the layout main_util.py:175 passes); pred, labels (B,N,3); mask (B,N), 1 = static;
    args.radar_res as dataset/vod.py:21-23."""
    B, _, N = pc.shape
    m = _metrics(B, N, pc.device, pc=pc, pred=pred, labels=labels, mask=mask, res=getattr(args, "radar_res", None))
    return {k: m[i] for i, k in enumerate(SF_KEYS)}


def eval_motion_seg(pre, gt):
    B, N = gt.shape
    m = _metrics(B, N, gt.device, mask=gt, pred_m=pre)
    return {k: m[9 + i] for i, k in enumerate(SEG_KEYS)}


def eval_trans_RPE(gt_trans, rigid_trans):
    B = gt_trans.shape[0]
    m = _metrics(B, 1, gt_trans.device, gt_trans=gt_trans, pred_trans=rigid_trans)
    return {k: m[12 + i] for i, k in enumerate(POSE_KEYS)}


def eval_batch(pc, pred, labels, mask, pred_m, gt_trans, pred_trans, args=None):
    """All three groups of main_util.py:175-192 from one kernel call -> (sf_metric, seg_metric, pose_metric)."""
    B, _, N = pc.shape
    m = _metrics(B, N, pc.device, pc, pred, labels, mask, pred_m, gt_trans, pred_trans, getattr(args, "radar_res", None))
    return ({k: m[i] for i, k in enumerate(SF_KEYS)}, {k: m[9 + i] for i, k in enumerate(SEG_KEYS)},
            {k: m[12 + i] for i, k in enumerate(POSE_KEYS)})
