"""The seven cross-modal loss terms and the label prep of the reference's training step.

Mirrors ``losses/radar_loss.py`` (RadarFlowLoss :260-292 and its components :17-258) and the label prep of
``main_util.py`` (:209-225 extract_dynamic_from_fg, :253-265 mseg_label_RRV).

ONE execution path: ``RadarFlowLoss`` runs the fused HIP kernels behind ``cmf_radar_loss`` (csrc/loss.hip: all seven terms and
their gradients w.r.t. the network outputs in one call), ``make_labels`` runs ``cmf_pseudo_labels`` (csrc/eval.hip).  Like the
reference's loss (radar_loss.py:17-97) the call takes any cloud size: up to 704 points with the reference's 8 smoothness
neighbours (it trains at N = 256) a sample lives in one workgroup's LDS (3 launches); larger clouds and num_nb in {4, 16} take
the tiled kernels of the same file (the cloud streamed through LDS, a sample's lists in a workspace; 4 launches) -- same terms,
same per-point arithmetic.  On CPU tensors the call RAISES: there is no torch-op fallback in the product (the torch-op
restatement of the terms is a test fixture, tests/loss_torch.py).  Unlike the reference the loss items stay on the device (no 8
``.item()`` host syncs per step, radar_loss.py:156-159,285-288) until the caller asks.
"""
import ctypes

import torch
from torch.autograd import Function
from torch.nn import Module

from . import _lib


def make_labels(batch, vr_thres):
    """main_util.py:63-67: dyn_mask and the merged pseudo motion-segmentation label -- one launch of
    cmf_pseudo_labels (csrc/eval.hip) on device tensors."""
    pc1 = batch["pc1"]
    if not pc1.is_cuda:
        raise RuntimeError("cmflow_amd.losses.make_labels runs on the GPU only (got %s tensors)" % pc1.device)
    B, _, N = pc1.shape
    f32 = torch.float32
    c = lambda t: t.to(f32).contiguous()
    pc, T, vel, dt = c(pc1), c(batch["gt_trans"]), c(batch["ft1"][:, 0]), c(batch["interval"])
    fg, fl = c(batch["fg_mask"]), c(batch["flow_label"])
    dyn_mask, mseg_gt = torch.empty(B, N, dtype=f32, device=pc1.device), torch.empty(B, N, dtype=f32, device=pc1.device)
    p = lambda t: _lib.dev_ptr(t, f32)
    _lib.check(_lib.lib().cmf_pseudo_labels(B, N, p(pc), p(T), p(vel), p(dt), p(fg), p(fl), vr_thres, p(dyn_mask),
                                            p(mseg_gt), None, _lib.stream_ptr()), "cmf_pseudo_labels")
    return dyn_mask.to(batch["fg_mask"].dtype), mseg_gt



MAX_N = 65536                                # include/cmflow_hip.h CMF_RADAR_LOSS_MAX_N
NUM_NB = (4, 8, 16)                          # csrc/loss.hip: the neighbour counts the kernels are instantiated for
ITEM_KEYS = ('Loss', 'smoothnessLoss', 'chamferLoss', 'veloLoss', 'egoLoss', 'maskLoss', 'opticalLoss', 'superviseLoss')


class RadarFlowLossFn(Function):
    """total, items = cmf_radar_loss(...); the kernel writes d total / d (pred_f, pre_trans, mseg_pre) in the same
    pass, backward only scales them by the incoming gradient."""

    @staticmethod
    def forward(ctx, pred_f, pre_trans, mseg_pre, data, hyper):
        pc1 = data["pc1"]
        B, _, N = pc1.shape
        dev = pc1.device
        f32 = torch.float32
        t = {k: v.contiguous() for k, v in data.items()}
        pred_f = pred_f.contiguous()
        pre_trans = pre_trans.contiguous() if pre_trans is not None else None
        mseg_pre = mseg_pre.contiguous() if mseg_pre is not None else None
        for v in list(t.values()) + [pred_f, pre_trans, mseg_pre]:
            _lib.dev_ptr(v, f32)                                  # device / dtype / density checks (no CPU fallback)
        need = any(ctx.needs_input_grad[:3])
        d = _lib.RadarLossDesc()
        d.B, d.N = B, N
        d.self_only = int(hyper.get("self_only", False))
        d.pc1, d.pc2, d.pred_f, d.vel1 = t["pc1"].data_ptr(), t["pc2"].data_ptr(), pred_f.data_ptr(), t["vel1"].data_ptr()
        if not d.self_only:
            d.gt_f, d.mseg_pre, d.mseg_gt = t["gt_f"].data_ptr(), mseg_pre.data_ptr(), t["mseg_gt"].data_ptr()
            d.dyn_mask, d.radar_u, d.radar_v = t["dyn_mask"].data_ptr(), t["radar_u"].data_ptr(), t["radar_v"].data_ptr()
            d.opt, d.pre_trans, d.gt_trans = t["opt"].data_ptr(), pre_trans.data_ptr(), t["gt_trans"].data_ptr()
            d.camera_inverse, d.t_camera_radar = t["camera_inverse"].data_ptr(), t["t_camera_radar"].data_ptr()
        d.w_self, d.w_em, d.w_ms, d.w_opt, d.w_dyn = hyper["w"]
        d.zeta, d.alpha, d.num_nb, d.lower_bound = hyper["zeta"], hyper["alpha"], hyper["num_nb"], hyper["lower_bound"]
        items = torch.empty(9, dtype=f32, device=dev)
        tiled = bool(hyper.get("tiled", False))              # tests: the tiled kernels at a size the LDS kernel would take
        size = _lib.lib().cmf_radar_loss_workspace_tiled if tiled else _lib.lib().cmf_radar_loss_workspace_nb
        ws = torch.empty(size(B, N, int(hyper["num_nb"])), dtype=f32, device=dev)
        d.items, d.workspace = items.data_ptr(), ws.data_ptr()
        if need:
            g_f = torch.empty(B, 3, N, dtype=f32, device=dev)
            g_t = torch.empty(B, 4, 4, dtype=f32, device=dev) if pre_trans is not None else None
            g_m = torch.empty(mseg_pre.shape, dtype=f32, device=dev) if mseg_pre is not None else None
            d.d_pred_f = g_f.data_ptr()
            d.d_pre_trans = g_t.data_ptr() if g_t is not None else None
            d.d_mseg_pre = g_m.data_ptr() if g_m is not None else None
            ctx.grads = (g_f, g_t, g_m)
        call = _lib.lib().cmf_radar_loss_tiled if tiled else _lib.lib().cmf_radar_loss
        _lib.check(call(ctypes.addressof(d), _lib.stream_ptr()), "cmf_radar_loss")
        ctx.mark_non_differentiable(items)
        return items[0], items

    @staticmethod
    def backward(ctx, g_total, _g_items):
        g_f, g_t, g_m = ctx.grads
        need = ctx.needs_input_grad
        live = [i for i, (g, n) in enumerate(((g_f, need[0]), (g_t, need[1]), (g_m, need[2]))) if n and g is not None]
        # out of place (one launch for the three): the saved buffers stay what the kernel wrote, so a second backward through
        # the node (retain_graph=True) scales them by ITS incoming gradient, not by the product of both
        scaled = torch._foreach_mul([(g_f, g_t, g_m)[i] for i in live], g_total)
        out = [None, None, None]
        for i, g in zip(live, scaled):
            out[i] = g
        return out[0], out[1], out[2], None, None


SELF_ITEM_KEYS = ITEM_KEYS[:4]


class RadarFlowLoss(Module):
    """radar_loss.py:260-292 for model in {'cmflow','cmflow_t'}; weights (1,1,1,0.1,1) (:262); hyper-parameters of the terms as the
    reference constructs them (zeta = 0.005 :20, alpha = 0.5 / num_nb = 8 :63, lower_bound = 0.25 :210).
    Returns (total_loss, items) with items as 0-d device tensors (call .item() when needed)."""

    tiled = False                # tests: force the tiled kernels (cmf_radar_loss_tiled) at any size

    def __init__(self, camera_projection, t_camera_radar, w_self=1, w_em=1, w_ms=1, w_opt=0.1, w_dyn=1,
                 zeta=0.005, alpha=0.5, num_nb=8, lower_bound=0.25):
        super().__init__()
        self.w_self, self.w_em, self.w_ms, self.w_opt, self.w_dyn = w_self, w_em, w_ms, w_opt, w_dyn
        self.zeta, self.alpha, self.num_nb, self.lower_bound = zeta, alpha, num_nb, lower_bound
        self.register_buffer("camera_projection", torch.as_tensor(camera_projection, dtype=torch.float32))
        self.register_buffer("camera_inverse", torch.inverse(self.camera_projection[:3, :3].cpu()))
        self.register_buffer("t_camera_radar", torch.as_tensor(t_camera_radar, dtype=torch.float32))

    def _check(self, pc1):
        N = pc1.shape[2]
        if not pc1.is_cuda:
            raise RuntimeError("cmflow_amd.losses.RadarFlowLoss runs on the GPU only (got %s tensors)" % pc1.device)
        if self.num_nb not in NUM_NB or not (self.num_nb < N <= MAX_N):
            raise RuntimeError("cmf_radar_loss: num_nb in %s and num_nb < N <= %d points (got N = %d, num_nb = %d)"
                               % (NUM_NB, MAX_N, N, self.num_nb))

    def forward(self, pc1, pc2, pred_f, vel1, gt_f=None, pre_trans=None, mseg_pre=None, gt_trans=None, mseg_gt=None,
                dyn_mask=None, radar_u=None, radar_v=None, opt=None):
        """With only (pc1, pc2, pred_f, vel1) this is the loss of model 'raflow' (radar_loss.py:274-276): the three
        self-supervised terms; items then has the four keys of SelfSupervisedLoss (:153-158)."""
        self._check(pc1)
        if gt_f is None:
            hyper = dict(w=(self.w_self, 0.0, 0.0, 0.0, 0.0), zeta=self.zeta, alpha=self.alpha, num_nb=self.num_nb,
                         lower_bound=0.0, self_only=True, tiled=self.tiled)
            total, items = RadarFlowLossFn.apply(pred_f, None, None, dict(pc1=pc1, pc2=pc2, vel1=vel1), hyper)
            return total, {k: items[i + 1] for i, k in enumerate(SELF_ITEM_KEYS)}
        data = dict(pc1=pc1, pc2=pc2, gt_f=gt_f, vel1=vel1, gt_trans=gt_trans, mseg_gt=mseg_gt.to(pc1.dtype),
                    dyn_mask=dyn_mask.to(pc1.dtype), radar_u=radar_u, radar_v=radar_v, opt=opt,
                    camera_inverse=self.camera_inverse, t_camera_radar=self.t_camera_radar)
        hyper = dict(w=(self.w_self, self.w_em, self.w_ms, self.w_opt, self.w_dyn), zeta=self.zeta, alpha=self.alpha,
                     num_nb=self.num_nb, lower_bound=self.lower_bound, tiled=self.tiled)
        total, items = RadarFlowLossFn.apply(pred_f, pre_trans, mseg_pre, data, hyper)
        return total, {k: items[i + 1] for i, k in enumerate(ITEM_KEYS)}
