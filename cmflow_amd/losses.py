"""The seven cross-modal loss terms and the label prep of the reference's training step.

Mirrors ``losses/radar_loss.py`` (RadarFlowLoss :260-292 and its components) and the label
prep of ``main_util.py`` (:209-225 extract_dynamic_from_fg, :253-265 mseg_label_RRV).

``RadarFlowLoss`` runs the fused HIP kernel ``cmf_radar_loss`` (csrc/loss.hip: all seven terms and
their gradients w.r.t. the network outputs in one call, 3 launches) whenever 9 <= N <= 704; the
component modules below are the same terms as torch ops on device tensors (the reference's own
structure; used for larger clouds and as the second implementation the tests compare against).
Unlike the reference the loss items stay on the device (no 8 ``.item()`` host syncs per step,
radar_loss.py:156-159,285-288) until the caller asks for them.
"""
import ctypes

import torch
import torch.nn.functional as F
from torch.autograd import Function
from torch.nn import Module

from . import _lib
from .radarflow_util import index_points_group, square_distance
from .cmflow import CMFlow

rigid_to_flow = CMFlow.rigid_to_flow


def extract_dynamic_from_fg(mask, pc1, trans, gt):
    """main_util.py:209-225.  mask (B,N): 1 = background; gt (B,3,N).  -> (B,N), 1 = static."""
    flow_nr = rigid_to_flow(pc1, trans).transpose(2, 1) - gt.transpose(2, 1)
    fg_mask = (mask != 1)
    static = torch.norm(flow_nr * fg_mask.unsqueeze(2), dim=2) < 0.05
    return ((mask == 1) | static).to(mask.dtype)


def mseg_label_RRV(pc1, trans, vel1, interval, vr_thres):
    """main_util.py:253-265: 1 = static, 0 = moving."""
    gt_sf_rg = rigid_to_flow(pc1, trans)
    proj = torch.sum(gt_sf_rg * pc1, dim=1) / torch.norm(pc1, dim=1)
    residual = torch.abs(vel1 - proj / interval.unsqueeze(1))
    bs_residual = torch.mean(residual, dim=1).unsqueeze(1)
    return ((residual - bs_residual) < vr_thres).to(pc1.dtype), residual


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


def make_labels_torch(batch, vr_thres):
    """The same label prep as torch ops (the reference's own structure)."""
    pc1 = batch["pc1"]
    dyn_mask = extract_dynamic_from_fg(batch["fg_mask"], pc1, batch["gt_trans"], batch["flow_label"].transpose(2, 1))
    mseg_gt, _ = mseg_label_RRV(pc1, batch["gt_trans"], batch["ft1"][:, 0], batch["interval"], vr_thres)
    mseg_gt = torch.where(dyn_mask == 1, mseg_gt, dyn_mask)
    return dyn_mask, mseg_gt


def compute_density_loss(xyz1, xyz2, bandwidth):
    """utils/util.py:172-182"""
    sqrdists = square_distance(xyz1, xyz2)
    return (torch.exp(-sqrdists / (2.0 * bandwidth * bandwidth)) / (2.5 * bandwidth)).mean(dim=-1)


class SoftChamferLoss(Module):
    """radar_loss.py:17-58"""

    def __init__(self, zeta=0.005):
        super().__init__()
        self.zeta = zeta

    def forward(self, pc1, pc2, pc1_warp):
        pc1, pc2, pc1_warp = pc1.permute(0, 2, 1), pc2.permute(0, 2, 1), pc1_warp.permute(0, 2, 1)
        mask1 = (compute_density_loss(pc1, pc2, 1) > self.zeta).type(torch.int32)
        mask2 = (compute_density_loss(pc2, pc1, 1) > self.zeta).type(torch.int32)
        sqrdist12w = square_distance(pc1_warp, pc2)
        dist1_w = F.relu(torch.min(sqrdist12w, dim=-1)[0] - 0.01) * mask1
        dist2_w = F.relu(torch.min(sqrdist12w, dim=1)[0] - 0.01) * mask2
        return torch.mean(dist1_w) + torch.mean(dist2_w)


class SpatialSmoothnessLoss(Module):
    """radar_loss.py:60-97"""

    def __init__(self, alpha=0.5, num_nb=8):
        super().__init__()
        self.alpha, self.num_nb = alpha, num_nb

    def forward(self, pc1, pred_flow):
        B, _, N = pc1.shape
        pc1 = pc1.permute(0, 2, 1)
        pred_flow = pred_flow.permute(0, 2, 1)
        sqrdist = square_distance(pc1, pc1)
        dists, kidx = torch.topk(sqrdist, self.num_nb + 1, dim=-1, largest=False, sorted=True)
        dists, kidx = torch.clamp_min(dists[:, :, 1:], 0.0), kidx[:, :, 1:]
        weights = torch.softmax(torch.exp(-dists / self.alpha).view(B, N * self.num_nb), dim=1).view(B, N, self.num_nb)
        grouped_flow = index_points_group(pred_flow, kidx)
        diff_flow = (N * weights * torch.norm(grouped_flow - pred_flow.unsqueeze(2), dim=3)).sum(dim=2)
        return torch.mean(diff_flow)


class RadialDisplacementLoss(Module):
    """radar_loss.py:99-122 (interval is hard-coded to 0.1 at :103)"""

    def forward(self, pc1, pred_f, vel1):
        pred_fr = torch.sum(pred_f * pc1, dim=1) / torch.norm(pc1, dim=1)
        return torch.mean(torch.abs(vel1 * 0.1 - pred_fr))


class EgoMotionLoss(Module):
    """radar_loss.py:162-183"""

    def forward(self, pc1, pre_trans, gt_trans):
        pc1_pre = torch.matmul(pre_trans[:, :3, :3], pc1) + pre_trans[:, :3, 3].unsqueeze(2)
        pc1_gt = torch.matmul(gt_trans[:, :3, :3], pc1) + gt_trans[:, :3, 3].unsqueeze(2)
        return torch.mean(torch.norm(pc1_pre - pc1_gt, dim=1))


class MotionSegLoss(Module):
    """radar_loss.py:185-205: BCE averaged separately over the two classes.  Written with masked
    means instead of boolean indexing (no host sync); equal whenever both classes are present."""

    def forward(self, mseg_pre, mseg_gt):
        p = mseg_pre.squeeze(1)
        bce = F.binary_cross_entropy(p, mseg_gt, reduction="none")
        m0, m1 = (mseg_gt == 0).to(bce.dtype), (mseg_gt == 1).to(bce.dtype)
        return ((bce * m0).sum() / m0.sum() + (bce * m1).sum() / m1.sum()) / 2


def point_ray_distance(warped_pcs, pixels, camera_inverse, t_camera_radar):
    """utils/util.py:31-58.  The reference inverts the constant 3x3 intrinsics on every call (:41); on a
    GPU that is a blocking solver call (14 ms per step measured), so the inverse is computed once."""
    B, _, N = warped_pcs.shape
    one = torch.ones((B, N, 1), dtype=pixels.dtype, device=pixels.device)
    pixels_h = torch.cat((pixels, one), dim=2).transpose(2, 1)
    cam_pcs = camera_inverse.unsqueeze(0) @ pixels_h
    unit_vector = cam_pcs / torch.norm(cam_pcs, dim=1).unsqueeze(1)
    warped_h = torch.cat((warped_pcs, one.transpose(2, 1)), dim=1)
    warped_cam = t_camera_radar.unsqueeze(0) @ warped_h
    return torch.norm(torch.linalg.cross(unit_vector, warped_cam[:, :3], dim=1), dim=1)


class OpticalFlowLoss(Module):
    """radar_loss.py:207-243"""
    lower_bound = 0.25

    def forward(self, opt, radar_u, radar_v, pc1_warp, mseg_gt, camera_inverse, t_camera_radar):
        end_pixels = torch.cat((radar_u.unsqueeze(2), radar_v.unsqueeze(2)), dim=2) + opt
        opt_div = F.relu(point_ray_distance(pc1_warp, end_pixels, camera_inverse, t_camera_radar) - self.lower_bound)
        m = mseg_gt.to(opt_div.dtype).detach()
        return torch.sum((1 - m) * opt_div) / torch.clamp_min(torch.sum(1 - m), 1.0)


class DynamicFlowLoss(Module):
    """radar_loss.py:245-258"""

    def forward(self, pred_f, gt_f, dyn_mask):
        return torch.sum((1 - dyn_mask) * torch.norm(gt_f - pred_f, dim=1)) / torch.clamp_min(torch.sum(1 - dyn_mask), 1.0)


NATIVE_MIN_N, NATIVE_MAX_N = 9, 704          # csrc/loss.hip: a sample's working set lives in LDS
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
        ws = torch.empty(_lib.lib().cmf_radar_loss_workspace(B, N), dtype=f32, device=dev)
        d.items, d.workspace = items.data_ptr(), ws.data_ptr()
        if need:
            g_f = torch.empty(B, 3, N, dtype=f32, device=dev)
            g_t = torch.empty(B, 4, 4, dtype=f32, device=dev) if pre_trans is not None else None
            g_m = torch.empty(mseg_pre.shape, dtype=f32, device=dev) if mseg_pre is not None else None
            d.d_pred_f = g_f.data_ptr()
            d.d_pre_trans = g_t.data_ptr() if g_t is not None else None
            d.d_mseg_pre = g_m.data_ptr() if g_m is not None else None
            ctx.grads = (g_f, g_t, g_m)
        _lib.check(_lib.lib().cmf_radar_loss(ctypes.addressof(d), _lib.stream_ptr()), "cmf_radar_loss")
        ctx.mark_non_differentiable(items)
        return items[0], items

    @staticmethod
    def backward(ctx, g_total, _g_items):
        g_f, g_t, g_m = ctx.grads
        need = ctx.needs_input_grad
        return (g_f * g_total if need[0] else None, g_t * g_total if (need[1] and g_t is not None) else None,
                g_m * g_total if (need[2] and g_m is not None) else None, None, None)


SELF_ITEM_KEYS = ITEM_KEYS[:4]


class RadarFlowLoss(Module):
    """radar_loss.py:260-292 for model in {'cmflow','cmflow_t'}; weights (1,1,1,0.1,1) (:262).
    Returns (total_loss, items) with items as 0-d device tensors (call .item() when needed).
    native=True (default): the fused HIP kernel when the cloud size allows; native=False: the torch-op terms."""

    def __init__(self, camera_projection, t_camera_radar, w_self=1, w_em=1, w_ms=1, w_opt=0.1, w_dyn=1, native=True):
        super().__init__()
        self.native = native
        self.w_self, self.w_em, self.w_ms, self.w_opt, self.w_dyn = w_self, w_em, w_ms, w_opt, w_dyn
        self.register_buffer("camera_projection", torch.as_tensor(camera_projection, dtype=torch.float32))
        self.register_buffer("camera_inverse", torch.inverse(self.camera_projection[:3, :3].cpu()))
        self.register_buffer("t_camera_radar", torch.as_tensor(t_camera_radar, dtype=torch.float32))
        self.sc_loss, self.ss_loss, self.rd_loss = SoftChamferLoss(), SpatialSmoothnessLoss(), RadialDisplacementLoss()
        self.ego_motion_loss, self.motion_seg_loss = EgoMotionLoss(), MotionSegLoss()
        self.opt_flow_loss, self.dyn_flow_loss = OpticalFlowLoss(), DynamicFlowLoss()

    def _self_terms(self, pc1, pc2, pred_f, vel1):
        sc = self.sc_loss(pc1, pc2, pc1 + pred_f)
        ss = self.ss_loss(pc1, pred_f)
        rd = self.rd_loss(pc1, pred_f, vel1)
        return sc + ss + rd, sc, ss, rd

    def _self_supervised(self, pc1, pc2, pred_f, vel1):
        N = pc1.shape[2]
        if self.native and pc1.is_cuda and NATIVE_MIN_N <= N <= NATIVE_MAX_N and self.ss_loss.num_nb == 8:
            hyper = dict(w=(self.w_self, 0.0, 0.0, 0.0, 0.0), zeta=self.sc_loss.zeta, alpha=self.ss_loss.alpha,
                         num_nb=self.ss_loss.num_nb, lower_bound=0.0, self_only=True)
            total, items = RadarFlowLossFn.apply(pred_f, None, None, dict(pc1=pc1, pc2=pc2, vel1=vel1), hyper)
            return total, {k: items[i + 1] for i, k in enumerate(SELF_ITEM_KEYS)}
        self_sup, sc, ss, rd = self._self_terms(pc1, pc2, pred_f, vel1)
        return self.w_self * self_sup, {'Loss': self_sup.detach(), 'smoothnessLoss': ss.detach(),
                                        'chamferLoss': sc.detach(), 'veloLoss': rd.detach()}

    def forward(self, pc1, pc2, pred_f, vel1, gt_f=None, pre_trans=None, mseg_pre=None, gt_trans=None, mseg_gt=None,
                dyn_mask=None, radar_u=None, radar_v=None, opt=None):
        """With only (pc1, pc2, pred_f, vel1) this is the loss of model 'raflow' (radar_loss.py:274-276): the three
        self-supervised terms; items then has the four keys of SelfSupervisedLoss (:153-158)."""
        N = pc1.shape[2]
        if gt_f is None:
            return self._self_supervised(pc1, pc2, pred_f, vel1)
        if self.native and pc1.is_cuda and NATIVE_MIN_N <= N <= NATIVE_MAX_N:
            data = dict(pc1=pc1, pc2=pc2, gt_f=gt_f, vel1=vel1, gt_trans=gt_trans, mseg_gt=mseg_gt.to(pc1.dtype),
                        dyn_mask=dyn_mask.to(pc1.dtype), radar_u=radar_u, radar_v=radar_v, opt=opt,
                        camera_inverse=self.camera_inverse, t_camera_radar=self.t_camera_radar)
            hyper = dict(w=(self.w_self, self.w_em, self.w_ms, self.w_opt, self.w_dyn), zeta=self.sc_loss.zeta,
                         alpha=self.ss_loss.alpha, num_nb=self.ss_loss.num_nb, lower_bound=self.opt_flow_loss.lower_bound)
            if self.ss_loss.num_nb == 8:
                total, items = RadarFlowLossFn.apply(pred_f, pre_trans, mseg_pre, data, hyper)
                return total, {k: items[i + 1] for i, k in enumerate(ITEM_KEYS)}
        self_sup, sc, ss, rd = self._self_terms(pc1, pc2, pred_f, vel1)
        pc1_warp = pc1 + pred_f
        em = self.ego_motion_loss(pc1, pre_trans, gt_trans)
        ms = self.motion_seg_loss(mseg_pre, mseg_gt)
        dyn = self.dyn_flow_loss(pred_f, gt_f, dyn_mask)
        of = self.opt_flow_loss(opt, radar_u, radar_v, pc1_warp, mseg_gt, self.camera_inverse, self.t_camera_radar)
        total = self.w_self * self_sup + self.w_em * em + self.w_ms * ms + self.w_opt * of + self.w_dyn * dyn
        items = {'Loss': self_sup.detach(), 'smoothnessLoss': ss.detach(), 'chamferLoss': sc.detach(),
                 'veloLoss': rd.detach(), 'egoLoss': em.detach(), 'maskLoss': ms.detach(),
                 'opticalLoss': of.detach(), 'superviseLoss': dyn.detach()}
        return total, items
