"""torch-CPU restatement of the reference's training-step sequence -- TEST INFRASTRUCTURE ONLY.

Label prep (main_util.py:209-225, 253-265), the seven loss terms (losses/radar_loss.py:17-292
with utils/util.py:31-58,148-182) and the step order of main_util.py:63-76.  Checked against
tests/golden/cmflow_train_synth_b4.npz (loss, items, gradient norms produced by the
reference's own code) in tests/test_oracle.py.
"""
import torch
import torch.nn.functional as F

from .cmflow_oracle import index_points_group, rigid_to_flow


def square_distance(src, dst):
    """utils/util.py:148-169 (torch form; used by the losses)."""
    B, N, _ = src.shape
    M = dst.shape[1]
    dist = -2 * torch.matmul(src, dst.permute(0, 2, 1))
    dist = dist + torch.sum(src ** 2, -1).view(B, N, 1)
    dist = dist + torch.sum(dst ** 2, -1).view(B, 1, M)
    return torch.clamp_min(dist, 0.0)


def extract_dynamic_from_fg(mask, pc1, trans, gt):
    """main_util.py:209-225.  mask (B,N) 1=background; gt (B,3,N) flow labels.  -> (B,N) 1=static."""
    flow_nr = rigid_to_flow(pc1, trans).transpose(2, 1) - gt.transpose(2, 1)
    fg = (mask != 1)
    mask = mask.clone()
    mask[torch.norm(flow_nr * fg.unsqueeze(2), dim=2) < 0.05] = 1
    mask[mask != 1] = 0
    return mask


def mseg_label_RRV(pc1, trans, vel1, interval, vr_thres):
    """main_util.py:253-265: 1 = static, 0 = moving, from the radial-velocity residual."""
    rg = rigid_to_flow(pc1, trans)
    proj = torch.sum(rg * pc1, dim=1) / torch.norm(pc1, dim=1)
    residual = torch.abs(vel1 - proj / interval.unsqueeze(1))
    bs = torch.mean(residual, dim=1).unsqueeze(1)
    return ((residual - bs) < vr_thres).to(pc1.dtype)


def make_labels(batch, vr_thres=0.3):
    """main_util.py:63-67"""
    pc1 = batch["pc1"]
    dyn = extract_dynamic_from_fg(batch["fg_mask"], pc1, batch["gt_trans"], batch["flow_label"].transpose(2, 1))
    mseg = mseg_label_RRV(pc1, batch["gt_trans"], batch["ft1"][:, 0], batch["interval"], vr_thres)
    sel = torch.logical_not(dyn == 1)
    mseg[sel] = dyn[sel]
    return dyn, mseg


def density(xyz1, xyz2, bandwidth=1.0):
    """utils/util.py:172-182"""
    d = square_distance(xyz1, xyz2)
    return (torch.exp(-d / (2.0 * bandwidth * bandwidth)) / (2.5 * bandwidth)).mean(dim=-1)


def soft_chamfer(pc1, pc2, pc1_warp, zeta=0.005):
    """losses/radar_loss.py:17-58"""
    pc1, pc2, pc1_warp = pc1.permute(0, 2, 1), pc2.permute(0, 2, 1), pc1_warp.permute(0, 2, 1)
    mask1 = (density(pc1, pc2) > zeta).int()
    mask2 = (density(pc2, pc1) > zeta).int()
    d = square_distance(pc1_warp, pc2)
    d1 = F.relu(torch.min(d, dim=-1)[0] - 0.01) * mask1
    d2 = F.relu(torch.min(d, dim=1)[0] - 0.01) * mask2
    return torch.mean(d1) + torch.mean(d2)


def smoothness(pc1, pred_flow, alpha=0.5, num_nb=8):
    """losses/radar_loss.py:60-97"""
    B, _, N = pc1.shape
    pc1 = pc1.permute(0, 2, 1)
    flow = pred_flow.permute(0, 2, 1)
    d = square_distance(pc1, pc1)
    dists, kidx = torch.topk(d, num_nb + 1, dim=-1, largest=False, sorted=True)
    dists, kidx = torch.clamp_min(dists[:, :, 1:], 0.0), kidx[:, :, 1:]
    w = torch.softmax(torch.exp(-dists / alpha).view(B, N * num_nb), dim=1).view(B, N, num_nb)
    grouped = index_points_group(flow, kidx)
    diff = (N * w * torch.norm(grouped - flow.unsqueeze(2), dim=3)).sum(dim=2)
    return torch.mean(diff)


def radial_displacement(pc1, pred_f, vel1, interval=0.1):
    """losses/radar_loss.py:99-122 (interval hard-coded 0.1 at :103)"""
    fr = torch.sum(pred_f * pc1, dim=1) / torch.norm(pc1, dim=1)
    return torch.mean(torch.abs(vel1 * interval - fr))


def ego_motion(pc1, pre_trans, gt_trans):
    """losses/radar_loss.py:162-183"""
    a = torch.matmul(pre_trans[:, :3, :3], pc1) + pre_trans[:, :3, 3].unsqueeze(2)
    b = torch.matmul(gt_trans[:, :3, :3], pc1) + gt_trans[:, :3, 3].unsqueeze(2)
    return torch.mean(torch.norm(a - b, dim=1))


def motion_seg(mseg_pre, mseg_gt):
    """losses/radar_loss.py:185-205: class-balanced BCE"""
    p = mseg_pre.squeeze(1)
    l0 = F.binary_cross_entropy(p[mseg_gt == 0], mseg_gt[mseg_gt == 0])
    l1 = F.binary_cross_entropy(p[mseg_gt == 1], mseg_gt[mseg_gt == 1])
    return (l0 + l1) / 2


def point_ray_distance(warped, pixels, P, Tcr):
    """utils/util.py:31-58"""
    B, _, N = warped.shape
    ph = torch.cat((pixels, torch.ones(B, N, 1, dtype=pixels.dtype)), dim=2).transpose(2, 1)
    cam = torch.inverse(P[:3, :3].unsqueeze(0)) @ ph
    unit = cam / torch.norm(cam, dim=1).unsqueeze(1)
    wc = Tcr.unsqueeze(0) @ torch.cat((warped, torch.ones(B, 1, N, dtype=warped.dtype)), dim=1)
    return torch.norm(torch.linalg.cross(unit, wc[:, :3], dim=1), dim=1)


def optical_flow(opt, radar_u, radar_v, pc1_warp, mseg_gt, P, Tcr, lower=0.25):
    """losses/radar_loss.py:207-243"""
    end = torch.cat((radar_u.unsqueeze(2), radar_v.unsqueeze(2)), dim=2) + opt
    div = F.relu(point_ray_distance(pc1_warp, end, P, Tcr) - lower)
    m = mseg_gt.to(div.dtype).detach()
    return torch.sum((1 - m) * div) / torch.clamp_min(torch.sum(1 - m), 1.0)


def dynamic_flow(pred_f, gt_f, dyn_mask):
    """losses/radar_loss.py:245-258"""
    return torch.sum((1 - dyn_mask) * torch.norm(gt_f - pred_f, dim=1)) / torch.clamp_min(torch.sum(1 - dyn_mask), 1.0)


def radar_flow_loss(batch, pred_f, pre_trans, mseg_pre, mseg_gt, dyn_mask, P, Tcr):
    """losses/radar_loss.py:260-292, weights (1,1,1,0.1,1) (:262)."""
    pc1, pc2 = batch["pc1"], batch["pc2"]
    vel1 = batch["ft1"][:, 0]
    warp = pc1 + pred_f
    sc = soft_chamfer(pc1, pc2, warp)
    ss = smoothness(pc1, pred_f)
    rd = radial_displacement(pc1, pred_f, vel1)
    self_sup = sc + ss + rd
    em = ego_motion(pc1, pre_trans, batch["gt_trans"])
    ms = motion_seg(mseg_pre, mseg_gt)
    dyn = dynamic_flow(pred_f, batch["flow_label"].transpose(2, 1), dyn_mask)
    of = optical_flow(batch["opt_flow"], batch["radar_u"], batch["radar_v"], warp, mseg_gt, P, Tcr)
    total = self_sup + em + ms + 0.1 * of + dyn
    items = {"Loss": self_sup.item(), "smoothnessLoss": ss.item(), "chamferLoss": sc.item(),
             "veloLoss": rd.item(), "egoLoss": em.item(), "maskLoss": ms.item(),
             "opticalLoss": of.item(), "superviseLoss": dyn.item()}
    return total, items


def self_supervised_loss(batch, pred_f):
    """losses/radar_loss.py:124-160 -- all RaFlow trains on (:274-276)."""
    pc1, pc2 = batch["pc1"], batch["pc2"]
    sc = soft_chamfer(pc1, pc2, pc1 + pred_f)
    ss = smoothness(pc1, pred_f)
    rd = radial_displacement(pc1, pred_f, batch["ft1"][:, 0])
    total = sc + ss + rd
    return total, {"Loss": total.item(), "smoothnessLoss": ss.item(), "chamferLoss": sc.item(), "veloLoss": rd.item()}


def train_step(net, opt, batch, P, Tcr):
    """main_util.py:63-76: labels -> forward('train') -> loss -> zero_grad/backward/step."""
    dyn, mseg = make_labels(batch)
    pred_f, mseg_pre, pre_trans, mask = net(batch["pc1"], batch["pc2"], batch["ft1"], batch["ft2"], mseg, "train")
    loss, items = radar_flow_loss(batch, pred_f, pre_trans, mseg_pre, mseg, dyn, P, Tcr)
    opt.zero_grad()
    loss.backward()
    opt.step()
    return loss, items, (pred_f, mseg_pre, pre_trans, mask), (dyn, mseg)
