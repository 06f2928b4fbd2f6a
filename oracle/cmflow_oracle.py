"""torch-CPU restatement of the reference's CMFlow / CMFlow-T forward -- TEST INFRASTRUCTURE ONLY.

This is the checker for the HIP path, never the thing measured or shipped: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg import it.  It keeps the
reference's tensor layout ((B,C,N) features, (B,C,N,ns) grouped tensors) and op sequence so
that it agrees with the reference bit-for-bit wherever torch is deterministic; the three
native ops and the kNN come from oracle/cmf_oracle.c (oracle/ops.py).

Pinned against the reference: tests/test_oracle.py compares every output of this module
with tests/golden/*.npz, which were produced by the reference's own Python modules
(tests/golden/make_golden.py).  The reference itself ships no tests ("parity unpinned"
upstream).

Reference lines restated (paths relative to Toytiny/CMFlow):
  lib/pointnet2_utils.py:184-292        GroupingOperation / BallQuery / QueryAndGroup
  utils/model_utils/radarflow_util.py   :52-63 index_points_group, :88-99 knn_point,
                                        :101-118 MultiScaleEncoder, :121-162 PointLocalFeature,
                                        :164-237 FeatureCorrelator, :240-285 heads, :287-318 WeightNet
  models/cmflow.py:10-197, models/cmflow_t.py:44-47,94-107,110-124,185-211
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


class _Group(torch.autograd.Function):
    """lib/pointnet2_utils.py:184-225 over the C oracle."""

    @staticmethod
    def forward(ctx, features, idx):
        idx = idx.int().contiguous()
        ctx.save_for_backward(idx)
        ctx.n = features.shape[2]
        return ops.group_points(features.contiguous(), idx)

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        return ops.group_points_grad(grad_out.contiguous(), idx, ctx.n), None


grouping_operation = _Group.apply


def query_and_group(radius, nsample, xyz, new_xyz, features):
    """lib/pointnet2_utils.py:269-292: -> (B, 3+C, npoint, nsample), relative xyz first."""
    idx = ops.ball_query(radius, nsample, xyz, new_xyz)
    grouped_xyz = grouping_operation(xyz.transpose(1, 2).contiguous(), idx)
    grouped_xyz = grouped_xyz - new_xyz.transpose(1, 2).unsqueeze(-1)
    return torch.cat([grouped_xyz, grouping_operation(features, idx)], dim=1), idx


def index_points_group(points, knn_idx):
    """radarflow_util.py:52-63: points (B,N,C), knn_idx (B,N,K) -> (B,N,K,C)."""
    return grouping_operation(points.permute(0, 2, 1).contiguous(), knn_idx.int()).permute(0, 2, 3, 1)


def knn_point(nsample, xyz, new_xyz):
    """radarflow_util.py:88-99 in canonical order (ascending distance, then index)."""
    return ops.knn(nsample, xyz.detach().contiguous(), new_xyz.detach().contiguous()).long()


class PointLocalFeature(nn.Module):
    def __init__(self, radius, nsample, in_channel, mlp, mlp2):
        super().__init__()
        self.radius, self.nsample = radius, nsample
        self.mlp_convs, self.mlp_bns = nn.ModuleList(), nn.ModuleList()
        self.mlp2_convs, self.mlp2_bns = nn.ModuleList(), nn.ModuleList()
        last = in_channel + 3
        for oc in mlp:
            self.mlp_convs.append(nn.Conv2d(last, oc, 1, bias=False))
            self.mlp_bns.append(nn.BatchNorm2d(oc))
            last = oc
        for oc in mlp2:
            self.mlp2_convs.append(nn.Conv2d(last, oc, 1, bias=False))
            self.mlp2_bns.append(nn.BatchNorm2d(oc))
            last = oc
        self.trace = None

    def forward(self, xyz, points):
        xyz_t = xyz.permute(0, 2, 1).contiguous()
        x, idx = query_and_group(self.radius, self.nsample, xyz_t, xyz_t, points)
        if self.trace is not None:
            self.trace.append(idx)
        for conv, bn in zip(self.mlp_convs, self.mlp_bns):
            x = F.relu(bn(conv(x)))
        x = torch.max(x, -1)[0].unsqueeze(2)
        for conv, bn in zip(self.mlp2_convs, self.mlp2_bns):
            x = F.relu(bn(conv(x)))
        return x.squeeze(2)


class MultiScaleEncoder(nn.Module):
    def __init__(self, radius, nsample, in_channel, mlp, mlp2):
        super().__init__()
        self.ms_ls = nn.ModuleList(
            [PointLocalFeature(r, k, in_channel, mlp, mlp2) for r, k in zip(radius, nsample)])

    def forward(self, xyz, features):
        return torch.cat([sa(xyz, features) for sa in self.ms_ls], dim=1)


class WeightNet(nn.Module):
    def __init__(self, in_channel, out_channel, hidden_unit=(8, 8)):
        super().__init__()
        self.mlp_convs, self.mlp_bns = nn.ModuleList(), nn.ModuleList()   # BNs exist but are unused (bn=False)
        chans = [in_channel, *hidden_unit, out_channel]
        for a, b in zip(chans[:-1], chans[1:]):
            self.mlp_convs.append(nn.Conv2d(a, b, 1))
            self.mlp_bns.append(nn.BatchNorm2d(b))

    def forward(self, x):
        for conv in self.mlp_convs:
            x = F.relu(conv(x))
        return x


class FeatureCorrelator(nn.Module):
    def __init__(self, nsample, in_channel, mlp):
        super().__init__()
        self.nsample = nsample
        self.mlp_convs = nn.ModuleList()
        last = in_channel
        for oc in mlp:
            self.mlp_convs.append(nn.Conv2d(last, oc, 1))
            last = oc
        self.weightnet1 = WeightNet(3, last)
        self.weightnet2 = WeightNet(3, last)
        self.trace = None

    def forward(self, xyz1, xyz2, points1, points2):
        B, C, N1 = xyz1.shape
        D1 = points1.shape[1]
        xyz1, xyz2 = xyz1.permute(0, 2, 1), xyz2.permute(0, 2, 1)
        points1, points2 = points1.permute(0, 2, 1), points2.permute(0, 2, 1)
        K = self.nsample
        # point-to-patch
        knn_idx = knn_point(K, xyz2, xyz1)
        if self.trace is not None:
            self.trace.append(knn_idx)
        direction = index_points_group(xyz2, knn_idx) - xyz1.view(B, N1, 1, C)
        g2 = index_points_group(points2, knn_idx)
        g1 = points1.view(B, N1, 1, D1).repeat(1, 1, K, 1)
        x = torch.cat([g1, g2, direction], dim=-1).permute(0, 3, 2, 1)     # (B, D1+D2+3, K, N1)
        for conv in self.mlp_convs:
            x = F.leaky_relu(conv(x), 0.1)
        w = self.weightnet1(direction.permute(0, 3, 2, 1))
        p2p = torch.sum(w * x, dim=2)                                       # (B, C, N1)
        # patch-to-patch
        knn_idx = knn_point(K, xyz1, xyz1)
        if self.trace is not None:
            self.trace.append(knn_idx)
        direction = index_points_group(xyz1, knn_idx) - xyz1.view(B, N1, 1, C)
        w = self.weightnet2(direction.permute(0, 3, 2, 1))
        g = index_points_group(p2p.permute(0, 2, 1), knn_idx)
        return torch.sum(w * g.permute(0, 3, 2, 1), dim=2)


class _Head(nn.Module):
    def __init__(self, in_channel, mlp, out_ch):
        super().__init__()
        self.sf_mlp = nn.ModuleList()
        last = in_channel
        for oc in mlp:
            self.sf_mlp.append(nn.Sequential(nn.Conv2d(last, oc, 1, bias=False), nn.BatchNorm2d(oc),
                                             nn.ReLU(inplace=False)))
            last = oc
        self.conv2 = nn.Conv2d(last, out_ch, 1, bias=False)


class FlowHead(_Head):
    def __init__(self, in_channel, mlp):
        super().__init__(in_channel, mlp, 3)

    def forward(self, feat):
        x = feat.unsqueeze(3)
        for blk in self.sf_mlp:
            x = blk(x)
        return self.conv2(x).squeeze(3)


class MotionHead(_Head):
    def __init__(self, in_channel, mlp):
        super().__init__(in_channel, mlp, 1)

    def forward(self, feat):
        x = feat.unsqueeze(3)
        for blk in self.sf_mlp:
            x = blk(x)
        return torch.sigmoid(self.conv2(x)).squeeze(3)


def rigid_to_flow(pc, trans):
    """models/cmflow.py:51-55 / utils/util.py:184-189"""
    h = torch.cat((pc, torch.ones((pc.size(0), 1, pc.size(2)), dtype=pc.dtype, device=pc.device)), dim=1)
    return torch.matmul(trans, h)[:, :3] - pc


def weighted_kabsch(A, B, W):
    """models/cmflow.py:128-169.  A, B (b,3,N), W (b,N) normalised weights -> (b,4,4).
    Note the reference negates ROW 2 of V in the reflection case (cmflow.py:161-162)."""
    b = A.size(0)
    W = W.unsqueeze(2)
    cA = torch.sum(A.transpose(2, 1).contiguous() * W, dim=1).reshape(b, 3, 1)
    cB = torch.sum(B.transpose(2, 1).contiguous() * W, dim=1).reshape(b, 3, 1)
    Am, Bm = A - cA, B - cB
    H = torch.matmul(Am, Bm.transpose(2, 1).contiguous() * W)
    U, _, V = torch.svd(H)
    Z = torch.matmul(V, U.transpose(2, 1).contiguous())
    d = (torch.linalg.det(Z) < 0).type(torch.int8) * 2 - 1
    Vc = V.clone()
    Vc[:, 2, :] *= -d.view(b, 1)
    R = torch.matmul(Vc, U.transpose(2, 1).contiguous())
    t = torch.matmul(-R, cA) + cB
    last = torch.tensor([0.0, 0.0, 0.0, 1.0], dtype=A.dtype, device=A.device).repeat(b, 1).view(b, 1, 4)
    return torch.cat((torch.cat((R, t), dim=2), last), dim=1)


class CMFlow(nn.Module):
    """models/cmflow.py:10-197"""
    add_eps = True          # cmflow.py:105 adds 1e-4 to the scores; cmflow_t.py:119 does not

    def __init__(self, args):
        super().__init__()
        self.npoints = args.num_points
        self.stat_thres = args.stat_thres
        radius, nsamples = [2.0, 4.0, 8.0, 16.0], [4, 8, 16, 32]
        self.mse_layer = MultiScaleEncoder(radius, nsamples, 3, [32, 32, 64], [64, 64, 64])
        fc = 4 * 64 * 2
        self.fc_layer = FeatureCorrelator(8, fc * 2 + 3, [fc, fc, fc])
        self.mse_layer2 = MultiScaleEncoder(radius, nsamples, fc * 2 + 3, [fc, fc // 2, fc // 8],
                                            [fc // 8, fc // 8, fc // 8])
        self._pre_head_modules()
        sf = 4 * (fc // 8) * 2
        self.fp = FlowHead(sf, [sf // 2, sf // 4, sf // 8])
        self.mp = MotionHead(sf, [sf // 2, sf // 4, sf // 8])

    def _pre_head_modules(self):
        pass

    def _embed(self, pc1, pc2, feature1, feature2):
        f1 = self.mse_layer(pc1, feature1)
        f2 = self.mse_layer(pc2, feature2)
        N = pc1.size(2)
        f1 = torch.cat((f1, torch.max(f1, -1)[0].unsqueeze(2).expand(-1, -1, N)), dim=1)
        f2 = torch.cat((f2, torch.max(f2, -1)[0].unsqueeze(2).expand(-1, -1, pc2.size(2))), dim=1)
        cor = self.fc_layer(pc1, pc2, f1, f2)
        prop = self.mse_layer2(pc1, torch.cat((feature1, f1, cor), dim=1))
        self.last = {"pc1_features": f1[:, :256], "pc2_features": f2[:, :256], "cor_features": cor,
                     "prop_features": prop}
        return prop

    def Backbone(self, pc1, pc2, feature1, feature2):
        prop = self._embed(pc1, pc2, feature1, feature2)
        g = torch.max(prop, -1)[0].unsqueeze(2).expand(-1, -1, pc1.size(2))
        return torch.cat((prop, g), dim=1)

    def EgoMotionHead(self, flow, pc1, score):
        score = score.squeeze(1)
        if self.add_eps:
            score = score + 1e-4
        weight = score / score.sum(dim=1).unsqueeze(1)
        return weighted_kabsch(pc1, pc1 + flow, weight)

    @staticmethod
    def refine_with_transform(flow, pc1, trans, mask):
        sf_rg = rigid_to_flow(pc1, trans)
        return torch.where(mask.unsqueeze(1), sf_rg, flow)      # == the per-sample loop cmflow.py:120-123

    def _heads(self, final_features, pc1, label_m, mode):
        output = self.fp(final_features)
        stat_cls = self.mp(final_features)
        scores = label_m.unsqueeze(1) if (mode == "train" and label_m is not None) else stat_cls
        mask = (scores > self.stat_thres).squeeze(1)
        pre_trans = self.EgoMotionHead(output, pc1, scores)
        sf_agg = self.refine_with_transform(output, pc1, pre_trans, mask)
        return sf_agg, stat_cls, pre_trans, mask

    def forward(self, pc1, pc2, feature1, feature2, label_m, mode):
        return self._heads(self.Backbone(pc1, pc2, feature1, feature2), pc1, label_m, mode)


class CMFlow_T(CMFlow):
    """models/cmflow_t.py: CMFlow + GRU(256,256) on the global feature; stat_thres fixed to 0.5
    (:18); no +1e-4 on the scores (:119)."""
    add_eps = False

    def __init__(self, args):
        super().__init__(args)
        self.stat_thres = 0.5

    def _pre_head_modules(self):                       # cmflow_t.py:46 declares the GRU before the heads
        self.gru = nn.GRU(input_size=256, hidden_size=256, num_layers=1)

    def Backbone(self, pc1, pc2, feature1, feature2, gfeat_prev):
        prop = self._embed(pc1, pc2, feature1, feature2)
        gfeat = torch.max(prop, -1)[0]
        if gfeat_prev is None:
            gfeat_prev = torch.zeros(gfeat.shape, dtype=gfeat.dtype, device=gfeat.device)
        gnew = self.gru(gfeat.unsqueeze(0), gfeat_prev.unsqueeze(0))[0].squeeze(0)
        return torch.cat((prop, gnew.unsqueeze(2).expand(-1, -1, pc1.size(2))), dim=1), gnew

    def forward(self, pc1, pc2, feature1, feature2, label_m, mode, gfeat):
        final, gfeat = self.Backbone(pc1, pc2, feature1, feature2, gfeat)
        return (*self._heads(final, pc1, label_m, mode), gfeat)


class FlowDecoder(nn.Module):
    """utils/model_utils/radarflow_util.py:321-350 (FlowPredictor :388-409 has the layout of FlowHead)"""

    def __init__(self, fc_inch):
        super().__init__()
        radius, nsamples = [2.0, 4.0, 8.0, 16.0], [4, 8, 16, 32]
        self.mse = MultiScaleEncoder(radius, nsamples, fc_inch * 2 + 3, [fc_inch, fc_inch // 2, fc_inch // 8],
                                     [fc_inch // 8, fc_inch // 8, fc_inch // 8])
        sf = 4 * (fc_inch // 8) * 2
        self.fp = FlowHead(sf, [sf // 2, sf // 4, sf // 8])

    def forward(self, pc1, feature1, pc1_features, cor_features):
        prop = self.mse(pc1, torch.cat((feature1, pc1_features, cor_features), dim=1))
        g = torch.max(prop, -1)[0].unsqueeze(2).expand(-1, -1, pc1.size(2))
        return self.fp(torch.cat((prop, g), dim=1))


def rigid_transform_masked(A, B, M):
    """models/raflow.py:121-157: Kabsch over the points selected by M.  The centroids are torch.mean over ALL N
    points of the masked coordinates (:132-133), i.e. sum over the mask / N -- weights M/N in weighted_kabsch."""
    return weighted_kabsch(A, B, M.to(A.dtype) / A.size(2))


class RaFlow(nn.Module):
    """models/raflow.py:10-165"""

    def __init__(self, args):
        super().__init__()
        self.rigid_thres = args.rigid_thres
        self.rigid_pcs = 0.25
        self.npoints = args.num_points
        radius, nsamples = [2.0, 4.0, 8.0, 16.0], [4, 8, 16, 32]
        self.mse_layer = MultiScaleEncoder(radius, nsamples, 3, [32, 32, 64], [64, 64, 64])
        fc = 4 * 64 * 2
        self.fc_layer = FeatureCorrelator(8, fc * 2 + 3, [fc, fc, fc])
        self.fd_layer = FlowDecoder(fc)

    def ROFE_module(self, pc1, pc2, feature1, feature2):
        f1 = self.mse_layer(pc1, feature1)
        f2 = self.mse_layer(pc2, feature2)
        f1 = torch.cat((f1, torch.max(f1, -1)[0].unsqueeze(2).expand(-1, -1, pc1.size(2))), dim=1)
        f2 = torch.cat((f2, torch.max(f2, -1)[0].unsqueeze(2).expand(-1, -1, pc2.size(2))), dim=1)
        cor = self.fc_layer(pc1, pc2, f1, f2)
        return self.fd_layer(pc1, feature1, f1, cor)

    def SFR_module(self, output, pc1, feature1, interval):
        N = pc1.size(2)
        warp = pc1 + output
        trans = rigid_transform_masked(pc1, warp, torch.ones(pc1.size(0), N, dtype=pc1.dtype, device=pc1.device))
        sf_rg = rigid_to_flow(pc1, trans)
        vel = feature1[:, 0]
        proj = torch.sum(sf_rg * pc1, dim=1) / torch.norm(pc1, dim=1)
        residual = vel * interval.unsqueeze(1) - proj
        mask_s = torch.abs(residual / vel) < self.rigid_thres
        # raflow.py:106-116: per sample, when more than rigid_pcs of the points are inliers, re-fit on the inliers
        # and replace their flow vectors by the rigid flow
        refit = rigid_transform_masked(pc1, warp, mask_s)
        use = (mask_s.sum(dim=1).to(pc1.dtype) / N) > self.rigid_pcs
        pre_trans = torch.where(use.view(-1, 1, 1), refit, trans)
        sf_agg = torch.where(use.view(-1, 1, 1) & mask_s.unsqueeze(1), rigid_to_flow(pc1, pre_trans), output)
        return sf_agg, pre_trans, mask_s

    def forward(self, pc1, pc2, feature1, feature2, interval):
        output = self.ROFE_module(pc1, pc2, feature1, feature2)
        return (output, *self.SFR_module(output, pc1, feature1, interval))


def set_trace(net, on=True):
    """Collect ball-query / kNN index tensors in call order (for index-level parity tests)."""
    bq, knn = ([] if on else None), ([] if on else None)
    for m in net.modules():
        if isinstance(m, PointLocalFeature):
            m.trace = bq
        if isinstance(m, FeatureCorrelator):
            m.trace = knn
    return bq, knn
