"""RaFlow with the reference's module API and checkpoint layout -- mirror of ``models/raflow.py``.

``RaFlow(args)`` takes ``args.num_points`` / ``args.rigid_thres`` (:16-18); ``forward(pc1, pc2, feature1,
feature2, interval)`` returns ``(output (B,3,N), sf_agg (B,3,N), pre_trans (B,4,4), mask_s (B,N) bool)``
(:159-165).  The backbone (ROFE module :46-76) is CMFlow's: the same set-conv / cost-volume kernels and the
same fused point-major path; the static-flow refinement (SFR module :78-119) fits rigid transforms with the
Kabsch kernel and selects per sample on the device (the reference loops over the batch with a host sync
per sample).
"""
import torch
import torch.nn as nn

from .cmflow import CMFlow
from .radarflow_util import FeatureCorrelator, FlowDecoder, MultiScaleEncoder, weighted_kabsch


class RaFlow(CMFlow):
    def __init__(self, args):
        nn.Module.__init__(self)
        self.rigid_thres = args.rigid_thres
        self.rigid_pcs = 0.25
        self.npoints = args.num_points
        sa_radius = [2.0, 4.0, 8.0, 16.0]
        sa_nsamples = [4, 8, 16, 32]
        sa_mlp2s = [64, 64, 64]
        self.mse_layer = MultiScaleEncoder(sa_radius, sa_nsamples, in_channel=3, mlp=[32, 32, 64], mlp2=sa_mlp2s)
        fc_inch = len(sa_radius) * sa_mlp2s[-1] * 2
        self.fc_layer = FeatureCorrelator(8, in_channel=fc_inch * 2 + 3, mlp=[fc_inch, fc_inch, fc_inch])
        self.fd_layer = FlowDecoder(fc_inch=fc_inch)

    def _second_encoder(self):
        return self.fd_layer.mse

    def ROFE_module(self, pc1, pc2, feature1, feature2):
        """raflow.py:46-76"""
        final_features = self.Backbone(pc1, pc2, feature1, feature2)          # (B,512,N): prop features + global max
        return self.fd_layer.fp.forward_pm(final_features.transpose(1, 2)).transpose(1, 2)

    @staticmethod
    def rigid_transform_torch(A, B, M):
        """raflow.py:121-157: Kabsch over the points selected by M.  The reference's centroids are torch.mean over
        all N points of the masked coordinates (:132-133) -- sum over the mask divided by N -- which is the
        weighted Kabsch with weights M / N."""
        return weighted_kabsch(A, B, M.to(A.dtype) / A.size(2))

    def SFR_module(self, output, pc1, feature1, interval):
        """raflow.py:78-119"""
        B, _, N = pc1.shape
        pc1_warp = pc1 + output
        trans = self.rigid_transform_torch(pc1, pc1_warp, torch.ones((B, N), dtype=pc1.dtype, device=pc1.device))
        sf_rg = self.rigid_to_flow(pc1, trans)
        vel_1 = feature1[:, 0]
        sf_proj = torch.sum(sf_rg * pc1, dim=1) / torch.norm(pc1, dim=1)
        residual = vel_1 * interval.unsqueeze(1) - sf_proj
        mask_s = torch.abs(residual / vel_1) < self.rigid_thres
        # :106-116 without the per-sample loop: where more than rigid_pcs of a sample's points are inliers, re-fit
        # on the inliers and replace their flow vectors by the rigid flow
        refit = self.rigid_transform_torch(pc1, pc1_warp, mask_s)
        use = ((mask_s.sum(dim=1).to(pc1.dtype) / N) > self.rigid_pcs).view(B, 1, 1)
        pre_trans = torch.where(use, refit, trans)
        sf_agg = torch.where(use & mask_s.unsqueeze(1), self.rigid_to_flow(pc1, pre_trans), output)
        return sf_agg, pre_trans, mask_s

    def forward(self, pc1, pc2, feature1, feature2, interval):
        output = self.ROFE_module(pc1, pc2, feature1, feature2)
        sf_agg, pre_trans, mask_s = self.SFR_module(output, pc1, feature1, interval)
        return output, sf_agg, pre_trans, mask_s
