"""CMFlow / CMFlow_T with the reference's module API and checkpoint layout.

``CMFlow(args)`` takes ``args.num_points`` / ``args.stat_thres`` (models/cmflow.py:17-18);
``forward(pc1, pc2, feature1, feature2, label_m, mode)`` returns
``(sf_agg (B,3,N), stat_cls (B,1,N), pre_trans (B,4,4), mask (B,N) bool)`` (:171-197); the
state_dict has the reference's 374 (CMFlow) / 378 (CMFlow_T) tensors, so reference
checkpoints load unchanged (models/model.py:38 uses strict=False).
"""
import os

import torch
import torch.nn as nn

from . import fused_blocks as FB
from .radarflow_util import (FeatureCorrelator, FlowHead, MotionHead, MultiScaleEncoder, ego_refine, weighted_kabsch)


class CMFlow(nn.Module):
    score_eps = 1e-4            # models/cmflow.py:105; CMFlow_T has none (cmflow_t.py:119)
    # One execution path: the point-major fused path on the hand-written GEMM / BN / pooling kernels.  (Two cross-checks are
    # TEST fixtures, not product code: the same layout with its dense math through torch -- tests/pm_torch.py -- and the
    # reference's own op sequence in its (B,C,N,ns) layout over the drop-in kernels, the oracle's modules bound to
    # pointnet2_utils -- tests/hip_ops.py.)
    path = "pm"

    def __init__(self, args):
        super().__init__()
        self.npoints = args.num_points
        self.stat_thres = args.stat_thres
        # multi-scale set feature abstraction (cmflow.py:20-27)
        sa_radius = [2.0, 4.0, 8.0, 16.0]
        sa_nsamples = [4, 8, 16, 32]
        sa_mlps = [32, 32, 64]
        sa_mlp2s = [64, 64, 64]
        num_sas = len(sa_radius)
        self.mse_layer = MultiScaleEncoder(sa_radius, sa_nsamples, in_channel=3, mlp=sa_mlps, mlp2=sa_mlp2s)
        # feature correlation layer / cost volume (cmflow.py:29-32)
        fc_inch = num_sas * sa_mlp2s[-1] * 2
        self.fc_layer = FeatureCorrelator(8, in_channel=fc_inch * 2 + 3, mlp=[fc_inch, fc_inch, fc_inch])
        # second multi-scale encoder on the embeddings (cmflow.py:34-42)
        ep_inch = fc_inch * 2 + 3
        ep_mlps = [fc_inch, fc_inch // 2, fc_inch // 8]
        ep_mlp2s = [fc_inch // 8, fc_inch // 8, fc_inch // 8]
        self.mse_layer2 = MultiScaleEncoder(sa_radius, sa_nsamples, in_channel=ep_inch, mlp=ep_mlps, mlp2=ep_mlp2s)
        self._declare_recurrent(len(sa_radius) * ep_mlp2s[-1])
        # heads (cmflow.py:44-48)
        sf_inch = len(sa_radius) * ep_mlp2s[-1] * 2
        sf_mlps = [sf_inch // 2, sf_inch // 4, sf_inch // 8]
        self.fp = FlowHead(in_channel=sf_inch, mlp=sf_mlps)
        self.mp = MotionHead(in_channel=sf_inch, mlp=sf_mlps)

    def _declare_recurrent(self, width):
        pass

    head_streams = True
    fused_tail = True                               # _heads: cmf_ego_refine instead of the torch ops around weighted_kabsch (A/B, tests)
    _head_stream = None

    def _second_encoder(self):
        """The encoder over the flow embeddings (mse_layer2 here, fd_layer.mse in RaFlow)."""
        return self.mse_layer2

    @staticmethod
    def rigid_to_flow(pc, trans):
        """cmflow.py:51-55"""
        # R pc + t - pc as one batched multiply-add and a subtraction (the reference appends a row of ones and multiplies by the 4 x 4
        # matrix: the same sum with t added last instead of first)
        return torch.baddbmm(trans[:, :3, 3:4], trans[:, :3, :3], pc) - pc

    def _propagate(self, pc1, pc2, feature1, feature2):
        """cmflow.py:59-88: everything of Backbone up to prop_features."""
        if self.path != "pm":
            raise ValueError("unknown path %r: the product has one path, 'pm'" % (self.path,))
        return self._propagate_pm(pc1, pc2, feature1, feature2)

    def _propagate_pm(self, pc1, pc2, feature1, feature2):
        """Same computation in point-major layout; returns prop_features as (B,256,N)."""
        # (B,N,3) coordinates and the input channels with zero columns behind them (rows of a multiple of 4 floats for the stacked
        # first-conv GEMM; a1 / a2 are views of those): one launch for the four tensors
        x1, x2, a1p, a2p = FB.inputs_point_major(pc1, pc2, feature1, feature2)
        a1, a2 = a1p[:, :, :feature1.shape[1]], a2p[:, :, :feature2.shape[1]]
        if not self.mse_layer.training and x1.shape == x2.shape:
            # eval-mode BN has no batch statistics: the two clouds share one call of the (weight-shared) encoder
            B = x1.shape[0]
            f12 = self.mse_layer.forward_pm(torch.cat((x1, x2), dim=0), torch.cat((a1p, a2p), dim=0))
            f1, f2 = f12[:B], f12[B:]
        else:                                                   # one zero column: rows of 4 floats for the stacked first-conv GEMM
            f1, f2 = self.mse_layer.forward_pm_pair(x1, a1p, x2, a2p)                      # (B,N,256) each
        # train.TrainStep reduces the gradient bucket in segments as backward completes them.  Every autograd node of the cost
        # volume, the second encoder and the heads is created AFTER the first encoder's node and depends only on gradients that
        # exist before that node becomes ready, so the engine (highest sequence number first among ready nodes, AccumulateGrad
        # ahead of everything) has run all of them when the first encoder's output gradient is handed over: segment 1 (cost
        # volume) is complete there, segment 0 (heads, GRU, second encoder) when the embedding's gradient is.
        ready = getattr(self, "_grad_ready", None)
        if ready is not None:
            for t in (f1, f2):
                if t.requires_grad:
                    t.register_hook(lambda g: ready(1))
        f1, f2 = FB.global_max_cat(f1), FB.global_max_cat(f2)                              # (B,N,512): features + global max
        cor = self.fc_layer.forward_pm(x1, x2, f1, f2)                                     # (B,N,512)
        # embeddings as [f1 | cor | ft1 | zero pad], K = 1040: the columns that need a gradient come first (the
        # data-gradient GEMM of the stacked first conv is 1024 wide and f1 / cor read their blocks of it in place),
        # the raw input channels (cmflow.py:82 puts them first) go behind, rows padded to a multiple of 16 floats
        n_grad, n_tail = f1.shape[2] + cor.shape[2], a1.shape[2]
        pad = -(n_grad + n_tail) % 16
        emb = torch.cat((f1, cor, a1, a1.new_zeros(a1.shape[0], a1.shape[1], pad)), dim=2)
        if ready is not None and emb.requires_grad:
            emb.register_hook(lambda g: ready(0))      # heads + second encoder have run their backward
        prop = self._second_encoder().forward_pm(x1, emb, n_tail=n_tail, n_grad=n_grad)       # (B,N,256)
        self.last = {"pc1_features": f1[:, :, :256].transpose(1, 2), "pc2_features": f2[:, :, :256].transpose(1, 2),
                     "cor_features": cor.transpose(1, 2), "prop_features": prop.transpose(1, 2)}
        return prop.transpose(1, 2)

    def Backbone(self, pc1, pc2, feature1, feature2):
        prop_features = self._propagate(pc1, pc2, feature1, feature2)
        return FB.global_max_cat(prop_features.transpose(1, 2)).transpose(1, 2)      # prop_features is the (B,256,N) view of point-major rows

    def EgoMotionHead(self, flow, pc1, score):
        """cmflow.py:96-110"""
        pc1_warp = pc1 + flow
        score = score.squeeze(1) + self.score_eps if self.score_eps else score.squeeze(1)
        weight = score / score.sum(dim=1).unsqueeze(1)
        return self.WeightedKabsch(pc1, pc1_warp, weight)

    def refine_with_transform(self, flow, pc1, trans, mask):
        """cmflow.py:112-125 (the per-sample boolean-index loop is one select; no host syncs)."""
        return torch.where(mask.unsqueeze(1), self.rigid_to_flow(pc1, trans), flow)

    @staticmethod
    def WeightedKabsch(A, B, W):
        """cmflow.py:128-169"""
        return weighted_kabsch(A, B, W)

    def _heads(self, final_features, pc1, label_m, mode):
        if self.head_streams and final_features.is_cuda:
            # the two heads are independent chains of small GEMMs (N = 256: a third of the CUs each): the motion head
            # runs on a side stream next to the flow head; autograd replays each backward on its forward stream
            ff = final_features.transpose(1, 2)                       # (B,N,512) view
            main = torch.cuda.current_stream()
            if self._head_stream is None:
                self._head_stream = FB.side_stream(0)
            side = self._head_stream
            side.wait_stream(main)
            FB.stress_point([side, main])
            with torch.cuda.stream(side):
                stat_cls = FB.stress_mark(self.mp.forward_pm(ff)).transpose(1, 2)
            output = FB.stress_mark(self.fp.forward_pm(ff)).transpose(1, 2)
            main.wait_stream(side)
            ff.record_stream(side)
            stat_cls.record_stream(main)
        else:
            ff = final_features.transpose(1, 2)                       # (B,N,512) view
            output = self.fp.forward_pm(ff).transpose(1, 2)
            stat_cls = self.mp.forward_pm(ff).transpose(1, 2)
        if (mode == 'train') and (label_m is not None):
            scores = label_m.unsqueeze(1)
        else:
            scores = stat_cls
        if self.fused_tail and output.is_cuda:
            # cmflow.py:96-125 (ego-motion weights, weighted Kabsch, rigid refinement, select) as one native call per direction
            pre_trans, sf_agg, mask = ego_refine(output, pc1, scores.squeeze(1), self.score_eps or 0.0, self.stat_thres)
            return sf_agg, stat_cls, pre_trans, mask
        mask = (scores > self.stat_thres).squeeze(1)
        pre_trans = self.EgoMotionHead(output, pc1, scores)
        sf_agg = self.refine_with_transform(output, pc1, pre_trans, mask)
        return sf_agg, stat_cls, pre_trans, mask

    def forward(self, pc1, pc2, feature1, feature2, label_m, mode):
        final_features = self.Backbone(pc1, pc2, feature1, feature2)
        return self._heads(final_features, pc1, label_m, mode)


class CMFlow_T(CMFlow):
    """models/cmflow_t.py: CMFlow + nn.GRU(256,256) on the global feature, carried across the
    frames of a mini-clip (clip_util.py:34-62)."""
    score_eps = 0.0

    def __init__(self, args):
        super().__init__(args)
        self.stat_thres = 0.50                       # cmflow_t.py:18

    def _declare_recurrent(self, width):             # cmflow_t.py:46 (declared before the heads)
        self.gru = nn.GRU(input_size=width, hidden_size=width, num_layers=1)

    def Backbone(self, pc1, pc2, feature1, feature2, gfeat_prev):
        prop_features = self._propagate(pc1, pc2, feature1, feature2)
        gfeat = torch.max(prop_features, -1)[0]
        if gfeat_prev is None:
            gfeat_prev = torch.zeros_like(gfeat)
        gfeat_new = self.gru(gfeat.unsqueeze(0), gfeat_prev.unsqueeze(0))[0].squeeze(0)
        expand = gfeat_new.unsqueeze(2).expand(-1, -1, pc1.size(2))
        return torch.cat((prop_features, expand), dim=1), gfeat_new

    def forward(self, pc1, pc2, feature1, feature2, label_m, mode, gfeat):
        final_features, gfeat = self.Backbone(pc1, pc2, feature1, feature2, gfeat)
        return (*self._heads(final_features, pc1, label_m, mode), gfeat)


def init_model(args, device="cuda"):
    """models/model.py:19-47 without nn.DataParallel: multi-GPU is one process per GPU with an
    RCCL gradient all-reduce (cmflow_amd/dp.py)."""
    if args.model == 'cmflow':
        net = CMFlow(args)
    elif args.model == 'raflow':
        from .raflow import RaFlow
        net = RaFlow(args)
    elif args.model == 'cmflow_t':
        net = CMFlow_T(args)
    else:
        raise Exception('Not implemented')
    net = net.to(device)
    path = getattr(args, 'model_path', '')
    if getattr(args, 'eval', False) or getattr(args, 'load_checkpoint', False):
        if path == '':
            path = 'checkpoints/' + args.exp_name + '/models/model.best.t7'
        if not os.path.exists(path):
            print("can't find pretrained model")
            return None
        net.load_state_dict(torch.load(path, map_location=device), strict=False)
    return net
