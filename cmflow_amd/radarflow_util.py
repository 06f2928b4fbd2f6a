"""Host-side mirror of the reference's building blocks (utils/model_utils/radarflow_util.py).

Module names, constructor arguments, parameter names (=> state_dict keys) and the tensor contracts of
``forward`` (channel-major (B,C,N) in and out) follow the reference.  The computation does not: every
``forward`` is an adapter onto the point-major path (``forward_pm``: rows = points, hoisted first conv,
fused grouping / BN / pooling kernels of libcmflow_hip.so), so a caller that uses a block on its own gets
the same kernels the whole model runs on.

  knn_point (:88-99)            -> cmf_knn            (canonical distance + ordered top-k)
  index_points_group (:52-63)   -> cmf_group_rows     (+ cmf_group_rows_grad in backward)
  PointLocalFeature (:121-162)  -> cmf_setconv_forward/_backward (ball query, group, 6 x conv+BN+ReLU, max)
  FeatureCorrelator (:164-237)  -> cmf_knn, cmf_group_affine, cmf_gemm, cmf_weightnet_ksum
  WeightedKabsch (models/cmflow.py:128-169) -> cmf_weighted_kabsch(+_grad)

The reference's own op sequence in its (B,C,N,ns) layout over the drop-in kernels (cmf_ball_query /
cmf_group_points / cmf_group_points_grad through ``pointnet2_utils``) is exercised by the tests with the
oracle's modules bound to those ops (tests/hip_ops.py), not from this package.
"""
import contextlib
import os

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function

from . import _lib
from . import pointnet2_utils as pointutils
from .fused import Neighbors, group_rows, w2d
from . import fused_blocks as FB

_f32, _i32 = torch.float32, torch.int32


def square_distance(src, dst):
    """radarflow_util.py:8-30: (B,N,C),(B,M,C) -> (B,N,M) squared distances |s|^2 + |d|^2 - 2 s.d, clamped at 0.
    Evaluated as ((-2 s.d) + |s|^2) + |d|^2 -- the operation order of the reference expression, which the canonical
    distance of the kNN kernel and of the oracle reproduces bit for bit."""
    cross = torch.matmul(src, dst.transpose(1, 2)).mul_(-2.0)
    cross.add_(src.square().sum(-1).unsqueeze(2)).add_(dst.square().sum(-1).unsqueeze(1))
    return cross.clamp_min_(0.0)


def index_points_group(points, knn_idx):
    """radarflow_util.py:52-63: points (B,N,C), knn_idx (B,S,K) -> (B,S,K,C).  Rows are contiguous in this layout, so
    the gather is the point-major row copy (cmf_group_rows; backward: segmented sum over the inverse index) -- the
    reference transposes to (B,C,N), gathers per channel and transposes back."""
    return group_rows(points, Neighbors(knn_idx.int().contiguous(), points.shape[1]))


def knn_point(nsample, xyz, new_xyz, return_dist=False, i32=False):
    """radarflow_util.py:88-99: xyz (B,N,3) database, new_xyz (B,S,3) queries -> (B,S,nsample)
    int64 (torch.topk's dtype; i32=True: the kernel's own int32 tensor, for the blocks).  Order is canonical (ascending distance,
    lowest index first); the reference's topk(sorted=False) order is unspecified."""
    xyz = xyz.detach().contiguous()
    new_xyz = new_xyz.detach().contiguous()
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    idx = torch.empty(B, S, nsample, dtype=_i32, device=xyz.device)
    dist = torch.empty(B, S, nsample, dtype=_f32, device=xyz.device) if return_dist else None
    err = _lib.lib().cmf_knn(B, N, S, nsample, _lib.dev_ptr(xyz, _f32), _lib.dev_ptr(new_xyz, _f32),
                             _lib.dev_ptr(idx, _i32), _lib.dev_ptr(dist, _f32), _lib.stream_ptr())
    _lib.check(err, "cmf_knn")
    if i32:
        return (idx, dist) if return_dist else idx
    return (idx.long(), dist) if return_dist else idx.long()


def rel_xyz(xyz_t, centre_t, idx):
    """(B,N,3) points, (B,P,3) centres, idx (B,P,S) int32 -> (B,P,S,4): neighbour minus centre with a zero fourth column
    (radarflow_util.py:207-208 + the row padding of the narrow GEMMs), one launch.  No gradient (coordinates are inputs)."""
    B, N, _ = xyz_t.shape
    P, S = idx.shape[1], idx.shape[2]
    out = torch.empty(B, P, S, 4, dtype=_f32, device=xyz_t.device)
    err = _lib.lib().cmf_rel_xyz(B, N, P, S, _lib.dev_ptr(xyz_t.detach().contiguous(), _f32), _lib.dev_ptr(centre_t.detach().contiguous(), _f32),
                                 _lib.dev_ptr(idx, _i32), _lib.dev_ptr(out, _f32), _lib.stream_ptr())
    _lib.check(err, "cmf_rel_xyz")
    return out


class _WeightedKabsch(Function):
    @staticmethod
    def forward(ctx, A, Bm, W):
        A, Bm, W = A.contiguous(), Bm.contiguous(), W.contiguous()
        b, _, n = A.shape
        trans = torch.empty(b, 4, 4, dtype=_f32, device=A.device)
        aux = torch.empty(b, 32, dtype=torch.float64, device=A.device)
        err = _lib.lib().cmf_weighted_kabsch(b, n, _lib.dev_ptr(A, _f32), _lib.dev_ptr(Bm, _f32), _lib.dev_ptr(W, _f32),
                                             _lib.dev_ptr(trans, _f32), _lib.dev_ptr(aux, torch.float64),
                                             _lib.stream_ptr())
        _lib.check(err, "cmf_weighted_kabsch")
        ctx.save_for_backward(A, Bm, W, aux)
        return trans

    @staticmethod
    def backward(ctx, grad_trans):
        A, Bm, W, aux = ctx.saved_tensors
        b, _, n = A.shape
        need = ctx.needs_input_grad
        gA = torch.empty_like(A) if need[0] else None
        gB = torch.empty_like(Bm) if need[1] else None
        gW = torch.empty_like(W) if need[2] else None
        err = _lib.lib().cmf_weighted_kabsch_grad(
            b, n, _lib.dev_ptr(A, _f32), _lib.dev_ptr(Bm, _f32), _lib.dev_ptr(W, _f32),
            _lib.dev_ptr(aux, torch.float64), _lib.dev_ptr(grad_trans.contiguous(), _f32),
            _lib.dev_ptr(gA, _f32), _lib.dev_ptr(gB, _f32), _lib.dev_ptr(gW, _f32), _lib.stream_ptr())
        _lib.check(err, "cmf_weighted_kabsch_grad")
        return gA, gB, gW


def weighted_kabsch(A, B, W):
    """models/cmflow.py:128-169: A, B (b,3,N), W (b,N) -> (b,4,4)."""
    assert A.size() == B.size()
    return _WeightedKabsch.apply(A, B, W)


class _EgoRefine(Function):
    """EgoMotionHead + refine_with_transform (models/cmflow.py:96-125) as one native call per direction (cmf_ego_refine):
    flow (b,3,N), pc1 (b,3,N), score (b,N) -> pre_trans (b,4,4), sf_agg (b,3,N), mask (b,N) bool.  pc1 takes no gradient."""

    @staticmethod
    def forward(ctx, flow, pc1, score, eps, thres):
        flow, pc1, score = flow.contiguous(), pc1.contiguous(), score.contiguous()
        b, _, n = pc1.shape
        dev = pc1.device
        W = torch.empty(b, n, dtype=_f32, device=dev); Bm = torch.empty(b, 3, n, dtype=_f32, device=dev)
        trans = torch.empty(b, 4, 4, dtype=_f32, device=dev); aux = torch.empty(b, 32, dtype=torch.float64, device=dev)
        sf = torch.empty(b, 3, n, dtype=_f32, device=dev); mask = torch.empty(b, n, dtype=torch.uint8, device=dev)
        err = _lib.lib().cmf_ego_refine(b, n, float(eps), float(thres), _lib.dev_ptr(pc1, _f32), _lib.dev_ptr(flow, _f32),
                                        _lib.dev_ptr(score, _f32), _lib.dev_ptr(W, _f32), _lib.dev_ptr(Bm, _f32), _lib.dev_ptr(trans, _f32),
                                        _lib.dev_ptr(aux, torch.float64), _lib.dev_ptr(sf, _f32), mask.data_ptr(), _lib.stream_ptr())
        _lib.check(err, "cmf_ego_refine")
        ctx.save_for_backward(pc1, score, W, Bm, mask, aux)
        ctx.eps = float(eps)
        mask_b = mask.view(torch.bool)
        ctx.mark_non_differentiable(mask_b)
        return trans, sf, mask_b

    @staticmethod
    def backward(ctx, g_trans, g_sf, _g_mask):
        pc1, score, W, Bm, mask, aux = ctx.saved_tensors
        b, _, n = pc1.shape
        if g_sf is None:
            g_sf = torch.zeros_like(Bm)
        g_flow = torch.empty_like(Bm); g_w = torch.empty_like(W)
        g_score = torch.empty_like(W) if ctx.needs_input_grad[2] else None
        err = _lib.lib().cmf_ego_refine_grad(
            b, n, ctx.eps, _lib.dev_ptr(pc1, _f32), _lib.dev_ptr(score, _f32), _lib.dev_ptr(W, _f32), _lib.dev_ptr(Bm, _f32), mask.data_ptr(),
            _lib.dev_ptr(aux, torch.float64), _lib.dev_ptr(g_sf.contiguous(), _f32),
            _lib.dev_ptr(g_trans.contiguous(), _f32) if g_trans is not None else None,
            _lib.dev_ptr(g_flow, _f32), _lib.dev_ptr(g_w, _f32), _lib.dev_ptr(g_score, _f32), _lib.stream_ptr())
        _lib.check(err, "cmf_ego_refine_grad")
        return g_flow, None, g_score, None, None


def ego_refine(flow, pc1, score, eps, thres):
    """-> (pre_trans, sf_agg, mask): cmflow.py:96-125 with score (b,N) the (detached label or predicted) motion scores."""
    return _EgoRefine.apply(flow, pc1, score, eps, thres)


def _rows(t, pad4=False):
    """(B,C,N) channel-major -> (B,N,C) point-major rows (contiguous); pad4: zero columns up to a multiple of 4 floats
    (16-byte rows for the stacked first-conv GEMM)."""
    r = t.transpose(1, 2)
    if pad4 and r.shape[2] % 4:
        return F.pad(r, (0, -r.shape[2] % 4))
    return r.contiguous()


class MultiScaleEncoder(nn.Module):
    """radarflow_util.py:101-118"""

    def __init__(self, radius, nsample, in_channel, mlp, mlp2):
        super().__init__()
        self.ms_ls = nn.ModuleList()
        for l in range(len(radius)):
            self.ms_ls.append(PointLocalFeature(radius[l], nsample[l], in_channel=in_channel, mlp=mlp, mlp2=mlp2))

    def forward(self, xyz, features):
        """Reference contract (:111-118): xyz (B,3,N), features (B,C,N) -> (B, 64*scales, N)."""
        return self.forward_pm(_rows(xyz), _rows(features, pad4=True)).transpose(1, 2)

    def forward_pm(self, xyz_t, feats, n_tail=0, n_grad=0):
        """Point-major: xyz_t (B,N,3), feats (B,N,C) -> (B,N,4*64).  The feature half of the four
        scales' first convs is ONE GEMM over the shared input (W_f of all scales stacked).
        n_tail / n_grad (fused path only): feats holds the module's input channels as [head, first n_tail channels,
        zero pad] and only its first n_grad columns need a gradient (fused_blocks.StackedFirstConvFn)."""
        o1 = self.ms_ls[0].mlp_convs[0].weight.shape[0]
        if self.multi_stream and FB.USE_BLOCK_CALLS and self.threaded_enqueue and feats.shape[2] % 4 == 0:
            B, N, Kp = feats.shape
            y_all = FB.StackedFirstConvFn.apply(feats.reshape(B * N, Kp), n_tail, n_grad,
                                                *[sa.mlp_convs[0].weight for sa in self.ms_ls]).view(B, N, -1)
            if self._streams is None:
                self._streams = FB.scale_streams(len(self.ms_ls))
            return FB.multi_scale_set_conv(self, list(self.ms_ls), self._streams, xyz_t, y_all)
        if n_tail:                      # permuted layout handed to one of the unstacked paths: back to the module's own order
            cin = self.ms_ls[0].mlp_convs[0].weight.shape[1] - 3
            feats = torch.cat((feats[:, :, cin - n_tail:cin], feats[:, :, :cin - n_tail]), dim=2)
            if cin % 4:
                feats = F.pad(feats, (0, 4 - cin % 4))
        wf_all = torch.cat([w2d(sa.mlp_convs[0])[:, 3:] for sa in self.ms_ls], dim=0)
        kpad = feats.shape[2] - wf_all.shape[1]            # caller may hand over K zero-padded to a multiple of 4
        if kpad:
            wf_all = F.pad(wf_all, (0, kpad))
        y_all = FB.linear(feats, wf_all)
        if not self.multi_stream:
            outs = [FB.set_conv(sa, xyz_t, y_all[:, :, i * o1:(i + 1) * o1]) for i, sa in enumerate(self.ms_ls)]
            return torch.cat(outs, dim=2)
        # The four scales are independent until the concat and most of their kernels are far too small to
        # fill 256 CUs (N = 256): run each scale on its own HIP stream so they overlap.  Autograd replays
        # each block's backward on the stream its forward ran on and orders the streams itself.
        if self._streams is None:
            self._streams = FB.scale_streams(len(self.ms_ls))
        if FB.USE_BLOCK_CALLS and self.threaded_enqueue:
            # one host thread per scale as well: see fused_blocks.MultiScaleBlockFn
            return FB.multi_scale_set_conv(self, list(self.ms_ls), self._streams, xyz_t, y_all)
        main = torch.cuda.current_stream()
        outs = []
        for i, (sa, st) in enumerate(zip(self.ms_ls, self._streams)):
            st.wait_stream(main)
            with torch.cuda.stream(st):
                o = FB.set_conv(sa, xyz_t, y_all[:, :, i * o1:(i + 1) * o1])
            o.record_stream(main)
            outs.append(o)
        for st in self._streams:
            main.wait_stream(st)
        return torch.cat(outs, dim=2)

    threaded_enqueue = True
    def forward_pm_pair(self, xyz1_t, feats1, xyz2_t, feats2):
        """Two calls of this (weight-shared) encoder -- forward_pm(xyz1_t, feats1), forward_pm(xyz2_t, feats2) in this
        order as far as BN running statistics go -- issued concurrently when that is possible (fused path, training with
        in-place gradient sinks: fused_blocks.DualCloudBlockFn)."""
        fused = self.multi_stream and FB.USE_BLOCK_CALLS and self.threaded_enqueue and \
            feats1.shape[2] % 4 == 0 and feats1.shape == feats2.shape
        if fused and self.training:
            ws = [sa.mlp_convs[0].weight for sa in self.ms_ls]
            B, N, Kp = feats1.shape
            y1 = FB.StackedFirstConvFn.apply(feats1.reshape(B * N, Kp), 0, 0, *ws).view(B, N, -1)
            y2 = FB.StackedFirstConvFn.apply(feats2.reshape(B * N, Kp), 0, 0, *ws).view(B, N, -1)
            if self._streams2 is None:
                self._streams2 = FB.scale_streams(len(self.ms_ls), 0) + FB.scale_streams(len(self.ms_ls), 1)
            out = FB.dual_cloud_set_conv(self, list(self.ms_ls), self._streams2, xyz1_t, y1, xyz2_t, y2)
            if out is not None:
                return out
            if self._streams is None:
                self._streams = FB.scale_streams(len(self.ms_ls))
            return (FB.multi_scale_set_conv(self, list(self.ms_ls), self._streams, xyz1_t, y1),
                    FB.multi_scale_set_conv(self, list(self.ms_ls), self._streams, xyz2_t, y2))
        return self.forward_pm(xyz1_t, feats1), self.forward_pm(xyz2_t, feats2)

    multi_stream = True
    _streams = None
    _streams2 = None


class PointLocalFeature(nn.Module):
    """radarflow_util.py:121-162 -- set-conv: group -> (conv1x1+BN+ReLU)x3 -> max over the ball
    -> (conv1x1+BN+ReLU)x3."""

    def __init__(self, radius, nsample, in_channel, mlp, mlp2):
        super().__init__()
        self.radius = radius
        self.nsample = nsample
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        self.mlp2_convs = nn.ModuleList()
        self.mlp2_bns = nn.ModuleList()
        last_channel = in_channel + 3
        for out_channel in mlp:
            self.mlp_convs.append(nn.Conv2d(last_channel, out_channel, 1, bias=False))
            self.mlp_bns.append(nn.BatchNorm2d(out_channel))
            last_channel = out_channel
        last_channel = mlp[-1]
        for out_channel in mlp2:
            self.mlp2_convs.append(nn.Conv2d(last_channel, out_channel, 1, bias=False))
            self.mlp2_bns.append(nn.BatchNorm2d(out_channel))
            last_channel = out_channel
        self.queryandgroup = pointutils.QueryAndGroup(radius, nsample)

    def forward(self, xyz, points):
        """Reference contract (:144-162): xyz (B,3,N), points (B,C,N) -> (B,64,N).  The feature columns of the first
        conv are applied per point (one GEMM), the rest is the fused block."""
        xyz_t, feats = _rows(xyz), _rows(points)
        return self.forward_pm(xyz_t, FB.linear(feats, w2d(self.mlp_convs[0])[:, 3:])).transpose(1, 2)

    def forward_pm(self, xyz_t, y):
        """Point-major set-conv.  xyz_t (B,N,3); y (B,N,O1) = feats @ W_f^T, the feature half of the first conv already
        applied per point (conv is linear: W [dxyz; f[idx]] = W_xyz dxyz + (W_f f)[idx]).  -> (B,N,64)."""
        return FB.set_conv(self, xyz_t, y)


class WeightNet(nn.Module):
    """radarflow_util.py:287-318 (bn=False: the BN modules exist for the state_dict only)."""

    def __init__(self, in_channel, out_channel, hidden_unit=(8, 8), bn=False):
        super().__init__()
        self.bn = bn
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        chans = [in_channel, *hidden_unit, out_channel]
        for a, b in zip(chans[:-1], chans[1:]):
            self.mlp_convs.append(nn.Conv2d(a, b, 1))
            self.mlp_bns.append(nn.BatchNorm2d(b))
        if not bn:      # never touched by forward: no gradient, and the optimizer must skip them (as torch
            for p in self.mlp_bns.parameters():      # skips grad=None params in the reference's Adam)
                p._cmf_unused = True

    def forward(self, localized_xyz):
        """Reference contract (:307-318): (B,3,K,N) -> (B,out,K,N)."""
        return self.forward_pm(localized_xyz.permute(0, 3, 2, 1)).permute(0, 3, 2, 1)

    def forward_pm(self, dxyz, preact_grad=False):
        """dxyz (B,N,K,3) -> (B,N,K,out).  preact_grad: the consumer (WeightedKSumFn relu_w=True) returns the gradient
        of the last layer's pre-activation."""
        assert not self.bn
        w = dxyz
        last = len(self.mlp_convs) - 1
        for i, conv in enumerate(self.mlp_convs):
            w = FB.linear(w, w2d(conv), conv.bias, act=1, preact_grad=preact_grad and i == last)
        return w

    def weighted_ksum(self, dxyz, x, nbr, leaky, x_bias=None, hidden=None):
        """sum_k WeightNet(dxyz)[b,n,k,:] * x[...] (radarflow_util.py:219-221,234-236) on the fused blocks.  With the
        reference's hidden width (8) the last layer is evaluated inside the weighting kernels and the (B,N,K,C) weights
        are never written; other widths materialise them."""
        last = self.mlp_convs[-1]
        if self.fuse_tail and FB.WeightNetKSumFn.supported(last.weight.shape[0], last.weight.shape[1]):
            h = hidden if hidden is not None else self.hidden_pm(dxyz)
            return FB.WeightNetKSumFn.apply(h, w2d(last), last.bias, x, nbr, leaky, x_bias, self._hidden_bias())
        weights = FB.linear(hidden if hidden is not None else self.hidden_pm(dxyz), w2d(last), last.bias, act=1, preact_grad=True)
        return FB.WeightedKSumFn.apply(weights, x, nbr, leaky, True, x_bias)

    def hidden_pm(self, dxyz):
        """All layers but the last on the fused blocks: (B,N,K,3|4) -> (B,N,K,hidden).  With the fused tail every ReLU
        mask and bias gradient of the chain is produced by the kernel that consumes the layer's output (the weighting
        kernel for the last hidden layer, the data-gradient GEMM epilogue of layer i+1 for layer i): the layers are
        built with detached biases and preact_grad=True, the real biases ride along as in_bias / h_bias."""
        convs = list(self.mlp_convs)[:-1]
        if not (self.fuse_tail and self._hidden_bias() is not None):
            h = dxyz
            for conv in convs:
                h = FB.linear(h, w2d(conv), conv.bias, act=1)
            return h
        h, prev = dxyz, None
        for conv in convs:
            h = FB.linear(h, w2d(conv), conv.bias.detach(), act=1, preact_grad=True, in_bias=prev)
            prev = conv.bias
        return h

    def _hidden_bias(self):
        """Bias of the last hidden layer when the fused chain applies (tail fused and every hidden width a multiple of 4)."""
        convs = list(self.mlp_convs)
        last = convs[-1]
        if not (self.fuse_tail and FB.WeightNetKSumFn.supported(last.weight.shape[0], last.weight.shape[1])):
            return None
        if any(c.weight.shape[0] % 4 for c in convs[:-1]) or len(convs) < 2:
            return None
        return convs[-2].bias

    fuse_tail = True


class FeatureCorrelator(nn.Module):
    """radarflow_util.py:164-237 -- cost volume (point-to-patch, then patch-to-patch)."""

    def __init__(self, nsample, in_channel, mlp, bn=False, use_leaky=True):
        super().__init__()
        self.nsample = nsample
        self.bn = bn
        self.mlp_convs = nn.ModuleList()
        if bn:
            self.mlp_bns = nn.ModuleList()
        last_channel = in_channel
        for out_channel in mlp:
            self.mlp_convs.append(nn.Conv2d(last_channel, out_channel, 1))
            if bn:
                self.mlp_bns.append(nn.BatchNorm2d(out_channel))
            last_channel = out_channel
        self.weightnet1 = WeightNet(3, last_channel)
        self.weightnet2 = WeightNet(3, last_channel)
        self.relu = nn.ReLU(inplace=True) if not use_leaky else nn.LeakyReLU(0.1, inplace=True)

    def forward(self, xyz1, xyz2, points1, points2):
        """Reference contract (:185-237): xyz (B,3,N), points (B,D,N) -> patch-to-patch cost (B,mlp[-1],N1)."""
        return self.forward_pm(_rows(xyz1), _rows(xyz2), _rows(points1), _rows(points2)).transpose(1, 2)

    def forward_pm(self, xyz1_t, xyz2_t, f1, f2):
        """Point-major cost volume.  xyz*_t (B,N,3), f1/f2 (B,N,D) -> (B,N,512).  The first conv
        over cat[f1, f2[idx], dxyz] is split by linearity into per-point GEMMs."""
        assert not self.bn
        return self._forward_blocks(xyz1_t, xyz2_t, f1, f2)


def _fc_blocks(self, xyz1_t, xyz2_t, f1, f2):
    """FeatureCorrelator on the fused blocks (own GEMMs, hoisted first conv, fused grouping)."""
    D1, D2 = f1.shape[2], f2.shape[2]
    K = self.nsample
    c0, c1, c2 = self.mlp_convs
    w0 = w2d(c0)
    # Two branches do not depend on the point-to-patch MLP: the per-point GEMM over the second cloud's features and the
    # whole neighbourhood side of the patch-to-patch stage (kNN in cloud 1, relative coordinates, WeightNet's hidden
    # layers).  They are small kernels; on side streams they run next to the large GEMMs of the main stream, and
    # autograd replays their backward on the same streams.
    side = self._side_streams() if (self.side_streams and xyz1_t.is_cuda) else None
    main = torch.cuda.current_stream() if side else None
    if side:
        for st in side:
            st.wait_stream(main)
        FB.stress_point(list(side) + [main])
    with torch.cuda.stream(side[0]) if side else contextlib.nullcontext():
        nbr = Neighbors(knn_point(K, xyz2_t, xyz1_t, i32=True), xyz2_t.shape[1])     # needed after p1: off the main stream too
        p2 = FB.stress_mark(FB.linear(f2, w0[:, D1:D1 + D2]))
    with torch.cuda.stream(side[1]) if side else contextlib.nullcontext():
        nbr2 = Neighbors(knn_point(K, xyz1_t, xyz1_t, i32=True), xyz1_t.shape[1])
        dxyz2 = rel_xyz(xyz1_t, xyz1_t, nbr2.idx)
        h2 = FB.stress_mark(self.weightnet2.hidden_pm(dxyz2))
    p1 = FB.linear(f1, w0[:, :D1], c0.bias)
    if side:
        main.wait_stream(side[0])
        p2.record_stream(main); nbr.idx.record_stream(main); f2.record_stream(side[0])
        xyz1_t.record_stream(side[0]); xyz2_t.record_stream(side[0])
    x, dxyz = FB.CostVolumeMLPFn.apply(xyz1_t, xyz2_t, p1, p2, nbr, w0[:, D1 + D2:], w2d(c1), c1.bias, w2d(c2), c2.bias, True)
    p2p = self.weightnet1.weighted_ksum(dxyz, x, None, True, c2.bias)                      # sum_k weights * x
    if side:
        main.wait_stream(side[1])
        xyz1_t.record_stream(side[1])
        for t in (h2, dxyz2, nbr2.idx):
            t.record_stream(main)
    return self.weightnet2.weighted_ksum(dxyz2, p2p, nbr2, False, hidden=h2)               # sum_k weights * p2p[idx]


def _fc_side_streams(self):
    if self._side is None:
        self._side = [FB.side_stream(0), FB.side_stream(1)]
    return self._side


FeatureCorrelator._side_streams = _fc_side_streams
FeatureCorrelator._side = None
FeatureCorrelator.side_streams = True


FeatureCorrelator._forward_blocks = _fc_blocks


class FlowHead(nn.Module):
    """radarflow_util.py:240-261"""

    def __init__(self, in_channel, mlp):
        super().__init__()
        self.sf_mlp = nn.ModuleList()
        last_channel = in_channel
        for out_channel in mlp:
            self.sf_mlp.append(nn.Sequential(nn.Conv2d(last_channel, out_channel, 1, bias=False),
                                             nn.BatchNorm2d(out_channel), nn.ReLU(inplace=False)))
            last_channel = out_channel
        self.conv2 = nn.Conv2d(mlp[-1], 3, 1, bias=False)

    def forward(self, feat):
        """Reference contract (:253-261): (B,C,N) -> (B,3,N)."""
        return self.forward_pm(feat.transpose(1, 2)).transpose(1, 2)

    def forward_pm(self, feat):
        """feat (B,N,512) -> (B,N,3)"""
        feat = FB.mlp_chain(feat, [(blk[0], blk[1]) for blk in self.sf_mlp], self.sf_mlp[0][1].training)
        return FB.linear(feat, w2d(self.conv2))


class MotionHead(nn.Module):
    """radarflow_util.py:263-285"""

    def __init__(self, in_channel, mlp):
        super().__init__()
        self.sf_mlp = nn.ModuleList()
        last_channel = in_channel
        for out_channel in mlp:
            self.sf_mlp.append(nn.Sequential(nn.Conv2d(last_channel, out_channel, 1, bias=False),
                                             nn.BatchNorm2d(out_channel), nn.ReLU(inplace=False)))
            last_channel = out_channel
        self.conv2 = nn.Conv2d(mlp[-1], 1, 1, bias=False)
        self.m = nn.Sigmoid()

    def forward(self, feat):
        """Reference contract (:276-285): (B,C,N) -> (B,1,N) in (0,1)."""
        return self.forward_pm(feat.transpose(1, 2)).transpose(1, 2)

    def forward_pm(self, feat):
        """feat (B,N,512) -> (B,N,1)"""
        feat = FB.mlp_chain(feat, [(blk[0], blk[1]) for blk in self.sf_mlp], self.sf_mlp[0][1].training)
        return FB.linear(feat, w2d(self.conv2), None, act=3)


FlowPredictor = FlowHead        # radarflow_util.py:388-409: same layers and parameter names as FlowHead (:240-261)


class FlowDecoder(nn.Module):
    """radarflow_util.py:321-350 (RaFlow): multi-scale propagation of the flow embeddings + flow predictor."""

    def __init__(self, fc_inch):
        super().__init__()
        ep_radius = [2.0, 4.0, 8.0, 16.0]
        ep_nsamples = [4, 8, 16, 32]
        ep_mlps = [fc_inch, int(fc_inch / 2), int(fc_inch / 8)]
        ep_mlp2s = [int(fc_inch / 8), int(fc_inch / 8), int(fc_inch / 8)]
        self.mse = MultiScaleEncoder(ep_radius, ep_nsamples, in_channel=fc_inch * 2 + 3, mlp=ep_mlps, mlp2=ep_mlp2s)
        sf_inch = len(ep_radius) * ep_mlp2s[-1] * 2
        self.fp = FlowPredictor(in_channel=sf_inch, mlp=[int(sf_inch / 2), int(sf_inch / 4), int(sf_inch / 8)])

    def forward(self, pc1, feature1, pc1_features, cor_features):
        """Reference contract (:339-350): (B,3,N), (B,3,N), (B,512,N), (B,512,N) -> flow (B,3,N)."""
        emb = torch.cat((_rows(feature1), _rows(pc1_features), _rows(cor_features)), dim=2)
        prop = self.mse.forward_pm(_rows(pc1), F.pad(emb, (0, -emb.shape[2] % 4)))
        return self.fp.forward_pm(FB.global_max_cat(prop)).transpose(1, 2)
