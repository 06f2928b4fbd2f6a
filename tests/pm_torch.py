"""Test fixture, not product code: the point-major path with its dense math through torch (F.linear, batch_norm, max, sum) on top
of the product's gather / neighbour-search kernels -- the cross-check the parity tests parametrise as "pm_torch".  It used to
live inside cmflow_amd/ behind `net.path = "pm_torch"`; the product package now contains the fused HIP path only.
install(net) replaces the forward bodies of the block modules of ONE model instance (same layout, same hoisting of the first
conv, same parameters) and returns the net."""
import types

import torch
import torch.nn.functional as F

from cmflow_amd import pointnet2_utils as pointutils
from cmflow_amd.fused import Neighbors, group_rows, w2d
from cmflow_amd import radarflow_util as RU


def bn_pm(bn, x):
    """BatchNorm2d semantics on a point-major tensor (..., C): statistics over every leading
    position (= over (B, N, ns) of the reference's (B,C,N,ns) tensor)."""
    shape = x.shape
    if bn.training and bn.track_running_stats:
        bn.num_batches_tracked.add_(1)
    y = F.batch_norm(x.reshape(-1, shape[-1]), bn.running_mean, bn.running_var, bn.weight, bn.bias,
                     bn.training, bn.momentum, bn.eps)
    return y.view(shape)


def _mse_forward_pm(self, xyz_t, feats, n_tail=0, n_grad=0):
    """MultiScaleEncoder: stacked feature half of the first convs as one F.linear, then each scale."""
    o1 = self.ms_ls[0].mlp_convs[0].weight.shape[0]
    cin = self.ms_ls[0].mlp_convs[0].weight.shape[1] - 3
    if n_tail:                      # permuted layout: back to the module's own channel order
        feats = torch.cat((feats[:, :, cin - n_tail:cin], feats[:, :, :cin - n_tail]), dim=2)
    feats = feats[:, :, :cin]
    wf_all = torch.cat([w2d(sa.mlp_convs[0])[:, 3:] for sa in self.ms_ls], dim=0)
    y_all = F.linear(feats, wf_all)                               # (B,N,4*o1)
    return torch.cat([sa.forward_pm(xyz_t, y_all[:, :, i * o1:(i + 1) * o1]) for i, sa in enumerate(self.ms_ls)], dim=2)


def _mse_forward(self, xyz, features):
    return self.forward_pm(RU._rows(xyz), RU._rows(features)).transpose(1, 2)


def _mse_forward_pm_pair(self, xyz1_t, feats1, xyz2_t, feats2):
    return self.forward_pm(xyz1_t, feats1), self.forward_pm(xyz2_t, feats2)


def _plf_forward(self, xyz, points):
    xyz_t, feats = RU._rows(xyz), RU._rows(points)
    return self.forward_pm(xyz_t, F.linear(feats, w2d(self.mlp_convs[0])[:, 3:])).transpose(1, 2)


def _plf_forward_pm(self, xyz_t, y):
    """PointLocalFeature (radarflow_util.py:144-162) with the feature half of the first conv already applied per point."""
    B, N, _ = xyz_t.shape
    idx = pointutils.ball_query(self.radius, self.nsample, xyz_t, xyz_t)
    nbr = Neighbors(idx, N)
    dxyz = group_rows(xyz_t, nbr) - xyz_t.unsqueeze(2)                # (B,N,ns,3) relative xyz
    x = group_rows(y, nbr) + F.linear(dxyz, w2d(self.mlp_convs[0])[:, :3])
    x = F.relu(bn_pm(self.mlp_bns[0], x))
    for conv, bn in zip(list(self.mlp_convs)[1:], list(self.mlp_bns)[1:]):
        x = F.relu(bn_pm(bn, F.linear(x, w2d(conv))))
    x = torch.max(x, dim=2)[0]                                        # over the ball
    for conv, bn in zip(self.mlp2_convs, self.mlp2_bns):
        x = F.relu(bn_pm(bn, F.linear(x, w2d(conv))))
    return x


def _wn_forward_pm(self, dxyz, preact_grad=False):
    assert not self.bn
    w = dxyz
    for conv in self.mlp_convs:
        w = F.relu(F.linear(w, w2d(conv), conv.bias))
    return w


def _fc_forward_pm(self, xyz1_t, xyz2_t, f1, f2):
    """FeatureCorrelator (radarflow_util.py:185-237), first conv split by linearity into per-point GEMMs."""
    assert not self.bn
    D1, D2 = f1.shape[2], f2.shape[2]
    K = self.nsample
    act = self.relu
    w0 = w2d(self.mlp_convs[0])
    nbr = Neighbors(RU.knn_point(K, xyz2_t, xyz1_t).int(), xyz2_t.shape[1])
    dxyz = group_rows(xyz2_t, nbr) - xyz1_t.unsqueeze(2)                               # (B,N1,K,3)
    p1 = F.linear(f1, w0[:, :D1], self.mlp_convs[0].bias)
    p2 = F.linear(f2, w0[:, D1:D1 + D2])
    x = act(p1.unsqueeze(2) + group_rows(p2, nbr) + F.linear(dxyz, w0[:, D1 + D2:]))
    for conv in list(self.mlp_convs)[1:]:
        x = act(F.linear(x, w2d(conv), conv.bias))
    p2p = torch.sum(self.weightnet1.forward_pm(dxyz) * x, dim=2)                      # (B,N1,512)
    nbr = Neighbors(RU.knn_point(K, xyz1_t, xyz1_t).int(), xyz1_t.shape[1])
    dxyz = group_rows(xyz1_t, nbr) - xyz1_t.unsqueeze(2)
    return torch.sum(self.weightnet2.forward_pm(dxyz) * group_rows(p2p, nbr), dim=2)


def _head_forward_pm(self, feat):
    for blk in self.sf_mlp:
        feat = F.relu(bn_pm(blk[1], F.linear(feat, w2d(blk[0]))))
    y = F.linear(feat, w2d(self.conv2))
    return torch.sigmoid(y) if hasattr(self, "m") else y


def _fd_forward(self, pc1, feature1, pc1_features, cor_features):
    emb = torch.cat((RU._rows(feature1), RU._rows(pc1_features), RU._rows(cor_features)), dim=2)
    prop = self.mse.forward_pm(RU._rows(pc1), emb)
    glob = prop.max(dim=1, keepdim=True)[0].expand(-1, prop.shape[1], -1)
    return self.fp.forward_pm(torch.cat((prop, glob), dim=2)).transpose(1, 2)


def _propagate(self, pc1, pc2, feature1, feature2):
    """CMFlow._propagate (cmflow.py:59-88) in point-major layout with torch ops."""
    x1, x2 = pc1.transpose(1, 2).contiguous(), pc2.transpose(1, 2).contiguous()
    a1, a2 = feature1.transpose(1, 2).contiguous(), feature2.transpose(1, 2).contiguous()
    f1 = self.mse_layer.forward_pm(x1, a1)
    f2 = self.mse_layer.forward_pm(x2, a2)
    f1 = torch.cat((f1, f1.max(dim=1, keepdim=True)[0].expand(-1, f1.shape[1], -1)), dim=2)
    f2 = torch.cat((f2, f2.max(dim=1, keepdim=True)[0].expand(-1, f2.shape[1], -1)), dim=2)
    cor = self.fc_layer.forward_pm(x1, x2, f1, f2)
    prop = self._second_encoder().forward_pm(x1, torch.cat((a1, f1, cor), dim=2))
    self.last = {"pc1_features": f1[:, :, :256].transpose(1, 2), "pc2_features": f2[:, :, :256].transpose(1, 2),
                 "cor_features": cor.transpose(1, 2), "prop_features": prop.transpose(1, 2)}
    return prop.transpose(1, 2)


def _backbone(self, pc1, pc2, feature1, feature2):
    prop_features = self._propagate(pc1, pc2, feature1, feature2)
    gfeat = torch.max(prop_features, -1)[0].unsqueeze(2).expand(-1, -1, pc1.size(2))
    return torch.cat((prop_features, gfeat), dim=1)


_BY_CLASS = {
    "MultiScaleEncoder": {"forward": _mse_forward, "forward_pm": _mse_forward_pm, "forward_pm_pair": _mse_forward_pm_pair},
    "PointLocalFeature": {"forward": _plf_forward, "forward_pm": _plf_forward_pm},
    "WeightNet": {"forward_pm": _wn_forward_pm},
    "FeatureCorrelator": {"forward_pm": _fc_forward_pm},
    "FlowHead": {"forward_pm": _head_forward_pm},
    "MotionHead": {"forward_pm": _head_forward_pm},
    "FlowDecoder": {"forward": _fd_forward},
}


def install(net):
    for m in net.modules():
        for name, fn in _BY_CLASS.get(type(m).__name__, {}).items():
            setattr(m, name, types.MethodType(fn, m))
    if hasattr(net, "_propagate"):
        net._propagate = types.MethodType(_propagate, net)
        net.Backbone = types.MethodType(_backbone, net)
        net.head_streams = False                      # the two heads one after the other on the caller's stream
    return net
