"""The slice of ``lib/pytorch_utils.py`` the PointNet++ modules need: ``SharedMLP`` (:5-33) built from
``Conv2d`` blocks (:36-100,177-207) -- 1x1 conv -> BatchNorm2d (wrapped once more, so parameter names read
``layer0.bn.bn.weight``) -> ReLU.  Same constructor arguments and the same state_dict keys, so checkpoints of
reference-built modules load unchanged.

``SharedMLP.forward`` on device tensors runs the library's kernels instead of ``nn.Conv2d`` / ``nn.BatchNorm2d``: the
``(B, C, npoint, nsample)`` input becomes a row-major ``[positions, channels]`` matrix and the [1x1 conv + BatchNorm + ReLU] layers
go through ``cmf_mlp_forward / _backward`` (fused_blocks.mlp_chain: GEMM with train-mode statistics, fold with the running-stat
update, hand-written backward), up to four layers per call; layers without BatchNorm (conv + bias + ReLU) go through ``cmf_gemm`` with
its bias / activation epilogue.  Pre-activation and instance-norm variants (unused by the reference's models) and BatchNorm widths
that are not a power of two keep the torch path.
"""
from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F


class BatchNorm2d(nn.Sequential):
    """lib/pytorch_utils.py:103-125: a Sequential around nn.BatchNorm2d (hence the doubled ``bn.bn`` key)."""

    def __init__(self, in_size: int, name: str = ""):
        super().__init__()
        self.add_module(name + "bn", nn.BatchNorm2d(in_size))
        nn.init.constant_(self[0].weight, 1.0)
        nn.init.constant_(self[0].bias, 0)


class Conv2d(nn.Sequential):
    """lib/pytorch_utils.py:36-100,177-207 (kernel (1,1), stride 1, no padding, kaiming-normal init)."""

    def __init__(self, in_size: int, out_size: int, *, kernel_size=(1, 1), stride=(1, 1), padding=(0, 0),
                 activation=nn.ReLU(inplace=True), bn: bool = False, init=nn.init.kaiming_normal_, bias: bool = True,
                 preact: bool = False, name: str = "", instance_norm: bool = False):
        super().__init__()
        bias = bias and (not bn)
        conv = nn.Conv2d(in_size, out_size, kernel_size=kernel_size, stride=stride, padding=padding, bias=bias)
        init(conv.weight)
        if bias:
            nn.init.constant_(conv.bias, 0)
        norm_size = in_size if preact else out_size

        def tail_or_head():
            if bn:
                self.add_module(name + 'bn', BatchNorm2d(norm_size))
            if activation is not None:
                self.add_module(name + 'activation', activation)
            if not bn and instance_norm:
                self.add_module(name + 'in', nn.InstanceNorm2d(norm_size, affine=False, track_running_stats=False))

        if preact:
            tail_or_head()
        self.add_module(name + 'conv', conv)
        if not preact:
            tail_or_head()


class SharedMLP(nn.Sequential):
    """lib/pytorch_utils.py:5-33"""

    def __init__(self, args: List[int], *, bn: bool = False, activation=nn.ReLU(inplace=True), preact: bool = False,
                 first: bool = False, name: str = "", instance_norm: bool = False):
        super().__init__()
        for i in range(len(args) - 1):
            plain = first and preact and i == 0
            self.add_module(name + 'layer{}'.format(i),
                            Conv2d(args[i], args[i + 1], bn=(not plain) and bn, activation=None if plain else activation,
                                   preact=preact, instance_norm=instance_norm))
        # the library's BN-backward passes tile rows x channels with C / 4 a power of two (csrc/pointwise.hip tile_ok): other widths
        # (e.g. 48, 96) keep the torch path for the whole stack
        pow2 = all(c >= 4 and c <= 1024 and (c & (c - 1)) == 0 for c in args[1:]) if bn else True
        self._native = (not preact) and (not instance_norm) and isinstance(activation, nn.ReLU) and pow2

    @staticmethod
    def _parts(layer):
        conv = bn = None
        for m in layer.children():
            if isinstance(m, nn.Conv2d):
                conv = m
            elif isinstance(m, BatchNorm2d):
                bn = m[0]
        return conv, bn

    def forward(self, x):
        if not (self._native and x.is_cuda and x.dim() == 4):
            return super().forward(x)
        from . import fused_blocks as FB
        B, C, P, S = x.shape
        rows = x.permute(0, 2, 3, 1).reshape(B * P * S, C)             # [positions, channels]
        rows = FB._pad_cols(rows) if C % 4 else rows.contiguous()
        layers = [self._parts(l) for l in self.children()]
        i = 0
        while i < len(layers):
            conv, bn = layers[i]
            w = conv.weight.view(conv.weight.shape[0], conv.weight.shape[1])
            if w.shape[1] != rows.shape[1]:                               # zero columns for the padded input channels
                w = F.pad(w, (0, rows.shape[1] - w.shape[1]))
            if bn is None:                                                # conv + bias + ReLU
                rows = FB.linear(rows, w, conv.bias, act=1)
                i += 1
                continue
            group = [(conv, bn, w)]                                       # up to four [conv + BN + ReLU] layers per library call
            while len(group) < 4 and i + len(group) < len(layers) and layers[i + len(group)][1] is not None:
                c2, b2 = layers[i + len(group)]
                group.append((c2, b2, c2.weight.view(c2.weight.shape[0], c2.weight.shape[1])))
            rows = FB.mlp_chain_w(rows, [(wt, b) for _, b, wt in group], group[0][1].training)
            i += len(group)
        return rows.view(B, P, S, rows.shape[1]).permute(0, 3, 1, 2).contiguous()
