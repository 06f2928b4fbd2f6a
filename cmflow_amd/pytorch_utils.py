"""The slice of ``lib/pytorch_utils.py`` the PointNet++ modules need: ``SharedMLP`` (:5-33) built from
``Conv2d`` blocks (:36-100,177-207) -- 1x1 conv -> BatchNorm2d (wrapped once more, so parameter names read
``layer0.bn.bn.weight``) -> ReLU.  Same constructor arguments and the same state_dict keys, so checkpoints of
reference-built modules load unchanged.
"""
from typing import List

import torch.nn as nn


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
