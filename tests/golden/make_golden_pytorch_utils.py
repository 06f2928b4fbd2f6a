#!/usr/bin/env python3
"""Golden vectors for lib/pytorch_utils.py (the classes CMFlow itself never builds): the REFERENCE's SharedMLP
(default, preact + first, without BN, odd widths), Conv1d, Conv2d, FC and BatchNorm1d run on torch-CPU with seeded weights -- outputs in eval
and train mode, the input / parameter gradients of a seeded loss and the BN buffers after the train-mode call.  Build container only
(imports /root/reference/lib/pytorch_utils.py, pure torch).

    python tests/golden/make_golden_pytorch_utils.py   ->  tests/golden/pytorch_utils_kat.npz
"""
import importlib.util
import os

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))

# name -> (class, positional args, keyword args, input shape)
CASES = {
    "mlp_default": ("SharedMLP", [[6, 16, 32]], dict(bn=True), (3, 6, 10, 4)),
    "mlp_preact_first": ("SharedMLP", [[6, 16, 32]], dict(bn=True, preact=True, first=True), (3, 6, 10, 4)),
    "mlp_preact": ("SharedMLP", [[8, 16]], dict(bn=True, preact=True), (3, 8, 10, 4)),
    "mlp_nobn": ("SharedMLP", [[6, 16, 8]], dict(bn=False), (3, 6, 10, 4)),
    "mlp_odd": ("SharedMLP", [[6, 12, 20]], dict(bn=True), (3, 6, 10, 4)),
    "conv1d_bn": ("Conv1d", [8, 16], dict(bn=True), (4, 8, 33)),
    "conv1d_preact": ("Conv1d", [8, 16], dict(bn=True, preact=True), (4, 8, 33)),
    "conv1d_plain": ("Conv1d", [8, 12], dict(activation=None), (4, 8, 33)),
    "conv2d_bias": ("Conv2d", [5, 8], dict(), (2, 5, 6, 7)),
    "fc_bn": ("FC", [8, 16], dict(bn=True), (37, 8)),
    "fc_bias": ("FC", [8, 12], dict(), (37, 8)),
    "fc_preact": ("FC", [8, 16], dict(bn=True, preact=True), (37, 8)),
    "fc_noact": ("FC", [8, 16], dict(activation=None), (37, 8)),
}


def build(mod, name, seed):
    """The module of CASES[name] from `mod` (the reference's or this repo's pytorch_utils) with seeded parameters / BN buffers."""
    cls, args, kw, shape = CASES[name]
    m = getattr(mod, cls)(*args, **kw)
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for k, p in sorted(m.named_parameters()):
            p.copy_(torch.randn(p.shape, generator=g) * (0.3 if p.dim() > 1 else 0.2) + (1.0 if (p.dim() == 1 and k.endswith("bn.weight")) else 0.0))
        for k, b in sorted(m.named_buffers()):
            if k.endswith("running_mean"):
                b.copy_(torch.randn(b.shape, generator=g) * 0.3)
            elif k.endswith("running_var"):
                b.copy_(torch.rand(b.shape, generator=g) * 1.5 + 0.5)
    x = torch.randn(shape, generator=g)
    wout = None
    return m, x, g


def run(m, x, g):
    """eval output; train output, gradients of sum(out * w) w.r.t. the input and every parameter, BN buffers afterwards."""
    res = {}
    m.eval()
    with torch.no_grad():
        res["eval"] = m(x.clone()).detach().clone()
    m.train()
    xi = x.clone().requires_grad_(True)
    y = m(xi)
    w = torch.randn(y.shape, generator=g)
    (y * w).sum().backward()
    res["train"], res["w"], res["dx"] = y.detach().clone(), w, xi.grad.detach().clone()
    for k, p in m.named_parameters():
        res["grad/" + k] = p.grad.detach().clone()
    for k, b in m.named_buffers():
        res["buf/" + k] = b.detach().clone()
    return res


def main():
    spec = importlib.util.spec_from_file_location("ref_pytorch_utils", "/root/reference/lib/pytorch_utils.py")
    R = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(R)
    out = {}
    for i, name in enumerate(sorted(CASES)):
        m, x, g = build(R, name, 100 + i)
        for k, v in m.state_dict().items():
            out["%s/state/%s" % (name, k)] = v.detach().clone().numpy()
        out["%s/x" % name] = x.numpy()
        for k, v in run(m, x, g).items():
            out["%s/%s" % (name, k)] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "pytorch_utils_kat.npz"), **out)
    print(len(out), "arrays;", {n: CASES[n][3] for n in sorted(CASES)})


if __name__ == "__main__":
    main()
