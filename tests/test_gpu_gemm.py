"""GPU: the fp32 MFMA GEMM (cmf_gemm) against a torch fp32/fp64 reference of the same op.

Floating-point kernel => tolerance, stated per check: fp32 MFMA is an exact fmaf chain, the only
difference to torch is summation order: |err| <= 2e-6 * sum|a*b| (checked against an fp64 product).
"""
import os

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    torch.backends.cuda.matmul.allow_tf32 = False
    return torch.device("cuda:0")


def _check(C, ref64, absprod64, tol=2e-6):
    err = (C.double() - ref64).abs()
    bound = tol * absprod64 + 1e-6
    assert bool((err <= bound).all()), float((err / bound).max())


SHAPES = [(256, 128, 64), (1000, 192, 100), (128, 64, 32), (130, 32, 1027 + 1), (4096, 256, 512), (77, 40, 8),
          (16384, 512, 1028), (300, 2048, 96)]


@pytest.mark.parametrize("M,N,K", SHAPES)
@pytest.mark.parametrize("a_t,b_t", [(False, True), (False, False), (True, False), (True, True)])
def test_gemm_layouts(dev, M, N, K, a_t, b_t):
    from cmflow_amd.fused import gemm
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    ldk = (K + 3) // 4 * 4 + 4                      # padded row strides (multiples of 4)
    ldm = (M + 3) // 4 * 4
    ldn = (N + 3) // 4 * 4 + 8
    A = torch.randn(M, K, generator=g).to(dev)
    B = torch.randn(K, N, generator=g).to(dev)      # asymmetric operands: a transposed write cannot pass
    if a_t:
        Abuf = torch.zeros(K, ldm, device=dev); Abuf[:, :M] = A.t(); Aarg = Abuf[:, :M]
    else:
        Abuf = torch.zeros(M, ldk, device=dev); Abuf[:, :K] = A; Aarg = Abuf[:, :K]
    if b_t:
        Bbuf = torch.zeros(N, ldk, device=dev); Bbuf[:, :K] = B.t(); Barg = Bbuf[:, :K]
    else:
        Bbuf = torch.zeros(K, ldn, device=dev); Bbuf[:, :N] = B; Barg = Bbuf[:, :N]
    C = gemm(Aarg, Barg, a_t=a_t, b_t=b_t)
    _check(C, A.double() @ B.double(), A.double().abs() @ B.double().abs())


def test_gemm_prologue_epilogue_stats(dev):
    from cmflow_amd.fused import gemm
    g = torch.Generator().manual_seed(0)
    M, N, K = 1000, 192, 96
    A, W = torch.randn(M, K, generator=g).to(dev), torch.randn(N, K, generator=g).to(dev)
    pa, pc = (torch.rand(K, generator=g) + 0.5).to(dev), torch.randn(K, generator=g).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    Ap = torch.relu(A * pa + pc)
    for act, fn in ((0, lambda x: x), (1, torch.relu), (2, lambda x: torch.nn.functional.leaky_relu(x, 0.1)),
                    (3, torch.sigmoid)):
        C, st = gemm(A, W, pro=(pa, pc), bias=bias, act=act, stats=True)
        ref = fn(Ap.double() @ W.double().t() + bias.double())
        np.testing.assert_allclose(C.cpu().numpy(), ref.float().cpu().numpy(), rtol=2e-5, atol=2e-5)
        s = st.double().sum(dim=0)                   # reduce the per-row-tile partials
        np.testing.assert_allclose(s[0].cpu().numpy(), ref.sum(0).cpu().numpy(), rtol=1e-4, atol=1e-3)
        np.testing.assert_allclose(s[1].cpu().numpy(), (ref * ref).sum(0).cpu().numpy(), rtol=1e-4, atol=1e-3)
    # write into a column slice of a wider buffer (concat without a copy) and accumulate
    wide = torch.zeros(M, 3 * N, device=dev)
    gemm(A, W, out=wide[:, N:2 * N])
    gemm(A, W, out=wide[:, N:2 * N], accumulate=True)
    np.testing.assert_allclose(wide[:, N:2 * N].cpu().numpy(), 2 * (A @ W.t()).cpu().numpy(), rtol=2e-5, atol=2e-4)
    assert float(wide[:, :N].abs().max()) == 0 and float(wide[:, 2 * N:].abs().max()) == 0


def test_gemm_backward_modes_and_splitk(dev):
    from cmflow_amd.fused import gemm
    g = torch.Generator().manual_seed(1)
    M, N, K = 2048, 128, 256
    dZ = torch.randn(M, N, generator=g).to(dev)          # grad wrt layer output
    W = torch.randn(N, K, generator=g).to(dev)           # layer weight (out=N, in=K)
    Zp = torch.randn(M, K, generator=g).to(dev)          # producer layer's pre-BN output
    ea, ec = (torch.rand(K, generator=g) + 0.5).to(dev), (0.3 * torch.randn(K, generator=g)).to(dev)
    mean, invstd = torch.randn(K, generator=g).to(dev), (torch.rand(K, generator=g) + 0.5).to(dev)
    # dX = dZ @ W masked by the producer's BN+ReLU, with the two BN-backward column sums
    dU, st = gemm(dZ, W, b_t=False, bwd=(1, Zp, ea, ec, mean, invstd))
    ref = (dZ.double() @ W.double()) * ((ea * Zp + ec) > 0)
    np.testing.assert_allclose(dU.cpu().numpy(), ref.float().cpu().numpy(), rtol=2e-5, atol=2e-4)
    s = st.double().sum(0)
    np.testing.assert_allclose(s[0].cpu().numpy(), ref.sum(0).cpu().numpy(), rtol=1e-4, atol=2e-3)
    np.testing.assert_allclose(s[1].cpu().numpy(), (ref * ((Zp - mean) * invstd).double()).sum(0).cpu().numpy(),
                               rtol=1e-4, atol=5e-3)
    dL = gemm(dZ, W, b_t=False, bwd=(2, Zp))
    ref2 = (dZ.double() @ W.double()) * torch.where(Zp > 0, 1.0, 0.1)
    np.testing.assert_allclose(dL.cpu().numpy(), ref2.float().cpu().numpy(), rtol=2e-5, atol=2e-4)
    # dW = dZ^T @ relu(a*Zp + c): contraction over M, split-K, activated B operand
    X = torch.relu(ea * Zp + ec)
    ref3 = dZ.double().t() @ X.double()
    for split in (1, 4, 16):
        dW = gemm(dZ, Zp, a_t=True, b_t=False, prob=(ea, ec), split_k=split)
        _check(dW, ref3, dZ.double().abs().t() @ X.double().abs(), tol=4e-6)
    dW2 = gemm(dZ, Zp, a_t=True, b_t=False, prob=(ea, ec), split_k=8)
    assert torch.equal(dW, dW) and torch.equal(dW2, gemm(dZ, Zp, a_t=True, b_t=False, prob=(ea, ec), split_k=8))


def test_gemm_throughput_report(dev):
    """Not a pass/fail perf gate: prints achieved TFLOP/s for the model's dominant shapes."""
    from cmflow_amd.fused import gemm
    for M, N, K in ((524288, 256, 512), (16384, 2048, 1028), (131072, 512, 512), (524288, 64, 256)):
        A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev)
        gemm(A, W)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            gemm(A, W)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        e0.record()
        for _ in range(5):
            torch.matmul(A, W.t())
        e1.record(); torch.cuda.synchronize()
        ms_t = e0.elapsed_time(e1) / 5
        print("gemm M=%d N=%d K=%d: %.3f ms = %.1f TFLOP/s (torch/hipBLASLt %.3f ms = %.1f)" %
              (M, N, K, ms, 2e-9 * M * N * K / ms, ms_t, 2e-9 * M * N * K / ms_t))


def test_gemm_backward_throughput_report(dev):
    """Prints TFLOP/s of the two backward GEMM forms on the dominant shapes."""
    from cmflow_amd.fused import gemm
    from cmflow_amd.fused_blocks import gemm_dw
    for M, N, K in ((524288, 256, 512), (131072, 512, 512)):
        dZ = torch.randn(M, N, device=dev); W = torch.randn(N, K, device=dev); X = torch.randn(M, K, device=dev)
        for name, fn in (("dX = dZ @ W", lambda: gemm(dZ, W, b_t=False)), ("dW = dZ^T @ X", lambda: gemm_dw(dZ, X))):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            print("%s M=%d N=%d K=%d: %.3f ms = %.1f TFLOP/s" % (name, M, N, K, ms, 2e-9 * M * N * K / ms))


@pytest.mark.parametrize("M,N,K", [(65536, 32, 32), (1000, 64, 64), (16384, 64, 32), (130, 32, 64), (524288, 64, 32), (77, 32, 8)])
def test_thin_gemm_forward_and_dx(dev, M, N, K):
    """Narrow layers (<= 64 channels) take the barrier-free per-wave kernels of thin_gemm.hip."""
    from cmflow_amd.fused import gemm
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(dev); W = torch.randn(N, K, generator=g).to(dev)
    pa, pc = (torch.rand(K, generator=g) + 0.5).to(dev), torch.randn(K, generator=g).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    C, st = gemm(A, W, pro=(pa, pc), bias=bias, act=1, stats=True)
    ref = torch.relu(torch.relu(A * pa + pc).double() @ W.double().t() + bias.double())
    np.testing.assert_allclose(C.cpu().numpy(), ref.float().cpu().numpy(), rtol=2e-5, atol=2e-5)
    s = st.double().sum(0)
    np.testing.assert_allclose(s[0].cpu().numpy(), ref.sum(0).cpu().numpy(), rtol=1e-4, atol=1e-2)
    np.testing.assert_allclose(s[1].cpu().numpy(), (ref * ref).sum(0).cpu().numpy(), rtol=1e-4, atol=1e-2)
    # data gradient with the BN+ReLU mask, the two BN sums and the three dxyz sums
    dZ = torch.randn(M, N, generator=g).to(dev)
    Zp = torch.randn(M, K, generator=g).to(dev)
    ea, ec = (torch.rand(K, generator=g) + 0.5).to(dev), (0.3 * torch.randn(K, generator=g)).to(dev)
    mean, invstd = torch.randn(K, generator=g).to(dev), (torch.rand(K, generator=g) + 0.5).to(dev)
    dxyz = torch.randn(M, 4, generator=g).to(dev)
    dU, st = gemm(dZ, W, b_t=False, bwd=(1, Zp, ea, ec, mean, invstd, dxyz))
    refu = (dZ.double() @ W.double()) * ((ea * Zp + ec) > 0)
    np.testing.assert_allclose(dU.cpu().numpy(), refu.float().cpu().numpy(), rtol=2e-5, atol=2e-4)
    s = st.double().sum(0)
    np.testing.assert_allclose(s[0].cpu().numpy(), refu.sum(0).cpu().numpy(), rtol=1e-4, atol=2e-2)
    np.testing.assert_allclose(s[1].cpu().numpy(), (refu * ((Zp - mean) * invstd).double()).sum(0).cpu().numpy(), rtol=1e-4, atol=5e-2)
    for k in range(3):
        np.testing.assert_allclose(s[2 + k].cpu().numpy(), (refu * dxyz[:, k:k + 1].double()).sum(0).cpu().numpy(), rtol=1e-4, atol=5e-2)


@pytest.mark.parametrize("M,N,K", [(65536, 32, 32), (16384, 64, 64), (1000, 64, 32), (524288, 32, 64)])
def test_thin_gemm_weight_gradient(dev, M, N, K):
    from cmflow_amd.fused_blocks import gemm_dw
    g = torch.Generator().manual_seed(M + N + K + 1)
    dZ = torch.randn(M, N, generator=g).to(dev); Zp = torch.randn(M, K, generator=g).to(dev)
    ea, ec = (torch.rand(K, generator=g) + 0.5).to(dev), (0.3 * torch.randn(K, generator=g)).to(dev)
    X = torch.relu(ea * Zp + ec)
    dW = gemm_dw(dZ, Zp, prob=(ea, ec))
    _check(dW, dZ.double().t() @ X.double(), dZ.double().abs().t() @ X.double().abs(), tol=4e-6)
    assert torch.equal(dW, gemm_dw(dZ, Zp, prob=(ea, ec)))            # deterministic
    # the same slabs accumulated into a row-strided buffer (gradient sinks: the [:, 3:] block of a conv weight's .grad)
    from cmflow_amd.fused import gemm
    from cmflow_amd.fused_blocks import dw_split
    base = torch.randn(N, K + 4, generator=g).to(dev)
    acc = base.clone()
    gemm(dZ, Zp, a_t=True, b_t=False, prob=(ea, ec), split_k=dw_split(M, N, K), out=acc[:, 4:], accumulate=True)
    assert torch.equal(acc[:, :4], base[:, :4])
    np.testing.assert_allclose(acc[:, 4:].cpu().numpy(), (base[:, 4:] + dW).cpu().numpy(), rtol=1e-6, atol=1e-4 * float(dW.abs().max()))


@pytest.mark.parametrize("rows,cout,cin", [(65536, 64, 32), (4096, 32, 32), (16384, 64, 64), (1000, 64, 64), (333, 24, 20), (128 * 1030 + 5, 32, 64)])
@pytest.mark.parametrize("train,in_mode,dxyz", [(True, 1, False), (True, 1, True), (True, 0, False), (False, 1, False)])
def test_thin_bwd_layer_matches_fp64_and_three_kernel_form(dev, rows, cout, cin, train, in_mode, dxyz):
    """cmf_thin_bwd_layer (BN backward + weight gradient + masked data gradient in one pass) against (a) the fp64
    formulas of its header comment, tolerance 2e-6 * sum|terms|, and (b) the three kernels it replaces
    (cmf_bn_bwd_apply, cmf_gemm a_t, cmf_gemm with the backward epilogue): the masks must agree exactly."""
    import ctypes
    from cmflow_amd import _lib
    from cmflow_amd.fused import gemm
    L = _lib.lib()
    g = torch.Generator(device="cpu").manual_seed(rows + cout * 7 + cin)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
    dU, z, x = rnd(rows, cout), rnd(rows, cout), rnd(rows, cin)
    w = rnd(cout, cin) * 0.2
    a, mean, invstd = rnd(cout), rnd(cout) * 0.3, torch.rand(cout, generator=g).to(dev) + 0.5
    sums = rnd(2, cout) * (rows ** 0.5) if train else None
    a_in, c_in, mean_in, invstd_in = rnd(cin), rnd(cin) * 0.3, rnd(cin) * 0.3, torch.rand(cin, generator=g).to(dev) + 0.5
    dxy = torch.cat((rnd(rows, 3), torch.zeros(rows, 1, device=dev)), 1).contiguous() if dxyz else None
    tpw = ctypes.c_int()
    nslab = L.cmf_thin_bwd_slabs(rows, ctypes.addressof(tpw))
    assert nslab * tpw.value * 128 >= rows and nslab <= max(2, min(rows // 128, 1024))
    tiles = (rows + 127) // 128
    nstat = 5 if dxyz else 2
    dx = torch.full((rows, cin), float("nan"), device=dev)
    stats = torch.full((tiles, nstat, cin), float("nan"), device=dev)
    dw0 = rnd(cout, cin)
    dw = dw0.clone()
    slabs = torch.empty(nslab, cout, cin, device=dev)
    p = lambda t: None if t is None else t.data_ptr()
    _lib.check(L.cmf_thin_bwd_layer(rows, cout, cin, p(dU), cout, p(z), cout, p(a), p(mean), p(invstd), p(sums), p(w), cin, p(x), cin,
                                    in_mode, p(a_in), p(c_in), p(mean_in), p(invstd_in), p(dxy), p(dx), cin,
                                    p(stats) if in_mode else None, p(dw), cin, 1, p(slabs), _lib.stream_ptr()), "cmf_thin_bwd_layer")
    # (a) fp64
    D = lambda t: t.double()
    if train:
        dZ = D(a) * (D(dU) - D(sums[0]) / rows - (D(z) - D(mean)) * D(invstd) * (D(sums[1]) / rows))
    else:
        dZ = D(a) * D(dU)
    act = torch.relu(D(a_in) * D(x) + D(c_in)) if in_mode else D(x)
    ref_dw = D(dw0) + dZ.t() @ act
    abs_dw = D(dw0).abs() + dZ.abs().t() @ act.abs()
    _check(dw, ref_dw, abs_dw, tol=4e-6)
    raw = dZ @ D(w)
    absraw = dZ.abs() @ D(w).abs()
    if in_mode:
        mask = (D(a_in) * D(x) + D(c_in)) > 0
        ref_dx = torch.where(mask, raw, torch.zeros_like(raw))
    else:
        ref_dx = raw
    edge = (D(a_in) * D(x) + D(c_in)).abs() < 1e-5 if in_mode else torch.zeros_like(raw, dtype=torch.bool)   # mask undecidable in fp64
    err = (D(dx) - ref_dx).abs()
    bound = 4e-6 * absraw + 1e-6
    assert bool(((err <= bound) | edge).all())
    if in_mode:
        got = D(dx)
        zh = (D(x) - D(mean_in)) * D(invstd_in)
        want = [got, got * zh] + ([got * D(dxy[:, k:k + 1]) for k in range(3)] if dxyz else [])
        for which, t in enumerate(want):
            pad = torch.zeros(tiles * 128 - rows, cin, dtype=torch.float64, device=dev)
            per_tile = torch.cat((t, pad)).view(tiles, 128, cin)
            _check(stats[:, which], per_tile.sum(1), per_tile.abs().sum(1), tol=4e-6)
    # (b) the three kernels
    dZ32 = dU.clone()
    _lib.check(L.cmf_bn_bwd_apply(rows, cout, p(dZ32), p(z), cout, p(a), p(mean), p(invstd), p(sums), _lib.stream_ptr()), "apply")
    if in_mode:
        dx3, part3 = gemm(dZ32, w, b_t=False, bwd=(1, x, a_in, c_in, mean_in, invstd_in) + ((dxy,) if dxyz else ()))
        assert float(((dx3 == 0) != (dx == 0)).float().mean()) < 1e-5
    else:
        dx3 = gemm(dZ32, w, b_t=False)
    assert float((dx3 - dx).abs().max()) <= 1e-4 * max(1.0, float(dx3.abs().max()))


@pytest.mark.parametrize("n_tail,cin,Kp", [(0, 3, 4), (0, 64, 64), (4, 1028, 1040), (2, 10, 12)])
def test_stacked_first_conv_weight_gather_and_gradient_scatter(dev, n_tail, cin, Kp):
    """cmf_stack_first_conv / cmf_unstack_first_conv_grad against the slice / cat / pad expressions they replace
    (radarflow_util.py:132-139 by linearity): pure data movement, so bit-exact."""
    import ctypes
    from cmflow_amd import _lib
    L = _lib.lib()
    n_w, o1 = 4, 32
    g = torch.Generator(device="cpu").manual_seed(cin)
    ws = [torch.randn(o1, cin + 3, 1, 1, generator=g).to(dev) for _ in range(n_w)]
    parts = []
    for w in ws:
        w2 = w.view(o1, cin + 3)[:, 3:]
        parts.append(torch.cat((w2[:, n_tail:], w2[:, :n_tail]), dim=1))
    want = torch.nn.functional.pad(torch.cat(parts, 0), (0, Kp - cin))
    wf = torch.full((n_w * o1, Kp), float("nan"), device=dev)
    ptrs = (ctypes.c_void_p * n_w)(*[w.data_ptr() for w in ws])
    _lib.check(L.cmf_stack_first_conv(n_w, o1, cin, n_tail, Kp, ctypes.addressof(ptrs), wf.data_ptr(), _lib.stream_ptr()), "stack")
    assert torch.equal(wf, want)
    dwf = torch.randn(n_w * o1, Kp, generator=g).to(dev)
    grads = [torch.randn(o1, cin + 3, 1, 1, generator=g).to(dev) for _ in range(n_w)]
    ref = [t.clone() for t in grads]
    for i, r in enumerate(ref):
        blk = dwf[i * o1:(i + 1) * o1]
        r2 = r.view(o1, cin + 3)
        r2[:, 3 + n_tail:] += blk[:, :cin - n_tail]
        if n_tail:
            r2[:, 3:3 + n_tail] += blk[:, cin - n_tail:cin]
    gp = (ctypes.c_void_p * n_w)(*[t.data_ptr() for t in grads])
    _lib.check(L.cmf_unstack_first_conv_grad(n_w, o1, cin, n_tail, Kp, dwf.data_ptr(), ctypes.addressof(gp), _lib.stream_ptr()), "unstack")
    for a, b in zip(grads, ref):
        assert torch.equal(a, b)


@pytest.mark.parametrize("P,S,cout,cin,train", [(4096, 4, 64, 32, True), (2048, 32, 64, 32, True), (1024, 8, 64, 64, True),
                                               (2048, 16, 32, 32, False)])
def test_pooled_thin_bwd_layer_matches_materialised_gradient(dev, P, S, cout, cin, train):
    """cmf_maxpool_bwd_point + cmf_thin_bwd_layer_pooled (the gradient of the pooled tensor kept per point) against
    cmf_maxpool_bwd + cmf_thin_bwd_layer on the materialised [P*S, C] gradient: same masks, same non-zero entries; the
    BN-backward sums are grouped differently (per 128 points vs per 128 rows), hence tolerances of a few fp32 ulps."""
    import ctypes
    from cmflow_amd import _lib, fused_blocks as FB
    L = _lib.lib()
    M = P * S
    g = torch.Generator(device="cpu").manual_seed(P + S)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
    z3, z2, dout = rnd(M, cout), rnd(M, cin), rnd(P, cout)
    w = rnd(cout, cin) * 0.2
    st = FB.BNState()
    st.a, st.c, st.mean, st.invstd = rnd(cout), rnd(cout) * 0.3, rnd(cout) * 0.3, torch.rand(cout, generator=g).to(dev) + 0.5
    st.training, st.count = train, M
    a_in, c_in, mean_in, invstd_in = rnd(cin), rnd(cin) * 0.3, rnd(cin) * 0.3, torch.rand(cin, generator=g).to(dev) + 0.5
    _, am = FB.bn_relu_maxpool(z3.view(P, S, cout), st)
    p = lambda t: None if t is None else t.data_ptr()
    tiles = (M + 127) // 128
    nslab = L.cmf_thin_bwd_slabs(M, None)
    slabs = torch.empty(max(nslab, 2), cout, cin, device=dev)

    def layer(pooled):
        dx = torch.empty(M, cin, device=dev); stats = torch.empty(tiles, 2, cin, device=dev); dw = torch.zeros(cout, cin, device=dev)
        if pooled:
            gp = torch.empty(P, cout, device=dev); part = torch.empty((P + 127) // 128, 2, cout, device=dev)
            _lib.check(L.cmf_maxpool_bwd_point(P, S, cout, p(dout), cout, p(z3), p(st.a), p(st.c), p(st.mean), p(st.invstd), p(am), p(gp),
                                               p(part), _lib.stream_ptr()), "maxpool_bwd_point")
            sums = FB.colsum_n(part)
            _lib.check(L.cmf_thin_bwd_layer_pooled(P, S, cout, cin, p(gp), p(am), p(z3), p(st.a), p(st.mean), p(st.invstd),
                                                   p(sums) if train else None, p(w), p(z2), p(a_in), p(c_in), p(mean_in), p(invstd_in),
                                                   p(dx), p(stats), p(dw), 0, p(slabs), _lib.stream_ptr()), "pooled")
        else:
            dU, part = FB.maxpool_bwd(dout, z3.view(P, S, cout), st, am)
            sums = FB.colsum_n(part)
            _lib.check(L.cmf_thin_bwd_layer(M, cout, cin, p(dU), cout, p(z3), cout, p(st.a), p(st.mean), p(st.invstd),
                                            p(sums) if train else None, p(w), cin, p(z2), cin, 1, p(a_in), p(c_in), p(mean_in),
                                            p(invstd_in), None, p(dx), cin, p(stats), p(dw), cin, 0, p(slabs), _lib.stream_ptr()), "dense")
        return sums, dx, stats, dw

    s0, dx0, st0, dw0 = layer(False)
    s1, dx1, st1, dw1 = layer(True)
    scale = lambda t: max(1.0, float(t.abs().max()))
    assert float((s0 - s1).abs().max()) <= 2e-6 * P               # sums of ~P non-zero terms of magnitude ~1, regrouped
    assert torch.equal(dx0 == 0, dx1 == 0)
    for a, b in ((dx0, dx1), (st0, st1), (dw0, dw1)):
        assert float((a - b).abs().max()) <= 2e-5 * scale(a)


@pytest.mark.parametrize("P,S,cin,train,pooled", [(2048, 8, 256, True, True), (1024, 32, 128, True, True), (4096, 4, 256, False, True),
                                                  (2048, 16, 256, True, False), (512, 4, 384, True, False)])
def test_wide_thin_bwd_layer_matches_three_kernel_form(dev, P, S, cin, train, pooled):
    """cmf_thin_bwd_wide_layer (64 <- cin in {128, 256, ...}: the 256 -> 64 conv of a second-encoder block, backward in one
    pass, optionally straight from the per-point max-pool gradient) against max-pool backward + BN backward in place + the
    two tiled GEMMs it replaces.  Same masks; sums regrouped: tolerance of a few fp32 ulps of the accumulated magnitudes."""
    import ctypes
    from cmflow_amd import _lib, fused_blocks as FB
    from cmflow_amd.fused import gemm
    L = _lib.lib()
    M, cout = P * S, 64
    g = torch.Generator(device="cpu").manual_seed(P + S + cin)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
    z3, z2, dout = rnd(M, cout), rnd(M, cin), rnd(P, cout)
    w = rnd(cout, cin) * 0.1
    st = FB.BNState()
    st.a, st.c, st.mean, st.invstd = rnd(cout), rnd(cout) * 0.3, rnd(cout) * 0.3, torch.rand(cout, generator=g).to(dev) + 0.5
    st.training, st.count = train, M
    a_in, c_in, mean_in, invstd_in = rnd(cin), rnd(cin) * 0.3, rnd(cin) * 0.3, torch.rand(cin, generator=g).to(dev) + 0.5
    _, am = FB.bn_relu_maxpool(z3.view(P, S, cout), st)
    p = lambda t: None if t is None else t.data_ptr()
    # reference: the three-kernel form on the materialised gradient
    dU, part = FB.maxpool_bwd(dout, z3.view(P, S, cout), st, am)
    sums0 = FB.colsum_n(part)
    dZ = dU.clone()
    _lib.check(L.cmf_bn_bwd_apply(M, cout, p(dZ), p(z3), cout, p(st.a), p(st.mean), p(st.invstd), p(sums0) if train else None,
                                  _lib.stream_ptr()), "apply")
    dw0 = FB.gemm_dw(dZ, z2, prob=(a_in, c_in))
    dx0, st0 = gemm(dZ, w, b_t=False, bwd=(1, z2, a_in, c_in, mean_in, invstd_in))
    # fused
    tiles = M // 128
    nslab = L.cmf_thin_bwd_wide_slabs(M, cin, None)
    slabs = torch.empty(nslab, cout, cin, device=dev)
    dx1 = torch.full((M, cin), float("nan"), device=dev); st1 = torch.full((tiles, 2, cin), float("nan"), device=dev)
    dw1 = torch.zeros(cout, cin, device=dev)
    if pooled:
        gp = torch.empty(P, cout, device=dev); part1 = torch.empty((P + 127) // 128, 2, cout, device=dev)
        _lib.check(L.cmf_maxpool_bwd_point(P, S, cout, p(dout), cout, p(z3), p(st.a), p(st.c), p(st.mean), p(st.invstd), p(am), p(gp),
                                           p(part1), _lib.stream_ptr()), "maxpool_bwd_point")
        sums1 = FB.colsum_n(part1)
        src = (None, cout, p(gp), p(am), S)
    else:
        sums1 = sums0
        src = (p(dU), cout, None, None, 0)
    _lib.check(L.cmf_thin_bwd_wide_layer(M, cin, *src, p(z3), cout, p(st.a), p(st.mean), p(st.invstd), p(sums1) if train else None,
                                         p(w), cin, p(z2), cin, p(a_in), p(c_in), p(mean_in), p(invstd_in), p(dx1), cin, p(st1), p(dw1), cin, 0,
                                         p(slabs), _lib.stream_ptr()), "wide")
    scale = lambda t: max(1.0, float(t.abs().max()))
    assert torch.equal(dx0 == 0, dx1 == 0)
    assert float((dx0 - dx1).abs().max()) <= 2e-5 * scale(dx0)
    assert float((st0 - st1).abs().max()) <= 2e-5 * scale(st0)
    assert float((dw0 - dw1).abs().max()) <= 3e-5 * scale(dw0)


@pytest.mark.parametrize("N,S", [(40, 16), (48, 16), (56, 16), (24, 32), (250, 4)])
def test_setconv_block_small_m_matches_python_sequence_and_stays_inside_its_arena(dev, N, S, monkeypatch):
    """A second-encoder set-conv block (512 -> 256 -> 64 | 64 -> 64 -> 64) whose neighbourhood matrix has 640 ... 1000
    rows: the fused wide backward layer writes tiles128(M) weight-gradient slabs of 64 x 256 floats, more than the split-K
    rule of the tiled kernels asks for at that size -- the arena of cmf_setconv_backward must be sized for them (round-2
    advisor finding).  The block call (one C-ABI call per direction) against the same kernels sequenced from Python
    (SetConvFn, separately allocated buffers), and a guard band behind the block's scratch arena that must stay untouched."""
    from cmflow_amd import fused_blocks as FB
    from cmflow_amd.radarflow_util import PointLocalFeature
    torch.manual_seed(N * 100 + S)
    B = 1
    mod = PointLocalFeature(4.0, S, in_channel=13, mlp=[512, 256, 64], mlp2=[64, 64, 64]).to(dev).train()
    xyz = (torch.rand(B, N, 3, device=dev) * torch.tensor([12.0, 12.0, 2.0], device=dev)).contiguous()
    y0 = torch.randn(B, N, 512, device=dev)
    dout = torch.randn(B, N, 64, device=dev)
    res = {}
    guard = {}
    real_empty = torch.empty
    for block in (False, True):
        monkeypatch.setattr(FB, "USE_BLOCK_CALLS", block)
        mod.zero_grad(set_to_none=True)
        y = y0.clone().requires_grad_(True)
        out = FB.set_conv(mod, xyz, y)
        if block:
            n_bwd = out.grad_fn.state["n_bwd"]
            big = torch.full((n_bwd + (1 << 18),), 12345.0, device=dev)

            def fake_empty(*a, **k):
                if len(a) == 1 and a[0] == n_bwd and k.get("dtype") == torch.float32:
                    return big[:n_bwd]
                return real_empty(*a, **k)
            monkeypatch.setattr(torch, "empty", fake_empty)
            guard["big"], guard["n"] = big, n_bwd
        out.backward(dout)
        monkeypatch.setattr(torch, "empty", real_empty)
        torch.cuda.synchronize()
        res[block] = (out.detach().clone(), y.grad.clone(), {k: p.grad.clone() for k, p in mod.named_parameters() if p.grad is not None})
    assert bool((guard["big"][guard["n"]:] == 12345.0).all()), "cmf_setconv_backward wrote past its scratch arena"
    a, b = res[False], res[True]
    assert torch.equal(a[0], b[0])
    scale = lambda t: max(1e-6, float(t.abs().max()))
    assert float((a[1] - b[1]).abs().max()) <= 1e-5 * scale(a[1])
    assert set(a[2]) == set(b[2]) and len(a[2]) >= 17
    for k in a[2]:
        assert float((a[2][k] - b[2][k]).abs().max()) <= 2e-5 * scale(a[2][k]), k


def test_gemm_relu_epilogues_select_on_non_finite_values(dev):
    """ReLU / masks in the fast epilogues are selects, not multiplications by a zero slope: a pre-activation of -inf gives 0
    (0 * -inf would be NaN) and an infinite gradient at a masked position gives 0, in interior tiles (fast path) and edge
    tiles (generic loop) alike -- the same values torch.relu and a boolean mask produce (round-2 advisor finding)."""
    from cmflow_amd.fused import gemm
    g = torch.Generator().manual_seed(9)
    M, N, K = 300, 256, 64                                           # two interior row tiles + one edge tile
    A = torch.randn(M, K, generator=g)
    A[5] = float("-inf"); A[200] = float("-inf"); A[290] = float("-inf")
    W = torch.rand(N, K, generator=g) + 0.1                          # positive: the row sums are -inf, not inf - inf
    got = gemm(A.to(dev), W.to(dev), act=1).cpu()
    want = torch.relu(A @ W.t())
    assert torch.isfinite(got).all() and float(got[5].abs().max()) == 0.0 and float(got[290].abs().max()) == 0.0
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-4, atol=1e-4)
    got2, _ = gemm(A.to(dev), W.to(dev), act=1, stats=True)          # the direct forward epilogue with statistics
    assert torch.equal(got2.cpu(), got)
    # backward through ReLU (mode 3): an infinite upstream gradient where the stored activation is 0 is masked to 0
    dY = torch.randn(M, N, generator=g)
    Wb = torch.rand(N, K, generator=g) + 0.1
    X = torch.randn(M, K, generator=g)
    X[7] = -1.0; X[295] = -1.0
    dY[7] = float("inf"); dY[295] = float("inf")
    dx = gemm(dY.to(dev), Wb.to(dev), b_t=False, bwd=(3, X.to(dev))).cpu()
    assert torch.isfinite(dx[7]).all() and float(dx[7].abs().max()) == 0.0 and float(dx[295].abs().max()) == 0.0
    keep = torch.ones(M, dtype=torch.bool); keep[7] = keep[295] = False
    want = torch.where(X > 0, dY @ Wb, torch.zeros(()))
    np.testing.assert_allclose(dx[keep].numpy(), want[keep].numpy(), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("M,chans,train", [(16384, (512, 256, 128, 64), True), (16384, (512, 256, 128, 64), False), (300, (16, 8, 4), True),
                                            (1000, (64, 64, 64, 64, 64), True)])
def test_mlp_chain_block_call_equals_python_sequence(dev, M, chans, train):
    """cmf_mlp_forward / _backward (the heads' [conv + BN + ReLU] stacks sequenced by the library) against the same kernels
    sequenced from Python (fused_blocks.MLPChainFn): outputs, input gradient, every parameter gradient and the BN running
    statistics bit-identical -- with gradients returned as tensors and with gradients accumulated into existing .grad."""
    from cmflow_amd import fused_blocks as FB
    torch.manual_seed(M + len(chans))

    def build():
        torch.manual_seed(7)
        layers = []
        for cin, cout in zip(chans[:-1], chans[1:]):
            conv, bn = torch.nn.Conv2d(cin, cout, 1, bias=False), torch.nn.BatchNorm2d(cout)
            with torch.no_grad():
                bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.2); bn.running_mean.normal_(0, 0.3); bn.running_var.uniform_(0.5, 2.0)
            layers.append((conv.to(dev), bn.to(dev).train(train)))
        return layers

    x0 = torch.randn(M, chans[0], device=dev)
    dy = torch.randn(M, chans[-1], device=dev)
    res = {}
    for block in (False, True):
        for sinks in (False, True):
            layers = build()
            FB.USE_BLOCK_CALLS = block
            try:
                params = [p for c, b in layers for p in (c.weight, b.weight, b.bias)]
                if sinks:
                    for p in params:
                        p.grad = torch.full_like(p, 0.25)
                    FB.enable_grad_sinks(params)
                x = x0.clone().requires_grad_(True)
                y = FB.mlp_chain(x, layers, train)
                y.backward(dy)
            finally:
                FB.USE_BLOCK_CALLS = True
            torch.cuda.synchronize()
            res[(block, sinks)] = (y.detach().clone(), x.grad.clone(), [p.grad.clone() for p in params],
                                   [t.clone() for c, b in layers for t in (b.running_mean, b.running_var, b.num_batches_tracked)])
    for sinks in (False, True):
        a, b = res[(False, sinks)], res[(True, sinks)]
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        for g0, g1 in zip(a[2], b[2]):
            assert torch.equal(g0, g1)
        for t0, t1 in zip(a[3], b[3]):
            assert torch.equal(t0, t1)
    # accumulation into existing gradients = returned gradient + what was there
    for g_ret, g_acc in zip(res[(True, False)][2], res[(True, True)][2]):
        assert float((g_acc - 0.25 - g_ret).abs().max()) <= 1e-5 * max(1.0, float(g_ret.abs().max()))


@pytest.mark.parametrize("rows,cout,cin,split", [(32768, 256, 512, 16), (65536, 256, 512, 96), (32768, 128, 128, 8), (49152, 256, 512, 1)])
def test_weight_gradient_gemm_with_fused_bn_backward(rows, cout, cin, split):
    """cmf_gemm_dw_bn_bwd (BN backward formed while the weight-gradient GEMM stages its A operand, dZ written as a by-product)
    against the two-kernel form it replaces -- cmf_bn_bwd_apply in place, then cmf_gemm(a_t, !b_t) with the producer's
    BN + ReLU on B: bit-identical dW and dZ (same operations, same loop), and against an fp64 evaluation."""
    from cmflow_amd import _lib
    from cmflow_amd.fused import gemm
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(rows + cout)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
    dU, Z, X = rnd(rows, cout), rnd(rows, cout) * 2 + 0.5, rnd(rows, cin)
    a, mean, invstd = torch.rand(cout, generator=g).to(dev) + 0.5, rnd(cout) * 0.3 + 0.5, torch.rand(cout, generator=g).to(dev) + 0.4
    pa, pc = torch.rand(cin, generator=g).to(dev) + 0.5, rnd(cin) * 0.2
    zhat = (Z - mean) * invstd
    sums = torch.stack((dU.sum(0), (dU * zhat).sum(0))).contiguous()            # what the producing kernel's epilogue leaves
    L = _lib.lib()
    # reference: the stand-alone pass, then the GEMM
    dZ_ref = dU.clone()
    _lib.check(L.cmf_bn_bwd_apply(rows, cout, dZ_ref.data_ptr(), Z.data_ptr(), cout, a.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                  sums.data_ptr(), _lib.stream_ptr()), "cmf_bn_bwd_apply")
    dW_ref = gemm(dZ_ref, X, a_t=True, b_t=False, prob=(pa, pc), split_k=split)
    for accumulate in (False, True):
        dW = torch.full((cout, cin), 0.5, device=dev)
        dZ = torch.full((rows, cout), float("nan"), device=dev)
        ws = torch.empty(split, cout, cin, device=dev) if split > 1 else None
        _lib.check(L.cmf_gemm_dw_bn_bwd(cout, cin, rows, dU.data_ptr(), cout, Z.data_ptr(), cout, a.data_ptr(), mean.data_ptr(),
                                        invstd.data_ptr(), sums.data_ptr(), dZ.data_ptr(), cout, X.data_ptr(), cin, pa.data_ptr(),
                                        pc.data_ptr(), dW.data_ptr(), cin, split, ws.data_ptr() if ws is not None else None,
                                        int(accumulate), _lib.stream_ptr()), "cmf_gemm_dw_bn_bwd")
        torch.cuda.synchronize()
        assert torch.equal(dZ, dZ_ref), float((dZ - dZ_ref).abs().max())
        if rows >= 32768 and not accumulate:
            assert torch.equal(dW, dW_ref), float((dW - dW_ref).abs().max())
        want = (dW_ref + 0.5) if accumulate else dW_ref
        assert float((dW - want).abs().max()) <= 1e-5 * float(dW_ref.abs().max())
    # fp64
    d64 = a.double() * (dU.double() - sums[0].double() / rows - zhat.double() * (sums[1].double() / rows))
    w64 = d64.t() @ torch.relu(pa.double() * X.double() + pc.double())
    assert float((dZ_ref.double() - d64).abs().max()) <= 2e-5 * float(d64.abs().max())
    assert float((dW_ref.double() - w64).abs().max()) <= 2e-5 * float(w64.abs().max())


@pytest.mark.parametrize("M,N,K,grid", [(2048, 256, 256, 8), (1024, 384, 512, 16), (4096, 128, 192, 8), (2048, 256, 1040, 24),
                                        (65536, 512, 256, 0)])
@pytest.mark.parametrize("mode,dxyz", [(1, False), (1, True), (2, False), (3, True), (2, True)])
def test_persistent_gemm_matches_tiled_kernel(dev, M, N, K, grid, mode, dxyz):
    """csrc/gemm_persist.hip (persistent workgroups, the epilogue of tile t between the MFMAs of tile t + 1) against the
    non-persistent kernel on the same call: the outputs must be BIT-identical (same MFMA sequence per element, masks decided
    from the same Z), the column statistics equal up to their fp32 summation order (bound: 1e-6 * sum |terms|) and
    bit-reproducible from run to run.  Small grids force several tiles per workgroup (tile switch, pipeline across tiles,
    first and last tile); grid 0 is the product's (two workgroups per CU) on one of the model's shapes."""
    from cmflow_amd import _lib
    from cmflow_amd.fused import gemm
    L = _lib.lib()
    g = torch.Generator().manual_seed(M + N + K + mode)
    ldz = N + 8
    dZ = torch.randn(M, K, generator=g).to(dev)
    W = torch.randn(K, N + 4, generator=g).to(dev)[:, :N]                  # B[K][N] with a padded row stride
    Zp = torch.randn(M, ldz, generator=g).to(dev)[:, :N]
    ea, ec = (torch.rand(N, generator=g) + 0.5).to(dev), (0.3 * torch.randn(N, generator=g)).to(dev)
    mean, invstd = torch.randn(N, generator=g).to(dev), (torch.rand(N, generator=g) + 0.5).to(dev)
    d4 = torch.randn(M, 4, generator=g).to(dev) if dxyz else None
    bwd = (mode, Zp, ea, ec, mean, invstd) if mode == 1 else (mode, Zp, None, None, None, None)
    if dxyz:
        bwd = bwd + (d4,)
    want_stats = mode == 1 or dxyz

    def run():
        out = torch.full((M, N + 12), 7.0, device=dev)                     # the padding columns must stay untouched
        r = gemm(dZ, W, b_t=False, bwd=bwd, stats=want_stats, out=out[:, 4:4 + N])
        torch.cuda.synchronize()
        return (out, r[1]) if want_stats else (out, None)

    try:
        assert L.cmf_gemm_persist_config(0, 0) == 0
        ref, ref_st = run()
        assert L.cmf_gemm_persist_config(2, grid) == 0
        got, got_st = run()
        got2, got2_st = run()
    finally:
        L.cmf_gemm_persist_config(1, 0)
    assert torch.equal(got, ref)
    assert torch.equal(got2, got)
    if want_stats:
        assert torch.equal(got_st, got2_st)
        x = ref[:, 4:4 + N].double()
        terms = [x.abs(), (x * ((Zp.double() - mean.double()) * invstd.double())).abs() if mode == 1 else x.abs() * 0]
        if dxyz:
            terms += [(x * d4[:, k:k + 1].double()).abs() for k in range(3)]
        tiles = ref_st.shape[0]
        for which, t in enumerate(terms):
            bound = 1e-6 * t.view(tiles, 128, N).sum(1) + 1e-6
            err = (got_st[:, which].double() - ref_st[:, which].double()).abs()
            assert bool((err <= bound).all()), (which, float((err / bound).max()))


@pytest.mark.parametrize("M,N,K,grid", [(2048, 256, 512, 8), (1024, 384, 256, 16), (4096, 128, 1024, 8), (131072, 512, 512, 0)])
@pytest.mark.parametrize("pro,bias,act,stats", [(False, False, 0, False), (True, True, 1, True), (True, False, 0, True), (False, True, 2, False)])
def test_persistent_gemm_forward_matches_tiled_kernel(dev, M, N, K, grid, pro, bias, act, stats):
    """The forward layout (A[M][K], W[N][K]) of the persistent kernel -- plain store and the bias / activation / BN-statistics
    epilogue, with and without the producer's BN + ReLU on A -- against the tiled kernel: outputs bit-identical, statistics
    equal up to their fp32 summation order (1e-6 * sum |terms|) and bit-reproducible."""
    from cmflow_amd import _lib
    from cmflow_amd.fused import gemm
    L = _lib.lib()
    g = torch.Generator().manual_seed(M + N + K + act)
    A = torch.randn(M, K + 8, generator=g).to(dev)[:, :K]
    W = torch.randn(N, K + 4, generator=g).to(dev)[:, :K]
    pa, pc = (torch.rand(K, generator=g) + 0.5).to(dev), (0.3 * torch.randn(K, generator=g)).to(dev)
    b = torch.randn(N, generator=g).to(dev)

    def run():
        out = torch.full((M, N + 12), 7.0, device=dev)
        r = gemm(A, W, pro=(pa, pc) if pro else None, bias=b if bias else None, act=act, stats=stats, out=out[:, 4:4 + N])
        torch.cuda.synchronize()
        return (out, r[1]) if stats else (out, None)

    try:
        assert L.cmf_gemm_persist_config(0, 0) == 0
        ref, ref_st = run()
        assert L.cmf_gemm_persist_config(2, grid) == 0
        got, got_st = run()
        got2, got2_st = run()
    finally:
        L.cmf_gemm_persist_config(1, 0)
    assert torch.equal(got, ref) and torch.equal(got2, got)
    if stats:
        assert torch.equal(got2_st, got_st)
        x = ref[:, 4:4 + N].double()
        tiles = ref_st.shape[0]
        for which, t in enumerate((x.abs(), x * x)):
            bound = 1e-6 * t.view(tiles, 128, N).sum(1) + 1e-6
            err = (got_st[:, which].double() - ref_st[:, which].double()).abs()
            assert bool((err <= bound).all()), (which, float((err / bound).max()))


@pytest.mark.parametrize("rows,M,N,split,grid", [(8192, 256, 256, 16, 64), (16384, 128, 384, 8, 24), (65536, 256, 512, 96, 0), (16384, 512, 512, 48, 0)])
@pytest.mark.parametrize("prob", [False, True])
def test_persistent_gemm_weight_gradient_matches_tiled_kernel(dev, rows, M, N, split, grid, prob):
    """The weight-gradient layout (A[K][M], B[K][N], split-K slabs, optional BN + ReLU on B) of the persistent kernel against an
    fp64 product (2e-6 * sum |a b|) and -- where it keeps the caller's slab count -- bit for bit against the tiled kernel."""
    from cmflow_amd import _lib
    from cmflow_amd.fused import gemm
    L = _lib.lib()
    g = torch.Generator().manual_seed(rows + M + N)
    dZ = torch.randn(rows, M + 4, generator=g).to(dev)[:, :M]
    X = torch.randn(rows, N + 8, generator=g).to(dev)[:, :N]
    qa, qc = (torch.rand(N, generator=g) + 0.5).to(dev), (0.3 * torch.randn(N, generator=g)).to(dev)
    Xa = torch.relu(X * qa + qc) if prob else X
    ref64 = dZ.double().t() @ Xa.double()
    bound = dZ.double().abs().t() @ Xa.double().abs()

    def run():
        r = gemm(dZ, X, a_t=True, b_t=False, prob=(qa, qc) if prob else None, split_k=split)
        torch.cuda.synchronize()
        return r

    try:
        assert L.cmf_gemm_persist_config(0, 0) == 0
        ref = run()
        assert L.cmf_gemm_persist_config(2, grid) == 0
        got = run()
        got2 = run()
    finally:
        L.cmf_gemm_persist_config(1, 0)
    _check(got, ref64, bound)
    assert torch.equal(got2, got)
    tiles = (M // 128) * (N // 128)
    g_eff = grid if grid else 512
    if min(split, g_eff // tiles) // 8 * 8 == split and not (prob and rows >= 32768):
        assert torch.equal(got, ref)             # same slabs, same loop arithmetic


@pytest.mark.parametrize("tiles,C", [(4096, 256), (2048, 64), (2047, 32), (700, 512), (33, 48), (5000, 12), (9, 6), (1, 16)])
def test_statistics_fold_kernels_match_fp64(dev, tiles, C):
    """cmf_bn_finalize / cmf_colsum over a [tiles][2][C] partial matrix in every form the dispatch picks (16 columns per
    1024-thread workgroup: columns % 16 == 0 and >= 2^20 partial sums; 4 columns: C % 4 == 0; generic otherwise) against
    an fp64 reduction: sums within 1e-6 of sum|terms| (fp64 accumulation inside, one rounding to fp32), the folded BatchNorm
    (mean, invstd, a, c, running statistics, counter) as nn.BatchNorm1d computes them from the same sums; twice = bit-identical."""
    from cmflow_amd import fused_blocks as FB
    g = torch.Generator().manual_seed(tiles * 7 + C)
    rows = tiles * 128
    s1 = torch.randn(tiles, C, generator=g, dtype=torch.float64) * 40 + 3.0            # per-tile sums of ~N(0.02, 1) values
    s2 = (torch.rand(tiles, C, generator=g, dtype=torch.float64) + 0.5) * 128 * 1.3 + s1 * s1 / 128
    part = torch.stack((s1, s2), dim=1).float().to(dev).contiguous()
    p64 = part.double()
    want = p64.sum(dim=0)
    # column sums, accumulated into / stored to gradient buffers
    acc0 = torch.ones(C, device=dev); acc1 = torch.full((C,), 2.0, device=dev)
    outs = [FB.colsum(part, acc0.clone(), acc1.clone()) for _ in range(2)]
    assert torch.equal(outs[0], outs[1])
    bound = 1e-6 * p64.abs().sum(dim=0) + 1e-6
    assert bool(((outs[0].double() - want).abs() <= bound).all())
    a0, a1 = acc0.clone(), acc1.clone()
    FB.colsum(part, a0, a1)
    assert bool(((a0.double() - 1.0 - want[0]).abs() <= bound[0] + 1e-6 * want[0].abs()).all())
    assert bool(((a1.double() - 2.0 - want[1]).abs() <= bound[1] + 1e-6 * want[1].abs()).all())
    outn = FB.colsum_n(part)
    assert torch.equal(outn, outs[0])
    # BatchNorm fold
    bn = torch.nn.BatchNorm1d(C).to(dev)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5, generator=None); bn.bias.normal_(); bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2.0)
    rm0, rv0 = bn.running_mean.double().clone(), bn.running_var.double().clone()
    st = FB.bn_fold(bn, part, rows)
    mean = want[0] / rows
    var = (want[1] / rows - mean * mean).clamp(min=0)
    invstd = 1.0 / torch.sqrt(var + bn.eps)
    np.testing.assert_allclose(st.mean.double().cpu().numpy(), mean.cpu().numpy(), rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(st.invstd.double().cpu().numpy(), invstd.cpu().numpy(), rtol=2e-6)
    np.testing.assert_allclose(st.a.double().cpu().numpy(), (bn.weight.detach().double() * invstd).cpu().numpy(), rtol=3e-6)
    np.testing.assert_allclose(st.c.double().cpu().numpy(), (bn.bias.detach().double() - mean * bn.weight.detach().double() * invstd).cpu().numpy(), rtol=1e-5, atol=1e-6)
    unb = var * rows / (rows - 1)
    np.testing.assert_allclose(bn.running_mean.double().cpu().numpy(), (0.9 * rm0 + 0.1 * mean).cpu().numpy(), rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(bn.running_var.double().cpu().numpy(), (0.9 * rv0 + 0.1 * unb).cpu().numpy(), rtol=2e-6)
    assert int(bn.num_batches_tracked) == 1


@pytest.mark.parametrize("B,N,S,K,NO,r", [(4, 256, 8, 512, 256, 4.0), (2, 256, 32, 64, 128, 16.0), (1, 128, 4, 16, 128, 2.0), (8, 256, 16, 512, 256, 8.0), (16, 256, 16, 512, 256, 8.0), (32, 256, 8, 256, 512, 4.0)])
def test_gather_affine_gemm_is_bit_identical_to_the_materialised_path(dev, B, N, S, K, NO, r):
    """cmf_group_prep + cmf_gemm_gather_affine (the set-conv first layer formed in the A-operand path of the next layer's GEMM,
    inference) against cmf_group_affine followed by cmf_gemm with the A prologue: the same operations in the same order, so the
    outputs must be equal bit for bit; twice = reproducible."""
    from cmflow_amd import _lib, synth, fused_blocks as FB, pointnet2_utils as pu
    from cmflow_amd.fused import gemm
    L = _lib.lib()
    torch.manual_seed(B * 100 + S)
    xyz = synth.make_batch(B, N=N, seed=7)["pc1"].to(dev).transpose(1, 2).contiguous()
    idx = pu.ball_query(r, S, xyz, xyz)
    M = B * N * S
    ybig = torch.randn(B, N, K + 48, device=dev)
    y = ybig[:, :, 16:16 + K]                               # a column slice of a wider per-point matrix, as the encoder passes it
    wx = torch.randn(K, 3, device=dev)
    pa, pc = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.3
    W = torch.randn(NO, K, device=dev)
    z, dxyz, _ = FB.group_affine(y, None, xyz, xyz, wx, idx, act=0, stats=False)
    want = gemm(z.view(M, K), W, pro=(pa, pc))
    rows = torch.empty(M, dtype=torch.int32, device=dev); dq = torch.empty(M, 4, device=dev); wx3 = torch.empty(3, K, device=dev)
    st = _lib.stream_ptr()
    _lib.check(L.cmf_group_prep(B, N, N, S, K, xyz.data_ptr(), xyz.data_ptr(), wx.data_ptr(), 3, idx.data_ptr(), rows.data_ptr(),
                                dq.data_ptr(), wx3.data_ptr(), st), "prep")
    assert torch.equal(dq, dxyz.view(M, 4)) and torch.equal(wx3, wx.t().contiguous())
    assert torch.equal(rows.view(B, N, S).long(), idx.long() + torch.arange(B, device=dev).view(B, 1, 1) * N)
    outs = []
    for _ in range(2):
        out = torch.empty(M, NO, device=dev)
        _lib.check(L.cmf_gemm_gather_affine(M, NO, K, y.data_ptr(), y.stride(1), rows.data_ptr(), dq.data_ptr(), wx3.data_ptr(),
                                            pa.data_ptr(), pc.data_ptr(), W.data_ptr(), K, out.data_ptr(), NO, None, st), "gather gemm")
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    assert torch.equal(outs[0], want), float((outs[0] - want).abs().max())


@pytest.mark.parametrize("B,N,S,K,NO,r,split", [(8, 256, 16, 512, 256, 8.0, 8), (8, 256, 16, 512, 256, 8.0, 128), (16, 256, 8, 256, 512, 4.0, 128), (4, 256, 32, 128, 128, 16.0, 1), (2, 256, 4, 256, 128, 2.0, 4)])
def test_gather_weight_gradient_is_bit_identical_to_the_materialised_path(dev, B, N, S, K, NO, r, split):
    """cmf_gemm_dw_gather (the set-conv first layer formed in the B-operand staging of the next layer's weight gradient) against
    cmf_gemm(a_t, prob) on the tensor cmf_group_affine writes, both in the register-staged loop: bit for bit, with and without
    split-K, written and accumulated."""
    from cmflow_amd import _lib, synth, fused_blocks as FB, pointnet2_utils as pu
    L = _lib.lib()
    torch.manual_seed(B * 10 + S)
    xyz = synth.make_batch(B, N=N, seed=11)["pc1"].to(dev).transpose(1, 2).contiguous()
    idx = pu.ball_query(r, S, xyz, xyz)
    M = B * N * S
    y = torch.randn(B, N, 2 * K, device=dev)[:, :, :K]
    wx = torch.randn(K, 3, device=dev)
    pa, pc = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.3
    dZ = torch.randn(M, NO, device=dev)
    z, dxyz, _ = FB.group_affine(y, None, xyz, xyz, wx, idx, act=0, stats=False)
    st = _lib.stream_ptr()
    rows = torch.empty(M, dtype=torch.int32, device=dev); dq = torch.empty(M, 4, device=dev); wx3 = torch.empty(3, K, device=dev)
    _lib.check(L.cmf_group_prep(B, N, N, S, K, xyz.data_ptr(), xyz.data_ptr(), wx.data_ptr(), 3, idx.data_ptr(), rows.data_ptr(),
                                dq.data_ptr(), wx3.data_ptr(), st), "prep")
    ws = torch.empty(max(split, 1) * NO * K, device=dev)
    for acc in (0, 1):
        want = torch.full((NO, K), 0.5, device=dev); got = want.clone()
        # the materialised form, forced onto the register-staged loop like the model's long weight gradients (cmf_gemm picks it for K >= 32768)
        g_no_direct = M >= 32768
        _lib.check(L.cmf_gemm(NO, K, M, 1, 0, dZ.data_ptr(), NO, z.data_ptr(), K, want.data_ptr(), K, None, None, pa.data_ptr(), pc.data_ptr(),
                              None, 0, None, 0, None, 0, None, None, None, None, None, split, ws.data_ptr() if split > 1 else None, acc, st), "dw")
        _lib.check(L.cmf_gemm_dw_gather(NO, K, M, dZ.data_ptr(), NO, y.data_ptr(), y.stride(1), rows.data_ptr(), dq.data_ptr(), wx3.data_ptr(),
                                        pa.data_ptr(), pc.data_ptr(), got.data_ptr(), K, split, ws.data_ptr() if split > 1 else None, acc, st), "dwg")
        if g_no_direct:
            assert torch.equal(got, want), float((got - want).abs().max())
        else:                                               # the LDS-direct loop applies the prologue to fragments: same values, same order
            assert torch.equal(got, want) or float((got - want).abs().max()) <= 2e-6 * float(want.abs().max()) * 50


@pytest.mark.parametrize("B,N,S,K,NO,r", [(8, 256, 16, 512, 256, 8.0), (16, 256, 32, 512, 256, 16.0), (4, 256, 32, 128, 128, 16.0)])
def test_gather_weight_gradient_with_fused_bn_backward(dev, B, N, S, K, NO, r):
    """cmf_gemm_dw_gather_bn_bwd (BOTH operands formed while staging: the BN backward of the output gradient on A, the gathered first layer
    on B) against cmf_bn_bwd_apply in place followed by cmf_gemm_dw_gather: dZ and dW bit for bit, at the split count the block calls use."""
    from cmflow_amd import _lib, synth, pointnet2_utils as pu
    L = _lib.lib()
    torch.manual_seed(B * 10 + S)
    xyz = synth.make_batch(B, N=N, seed=11)["pc1"].to(dev).transpose(1, 2).contiguous()
    idx = pu.ball_query(r, S, xyz, xyz)
    M = B * N * S
    y = torch.randn(B, N, 2 * K, device=dev)[:, :, :K]
    wx = torch.randn(K, 3, device=dev)
    pa, pc = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.3
    dU, Z = torch.randn(M, NO, device=dev), torch.randn(M, NO, device=dev) * 2 + 0.5
    a, mean, invstd = torch.rand(NO, device=dev) + 0.5, torch.randn(NO, device=dev) * 0.3 + 0.5, torch.rand(NO, device=dev) + 0.4
    sums = torch.stack((dU.sum(0), (dU * ((Z - mean) * invstd)).sum(0))).contiguous()
    st = _lib.stream_ptr()
    rows = torch.empty(M, dtype=torch.int32, device=dev); dq = torch.empty(M, 4, device=dev); wx3 = torch.empty(3, K, device=dev)
    _lib.check(L.cmf_group_prep(B, N, N, S, K, xyz.data_ptr(), xyz.data_ptr(), wx.data_ptr(), 3, idx.data_ptr(), rows.data_ptr(),
                                dq.data_ptr(), wx3.data_ptr(), st), "prep")
    split = L.cmf_gemm_dw_gather_split(NO, K, M) or 8
    ws = torch.empty(split * NO * K, device=dev)
    dZ_ref = dU.clone()
    _lib.check(L.cmf_bn_bwd_apply(M, NO, dZ_ref.data_ptr(), Z.data_ptr(), NO, a.data_ptr(), mean.data_ptr(), invstd.data_ptr(), sums.data_ptr(), st), "bnb")
    for acc in (0, 1):
        want, got = torch.full((NO, K), 0.5, device=dev), torch.full((NO, K), 0.5, device=dev)
        dZ = torch.full((M, NO), float("nan"), device=dev)
        _lib.check(L.cmf_gemm_dw_gather(NO, K, M, dZ_ref.data_ptr(), NO, y.data_ptr(), y.stride(1), rows.data_ptr(), dq.data_ptr(), wx3.data_ptr(),
                                        pa.data_ptr(), pc.data_ptr(), want.data_ptr(), K, split, ws.data_ptr(), acc, st), "dwg")
        _lib.check(L.cmf_gemm_dw_gather_bn_bwd(NO, K, M, dU.data_ptr(), NO, Z.data_ptr(), NO, a.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                               sums.data_ptr(), dZ.data_ptr(), NO, y.data_ptr(), y.stride(1), rows.data_ptr(), dq.data_ptr(),
                                               wx3.data_ptr(), pa.data_ptr(), pc.data_ptr(), got.data_ptr(), K, split, ws.data_ptr(), acc, st), "dwgb")
        torch.cuda.synchronize()
        assert torch.equal(dZ, dZ_ref), float((dZ - dZ_ref).abs().max())
        assert torch.equal(got, want), float((got - want).abs().max())


@pytest.mark.parametrize("B,N,S,K,NO,r", [(4, 256, 8, 512, 256, 4.0), (2, 256, 32, 128, 64, 16.0), (1, 128, 4, 256, 16, 2.0)])
def test_gather_data_gradient_is_bit_identical_to_the_materialised_path(dev, B, N, S, K, NO, r):
    """cmf_gemm_dx_gather (the first layer's pre-activations formed from the per-point rows in the backward epilogue) against
    cmf_gemm(bwd_mode 1, Z = the tensor cmf_group_affine writes, dxyz) in the non-persistent kernel: dU and the five partial sums
    bit for bit."""
    from cmflow_amd import _lib, synth, fused_blocks as FB, pointnet2_utils as pu
    L = _lib.lib()
    torch.manual_seed(B * 10 + S)
    xyz = synth.make_batch(B, N=N, seed=13)["pc1"].to(dev).transpose(1, 2).contiguous()
    idx = pu.ball_query(r, S, xyz, xyz)
    M = B * N * S
    y = torch.randn(B, N, 2 * K, device=dev)[:, :, :K]
    wx = torch.randn(K, 3, device=dev)
    ea, ec, em, ei = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.3, torch.randn(K, device=dev), torch.rand(K, device=dev) + 0.5
    dZ = torch.randn(M, NO, device=dev); W = torch.randn(NO, K, device=dev)
    z, dxyz, _ = FB.group_affine(y, None, xyz, xyz, wx, idx, act=0, stats=False)
    st = _lib.stream_ptr()
    rows = torch.empty(M, dtype=torch.int32, device=dev); dq = torch.empty(M, 4, device=dev); wx3 = torch.empty(3, K, device=dev)
    _lib.check(L.cmf_group_prep(B, N, N, S, K, xyz.data_ptr(), xyz.data_ptr(), wx.data_ptr(), 3, idx.data_ptr(), rows.data_ptr(),
                                dq.data_ptr(), wx3.data_ptr(), st), "prep")
    tiles = M // 128
    want, ws = torch.empty(M, K, device=dev), torch.empty(tiles, 5, K, device=dev)
    got, gs = torch.empty(M, K, device=dev), torch.empty(tiles, 5, K, device=dev)
    L.cmf_gemm_persist_config(0, 0)
    try:
        _lib.check(L.cmf_gemm(M, K, NO, 0, 0, dZ.data_ptr(), NO, W.data_ptr(), K, want.data_ptr(), K, None, None, None, None, None, 0,
                              ws.data_ptr(), 1, z.data_ptr(), K, ea.data_ptr(), ec.data_ptr(), em.data_ptr(), ei.data_ptr(), dq.data_ptr(),
                              1, None, 0, st), "dx")
    finally:
        L.cmf_gemm_persist_config(1, 0)
    _lib.check(L.cmf_gemm_dx_gather(M, K, NO, dZ.data_ptr(), NO, W.data_ptr(), K, got.data_ptr(), K, y.data_ptr(), y.stride(1), rows.data_ptr(),
                                    dq.data_ptr(), wx3.data_ptr(), ea.data_ptr(), ec.data_ptr(), em.data_ptr(), ei.data_ptr(), gs.data_ptr(), st), "dxg")
    assert torch.equal(got, want), float((got - want).abs().max())
    assert torch.equal(gs, ws)


@pytest.mark.parametrize("B,N,S,K,NO,r", [(4, 256, 8, 512, 256, 4.0), (2, 256, 32, 128, 64, 16.0), (8, 256, 16, 256, 128, 8.0), (1, 128, 4, 128, 16, 2.0)])
def test_gather_data_gradient_summed_per_source_point(dev, B, N, S, K, NO, r):
    """cmf_group_perm + cmf_gemm_dx_gather_sum + cmf_group_rows_grad_bn_cf_pieces (the data gradient into the first layer never
    stored: rows in inverse-index order, runs of equal source points reduced in the GEMM's epilogue) against cmf_gemm_dx_gather +
    cmf_group_rows_grad_bn_cf: the five statistics and the per-point result agree up to fp32 association (1e-5 of the column's
    absolute sums), twice = bit-reproducible."""
    from cmflow_amd import _lib, synth, fused_blocks as FB, pointnet2_utils as pu
    from cmflow_amd.fused import Neighbors
    L = _lib.lib()
    torch.manual_seed(B * 10 + S)
    xyz = synth.make_batch(B, N=N, seed=17)["pc1"].to(dev).transpose(1, 2).contiguous()
    idx = pu.ball_query(r, S, xyz, xyz)
    off, inv = Neighbors(idx, N).inverse()
    M, P, E = B * N * S, B * N, N * S
    y = torch.randn(B, N, 2 * K, device=dev)[:, :, :K]
    wx = torch.randn(K, 3, device=dev)
    ea, ec, em, ei = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.3, torch.randn(K, device=dev), torch.rand(K, device=dev) + 0.5
    dZ = torch.randn(M, NO, device=dev); W = torch.randn(NO, K, device=dev)
    st = _lib.stream_ptr()
    rows = torch.empty(M, dtype=torch.int32, device=dev); dq = torch.empty(M, 4, device=dev); wx3 = torch.empty(3, K, device=dev)
    _lib.check(L.cmf_group_prep(B, N, N, S, K, xyz.data_ptr(), xyz.data_ptr(), wx.data_ptr(), 3, idx.data_ptr(), rows.data_ptr(),
                                dq.data_ptr(), wx3.data_ptr(), st), "prep")
    tiles = M // 128
    dU, s_ref = torch.empty(M, K, device=dev), torch.empty(tiles, 5, K, device=dev)
    _lib.check(L.cmf_gemm_dx_gather(M, K, NO, dZ.data_ptr(), NO, W.data_ptr(), K, dU.data_ptr(), K, y.data_ptr(), y.stride(1), rows.data_ptr(),
                                    dq.data_ptr(), wx3.data_ptr(), ea.data_ptr(), ec.data_ptr(), em.data_ptr(), ei.data_ptr(), s_ref.data_ptr(), st), "dxg")
    sums = s_ref.double().sum(0).float().contiguous()
    want = torch.empty(B, N, K, device=dev)
    args = (y.data_ptr(), y.stride(1), wx.data_ptr(), 3, xyz.data_ptr(), xyz.data_ptr(), ea.data_ptr(), em.data_ptr(), ei.data_ptr(), sums.data_ptr(),
            1.0 / M, off.data_ptr(), inv.data_ptr())
    _lib.check(L.cmf_group_rows_grad_bn_cf(B, N, K, E, S, dU.data_ptr(), *args, want.data_ptr(), K, st), "cf")
    perm = torch.empty(M, dtype=torch.int32, device=dev); pts = torch.empty(M, dtype=torch.int32, device=dev); dq2 = torch.empty(M, 4, device=dev)
    _lib.check(L.cmf_group_perm(B, E, inv.data_ptr(), rows.data_ptr(), dq.data_ptr(), perm.data_ptr(), pts.data_ptr(), dq2.data_ptr(), st), "perm")
    assert torch.equal(pts, rows[perm.long()]) and torch.equal(dq2, dq[perm.long()]) and bool((pts[1:] >= pts[:-1]).all())
    outs = []
    for _ in range(2):
        pieces = torch.full((P + M // 64, K), float("nan"), device=dev)
        s_got = torch.empty(tiles, 5, K, device=dev)
        _lib.check(L.cmf_gemm_dx_gather_sum(M, K, NO, dZ.data_ptr(), NO, W.data_ptr(), K, y.data_ptr(), y.stride(1), perm.data_ptr(), pts.data_ptr(),
                                            dq2.data_ptr(), wx3.data_ptr(), ea.data_ptr(), ec.data_ptr(), em.data_ptr(), ei.data_ptr(),
                                            pieces.data_ptr(), s_got.data_ptr(), st), "dxs")
        got = torch.empty(B, N, K, device=dev)
        _lib.check(L.cmf_group_rows_grad_bn_cf_pieces(B, N, K, E, S, pieces.data_ptr(), *args, got.data_ptr(), K, st), "cfp")
        outs.append((got, s_got))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    got, s_got = outs[0]
    assert not bool(torch.isnan(got).any())
    # statistics: the same terms in another order
    absum = torch.zeros(5, K, dtype=torch.float64, device=dev)
    d64 = dU.double()
    zhat = ((FB.group_affine(y, None, xyz, xyz, wx, idx, act=0, stats=False)[0].view(M, K).double() - em.double()) * ei.double())
    terms = [d64, d64 * zhat, d64 * dq[:, 0:1].double(), d64 * dq[:, 1:2].double(), d64 * dq[:, 2:3].double()]
    for k in range(5):
        absum[k] = terms[k].abs().sum(0)
        assert bool(((s_got[:, k].double().sum(0) - terms[k].sum(0)).abs() <= 2e-6 * absum[k] + 1e-5).all()), k
    # the per-point result: against the materialised form, scaled by the absolute sums that go into a point
    scale = torch.zeros(P, K, dtype=torch.float64, device=dev).index_add_(0, rows.long(), d64.abs())
    err = (got.view(P, K).double() - want.view(P, K).double()).abs()
    bound = 1e-5 * (scale * ea.double().abs() + 1.0) + 1e-4 * want.view(P, K).double().abs().clamp(max=1.0)
    assert bool((err <= bound).all()), float((err / bound).max())


@pytest.mark.parametrize("M,K,N", [(1024, 32, 64), (4096, 32, 32), (640, 64, 64), (256, 8, 32), (384, 16, 64), (128 * 37, 64, 32)])
@pytest.mark.parametrize("pro,stats,act,bias", [(True, True, 0, False), (False, True, 1, True), (True, False, 2, True), (False, False, 0, False)])
def test_narrow_forward_full_tile_body_equals_the_general_body(dev, M, K, N, pro, stats, act, bias):
    """The <= 64-channel forward layers have a full-tile body (all loads of a wave issued up front, accumulators side by side) beside
    the general one (bounds-checked, fragment by fragment): same MFMA sequence per accumulator, same summation order of the
    statistics -- outputs and partial sums must be equal bit for bit."""
    from cmflow_amd import _lib
    from cmflow_amd.fused import gemm
    L = _lib.lib()
    torch.manual_seed(M + K + N)
    A = torch.randn(M, K + 8, device=dev)[:, :K]            # a row-strided view, as the blocks pass column slices
    W = torch.randn(N, K, device=dev)
    pa, pc = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.3
    b = torch.randn(N, device=dev) if bias else None
    outs = []
    prev = L.cmf_thin_general(0)
    try:
        for general in (0, 1, 0):
            L.cmf_thin_general(general)
            r = gemm(A, W, pro=(pa, pc) if pro else None, bias=b, act=act, stats=stats)
            outs.append(r if stats else (r, None))
    finally:
        L.cmf_thin_general(prev)
    for o, s in outs[1:]:
        assert torch.equal(o, outs[0][0]), float((o - outs[0][0]).abs().max())
        if stats:
            assert torch.equal(s, outs[0][1]), float((s - outs[0][1]).abs().max())
    x = torch.relu(A * pa + pc) if pro else A
    want = x.double() @ W.double().t() + (b.double() if bias else 0.0)
    want = torch.relu(want) if act == 1 else torch.where(want > 0, want, 0.1 * want) if act == 2 else want
    assert float((outs[0][0].double() - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max()))
