"""GPU: op-level parity of the HIP kernels (through the C-ABI) against the CPU oracle.

Bit-exact for indices and copies; tolerances are written next to each floating-point check.
"""
import numpy as np
import pytest
import torch

from cmflow_amd import synth
from oracle import cmflow_oracle as O
from oracle import ops as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from cmflow_amd import _lib
    _lib.lib()          # fails loudly if the HIP extension is missing
    return torch.device("cuda:0")


def clouds(B, N, seed, lidar=False):
    b = synth.make_batch(B, N=N, seed=seed, lidar=lidar)
    return b["pc1"].permute(0, 2, 1).contiguous(), b["pc2"].permute(0, 2, 1).contiguous()


@pytest.mark.parametrize("B,N,r,ns", [(4, 256, 2.0, 4), (4, 256, 4.0, 8), (4, 256, 8.0, 16), (4, 256, 16.0, 32),
                                      (2, 100, 4.0, 8), (1, 1, 1.0, 4), (2, 1500, 2.0, 32), (1, 4096, 2.0, 64),
                                      (3, 300, 1000.0, 16)])
def test_ball_query_bit_exact(dev, B, N, r, ns):
    from cmflow_amd.pointnet2_utils import ball_query
    xyz, _ = clouds(B, N, seed=B * 1000 + N, lidar=N > 1000)
    ref = orc.ball_query(r, ns, xyz, xyz)
    got = ball_query(r, ns, xyz.to(dev), xyz.to(dev)).cpu()
    assert got.dtype == torch.int32 and torch.equal(got, ref)


@pytest.mark.parametrize("B,N,M,scales,nclouds", [(4, 256, 256, ((2.0, 4), (4.0, 8), (8.0, 16), (16.0, 32)), 2), (3, 256, 256, ((2.0, 4), (4.0, 8), (8.0, 16), (16.0, 32)), 1),
                                                  (2, 700, 123, ((0.01, 4), (3.0, 9), (1000.0, 64)), 1), (1, 1024, 1024, ((1.5, 32),), 2),
                                                  (2, 300, 300, ((4.0, 8), (0.0, 3)), 2)])
def test_ball_query_multi_equals_the_single_queries(dev, B, N, M, scales, nclouds):
    """cmf_ball_query_multi: the ball queries of one multi-scale grouping call (radarflow_util.py:111-118) over the same centres
    and cloud -- and over two clouds of equal geometry -- in ONE launch.  Every index tensor must equal the oracle's (and so
    cmf_ball_query's) bit for bit, empty balls included (zero rows with zero_empty, untouched rows without)."""
    import ctypes
    from cmflow_amd import _lib
    L = _lib.lib()
    nq = len(scales)
    radii = (ctypes.c_float * nq)(*[r for r, _ in scales])
    ns = (ctypes.c_int * nq)(*[k for _, k in scales])
    cl, ctrs = [], []
    for c in range(nclouds):
        xyz, other = clouds(B, N, seed=31 * B + N + c)
        cl.append(xyz)
        ctrs.append(xyz if M == N else (other[:, :M] + 0.5).contiguous())
    dcl, dct = [x.to(dev) for x in cl], [x.to(dev) for x in ctrs]
    for zero_empty in (1, 0):
        out = [[torch.full((B, M, k), -7, dtype=torch.int32, device=dev) for _, k in scales] for _ in range(nclouds)]
        pc = (ctypes.c_void_p * nclouds)(*[x.data_ptr() for x in dct])
        px = (ctypes.c_void_p * nclouds)(*[x.data_ptr() for x in dcl])
        pi = (ctypes.c_void_p * (nclouds * nq))(*[t.data_ptr() for row in out for t in row])
        _lib.check(L.cmf_ball_query_multi(B, N, M, nq, ctypes.addressof(radii), ctypes.addressof(ns), nclouds, ctypes.addressof(pc),
                                          ctypes.addressof(px), ctypes.addressof(pi), zero_empty, _lib.stream_ptr()), "cmf_ball_query_multi")
        for c in range(nclouds):
            for q, (r, k) in enumerate(scales):
                ref = orc.ball_query(r, k, cl[c], ctrs[c])                   # (empty balls: zero rows, the pre-zeroed idx of the reference)
                got = out[c][q].cpu()
                d2 = ((ctrs[c][:, :, None, :] - cl[c][:, None, :, :]) ** 2).sum(-1)
                empty = ~(d2 < r * r).any(-1)                                # (robust: rows the oracle left at zero AND that have no hit at all)
                empty &= (ref == 0).all(-1)
                if zero_empty:
                    assert torch.equal(got, ref), (c, q)
                else:
                    assert torch.equal(got[~empty], ref[~empty]), (c, q)
                    assert bool((got[empty] == -7).all()), (c, q)


@pytest.mark.parametrize("B,N,M,r,ns,kind", [(2, 4096, 4096, 2.0, 64, "lidar"), (1, 8192, 300, 1.0, 32, "lidar"), (2, 4096, 777, 0.05, 8, "lidar"),
                                             (1, 4096, 4096, 1.0e4, 16, "lidar"), (1, 5000, 64, 3.0, 5, "line"), (1, 4096, 100, 0.5, 4, "dup"),
                                             (1, 6000, 50, 2.0, 7, "far")])
def test_ball_query_cell_grid_bit_exact(dev, B, N, M, r, ns, kind):
    """Large clouds take the cell-grid kernel (4096 <= N <= 8192: bin, lane-parallel distance tests, index-ordered read-back
    of a hit bitmap): it must reproduce the scan's "first nsample in index order" exactly -- dense and empty balls, a
    radius larger than the cloud, degenerate clouds (all points on a line / identical), centres far outside the cloud,
    m != n (ball_query_gpu.cu:9-45)."""
    from cmflow_amd.pointnet2_utils import ball_query
    xyz, other = clouds(B, N, seed=N + ns, lidar=True)
    if kind == "line":
        xyz = xyz.clone(); xyz[:, :, 1:] = 0.25
    if kind == "dup":
        xyz = xyz[:, :1].expand(B, N, 3).contiguous()
    ctr = xyz[:, :M].contiguous() if M <= N else xyz
    if kind == "far":
        ctr = ctr.clone(); ctr[:, ::2] += 500.0; ctr[:, 1::4] -= 300.0
    elif M != N:
        ctr = (other[:, :M] * 1.0).contiguous()
    ref = orc.ball_query(r, ns, xyz, ctr)
    got = ball_query(r, ns, xyz.to(dev), ctr.to(dev)).cpu()
    assert torch.equal(got, ref)


def test_ball_query_edge_cases(dev):
    """strict '<', first-hit padding, empty ball leaves the pre-zeroed idx, duplicates, m != n."""
    from cmflow_amd.pointnet2_utils import ball_query
    xyz = torch.tensor([[[0.0, 0, 0], [1.0, 0, 0], [2.0, 0, 0], [0.5, 0, 0]]])
    assert ball_query(1.0, 3, xyz.to(dev), xyz.to(dev)).cpu()[0].tolist() == [[0, 3, 0], [1, 3, 1], [2, 2, 2], [0, 1, 3]]
    far = torch.tensor([[[100.0, 0, 0], [0.4, 0, 0]]])
    got = ball_query(1.0, 4, xyz.to(dev), far.to(dev)).cpu()
    assert torch.equal(got, orc.ball_query(1.0, 4, xyz, far)) and got[0, 0].tolist() == [0, 0, 0, 0]
    dup = torch.zeros(1, 70, 3)
    assert torch.equal(ball_query(0.5, 8, dup.to(dev), dup.to(dev)).cpu(), orc.ball_query(0.5, 8, dup, dup))
    # padded real-style cloud: duplicated points
    xyz, _ = clouds(2, 200, seed=5)
    xyz = torch.cat([xyz, xyz[:, :56]], dim=1).contiguous()
    assert torch.equal(ball_query(4.0, 8, xyz.to(dev), xyz.to(dev)).cpu(), orc.ball_query(4.0, 8, xyz, xyz))


@pytest.mark.parametrize("B,C,N,P,S", [(4, 3, 256, 256, 4), (2, 64, 256, 256, 32), (2, 1027, 256, 256, 8),
                                       (1, 5, 100, 37, 3), (1, 2, 5000, 64, 16), (2, 9, 4096, 512, 64),
                                       (2, 33, 256, 256, 16), (1, 20, 300, 250, 32), (1, 2, 35000, 16, 8),
                                       (1, 17, 700, 1, 1), (2, 6, 4096, 4096, 64), (1, 5, 4096, 1000, 37),
                                       (1, 3, 9000, 2000, 9), (2, 5, 2048, 1024, 32), (1, 6, 3000, 1024, 16), (1, 9, 4096, 130, 64),
                                       (1, 3, 2, 1100, 64), (2, 4, 8192, 2048, 32), (1, 2, 70, 4097, 16)])
def test_group_points_and_grad(dev, B, C, N, P, S):
    from cmflow_amd.pointnet2_utils import pointnet2_cuda as ext
    g = torch.Generator().manual_seed(B + C + N)
    pts = torch.randn(B, C, N, generator=g)
    idx = torch.randint(0, N, (B, P, S), generator=g, dtype=torch.int32)
    out = torch.empty(B, C, P, S, device=dev)
    ext.group_points_wrapper(B, C, N, P, S, pts.to(dev), idx.to(dev), out)
    assert torch.equal(out.cpu(), orc.group_points(pts, idx))                 # a copy: bit-exact
    go = torch.randn(B, C, P, S, generator=g)
    gp = torch.zeros(B, C, N, device=dev)
    ext.group_points_grad_wrapper(B, C, N, P, S, go.to(dev), idx.to(dev), gp)
    ref = orc.group_points_grad(go, idx, N)
    # the reference's atomicAdd order is undefined; ours is a fixed chunked order (balanced kernel: rows <= 8192 entries;
    # tiled kernel: longer rows, n <= 16384): equal to the oracle's scan-order sum to fp32 rounding of the partial sums
    np.testing.assert_allclose(gp.cpu().numpy(), ref.numpy(), rtol=1e-5, atol=1e-5 * float(S * P / N + 1) ** 0.5 * 4)
    # balanced / tiled kernels, and the pad-folded CSR gather when it is selected (lists of 16 / 32 / 64 slots, n <= 8192;
    # test_group_points_grad_csr_form runs this test under CMF_GROUP_GRAD_CSR=1): bit-reproducible run to run
    import os
    csr = os.environ.get("CMF_GROUP_GRAD_CSR") != "0" and S in (16, 32, 64) and N <= 8192
    if (P * S <= 8192 and N <= 4096) or 8192 < N <= 16384 or csr:
        gp2 = torch.zeros(B, C, N, device=dev)
        ext.group_points_grad_wrapper(B, C, N, P, S, go.to(dev), idx.to(dev), gp2)
        assert torch.equal(gp2.cpu(), gp.cpu())
    # accumulates INTO grad_points (lib/pointnet2_utils.py:218 zero-fills first)
    ext.group_points_grad_wrapper(B, C, N, P, S, go.to(dev), idx.to(dev), gp)
    np.testing.assert_allclose(gp.cpu().numpy(), 2 * ref.numpy(), rtol=1e-5, atol=2e-5 * max(1.0, float(S * P / N) ** 0.5))


@pytest.mark.parametrize("B,N,M,r,ns", [(3, 512, 512, 4.0, 16), (2, 1024, 300, 2.0, 64), (2, 700, 701 - 1, 8.0, 100), (5, 257, 33, 3.0, 7),
                                        (64, 256, 256, 2.0, 32), (1, 64, 64, 100.0, 200), (2, 1000, 5, 0.01, 4)])
def test_ball_query_ballot_kernel_bit_exact(dev, B, N, M, r, ns):
    """Small clouds (n <= 1024) take the ballot kernel: a wave per centre, the cloud in registers, four ballots per
    256-point chunk are the hit list in index order.  Chunk boundaries (n = 257, 512, 700, 1000, 1024), nsample larger
    than a wave, lists longer than the cloud, centres that are not the points, empty balls, m not a multiple of the
    centres-per-wave group."""
    from cmflow_amd.pointnet2_utils import ball_query
    xyz, other = clouds(B, N, seed=N + ns)
    ctr = xyz[:, :M].contiguous() if (M <= N and (M % 2 == 0)) else other[:, :M].contiguous()
    ref = orc.ball_query(r, ns, xyz, ctr)
    got = ball_query(r, ns, xyz.to(dev), ctr.to(dev)).cpu()
    assert torch.equal(got, ref)


@pytest.mark.parametrize("B,N,M,r,ns,C,use_xyz", [(64, 256, 256, 2.0, 32, 3, True), (4, 256, 256, 2.0, 4, 6, True), (2, 100, 37, 4.0, 8, 5, True),
                                                  (2, 700, 700, 8.0, 16, 64, True), (1, 1500, 300, 3.0, 16, 4, True),
                                                  (2, 256, 256, 16.0, 32, 200, True), (3, 256, 256, 4.0, 8, 0, True),
                                                  (3, 256, 200, 0.05, 8, 12, False), (2, 512, 512, 2.0, 64, 64, True)])
def test_query_and_group_fused_matches_the_three_op_sequence(dev, B, N, M, r, ns, C, use_xyz):
    """cmf_query_and_group (QueryAndGroup.forward, lib/pointnet2_utils.py:269-292, in one call) against the oracle's ball
    query + two grouping operations + subtraction + cat: indices and every output float bit-equal (gathers and ONE
    subtraction), empty balls group point 0, gradient w.r.t. the features equal to GroupingOperation.backward of the
    feature planes.  Covers the single-launch form, the wide-feature form (3 + C > 160) and large clouds (n > 1024)."""
    from cmflow_amd.pointnet2_utils import QueryAndGroup
    xyz, other = clouds(B, N, seed=N + ns + C)
    ctr = xyz[:, :M].contiguous() if M == N else other[:, :M].contiguous()
    g = torch.Generator().manual_seed(C + 1)
    feats = torch.randn(B, C, N, generator=g) if C else None
    idx = orc.ball_query(r, ns, xyz, ctr)
    gx = orc.group_points(xyz.transpose(1, 2).contiguous(), idx) - ctr.transpose(1, 2).unsqueeze(-1)
    want = gx if C == 0 else (torch.cat([gx, orc.group_points(feats, idx)], dim=1) if use_xyz else orc.group_points(feats, idx))
    mod = QueryAndGroup(r, ns, use_xyz=use_xyz)
    fd = feats.to(dev).requires_grad_(True) if C else None
    got = mod(xyz.to(dev), ctr.to(dev), fd)
    assert got.shape == want.shape and torch.equal(got.cpu(), want)
    if C:
        assert torch.equal(got.grad_fn.for_backwards[0].cpu(), idx)                 # the index tensor kept for backward

        go = torch.randn(want.shape, generator=g)
        got.backward(go.to(dev))
        ref = orc.group_points_grad(go[:, want.shape[1] - C:].contiguous(), idx, N)
        np.testing.assert_allclose(fd.grad.cpu().numpy(), ref.numpy(), rtol=1e-5, atol=1e-5 * float(ref.abs().max()) + 1e-6)
    # the unfused module path gives the same tensor
    mod.fused = False
    assert torch.equal(mod(xyz.to(dev), ctr.to(dev), feats.to(dev) if C else None).cpu(), want)


@pytest.mark.parametrize("C", [64, 128])
def test_config5_shapes_match_oracle(dev, C):
    """BASELINE config 5 at its own op shape -- N = 4096 LiDAR-like points, K = 64, r = 2.0, C in {64, 128} -- the first
    two samples of the very tensors bench.py's `roofline_hbm` times (same generator, seed 1234, B = 32): cell-grid ball
    query bit-exact, group_points a bit-exact copy, group_points_grad within fp32 rounding of the oracle's scan-order sum
    (the default kernel here is the pad-folded CSR gather: a fixed order of additions, bit-reproducible, but not the
    oracle's scan order; the reference's atomicAdd order is undefined, group_points_gpu.cu:8-25)."""
    from cmflow_amd.pointnet2_utils import pointnet2_cuda as ext
    B, N, K, r = 2, 4096, 64, 2.0
    xyz = synth.make_batch(32, N=N, seed=1234, lidar=True)["pc1"][:B].contiguous()       # (B,3,N) as bench.py builds it
    xyz_t = xyz.transpose(1, 2).contiguous()
    idx = torch.zeros(B, N, K, dtype=torch.int32, device=dev)
    ext.ball_query_wrapper(B, N, N, r, K, xyz_t.to(dev), xyz_t.to(dev), idx)
    want_idx = orc.ball_query(r, K, xyz_t, xyz_t)
    assert torch.equal(idx.cpu(), want_idx)
    gx = torch.empty(B, 3, N, K, device=dev)
    ext.group_points_wrapper(B, 3, N, N, K, xyz.to(dev), idx, gx)
    assert torch.equal(gx.cpu(), orc.group_points(xyz, want_idx))
    g = torch.Generator().manual_seed(C)
    feats = torch.randn(B, C, N, generator=g)
    out = torch.empty(B, C, N, K, device=dev)
    ext.group_points_wrapper(B, C, N, N, K, feats.to(dev), idx, out)
    assert torch.equal(out.cpu(), orc.group_points(feats, want_idx))
    go = torch.randn(B, C, N, K, generator=g)
    gp = torch.zeros(B, C, N, device=dev)
    ext.group_points_grad_wrapper(B, C, N, N, K, go.to(dev), idx, gp)
    ref = orc.group_points_grad(go, want_idx, N)
    np.testing.assert_allclose(gp.cpu().numpy(), ref.numpy(), rtol=1e-5, atol=1e-5 * float(ref.abs().max()))


def test_config5_full_batch_matches_oracle(dev):
    """BASELINE config 5 at its FULL batch: all 32 samples of the tensors bench.py's `roofline_hbm` times (seed 1234, N = 4096,
    K = 64, r = 2.0) -- the cell-grid ball query bit-exact on every sample, and the whole (32,128,4096,64) gather of the C = 128
    row against the oracle, sample by sample (the 4.3 GB output stays on the device)."""
    from cmflow_amd.pointnet2_utils import pointnet2_cuda as ext
    B, N, K, r, C = 32, 4096, 64, 2.0, 128
    xyz = synth.make_batch(B, N=N, seed=1234, lidar=True)["pc1"].contiguous()            # (B,3,N) as bench.py builds it
    xyz_t = xyz.transpose(1, 2).contiguous()
    idx = torch.zeros(B, N, K, dtype=torch.int32, device=dev)
    ext.ball_query_wrapper(B, N, N, r, K, xyz_t.to(dev), xyz_t.to(dev), idx)
    want_idx = orc.ball_query(r, K, xyz_t, xyz_t)
    assert torch.equal(idx.cpu(), want_idx)
    feats = torch.randn(B, C, N, generator=torch.Generator().manual_seed(C))
    out = torch.empty(B, C, N, K, device=dev)
    ext.group_points_wrapper(B, C, N, N, K, feats.to(dev), idx, out)
    for i in range(B):
        assert torch.equal(out[i].cpu(), orc.group_points(feats[i:i + 1], want_idx[i:i + 1])[0]), i


@pytest.mark.parametrize("B,C,N,P,S", [(64, 3, 256, 256, 32), (8, 64, 256, 256, 32), (2, 70, 256, 128, 64), (3, 5, 100, 256, 8), (2, 9, 290, 256, 4),
                                       (1, 130, 256, 256, 17), (2, 4, 600, 100, 50), (5, 1, 16, 3, 1), (2, 6, 1500, 300, 20), (1, 3, 2048, 128, 64),
                                       (2, 5, 1, 40, 9), (2, 3, 2049, 64, 32), (2, 7, 256, 256, 31)])
def test_group_points_grad_plan_form(dev, B, C, N, P, S):
    """Rows of <= 8192 entries over <= 2048 targets (the model's ball-query / kNN shapes) take the plan form: one kernel turns idx
    into the register image of the scatter workgroups, no separate inverse index (2049 targets: the inverse-index form).  Real ball-query lists (first-hit padding: heavy
    low-numbered targets, empty targets) and uniform random ones; result within fp32 rounding of the oracle's scan-order sum,
    bit-reproducible run to run, accumulating into grad_points."""
    from cmflow_amd.pointnet2_utils import pointnet2_cuda as ext
    g = torch.Generator().manual_seed(B * 7 + C + N + S)
    if N >= 100 and P <= N:
        xyz, _ = clouds(B, N, seed=N + S)
        idx = orc.ball_query(4.0, S, xyz, xyz[:, :P].contiguous())
    else:
        idx = torch.randint(0, N, (B, P, S), generator=g, dtype=torch.int32)
    go = torch.randn(B, C, P, S, generator=g)
    ref = orc.group_points_grad(go, idx, N)
    outs = []
    for _ in range(2):
        gp = torch.zeros(B, C, N, device=dev)
        ext.group_points_grad_wrapper(B, C, N, P, S, go.to(dev), idx.to(dev), gp)
        outs.append(gp)
    assert torch.equal(outs[0], outs[1])
    np.testing.assert_allclose(outs[0].cpu().numpy(), ref.numpy(), rtol=1e-5, atol=1e-5 * float(ref.abs().max()) + 1e-6)
    ext.group_points_grad_wrapper(B, C, N, P, S, go.to(dev), idx.to(dev), outs[0])
    np.testing.assert_allclose(outs[0].cpu().numpy(), 2 * ref.numpy(), rtol=1e-5, atol=2e-5 * float(ref.abs().max()) + 1e-6)


@pytest.mark.parametrize("r,S", [(2.0, 32), (16.0, 32), (4.0, 8), (0.01, 4)])
def test_group_points_grad_skewed_index(dev, r, S):
    """Real ball-query indices: first-hit padding makes the inverse lists of low-numbered points ~10x the
    mean (and r=0.01 leaves every list at exactly S entries of the point itself)."""
    from cmflow_amd.pointnet2_utils import pointnet2_cuda as ext
    xyz, _ = clouds(3, 256, seed=11)
    idx = orc.ball_query(r, S, xyz, xyz)
    go = torch.randn(3, 40, 256, S, generator=torch.Generator().manual_seed(3))
    gp = torch.zeros(3, 40, 256, device=dev)
    ext.group_points_grad_wrapper(3, 40, 256, 256, S, go.to(dev), idx.to(dev), gp)
    ref = orc.group_points_grad(go, idx, 256)
    scale = float(ref.abs().max())
    np.testing.assert_allclose(gp.cpu().numpy(), ref.numpy(), rtol=1e-5, atol=2e-6 * scale)


def test_grouping_operation_autograd(dev):
    from cmflow_amd.pointnet2_utils import grouping_operation
    g = torch.Generator().manual_seed(1)
    pts = torch.randn(2, 7, 64, generator=g)
    idx = torch.randint(0, 64, (2, 64, 8), generator=g, dtype=torch.int32)
    a = pts.clone().to(dev).requires_grad_(True)
    b = pts.clone().requires_grad_(True)
    w = torch.randn(2, 7, 64, 8, generator=g)
    (grouping_operation(a, idx.to(dev)) * w.to(dev)).sum().backward()
    (O.grouping_operation(b, idx) * w).sum().backward()
    np.testing.assert_allclose(a.grad.cpu().numpy(), b.grad.numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("B,N,S,K", [(4, 256, 256, 8), (2, 100, 77, 8), (2, 256, 256, 9), (1, 256, 256, 1),
                                     (1, 2000, 300, 16), (1, 5, 5, 8), (2, 256, 256, 32)])
def test_knn_bit_exact(dev, B, N, S, K):
    from cmflow_amd.radarflow_util import knn_point
    p1, p2 = clouds(B, max(N, S), seed=N + S + K)
    xyz, new = p2[:, :N].contiguous(), p1[:, :S].contiguous()
    ref_i, ref_d = orc.knn(K, xyz, new, return_dist=True)
    got_i, got_d = knn_point(K, xyz.to(dev), new.to(dev), return_dist=True)
    assert got_i.dtype == torch.int64
    assert torch.equal(got_i.cpu().int(), ref_i) and torch.equal(got_d.cpu(), ref_d)


def test_knn_duplicates_and_self(dev):
    from cmflow_amd.radarflow_util import knn_point
    xyz, _ = clouds(2, 200, seed=9)
    xyz = torch.cat([xyz, xyz[:, :56]], dim=1).contiguous()          # dataset-style duplicate padding
    assert torch.equal(knn_point(8, xyz.to(dev), xyz.to(dev)).cpu().int(), orc.knn(8, xyz, xyz))


def test_kabsch_kat_and_grad(dev, golden_dir):
    """Row a13: KATs produced by the reference's own WeightedKabsch (identity, equal weights,
    mirrored cloud = reflection branch, noisy, one-hot-ish weights); tolerance 1e-5 on R and t."""
    import os
    from cmflow_amd.radarflow_util import weighted_kabsch
    with np.load(os.path.join(golden_dir, "kabsch_kat.npz")) as z:
        A, Bm, W, T = (torch.from_numpy(z[k]) for k in ("A", "B", "W", "trans"))
    got = weighted_kabsch(A.to(dev), Bm.to(dev), W.to(dev)).cpu()
    # R within 1e-5; t = -R cA + cB cancels centroids of ~50 m, where the reference's own fp32
    # rounding is ~1e-5 m (its "identity" KAT returns t = 3.8e-6, the fp64-accumulating kernel 7e-15)
    np.testing.assert_allclose(got.numpy()[:, :3, :3], T.numpy()[:, :3, :3], rtol=0, atol=1e-5)
    np.testing.assert_allclose(got.numpy()[:, :, 3], T.numpy()[:, :, 3], rtol=0, atol=5e-5)
    # backward vs torch autograd through the oracle's svd-based restatement (fp64 for a clean reference)
    g = torch.Generator().manual_seed(0)
    G = torch.randn(5, 4, 4, generator=g)
    a, b, w = (t.clone().to(dev).requires_grad_(True) for t in (A, Bm, W))
    (weighted_kabsch(a, b, w) * G.to(dev)).sum().backward()
    a64, b64, w64 = (t.double().clone().requires_grad_(True) for t in (A, Bm, W))
    (O.weighted_kabsch(a64, b64, w64) * G.double()).sum().backward()
    for got_g, ref_g, name in ((a.grad, a64.grad, "A"), (b.grad, b64.grad, "B"), (w.grad, w64.grad, "W")):
        ref = ref_g.float().numpy()
        scale = np.abs(ref).max()
        np.testing.assert_allclose(got_g.cpu().numpy(), ref, rtol=1e-4, atol=1e-5 * scale, err_msg=name)


def test_layout_helpers_equal_their_torch_forms(dev):
    """cmf_pad_rows, cmf_inputs_point_major, cmf_rel_xyz: one launch each where the host side used a fill, a copy and sometimes a
    subtraction -- pure data movement (and one fp32 subtraction per coordinate), so equality is exact."""
    from cmflow_amd import fused_blocks as FB
    from cmflow_amd.radarflow_util import rel_xyz, knn_point
    g = torch.Generator().manual_seed(5)
    for rows, k in ((7, 3), (64, 5), (1000, 13), (16, 8)):
        t = torch.randn(rows, k + 6, generator=g).to(dev)[:, 2:2 + k]                # a strided view
        got = FB._pad_cols(t)
        ld = (k + 3) // 4 * 4
        assert got.shape == (rows, ld) and torch.equal(got[:, :k], t) and (ld == k or bool((got[:, k:] == 0).all()))
    for B, N, C in ((3, 200, 3), (2, 256, 5), (1, 64, 4)):
        pc1, pc2 = torch.randn(B, 3, N, generator=g).to(dev), torch.randn(B, 3, N, generator=g).to(dev)
        f1, f2 = torch.randn(B, C, N, generator=g).to(dev), torch.randn(B, C, N, generator=g).to(dev)
        x1, x2, a1, a2 = FB.inputs_point_major(pc1, pc2, f1, f2)
        cp = a1.shape[2]
        assert cp % 4 == 0 and cp >= C and cp - C < 4
        assert torch.equal(x1, pc1.transpose(1, 2)) and torch.equal(x2, pc2.transpose(1, 2))
        for a, f in ((a1, f1), (a2, f2)):
            assert torch.equal(a[:, :, :C], f.transpose(1, 2)) and bool((a[:, :, C:] == 0).all())
    xyz = (torch.rand(2, 300, 3, generator=g) * 40).to(dev)
    ctr = xyz[:, :100].contiguous()
    idx = knn_point(8, xyz, ctr, i32=True)
    d = rel_xyz(xyz, ctr, idx)
    want = torch.gather(xyz.unsqueeze(1).expand(-1, 100, -1, -1), 2, idx.long().unsqueeze(-1).expand(-1, -1, -1, 3)) - ctr.unsqueeze(2)
    assert d.shape == (2, 100, 8, 4) and torch.equal(d[..., :3], want) and bool((d[..., 3] == 0).all())


@pytest.mark.parametrize("B,N,eps,thres,score_grad", [(8, 256, 1e-4, 0.3, True), (3, 200, 0.0, 0.5, True), (5, 256, 1e-4, 0.3, False)])
def test_ego_refine_equals_the_torch_ops_around_the_solve(dev, B, N, eps, thres, score_grad):
    """cmf_ego_refine (cmflow.py:96-125 as one call per direction: ego-motion weights, weighted Kabsch, rigid refinement, select)
    against the same lines as torch ops around weighted_kabsch: outputs 2e-5 (t cancels 50 m centroids), mask equal, gradients
    w.r.t. the flow and the scores 1e-4 of their largest entry."""
    from cmflow_amd.radarflow_util import ego_refine, weighted_kabsch
    from cmflow_amd.cmflow import CMFlow
    g = torch.Generator().manual_seed(B + N)
    pc1 = (torch.rand(B, 3, N, generator=g) * torch.tensor([100.0, 60.0, 6.0]).view(1, 3, 1) - torch.tensor([10.0, 30.0, 3.0]).view(1, 3, 1)).to(dev)
    flow = (torch.randn(B, 3, N, generator=g) * 0.5 + torch.tensor([1.0, 0.2, 0.0]).view(1, 3, 1)).to(dev)
    score = torch.rand(B, 1, N, generator=g).to(dev)
    Gt, Gs = torch.randn(B, 4, 4, generator=g).to(dev), torch.randn(B, 3, N, generator=g).to(dev)
    Gt[:, 3] = 0

    def torch_tail(f, s):
        mask = (s > thres).squeeze(1)
        sc = s.squeeze(1) + eps if eps else s.squeeze(1)
        w = sc / sc.sum(dim=1).unsqueeze(1)
        T = weighted_kabsch(pc1, pc1 + f, w)
        return T, torch.where(mask.unsqueeze(1), CMFlow.rigid_to_flow(pc1, T), f), mask

    f0, s0 = flow.clone().requires_grad_(True), score.clone().requires_grad_(score_grad)
    T0, sf0, m0 = torch_tail(f0, s0)
    ((T0 * Gt).sum() + (sf0 * Gs).sum()).backward()
    f1, s1 = flow.clone().requires_grad_(True), score.clone().requires_grad_(score_grad)
    T1, sf1, m1 = ego_refine(f1, pc1, s1.squeeze(1), eps, thres)
    ((T1 * Gt).sum() + (sf1 * Gs).sum()).backward()
    assert torch.equal(m0, m1) and 0 < int(m0.sum()) < m0.numel()
    np.testing.assert_allclose(T1.detach().cpu().numpy(), T0.detach().cpu().numpy(), rtol=0, atol=2e-5)
    np.testing.assert_allclose(sf1.detach().cpu().numpy(), sf0.detach().cpu().numpy(), rtol=0, atol=5e-5)
    for got, ref, name in ((f1.grad, f0.grad, "flow"),) + (((s1.grad, s0.grad, "score"),) if score_grad else ()):
        ref = ref.cpu().numpy()
        np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=1e-3, atol=1e-4 * np.abs(ref).max(), err_msg=name)
    if not score_grad:
        assert s1.grad is None


@pytest.mark.parametrize("B,N,C,P,S,ld", [(4, 256, 512, 256, 32, 2048), (2, 256, 32, 256, 4, 32), (2, 100, 3, 100, 8, 3),
                                          (1, 300, 64, 50, 5, 64), (2, 256, 1027, 256, 8, 1027)])
def test_group_rows_and_grad(dev, B, N, C, P, S, ld):
    """Point-major grouping: bit-exact copy forward; backward is a deterministic segmented sum in
    ascending entry order == the oracle's scan-order scatter (bit-exact, unlike atomics)."""
    from cmflow_amd.fused import Neighbors, group_rows
    g = torch.Generator().manual_seed(B * N + C)
    big = torch.randn(B, N, ld, generator=g)
    col0 = (ld - C) // 2
    idx = torch.randint(0, N, (B, P, S), generator=g, dtype=torch.int32)
    idx[:, :, 0] = torch.arange(P).remainder(N).int()               # every point referenced, some hubs
    feat = big.to(dev)[:, :, col0:col0 + C].requires_grad_(True)    # strided row view, like y_all slices
    nbr = Neighbors(idx.to(dev), N)
    out = group_rows(feat, nbr)
    ref_in = big[:, :, col0:col0 + C].permute(0, 2, 1).contiguous()          # (B,C,N) for the oracle
    ref = orc.group_points(ref_in, idx).permute(0, 2, 3, 1)                  # (B,P,S,C)
    assert torch.equal(out.cpu(), ref)
    go = torch.randn(B, P, S, C, generator=g)
    out.backward(go.to(dev))
    ref_g = orc.group_points_grad(go.permute(0, 3, 1, 2).contiguous(), idx, N).permute(0, 2, 1)   # (B,N,C)
    assert torch.equal(feat.grad.cpu(), ref_g)
    feat.grad = None
    group_rows(feat, nbr).backward(go.to(dev))                       # reproducible run to run
    assert torch.equal(feat.grad.cpu(), ref_g)


def test_product_refuses_cpu_tensors():
    """No CPU fallback: the product ops raise on CPU tensors instead of computing elsewhere."""
    from cmflow_amd.pointnet2_utils import ball_query
    x = torch.zeros(1, 4, 3)
    with pytest.raises(RuntimeError):
        ball_query(1.0, 2, x, x)


@pytest.mark.parametrize("B,N,S,C,r", [(4, 256, 16, 64, 6.0), (2, 100, 8, 32, 3.0), (3, 64, 32, 512, 50.0), (1, 50, 4, 8, 0.01),
                                        (8, 256, 32, 32, 16.0), (3, 77, 16, 16, 4.0), (2, 256, 4, 128, 2.0), (5, 33, 8, 32, 100.0)])
def test_group_rows_grad_bn_closed_form(dev, B, N, S, C, r):
    """Set-conv first-layer backward (radarflow_util.py:148-151): the scatter with the BN backward folded in, once
    reading z (cmf_group_rows_grad_bn) and once from the closed form over per-point rows (cmf_group_rows_grad_bn_cf),
    against the dense torch expression -- train-mode and eval-mode BN."""
    from cmflow_amd import _lib
    from cmflow_amd.fused import Neighbors
    from cmflow_amd.fused_blocks import group_affine
    from cmflow_amd.pointnet2_utils import ball_query
    L = _lib.lib()
    p = lambda t, dt=None: _lib.dev_ptr(t, dt or t.dtype)
    g = torch.Generator().manual_seed(N + S + C)
    xyz = (torch.rand(B, N, 3, generator=g) * 20).to(dev)
    y = torch.randn(B, N, C, generator=g).to(dev)
    wx = torch.randn(C, 3, generator=g).to(dev)
    idx = ball_query(r, S, xyz, xyz)                                     # repeated slots when the ball holds < S points
    z, dxyz, _ = group_affine(y, None, xyz, xyz, wx, idx, act=0, stats=False)
    M = B * N * S
    dU = torch.randn(M, C, generator=g).to(dev)
    zf = z.view(M, C)
    mean, var = zf.mean(0), zf.var(0, unbiased=False)
    invstd = torch.rsqrt(var + 1e-5)
    a = (torch.rand(C, generator=g).to(dev) + 0.5) * invstd
    zhat = (zf - mean) * invstd
    sums = torch.stack((dU.sum(0), (dU * zhat).sum(0))).contiguous()
    nbr = Neighbors(idx.int(), N)
    off, inv = nbr.inverse()
    for train in (True, False):
        dZ = a * (dU - sums[0] / M - zhat * sums[1] / M) if train else a * dU
        want = torch.zeros(B, N, C, device=dev, dtype=torch.float64)
        want.scatter_add_(1, idx.long().view(B, N * S, 1).expand(-1, -1, C), dZ.view(B, N * S, C).double())
        got_z = torch.empty(B, N, C, device=dev)
        got_cf = torch.empty(B, N, C, device=dev)
        sp = p(sums, torch.float32) if train else None
        _lib.check(L.cmf_group_rows_grad_bn(B, N, C, N * S, p(dU), p(zf), p(a), p(mean), p(invstd), sp, 1.0 / M,
                                            p(off), p(inv), p(got_z), C, _lib.stream_ptr()), "z")
        _lib.check(L.cmf_group_rows_grad_bn_cf(B, N, C, N * S, S, p(dU), p(y), C, p(wx), 3, p(xyz), p(xyz), p(a), p(mean),
                                               p(invstd), sp, 1.0 / M, p(off), p(inv), p(got_cf), C, _lib.stream_ptr()), "cf")
        scale = float(want.abs().max())
        for name, got in (("z", got_z), ("closed form", got_cf)):
            err = float((got.double() - want).abs().max()) / scale
            assert err < 2e-6, (name, train, err)


@pytest.mark.parametrize("B,N,C", [(64, 256, 256), (3, 77, 12), (2, 1, 64), (1, 300, 100)])
def test_global_max_cat_matches_torch(dev, B, N, C):
    """cmf_global_max_cat(_grad) (cmflow.py:76-81,89-91: max over the points + expand + cat) against the torch ops:
    bit-exact output, gradient through a column block of a wider upstream gradient, first-row tie rule."""
    from cmflow_amd.fused_blocks import global_max_cat
    g = torch.Generator().manual_seed(B + N + C)
    f = torch.randn(B, N, C, generator=g)
    f[:, N // 2] = f[:, 0]                                                # duplicated point: ties between rows 0 and N//2
    f = f.to(dev).requires_grad_(True)
    wide = torch.randn(B, N, 2 * C + 8, generator=g).to(dev)             # upstream gradient lives in columns 4 : 4 + 2C
    out = global_max_cat(f)
    out.backward(wide[:, :, 4:4 + 2 * C])
    got = f.grad.clone()
    f.grad = None
    ref = torch.cat((f, f.max(dim=1, keepdim=True)[0].expand(-1, N, -1)), dim=2)
    assert torch.equal(out.detach(), ref.detach())
    # torch's own backward of max may pick either of two tied rows; the rule here is the first one
    gsum = wide[:, :, 4 + C:4 + 2 * C].sum(dim=1)
    first = (f.detach() == f.detach().max(dim=1, keepdim=True)[0]).float().argmax(dim=1)      # (B,C) first maximal row
    want = wide[:, :, 4:4 + C].clone()
    want.scatter_add_(1, first.unsqueeze(1), gsum.unsqueeze(1))
    np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=1e-6, atol=1e-5)


@pytest.mark.parametrize("B,N1,N2,K,C", [(2, 64, 64, 8, 512), (3, 51, 70, 5, 256), (1, 33, 40, 16, 1024), (4, 256, 256, 8, 512)])
def test_weightnet_ksum_matches_torch(dev, B, N1, N2, K, C):
    """cmf_weightnet_ksum(_grad): the weighting with WeightNet's last conv + ReLU (radarflow_util.py:307-318) evaluated
    inside the kernel, against the torch composition -- output, dx (dense and scattered), dh, and the conv's dW / db
    and the producer-bias column sums; the fused and the materialised-weights host paths agree."""
    from cmflow_amd.fused import Neighbors, group_rows
    from cmflow_amd.fused_blocks import WeightNetKSumFn, WeightedKSumFn
    assert WeightNetKSumFn.supported(C, 8) and not WeightNetKSumFn.supported(12, 8) and not WeightNetKSumFn.supported(C, 16)
    g = torch.Generator().manual_seed(B + N1 + K + C)
    mk = lambda *shape: torch.randn(*shape, generator=g).to(dev).requires_grad_(True)
    # h, wl, bl on a dyadic grid: the pre-activation is then exact in any summation order, so the kernel and torch
    # take the same side of the ReLU kink everywhere (a flipped slot would move dh / dwl / dbl by a whole term)
    grid = lambda t, q: (torch.round(t * q) / q).to(dev).requires_grad_(True)
    h = grid(torch.relu(torch.randn(B, N1, K, 8, generator=g)), 8)
    wl, bl, xb = grid(torch.randn(C, 8, generator=g), 16), grid(torch.randn(C, generator=g), 16), mk(C)
    go = torch.randn(B, N1, C, generator=g).to(dev)
    leaves = [h, wl, bl, xb]

    def grads_of(fn, x):
        for t in leaves + [x]:
            t.grad = None
        out = fn()
        out.backward(go)
        return [out.detach()] + [t.grad.clone() if t.grad is not None else None for t in leaves + [x]]

    def check(got, ref, leaky, x):
        names = ["out", "dh", "dwl", "dbl", "dxb", "dx"]
        for n, a, b in zip(names, got, ref):
            if n == "dxb" and b is None:
                continue
            scale = float(b.abs().max()) + 1e-6
            np.testing.assert_allclose(a.cpu().numpy() / scale, b.cpu().numpy() / scale, rtol=0, atol=2e-6, err_msg=n)

    # dense x = leaky(z + xb): the kernel returns the gradient w.r.t. the pre-activation and its column sums as dxb
    z = mk(B, N1, K, C)
    for leaky in (True, False):
        def ref_fn():
            xa = torch.nn.functional.leaky_relu(z + xb, 0.1) if leaky else z + xb
            return torch.sum(torch.relu(torch.nn.functional.linear(h, wl, bl)) * xa, dim=2)
        ref = grads_of(ref_fn, z)
        xa = (torch.nn.functional.leaky_relu(z + xb, 0.1) if leaky else z + xb).detach().requires_grad_(True)
        got = grads_of(lambda: WeightNetKSumFn.apply(h, wl, bl, xa, None, leaky, xb), xa)
        check(got, ref, leaky, z)
        if not leaky:                                                    # unfused host path (materialised weights), same numbers
            xa2 = xa.detach().requires_grad_(True)
            w_mat = lambda: torch.relu(torch.nn.functional.linear(h, wl, bl))
            got2 = grads_of(lambda: WeightedKSumFn.apply(w_mat(), xa2, None, False, False, None), xa2)
            check(got2[:4] + [None] + got2[5:], ref[:4] + [None] + ref[5:], leaky, z)
    # gathered x: per-point rows, gradient scattered back
    p = mk(B, N2, C)
    nbr = Neighbors(torch.randint(0, N2, (B, N1, K), generator=g, dtype=torch.int32).to(dev), N2)
    ref = grads_of(lambda: torch.sum(torch.relu(torch.nn.functional.linear(h, wl, bl)) * group_rows(p, nbr), dim=2), p)
    got = grads_of(lambda: WeightNetKSumFn.apply(h, wl, bl, p, nbr, False), p)
    check(got, ref, False, p)
    # run-to-run identical (fixed-order reductions)
    again = grads_of(lambda: WeightNetKSumFn.apply(h, wl, bl, p, nbr, False), p)
    for a, b in zip(got, again):
        assert (a is None and b is None) or torch.equal(a, b)


@pytest.mark.parametrize("gather", [False, True])
def test_weightnet_chain_gradients_match_torch(dev, gather):
    """WeightNet (radarflow_util.py:287-318: 3 -> 8 -> 8 -> 512 with ReLUs) through the fused chain: the last layer lives
    inside the weighting kernels, and every hidden layer's ReLU mask and bias gradient is produced by the kernel that
    consumes its output (weighting backward for the second hidden layer, the data-gradient GEMM epilogue for the first).
    Output and the gradients of all six parameters and of x against plain torch autograd.  All WeightNet operands sit on
    dyadic grids so that every pre-activation is exact in any summation order (no ReLU-kink ambiguity)."""
    from cmflow_amd.fused import Neighbors, group_rows
    from cmflow_amd.radarflow_util import WeightNet
    B, N, K, C = 3, 96, 8, 512
    g = torch.Generator().manual_seed(5 + int(gather))
    wn = WeightNet(3, C).to(dev)
    assert wn._hidden_bias() is wn.mlp_convs[1].bias
    with torch.no_grad():
        for conv, q in zip(wn.mlp_convs, (2, 2, 4)):
            conv.weight.copy_((torch.round(torch.randn(conv.weight.shape, generator=g) * q) / q).clamp(-2, 2))
            conv.bias.copy_(torch.round(torch.randn(conv.bias.shape, generator=g) * q) / q)
    dxyz = torch.nn.functional.pad((torch.round(torch.randn(B, N, K, 3, generator=g) * 4) / 4).clamp(-3, 3), (0, 1)).to(dev)
    go = torch.randn(B, N, C, generator=g).to(dev)
    if gather:
        x = torch.randn(B, N, C, generator=g).to(dev).requires_grad_(True)
        nbr = Neighbors(torch.randint(0, N, (B, N, K), generator=g, dtype=torch.int32).to(dev), N)
    else:
        x = torch.randn(B, N, K, C, generator=g).to(dev).requires_grad_(True)
        nbr = None
    params = [p for conv in wn.mlp_convs for p in (conv.weight, conv.bias)]

    def run(fn):
        for t in params + [x]:
            t.grad = None
        out = fn()
        out.backward(go)
        return [out.detach()] + [t.grad.clone() for t in params + [x]]

    def torch_path():
        w = dxyz[..., :3]
        for conv in wn.mlp_convs:
            w = torch.relu(torch.nn.functional.linear(w, conv.weight.view(conv.weight.shape[0], -1), conv.bias))
        return torch.sum(w * (group_rows(x, nbr) if gather else x), dim=2)

    ref = run(torch_path)
    got = run(lambda: wn.weighted_ksum(dxyz, x, nbr, False))
    names = ["out", "w1", "b1", "w2", "b2", "w3", "b3", "dx"]
    for n, a, b in zip(names, got, ref):
        scale = float(b.abs().max()) + 1e-6
        np.testing.assert_allclose(a.reshape(b.shape).cpu().numpy() / scale, b.cpu().numpy() / scale, rtol=0, atol=3e-6, err_msg=n)
    again = run(lambda: wn.weighted_ksum(dxyz, x, nbr, False))
    for a, b in zip(got, again):
        assert torch.equal(a, b)


@pytest.mark.parametrize("B,N1,N2,K,C", [(2, 64, 64, 8, 512), (3, 50, 70, 5, 12), (1, 33, 33, 16, 64)])
def test_weighted_ksum_matches_torch(dev, B, N1, N2, K, C):
    """cmf_weighted_ksum(_grad) (radarflow_util.py:219-221,234-236) against the torch expressions, dense and gathered."""
    from cmflow_amd.fused import Neighbors, group_rows
    from cmflow_amd.fused_blocks import WeightedKSumFn
    g = torch.Generator().manual_seed(B + N1 + K)
    w = torch.randn(B, N1, K, C, generator=g).to(dev).requires_grad_(True)
    x = torch.randn(B, N1, K, C, generator=g).to(dev).requires_grad_(True)
    go = torch.randn(B, N1, C, generator=g).to(dev)
    for leaky in (False, True):
        w.grad = x.grad = None
        WeightedKSumFn.apply(w, x, None, leaky, True).backward(go)           # relu_w: dw masked by w > 0
        gw_relu = w.grad.clone()
        w.grad = x.grad = None
        WeightedKSumFn.apply(w, x, None, leaky).backward(go)
        gw, gx = w.grad.clone(), x.grad.clone()
        assert torch.equal(gw_relu, torch.where(w > 0, gw, torch.zeros_like(gw)))
        w.grad = x.grad = None
        ref = torch.sum(w * x, dim=2)
        np.testing.assert_allclose(WeightedKSumFn.apply(w, x, None, leaky).detach().cpu().numpy(), ref.detach().cpu().numpy(),
                                   rtol=1e-5, atol=1e-5)
        ref.backward(go)
        rx = torch.where(x > 0, x.grad, 0.1 * x.grad) if leaky else x.grad
        np.testing.assert_allclose(gw.cpu().numpy(), w.grad.cpu().numpy(), rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(gx.cpu().numpy(), rx.detach().cpu().numpy(), rtol=1e-6, atol=1e-6)
    p = torch.randn(B, N2, C, generator=g).to(dev).requires_grad_(True)
    nbr = Neighbors(torch.randint(0, N2, (B, N1, K), generator=g, dtype=torch.int32).to(dev), N2)
    w.grad = None
    out = WeightedKSumFn.apply(w, p, nbr, False)
    out.backward(go)
    gw, gp = w.grad.clone(), p.grad.clone()
    w.grad = p.grad = None
    ref = torch.sum(w * group_rows(p, nbr), dim=2)
    ref.backward(go)
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().cpu().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(gw.cpu().numpy(), w.grad.cpu().numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(gp.cpu().numpy(), p.grad.cpu().numpy(), rtol=1e-5, atol=1e-5)


def test_group_points_grad_atomic_forms():
    """Long rows take the pad-folded CSR gather by default (test_group_points_and_grad covers it: random indices incl. whole
    tiles on two targets -- the workgroup-wide rank sort --, lists of 16 / 32 / 64 slots, n up to 8192, bit-reproducible).
    The LDS-atomic kernels behind it (CMF_GROUP_GRAD_CSR=0, read once per process, so a child process) stay tested on the
    same shapes."""
    import subprocess, sys, os
    env = dict(os.environ, CMF_GROUP_GRAD_CSR="0")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", __file__, "-m", "gpu", "-k", "test_group_points_and_grad"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_group_points_grad_long_rows_real_indices(dev):
    """BASELINE config 5's row length (N = 4096, K = 64: 262 144 entries per row) with real ball-query indices -- first-hit
    padding gives long runs of equal targets.  The default (pad-folded CSR gather) against the oracle, twice
    (bit-reproducible), incl. lists of 32 and 16 slots; the tiled deterministic kernel that serves the shapes the gather does
    not take (CMF_GROUP_GRAD_CSR=0 CMF_GROUP_GRAD_DETERMINISTIC=1: read once per process, so a child process) likewise."""
    import subprocess, sys, os
    from cmflow_amd.pointnet2_utils import pointnet2_cuda as ext
    from cmflow_amd import synth
    if os.environ.get("CMF_GROUP_GRAD_DETERMINISTIC") != "1":
        env = dict(os.environ, CMF_GROUP_GRAD_DETERMINISTIC="1", CMF_GROUP_GRAD_CSR="0")
        r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", __file__ + "::test_group_points_grad_long_rows_real_indices",
                            "-m", "gpu"], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    xyz = synth.make_batch(2, N=4096, seed=9, lidar=True)["pc1"].permute(0, 2, 1).contiguous()
    idx = orc.ball_query(2.0, 64, xyz, xyz)
    go = torch.randn(2, 7, 4096, 64, generator=torch.Generator().manual_seed(4))
    ref = orc.group_points_grad(go, idx, 4096)
    outs = []
    for _ in range(2):
        gp = torch.zeros(2, 7, 4096, device=dev)
        ext.group_points_grad_wrapper(2, 7, 4096, 4096, 64, go.to(dev), idx.to(dev), gp)
        outs.append(gp.cpu())
    assert torch.equal(outs[0], outs[1])                               # lists of 64 slots: the pad-folded CSR gather (default)
    np.testing.assert_allclose(outs[0].numpy(), ref.numpy(), rtol=1e-5, atol=2e-6 * float(ref.abs().max()))
    # lists of 32 and 16 slots (two / four lists per wave in the list-per-lanes kernel), padded tails included
    for r, ns in ((1.2, 32), (0.8, 16)):
        idx = orc.ball_query(r, ns, xyz, xyz)
        go = torch.randn(2, 5, 4096, ns, generator=torch.Generator().manual_seed(ns))
        ref = orc.group_points_grad(go, idx, 4096)
        gp = torch.zeros(2, 5, 4096, device=dev)
        ext.group_points_grad_wrapper(2, 5, 4096, 4096, ns, go.to(dev), idx.to(dev), gp)
        np.testing.assert_allclose(gp.cpu().numpy(), ref.numpy(), rtol=1e-5, atol=2e-6 * float(ref.abs().max()))


@pytest.mark.parametrize("B,C,N,P,S,hi", [(2, 5, 2000, 1000, 16, 2000), (1, 3, 8192, 300, 32, 8192), (2, 4, 4096, 4100, 16, 4096), (1, 6, 4096, 1024, 64, 3),
                                          (2, 3, 300, 2048, 32, 300), (1, 2, 5000, 700, 64, 1)])
def test_group_points_grad_csr_index_variants(dev, B, C, N, P, S, hi):
    """The pad-folded CSR gather on arbitrary (not ball-query) indices: every wave count of the index kernel (16 / 8 / 4 waves for
    n <= 2048 / 4096 / 8192), ragged last tiles, and a handful of targets taking whole tiles (hi = 3, 1: segments of ~10 000 entries,
    every list 'padded' by chance).  Against the oracle's scan-order sum, bit-reproducible."""
    from cmflow_amd.pointnet2_utils import pointnet2_cuda as ext
    g = torch.Generator().manual_seed(N + P + S + hi)
    idx = torch.randint(0, hi, (B, P, S), generator=g, dtype=torch.int32)
    go = torch.randn(B, C, P, S, generator=g)
    ref = orc.group_points_grad(go, idx, N)
    outs = []
    for _ in range(2):
        gp = torch.zeros(B, C, N, device=dev)
        ext.group_points_grad_wrapper(B, C, N, P, S, go.to(dev), idx.to(dev), gp)
        outs.append(gp.cpu())
    assert torch.equal(outs[0], outs[1])
    np.testing.assert_allclose(outs[0].numpy(), ref.numpy(), rtol=2e-5, atol=1e-5 * float(ref.abs().max()) + 1e-6)


@pytest.mark.parametrize("B,N,S,C,r", [(4, 256, 16, 512, 8.0), (2, 256, 32, 64, 16.0), (3, 100, 8, 32, 4.0)])
def test_group_affine_statistics_only_form(dev, B, N, S, C, r):
    """cmf_group_affine with z == NULL (the tensor is formed again where it is consumed): the BN partial sums, the z * d_k sums and
    dxyz must be bit-identical to the writing form's."""
    from cmflow_amd import fused_blocks as FB, pointnet2_utils as pu
    xyz, _ = clouds(B, N, seed=S + C)
    xyz = xyz.to(dev)
    idx = pu.ball_query(r, S, xyz, xyz)
    g = torch.Generator(device="cpu").manual_seed(C)
    y = torch.randn(B, N, C, generator=g).to(dev); wx = torch.randn(C, 3, generator=g).to(dev)
    z, d, p, px = FB.group_affine(y, None, xyz, xyz, wx, idx, extra=True)
    z0, d0, p0, px0 = FB.group_affine(y, None, xyz, xyz, wx, idx, extra=True, write_z=False)
    assert z0 is None and torch.equal(d, d0) and torch.equal(p, p0) and torch.equal(px, px0)
    z1, d1, p1 = FB.group_affine(y, None, xyz, xyz, wx, idx, write_z=False)
    assert z1 is None and torch.equal(d, d1) and torch.equal(p, p1)


@pytest.mark.parametrize("env", [dict(CMF_BALL_QUERY_GRID="0"), dict(CMF_BALL_QUERY_BALLOT="0"), dict(CMF_GROUP_GRAD_PLAN="0"),
                                 dict(CMF_GROUP_GRAD_CSR="0")],
                         ids=lambda e: ",".join("%s=%s" % kv for kv in e.items()))
def test_drop_in_ab_switches_change_nothing(tmp_path, env):
    """The A/B switches of the drop-in neighbour search / grouping calls (cell grid vs scan at N = 4096, ballot vs scan kernels at N = 256,
    plan / CSR forms of the grouping gradient vs their predecessors), each forced in a child process (the library reads its environment
    once): indices and grouped tensors bit-identical; the gradient bit-identical where both forms are deterministic sums in the same
    order, else within 1e-6 of its largest entry (the LDS-atomic predecessor adds in arrival order)."""
    import os, subprocess, sys
    outs = []
    for i, e in enumerate(({}, env)):
        f = str(tmp_path / ("ops%d.pt" % i))
        r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "op_dump.py"), f],
                           env=dict(os.environ, **e), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        outs.append(torch.load(f))
    a, b = outs
    for k in a:
        if k.endswith(".grad"):
            assert float((a[k] - b[k]).abs().max()) <= 1e-6 * float(a[k].abs().max()), (env, k)
        else:
            assert torch.equal(a[k], b[k]), (env, k)
