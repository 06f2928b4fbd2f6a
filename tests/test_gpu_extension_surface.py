"""GPU: the rest of the reference's `pointnet2_cuda` export table (lib/src/pointnet2_api.cpp:10-25) -- the seven
kernels CMFlow itself never calls -- against the C oracle.  Indices bit-exact, copies bit-exact, scatter-adds
to 1e-5 (the reference uses unordered atomics there)."""
import numpy as np
import pytest
import torch

from cmflow_amd import synth
from oracle import ops as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _clouds(B, N, M, seed):
    b = synth.make_batch(B, N=max(N, M), seed=seed)
    return b["pc1"].permute(0, 2, 1)[:, :N].contiguous(), b["pc2"].permute(0, 2, 1)[:, :M].contiguous()


def test_export_table_is_complete():
    from cmflow_amd.pointnet2_utils import pointnet2_cuda as ext
    for name in ("ball_query_wrapper", "group_points_wrapper", "group_points_grad_wrapper", "gather_points_wrapper",
                 "gather_points_grad_wrapper", "furthest_point_sampling_wrapper", "knn_wrapper", "three_nn_wrapper",
                 "three_interpolate_wrapper", "three_interpolate_grad_wrapper"):
        assert callable(getattr(ext, name))


@pytest.mark.parametrize("B,N,m", [(3, 256, 64), (2, 1000, 128), (1, 37, 37)])
def test_furthest_point_sampling(dev, B, N, m):
    from cmflow_amd.pointnet2_utils import furthest_point_sample
    xyz, _ = _clouds(B, N, N, seed=N + m)
    ref = torch.empty(B, m, dtype=torch.int32)
    orc.furthest_point_sampling_wrapper(B, N, m, xyz, torch.full((B, N), 1e10), ref)
    assert torch.equal(furthest_point_sample(xyz.to(dev), m).cpu(), ref)


@pytest.mark.parametrize("B,N,M,k", [(2, 256, 256, 8), (2, 100, 300, 16), (1, 64, 50, 3), (1, 128, 500, 40)])
def test_knn_and_three_nn(dev, B, N, M, k):
    from cmflow_amd.pointnet2_utils import knn, three_nn
    unknown, known = _clouds(B, N, M, seed=N + M + k)
    d_ref, i_ref = torch.empty(B, N, k), torch.empty(B, N, k, dtype=torch.int32)
    orc.knn_wrapper(B, N, M, k, unknown, known, d_ref, i_ref)
    d, i = knn(k, unknown.to(dev), known.to(dev))
    assert torch.equal(i.cpu(), i_ref)
    np.testing.assert_allclose(d.cpu().numpy(), np.sqrt(d_ref.numpy()), rtol=1e-6, atol=0)
    d3_ref, i3_ref = torch.empty(B, N, 3), torch.empty(B, N, 3, dtype=torch.int32)
    orc.three_nn_wrapper(B, N, M, unknown, known, d3_ref, i3_ref)
    d3, i3 = three_nn(unknown.to(dev), known.to(dev))
    assert torch.equal(i3.cpu(), i3_ref)
    np.testing.assert_allclose((d3 * d3).cpu().numpy(), d3_ref.numpy(), rtol=1e-5, atol=1e-7)


def test_gather_and_interpolate(dev):
    from cmflow_amd.pointnet2_utils import gather_operation, three_interpolate
    g = torch.Generator().manual_seed(2)
    B, C, N, P = 2, 19, 300, 77
    feats = torch.randn(B, C, N, generator=g)
    idx = torch.randint(0, N, (B, P), generator=g, dtype=torch.int32)
    a = feats.clone().to(dev).requires_grad_(True)
    out = gather_operation(a, idx.to(dev))
    ref = torch.empty(B, C, P)
    orc.gather_points_wrapper(B, C, N, P, feats, idx, ref)
    assert torch.equal(out.cpu(), ref)
    go = torch.randn(B, C, P, generator=g)
    out.backward(go.to(dev))
    gref = torch.zeros(B, C, N)
    orc.gather_points_grad_wrapper(B, C, N, P, go, idx, gref)
    np.testing.assert_allclose(a.grad.cpu().numpy(), gref.numpy(), rtol=1e-5, atol=1e-5)
    # three_interpolate: (B,C,M) known features -> (B,C,n)
    M, n = 40, 123
    known = torch.randn(B, C, M, generator=g)
    idx3 = torch.randint(0, M, (B, n, 3), generator=g, dtype=torch.int32)
    w = torch.rand(B, n, 3, generator=g)
    w = w / w.sum(-1, keepdim=True)
    k = known.clone().to(dev).requires_grad_(True)
    o = three_interpolate(k, idx3.to(dev), w.to(dev))
    oref = torch.empty(B, C, n)
    orc.three_interpolate_wrapper(B, C, M, n, known, idx3, w, oref)
    assert torch.equal(o.cpu(), oref)                       # same non-contracted (w0*p0 + w1*p1) + w2*p2
    go = torch.randn(B, C, n, generator=g)
    o.backward(go.to(dev))
    gk = torch.zeros(B, C, M)
    orc.three_interpolate_grad_wrapper(B, C, n, M, go, idx3, w, gk)
    np.testing.assert_allclose(k.grad.cpu().numpy(), gk.numpy(), rtol=1e-5, atol=1e-5)


def test_integration_md_stub_runs_as_written(dev):
    """The ctypes stub of INTEGRATION.md (seam 1: a drop-in `pointnet2_cuda.py`) is executed verbatim, only the library path
    filled in, and its three wrappers are checked against the oracle."""
    import os
    import re
    import types
    from cmflow_amd import _lib
    from oracle import ops as orc
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(repo, "INTEGRATION.md")).read()
    code = re.search(r"```python\n# pointnet2_cuda.py.*?```", text, flags=re.S).group(0)
    code = code[len("```python\n"):-3].replace("/path/to/cmflow_amd/libcmflow_hip.so", _lib.SO_PATH)
    mod = types.ModuleType("pointnet2_cuda_stub")
    exec(compile(code, "INTEGRATION.md", "exec"), mod.__dict__)
    xyz, new = _clouds(2, 200, 64, seed=9)
    idx = torch.zeros(2, 64, 8, dtype=torch.int32, device=dev)
    assert mod.ball_query_wrapper(2, 200, 64, 3.0, 8, new.to(dev), xyz.to(dev), idx) == 1
    assert torch.equal(idx.cpu(), orc.ball_query(3.0, 8, xyz, new))
    feats = torch.randn(2, 5, 200)
    out = torch.empty(2, 5, 64, 8, device=dev)
    mod.group_points_wrapper(2, 5, 200, 64, 8, feats.to(dev), idx, out)
    assert torch.equal(out.cpu(), orc.group_points(feats, idx.cpu()))
    go = torch.randn(2, 5, 64, 8)
    gp = torch.zeros(2, 5, 200, device=dev)
    mod.group_points_grad_wrapper(2, 5, 200, 64, 8, go.to(dev), idx, gp)
    torch.testing.assert_close(gp.cpu(), orc.group_points_grad(go, idx.cpu(), 200), rtol=1e-5, atol=1e-5)
    # the second stub: QueryAndGroup.forward bound to cmf_query_and_group, executed in the same namespace and attached to a
    # bare object carrying the module's three attributes; against the reference's op sequence on the oracle's kernels
    code2 = re.search(r"```python\n# lib/pointnet2_utils.py:269-292, body of QueryAndGroup.forward.*?```", text, flags=re.S).group(0)
    exec(compile(code2[len("```python\n"):-3], "INTEGRATION.md", "exec"), mod.__dict__)
    for use_xyz, with_feats in ((True, True), (True, False), (False, True)):
        qg = types.SimpleNamespace(radius=3.0, nsample=8, use_xyz=use_xyz)
        f = feats if with_feats else None
        got = mod.forward(qg, xyz.to(dev), new.to(dev), f.to(dev) if with_feats else None)
        ref_idx = orc.ball_query(3.0, 8, xyz, new)
        gx = orc.group_points(xyz.transpose(1, 2).contiguous(), ref_idx) - new.transpose(1, 2).unsqueeze(-1)
        parts = ([gx] if (use_xyz or not with_feats) else []) + ([orc.group_points(f, ref_idx)] if with_feats else [])
        assert torch.equal(got.cpu(), torch.cat(parts, dim=1))


def test_library_memory_statistics_count_the_drop_in_scratch():
    """cmf_mem_stats (bench.py's extra.peak_mem_bytes.library_scratch): the per-(stream, slot) working buffers the drop-in calls keep --
    the inverse index / plan of cmf_group_points_grad has no argument in the reference's signature, so the library owns it."""
    from cmflow_amd import _lib, pointnet2_utils as pu, synth
    dev = torch.device("cuda:0")
    L = _lib.lib()
    before = L.cmf_mem_stats()
    xyz = synth.make_batch(2, N=256, seed=5)["pc1"].to(dev).transpose(1, 2).contiguous()
    idx = pu.ball_query(2.0, 16, xyz, xyz)
    feats = torch.randn(2, 8, 256, device=dev, requires_grad=True)
    pu.grouping_operation(feats, idx).sum().backward()
    torch.cuda.synchronize()
    after = L.cmf_mem_stats()
    assert after >= before and after > 0 and after % (1 << 20) == 0          # grown in whole MiB, kept for the process
