#!/usr/bin/env python3
"""Headline benchmark: frame-pairs/s of the CMFlow training step (fwd+bwd, cross-modal losses,
Adam) on synthetic N=256 radar clouds at B=64 per GPU (BASELINE.json metric / configs[2]).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One process per GPU; batches shard over ranks (weak scaling: 64 pairs per GPU); the only
collective is one RCCL all-reduce of the flat gradient bucket per step.  Rank 0 prints ONE
JSON line.  The dominant hand-written kernel is cmf_gemm (the fp32 MFMA GEMM behind every 1x1 conv); every launch
>= 1 GFLOP is bracketed live with HIP events recorded INSIDE the library on the launch stream (the GEMMs of the
set-conv block calls count too).  `roofline` = what the STEP extracts from the MFMA pipe: sum of 2*M*N*K over the
bracketed launches of the timed region / the region's wall time (chains run on four streams, so per-launch
durations overlap and their sum exceeds the wall time: that contended per-launch figure is kept under `extra`);
`roofline_isolated` = the kernel's own rate (the same launches with every chain on one stream: flops / sum of
durations); peak = 157.3 TFLOP/s dense fp32 MFMA; `flop_share` = share of all cmf_gemm FLOPs the bracketed launches carry.  `roofline_hbm` (N=1 only, after the timed region): the drop-in ball_query + group_points
(+ group_points_grad) kernels at the op shapes of SURVEY 8d against the 8 TB/s HBM peak.
`cpu_baseline` is the CPU oracle ("port") timed on this box's host cores on a bounded sample of the
same workload; `cpu_baseline_config1` is BASELINE config 1 (B=1 forward, eval mode) on the same oracle.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs on this stack (already exported on the pool)
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
TRAFFIC_PROFILE = "r06_gemm_traffic_instep.json"   # committed FETCH_SIZE / WRITE_SIZE passes the `roofline.traffic` constant comes from
MFMA_F32_PEAK_TFLOPS = 157.3    # MI355X_MICROARCH.md: fp32-input MFMA dense peak
OP_TRAFFIC_PROFILE = "r06_op_hbm_pmc.json"        # committed PMC passes over the drop-in calls (tools/session.sh op_pmc)


class Args:
    num_points = 256
    stat_thres = 0.5
    vr_thres = 0.3
    rigid_thres = 0.15


def load_weights(model_name):
    from cmflow_amd import synth
    gold = os.path.join(REPO, "tests", "golden")
    man = json.load(open(os.path.join(gold, "state_manifest_%s.json" % model_name)))
    calib = os.path.join(gold, "bn_calib_%s.npz" % model_name)
    return synth.synth_state_dict(man, seed=1234, calib=calib if os.path.exists(calib) else None)


def cpu_baseline(mode, model_name, budget_s=20.0, B=4):
    """The CPU oracle (a port: own torch-CPU restatement + C ops) on this box's host cores."""
    from cmflow_amd import synth
    from oracle import cmflow_oracle as O
    from oracle import train_oracle as TO
    # host threads: the cores this process may run on, capped at 16 -- the tiny per-layer tensors of a
    # B=4 step do not scale further (256 threads measured 100x SLOWER than 16 on the GPU box's host)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(16, avail))
    torch.set_num_threads(cores)
    net = {"cmflow": O.CMFlow, "cmflow_t": O.CMFlow_T, "raflow": O.RaFlow}[model_name](Args())
    net.load_state_dict(load_weights(model_name))
    b = synth.make_batch(B, seed=1, train_extras=True)
    P, Tcr = torch.tensor(synth.CAMERA_PROJECTION), torch.tensor(synth.T_CAMERA_RADAR)
    if mode == "train":
        net.train()
        opt = torch.optim.Adam(net.parameters(), lr=1e-6 if model_name == "raflow" else 0.001, weight_decay=1e-4)
        if model_name == "raflow":
            def one():
                loss, _ = TO.self_supervised_loss(b, net(b["pc1"], b["pc2"], b["ft1"], b["ft2"], b["interval"])[1])
                opt.zero_grad(); loss.backward(); opt.step()
        elif model_name == "cmflow_t":
            def one():
                dyn, mseg = TO.make_labels(b)
                out = net(b["pc1"], b["pc2"], b["ft1"], b["ft2"], mseg, "train", None)
                loss, _ = TO.radar_flow_loss(b, out[0], out[2], out[1], mseg, dyn, P, Tcr)
                opt.zero_grad(); loss.backward(); opt.step()
        else:
            def one():
                TO.train_step(net, opt, b, P, Tcr)
        what = "fwd+bwd+%s losses+Adam, train-mode BN" % ("3 self-supervised" if model_name == "raflow" else "7")
    else:
        net.eval()

        def one():
            with torch.no_grad():
                if model_name == "cmflow_t":
                    net(b["pc1"], b["pc2"], b["ft1"], b["ft2"], None, "test", None)
                elif model_name == "raflow":
                    net(b["pc1"], b["pc2"], b["ft1"], b["ft2"], b["interval"])
                else:
                    net(b["pc1"], b["pc2"], b["ft1"], b["ft2"], None, "test")
        what = "fwd, eval-mode BN"
    t_end = time.perf_counter() + budget_s
    one()                                           # warm-up (counts against the budget)
    times = []
    while (time.perf_counter() < t_end and len(times) < 10) or not times:
        t0 = time.perf_counter()
        one()
        times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    return {"value": round(B / med, 3), "unit": "frame-pairs/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d steps of B=%d N=256 %s (%s), median; oracle/ torch-CPU + C ops" % (len(times), B, model_name, what)}


def gemm_shape_table(prof, title):
    """Markdown table of the bracketed cmf_gemm launches grouped by (shape, layout, epilogue kind)."""
    rows = []
    for (M, N, K, layout, kind, split, bm, bn), (n, ms) in prof["shapes"].items():
        fl = 2.0 * M * N * K * n
        rows.append((ms, "| %d x %d x %d | %s | %s | %d | %dx%s | %d | %.3f | %.1f | %.3f |" % (
            M, N, K, ("A[M,K] W[N,K]", "A[M,K] B[K,N]", "A[K,M] B[K,N]", "A[K,M] B[N,K]")[(1, 0, 2, 3)[layout & 3]],
            ("store", "fwd bias/act/stats", "bwd BN+ReLU", "bwd (leaky) ReLU", "bwd BN+ReLU + dxyz sums", "bwd (leaky) ReLU + dxyz sums")[kind], split, bm, ("%dp" % bn) if layout & 4 else str(bn), n, ms,
            fl / (ms * 1e-3) / 1e12, fl / (ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS)))
    rows.sort(reverse=True)
    tot_ms = prof["ms"]
    head = ("### %s\n\n%d launches, %.2f ms, %.1f TFLOP/s = %.3f of %.1f\n\n| M x N x K | operands | epilogue | split-K | tile | launches | "
            "ms (sum) | TFLOP/s | of peak |\n|---|---|---|---|---|---|---|---|---|\n"
            % (title, prof["launches"], tot_ms, prof["units"] / (tot_ms * 1e-3) / 1e12,
               prof["units"] / (tot_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, MFMA_F32_PEAK_TFLOPS))
    return head + "\n".join(r for _, r in rows) + "\n"


def hbm_op_rooflines(dev, iters=20):
    """The drop-in kernels of the boundary (cmf_ball_query, cmf_group_points, cmf_group_points_grad through the C-ABI)
    at the op-level shapes of SURVEY 8d / BASELINE.md section 4, inputs resident in HBM.  Per op: `iters` back-to-back
    launches between one HIP event pair on the launch stream (kernel duration incl. the ~1.5 us launch gap).
    Algorithmic bytes (every tensor touched once):
      ball_query+group = 2*B*N*12 (xyz, centres) + B*C*N*4 (features) + B*M*K*4 (idx) + B*(3+C)*M*K*4 (grouped out)
                         over t(ball_query) + t(group xyz) + t(group features); `query_and_group`: the same bytes over ONE
                         cmf_query_and_group call (QueryAndGroup.forward as the reference's module issues it)
      group_grad       = B*C*M*K*4 (grad_out) + B*M*K*4 (idx) + B*C*N*4 (grad_points)"""
    from cmflow_amd import _lib, synth
    L = _lib.lib()
    st = _lib.stream_ptr()

    def timed(fn):
        for _ in range(3):                  # synchronised warm-up calls: library scratch grown by the first call retires its old
            fn()                            # buffer behind an event and a LATER call frees it (a multi-ms hipFree) -- that one-off
            torch.cuda.synchronize()        # must happen here, not between the events below
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e-3

    def check_sample0(xyz, xyz_t, r, K, idx, feats, out, fused):
        """One batch row of the timed outputs against the definition of the ops written in plain torch on the device (no
        oracle here: that is tests/test_gpu_ops.py::test_config5_shapes_match_oracle's job): ball query = first K points in
        index order with ((dx*dx)+(dy*dy))+(dz*dz) < r*r, every operation individually rounded, padded with the first hit;
        grouping = a gather; the fused call = cat(xyz[idx] - centre, feats[idx])."""
        p = xyz_t[0]                                                                     # (N,3)
        d = p[:, None, :] - p[None, :, :]                                                # centre - point
        d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
        hit = d2 < (torch.tensor(r, device=dev) * torch.tensor(r, device=dev))
        rank = torch.cumsum(hit.int(), dim=1) - 1
        want = torch.zeros(p.shape[0], K, dtype=torch.int32, device=dev)
        sel = hit & (rank < K)
        ci, pi = sel.nonzero(as_tuple=True)
        want[ci, rank[ci, pi].long()] = pi.int()
        cnt = hit.sum(dim=1).clamp(max=K)
        pad = torch.arange(K, device=dev)[None, :] >= cnt[:, None]
        want = torch.where(pad & (cnt[:, None] > 0), want[:, :1].expand(-1, K), want)
        assert torch.equal(idx[0], want), "ball query: timed output differs from its definition"
        g = feats[0][:, want.long()]                                                     # (C,N,K)
        assert torch.equal(out[0], g), "group_points: timed output differs from its definition"
        rel = xyz[0][:, want.long()] - xyz[0][:, :, None]
        assert torch.equal(fused[0], torch.cat((rel, g), dim=0)), "query_and_group: timed output differs from its definition"

    # the launch floor of this box: an (almost) empty one-wave kernel back to back on the same stream, timed the same way -- what a
    # launch costs before it moves a byte; the latency-bound rows below are printed as multiples of it (`x_floor`)
    floor1 = timed(lambda: _lib.check(L.cmf_debug_spin(0.005, st), "spin"))

    def two():
        _lib.check(L.cmf_debug_spin(0.005, st), "spin"); _lib.check(L.cmf_debug_spin(0.005, st), "spin")
    floor2 = timed(two)
    rows = []
    for (B, N, K, r, lidar), Cs in (((64, 256, 32, 2.0, False), (3, 64, 1027)), ((32, 4096, 64, 2.0, True), (64, 128))):
        xyz = synth.make_batch(B, N=N, seed=1234, lidar=lidar)["pc1"].to(dev)          # (B,3,N) channel-major
        xyz_t = xyz.transpose(1, 2).contiguous()                                        # (B,N,3)
        idx = torch.zeros(B, N, K, dtype=torch.int32, device=dev)
        gx = torch.empty(B, 3, N, K, device=dev)
        bq = lambda: _lib.check(L.cmf_ball_query(B, N, N, r, K, xyz_t.data_ptr(), xyz_t.data_ptr(), idx.data_ptr(), st), "bq")
        t_bq = timed(bq)
        t_gx = timed(lambda: _lib.check(L.cmf_group_points(B, 3, N, N, K, xyz.data_ptr(), idx.data_ptr(), gx.data_ptr(), st), "gx"))
        for C in Cs:
            feats = torch.randn(B, C, N, device=dev)
            out = torch.empty(B, C, N, K, device=dev)
            t_gf = timed(lambda: _lib.check(L.cmf_group_points(B, C, N, N, K, feats.data_ptr(), idx.data_ptr(), out.data_ptr(), st), "gf"))
            # the same as ONE call (QueryAndGroup.forward, lib/pointnet2_utils.py:269-292)
            fused = torch.empty(B, 3 + C, N, K, device=dev)
            idx2 = torch.empty(B, N, K, dtype=torch.int32, device=dev)
            t_qg = timed(lambda: _lib.check(L.cmf_query_and_group(B, N, N, r, K, C, 1, xyz_t.data_ptr(), xyz_t.data_ptr(), feats.data_ptr(),
                                                                  idx2.data_ptr(), fused.data_ptr(), st), "qg"))
            if C == Cs[-1]:
                check_sample0(xyz, xyz_t, r, K, idx, feats, out, fused)
            del fused, idx2
            out.normal_()
            gp = torch.zeros(B, C, N, device=dev)
            t_gg = timed(lambda: _lib.check(L.cmf_group_points_grad(B, C, N, N, K, out.data_ptr(), idx.data_ptr(), gp.data_ptr(), st), "gg"))
            nb = 2 * B * N * 12 + B * C * N * 4 + B * N * K * 4 + B * (3 + C) * N * K * 4
            ng = B * C * N * K * 4 + B * N * K * 4 + B * C * N * 4
            t_f = t_bq + t_gx + t_gf
            rows.append({"shape_BNKC": [B, N, K, C],
                         "ball_query+group": {"bytes": nb, "us": round(t_f * 1e6, 1), "achieved": round(nb / t_f / 1e9, 1),
                                              "frac": round(nb / t_f / 1e9 / HBM_PEAK_GBS, 4),
                                              "us_parts": [round(t * 1e6, 1) for t in (t_bq, t_gx, t_gf)]},
                         "query_and_group": {"bytes": nb, "us": round(t_qg * 1e6, 1), "achieved": round(nb / t_qg / 1e9, 1),
                                             "frac": round(nb / t_qg / 1e9 / HBM_PEAK_GBS, 4)},
                         "group_grad": {"bytes": ng, "us": round(t_gg * 1e6, 1), "achieved": round(ng / t_gg / 1e9, 1),
                                        "frac": round(ng / t_gg / 1e9 / HBM_PEAK_GBS, 4)}})
            for k in ("ball_query+group", "query_and_group", "group_grad"):
                rows[-1][k]["x_floor"] = round(rows[-1][k]["us"] * 1e-6 / floor1, 1)
            del feats, out, gp
    # counted HBM bytes per call from the committed PMC passes over the same calls (tools/session.sh op_pmc): constants of a
    # profiled build, NOT measured by this run
    tj = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", OP_TRAFFIC_PROFILE)
    src_ok = None
    if os.path.exists(tj):
        rec = json.load(open(tj))
        src_ok = rec.get("source_id") == _lib.source_id()       # constants of another build of the kernels are not reported
        table = {(r["op"], tuple(r["shape_BNKC"])): (r["traffic_bytes"] if src_ok else None) for r in rec["rows"]}
        for row in rows:
            key = tuple(row["shape_BNKC"])
            row["query_and_group"]["traffic_committed"] = table.get(("query_and_group", key))
            row["group_grad"]["traffic_committed"] = table.get(("group_points_grad", key))
    return {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None,
            "launch_floor_us": round(floor1 * 1e6, 2), "launch_floor_pair_us": round(floor2 * 1e6, 2),
            "launch_floor_note": "one / two back-to-back launches of a one-wave kernel that returns at once (cmf_debug_spin with zero ticks), timed like the "
                                 "rows; x_floor = row us / launch_floor_us",
            "traffic_source": ("per-row traffic_committed: constants from the committed rocprofv3 PMC passes (profiles/%s), not measured by "
                               "this run; null when the profile was taken on other kernel sources" % OP_TRAFFIC_PROFILE),
            "traffic_profile": "profiles/" + OP_TRAFFIC_PROFILE, "traffic_profile_matches_sources": src_ok, "source_id": _lib.source_id(),
            "kernels": ["ball_query_ballot_kernel / bq_grid_*", "group_points_kernel", "query_and_group_kernel", "group_points_grad_*_kernel"],
            "method": "%d back-to-back launches per op between one HIP event pair; bytes = SURVEY 8d algorithmic bytes" % iters,
            "rows": rows}


def clip_from_loader(batch, clip_len, rank, dev):
    """`batch` synthetic mini-clips of `clip_len` frames written as sample files (ragged clouds of 200-340 points), read back through
    cmflow_amd.dataset.vodClipDataset + a DataLoader, one resident batch dict per frame."""
    import shutil
    import tempfile
    import numpy as np
    from cmflow_amd import dataset as D

    class A:
        num_points, eval, mini_clip_len, update_len = 256, False, clip_len, 1

    tmp = tempfile.mkdtemp(prefix="cmf_bench_clip_")
    try:
        clips = tuple(("train", "delft_%d" % c, tuple(200 + (37 * c + 53 * f) % 141 for f in range(clip_len))) for c in range(batch))
        D.write_synthetic_split(tmp, seed=1234 + 100003 * rank, clips=clips)
        ds = D.vodClipDataset(A(), root=tmp + "/", partition="train")
        assert len(ds) == batch, (len(ds), batch)
        np.random.seed(1234 + rank)
        data = next(iter(torch.utils.data.DataLoader(ds, batch_size=batch, shuffle=False)))
        return [D.as_batch_dict(D.extract_data_info_clip(data, j, device=dev)) for j in range(clip_len)]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 100: a 2 s timed region; the driver passes its own)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", choices=["train", "fwd"], default="train")
    ap.add_argument("--model", choices=["cmflow", "cmflow_t", "raflow"], default="cmflow")
    ap.add_argument("--batch", type=int, default=64, help="frame pairs per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--path", choices=["pm"], default="pm", help="pm: the fused point-major HIP path (the product has one path)")
    ap.add_argument("--serial", action="store_true",
                    help="diagnostic: every chain of the step on ONE stream (per-kernel durations free of contention)")
    ap.add_argument("--force-allreduce", action="store_true",
                    help="diagnostic at N=1: run the RCCL all-reduce of the gradient bucket (world size 1) inside every step")
    ap.add_argument("--no-op-rooflines", action="store_true", help="skip the roofline_hbm op benchmarks after the timed region")
    ap.add_argument("--no-config2", action="store_true", help="skip the fwd-only (BASELINE config 2) timing behind the timed regions (profiling runs: "
                                                              "keeps the kernel statistics of a training run free of inference steps)")
    ap.add_argument("--clip", type=int, default=5, help="cmflow_t: frames per mini-clip (clip_util.py:34-62)")
    ap.add_argument("--gemm-table", default=None, help="write a markdown table of the bracketed cmf_gemm launches by shape to this file")
    ap.add_argument("--host-cores", type=int, default=0,
                    help="restrict this process (and the library's chain threads) to K host cores: the budget a rank has when N ranks "
                         "share one host (nproc / N); applied with sched_setaffinity before anything touches the GPU")
    ap.add_argument("--watchdog", type=float, default=0.0,
                    help="seconds after which a rank that has not finished exits non-zero (default: 600 at world > 1, off at world 1)")
    a = ap.parse_args()
    if a.steps is None:
        a.steps = 100
    host_cores = None
    if a.host_cores > 0:
        # in-process, before any HIP call (never a re-exec): the cores are taken from the allowed set in order, offset by the local
        # rank so that ranks sharing a host get disjoint sets
        allowed = sorted(os.sched_getaffinity(0))
        k = min(a.host_cores, len(allowed))
        lr = int(os.environ.get("LOCAL_RANK", "0"))
        first = (lr * k) % max(1, len(allowed) - k + 1)
        os.sched_setaffinity(0, set(allowed[first:first + k]))
        host_cores = {"requested": a.host_cores, "granted": len(os.sched_getaffinity(0)), "of": len(allowed)}
        torch.set_num_threads(max(1, min(torch.get_num_threads(), k)))

    # stdout carries exactly ONE line (the JSON record): libraries that print to the C-level stdout (RCCL's version banner
    # at communicator creation) are sent to stderr, and the record is written to the saved descriptor at the end.  Done
    # here, not at import: tools and tests import this module for Args / load_weights and keep their own stdout.
    result_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        print("warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (a.gpus, world), file=sys.stderr)
    import torch.distributed as dist
    # CMF_BENCH_ONE_GPU=1: every rank on cuda:0 over gloo -- exercises the multi-rank control flow (collectives entered by
    # all ranks, one JSON line, clean exit) on a 1-GPU box; its numbers mean nothing (tests/test_gpu_model.py)
    one_gpu = os.environ.get("CMF_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 and not one_gpu:
        from cmflow_amd.dp import pin_rank_to_cores
        # opt-in (CMF_PIN_CORES=1), before the library creates its chain-worker threads (they inherit the mask); ranks on THIS host
        pin_rank_to_cores(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    if world == 1 and (a.force_allreduce or os.environ.get("CMF_BENCH_PG_ONLY") == "1"):    # (PG_ONLY: diagnostic -- the process group exists, no collective runs)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.backends.cudnn.allow_tf32 = False
    # a rank stuck in a collective (a peer died, a link is down) must not hang the launcher: a daemon thread ends the process
    # with a non-zero code after the limit -- a fresh exit, nothing is re-executed
    wd_limit = a.watchdog if a.watchdog > 0 else (600.0 if world > 1 else 0.0)
    if wd_limit > 0:
        import threading

        def _watchdog():
            time.sleep(wd_limit)
            print("bench.py watchdog: rank %d still running after %.0f s, exiting 3" % (rank, wd_limit), file=sys.stderr, flush=True)
            os._exit(3)
        threading.Thread(target=_watchdog, daemon=True).start()
    # who takes part: every rank's (host, device) identity gathered over the process group the step uses -- evidence a driver can
    # check that RCCL saw `world` ranks on `world` distinct devices
    ranks_seen, devices, backend = 1, None, None
    props = torch.cuda.get_device_properties(dev)
    me = "%s/%s/%s" % (os.uname().nodename, getattr(props, "uuid", "?"), getattr(props, "pci_bus_id", "?"))
    if dist.is_initialized():
        backend = dist.get_backend()
        got = [None] * dist.get_world_size()
        dist.all_gather_object(got, me)
        ranks_seen, devices = len(got), got
        if world > 1 and not one_gpu:
            assert len(set(got)) == world, "ranks share a device: %s" % got
    else:
        devices = [me]

    from cmflow_amd import _lib, synth
    from cmflow_amd.cmflow import CMFlow, CMFlow_T
    from cmflow_amd.dp import broadcast_module
    from cmflow_amd.train import TrainStep
    _lib.lib()                                       # fail loudly if the HIP extension is missing

    from cmflow_amd.raflow import RaFlow
    net = {"cmflow": CMFlow, "cmflow_t": CMFlow_T, "raflow": RaFlow}[a.model](Args())
    net.load_state_dict(load_weights(a.model))
    net = net.to(dev)
    net.path = a.path
    broadcast_module(net)
    batch = {k: v.to(dev) for k, v in synth.make_batch(a.batch, seed=1234 + rank, train_extras=True).items()}
    # CMFlow-T trains on mini-clips (clip_util.py:34-62): `--clip` consecutive frames, the GRU state handed from frame to
    # frame as gfeat.detach() (:54), one optimizer step per frame, gfeat = None at the first frame of a clip (:51-52).
    # A bench "step" is one frame; the clip's frames are distinct resident batches.
    # The clip comes through the reference's own input path (dataset/vod_clip.py vodClipDataset -> DataLoader -> extract_data_info_clip,
    # clip_util.py:81-96): `batch` mini-clips of `--clip` ragged synthetic sample files, resampled to 256 points by the loader.
    if a.model == "cmflow_t" and a.mode == "train":
        clip = clip_from_loader(a.batch, a.clip, rank, dev)
    else:
        clip = [batch]
    frame = [0]

    if a.mode == "train":
        net.train()
        # RaFlow's purely self-supervised loss diverges within a few Adam steps on seeded random weights (NaN -> the
        # CPU SVD of the baseline leg throws): same work per step with a small learning rate
        step = TrainStep(net, vr_thres=Args.vr_thres, lr=1e-6 if a.model == "raflow" else 0.001)
        step.force_allreduce = bool(a.force_allreduce)

        def one():
            f = frame[0] % len(clip)
            if a.model == "cmflow_t" and f == 0:
                step.reset_clip()
            step(clip[f])
            frame[0] += 1
    else:
        net.eval()

        def one():
            with torch.no_grad():
                if a.model == "cmflow_t":
                    net(batch["pc1"], batch["pc2"], batch["ft1"], batch["ft2"], None, "test", None)
                elif a.model == "raflow":
                    net(batch["pc1"], batch["pc2"], batch["ft1"], batch["ft2"], batch["interval"])
                else:
                    net(batch["pc1"], batch["pc2"], batch["ft1"], batch["ft2"], None, "test")

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if a.serial:
        from cmflow_amd import fused_blocks as _FB
        _FB.set_serial(net, True)
    for _ in range(a.warmup):
        one()
    frame[0] = 0                                     # the timed region starts at the first frame of a clip
    # HIP-event pairs around every 5th cmf_gemm launch >= 1 GFLOP, inside the library (all of them are counted; around every one the
    # pairs themselves cost 0.13-0.18 ms per step: measured as the bracketed region against the unbracketed regions behind it)
    _lib.SAMPLE_EVERY = 5                          # (coprime with the 44 such launches of a step: every launch position is sampled in turn)
    _lib.profile_begin()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        one()
    fence()
    dt = time.perf_counter() - t0
    prof = _lib.profile_end()
    # run-to-run spread: two more timed regions of the same length, back to back (not part of `value`)
    spread = [dt / a.steps * 1e3]
    for _ in range(2):
        fence()
        t1 = time.perf_counter()
        for _ in range(a.steps):
            one()
        fence()
        spread.append((time.perf_counter() - t1) / a.steps * 1e3)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    def roofline_of(prof):
        if not (prof and prof["launches"]):
            return None
        avg_ms = prof["ms"] / prof["launches"]
        per_launch = prof["units"] / prof["launches"]
        if prof["bound"] == "hbm":
            achieved, peak, unit = per_launch / (avg_ms * 1e-3) / 1e9, HBM_PEAK_GBS, "GB/s"
        else:
            peak = MFMA_F32_PEAK_TFLOPS
            achieved, unit = per_launch / (avg_ms * 1e-3) / 1e12, "TFLOP/s"
        # PMC counters need rocprofv3 around the process: `traffic` is the per-launch HBM byte count of this kernel's
        # >= 256-workgroup launches from the committed FETCH_SIZE / WRITE_SIZE passes over this same command (separate --pmc
        # runs, gfx950 x2 correction on FETCH_SIZE; tools/session.sh pmc_gemm), beside the algorithmic bytes of the launches
        # bracketed live (A + B + C once, + the Z operand of the backward epilogues)
        traffic = alg_bytes = traffic_source = None
        tp = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", TRAFFIC_PROFILE)
        if prof["bound"] == "mfma" and a.mode == "train" and a.model == "cmflow" and os.path.exists(tp):
            rec = json.load(open(tp))
            if rec.get("source_id") == _lib.source_id():
                traffic = rec["traffic_bytes_per_launch"]
                traffic_source = ("NOT measured by this run: constant from the committed rocprofv3 PMC passes in profiles/%s over this same "
                                  "command, taken on the same kernel sources (source_id %s)" % (TRAFFIC_PROFILE, rec.get("source_id")))
            else:
                traffic_source = ("null: profiles/%s was taken on other kernel sources (source_id %s, this library %s)"
                                  % (TRAFFIC_PROFILE, rec.get("source_id"), _lib.source_id()))
        if prof.get("shapes"):
            tot = sum(cnt * 4.0 * (M * K + K * N + M * N * (2 if kind >= 2 else 1))
                      for (M, N, K, layout, kind, split_k, bm, bn), (cnt, _) in prof["shapes"].items())
            alg_bytes = round(tot / prof["launches"])
        out = {"kernel": prof["kernel"], "bound": prof["bound"], "achieved": round(achieved, 2), "peak": peak,
               "unit": unit, "frac": round(achieved / peak, 4), "traffic": traffic,
               "traffic_source": traffic_source, "traffic_profile": "profiles/" + TRAFFIC_PROFILE, "algorithmic_bytes_per_launch": alg_bytes,
               "algorithmic_bytes_note": "A + B + C once (+ the Z operand of the backward epilogues) at their materialised sizes; the "
                                         "gathering GEMMs of the second encoder read per-point rows (L2-resident) instead of an M x K operand, "
                                         "so the counted traffic can be below this figure",
               "launches": prof["launches"], "avg_us": round(avg_ms * 1e3, 2),
               "algorithmic_per_launch": per_launch,
               "launch_filter": "every %s tiled-kernel launch >= %.0e flop, bracketed inside libcmflow_hip.so (block-internal "
                                "GEMMs included); thin (<= 64-channel) kernels are counted in flop_share only" %
                                ("" if prof.get("sample_every", 1) == 1 else "%d-th" % prof["sample_every"], _lib.TRACK_MIN_UNITS.get(prof["kernel"], 0.0))}
        if prof.get("units_all"):
            out["flop_share"] = round(prof.get("units_eligible", prof["units"]) / prof["units_all"], 4)
            out["launches_all"] = prof["launches_all"]
        return out

    # In the timed region the four scales of an encoder run on four HIP streams, so a bracketed launch shares
    # the chip with kernels of the other streams and its duration is a contended one.  A short extra pass with
    # the scales serialised (outside the timed region, not part of `value`) gives the kernel's own rate.
    # Every rank runs it: a training step contains the gradient all-reduce, a collective all ranks must enter.
    iso = iso_prof = None
    if a.path == "pm":
        from cmflow_amd import fused_blocks as _FB
        _FB.set_serial(net, True)                   # the SAME launches, every chain on the caller's stream
        one(); torch.cuda.synchronize()
        _lib.SAMPLE_EVERY = 1                       # (untimed pass: every launch bracketed)
        _lib.profile_begin()
        for _ in range(3):
            one()
        iso_prof = _lib.profile_end()
        iso = roofline_of(iso_prof)
        if iso:
            iso["note"] = "3 extra steps with every chain on one stream (outside the timed region): the kernel's own rate"

    # peak device memory of the timed regions: torch's allocator (every tensor and arena of the step) + what the library holds itself
    peak_mem = {"torch_max_allocated": int(torch.cuda.max_memory_allocated(dev)), "torch_max_reserved": int(torch.cuda.max_memory_reserved(dev)),
                "library_scratch": int(_lib.lib().cmf_mem_stats())}
    peak_mem["total"] = peak_mem["torch_max_allocated"] + peak_mem["library_scratch"]

    # BASELINE config 2 (fwd only, eval-mode BN, same batch) timed in the same run: 20 steps behind 3 warm-up steps, the chains back
    # on the stream pool.  Not part of `value`.  (Every rank runs it: no collective inside.)
    config2 = None
    if a.mode == "train" and a.model == "cmflow" and not a.no_config2:
        from cmflow_amd import fused_blocks as _FB
        _FB.set_serial(net, bool(a.serial))
        net.eval()

        def fwd():
            with torch.no_grad():
                net(batch["pc1"], batch["pc2"], batch["ft1"], batch["ft2"], None, "test")
        for _ in range(3):
            fwd()
        fence()
        t2 = time.perf_counter()
        for _ in range(20):
            fwd()
        fence()
        d2 = (time.perf_counter() - t2) / 20
        config2 = {"ms_per_step": round(d2 * 1e3, 3), "frame_pairs_per_s": round(a.batch * world / d2, 1), "steps": 20,
                   "workload": "cmflow fwd-only inference (eval-mode BN), N=256, B=%d per GPU: BASELINE config 2" % a.batch}
        net.train()

    # the gradient all-reduce alone (world > 1, or forced at world 1): 20 back-to-back all-reduces of the flat bucket between one
    # event pair on the current stream -- every rank enters them
    allreduce_ms = None
    if a.mode == "train" and dist.is_initialized() and not one_gpu:
        bucket = step.bucket.flat if hasattr(step, "bucket") and hasattr(step.bucket, "flat") else None
        if bucket is not None:
            scratch = torch.zeros_like(bucket)
            for _ in range(3):
                dist.all_reduce(scratch)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                dist.all_reduce(scratch)
            e1.record(); torch.cuda.synchronize()
            allreduce_ms = round(e0.elapsed_time(e1) / 20, 4)
            del scratch

    if rank == 0:
        pairs = a.batch * world * a.steps
        contended = roofline_of(prof)
        roof = None
        if contended:
            # the step figure: flops of the bracketed launches / wall time of the timed region (rank 0's launches over the
            # max-over-ranks wall time)
            tf = prof["units_eligible"] / dt / 1e12
            roof = dict(contended)
            roof.update({"achieved": round(tf, 2), "frac": round(tf / MFMA_F32_PEAK_TFLOPS, 4),
                         "launches": prof["launches_eligible"], "avg_us": round(dt / prof["launches_eligible"] * 1e6, 2),
                         "definition": "sum of 2*M*N*K over the launches >= min_units_per_launch of the timed region / the region's wall time: "
                                       "what the step extracts from the MFMA pipe (avg_us = wall time per such launch); every %d-th of them is "
                                       "bracketed with a HIP event pair on its launch stream (extra.contended_per_launch: an event pair around "
                                       "every one costs 0.13-0.18 ms per step); the kernel's own rate is roofline_isolated" % prof["sample_every"]})
            roof.pop("note", None)
        cpu = cpu1 = hbm = None
        if world == 1 and not a.no_op_rooflines:
            hbm = hbm_op_rooflines(dev)
        if not a.no_cpu_baseline and world == 1:           # the CPU leg is timed at N=1 only
            cpu = cpu_baseline(a.mode, a.model)
            # BASELINE config 1 (the reference's own CPU-runnable case): one frame pair, forward, eval mode (main_util.py:142)
            cpu1 = cpu_baseline("fwd", a.model, budget_s=8.0, B=1)
        line = {
            "metric": "frame-pairs/sec %s %s" % ({"cmflow": "CMFlow", "cmflow_t": "CMFlow-T", "raflow": "RaFlow"}[a.model],
                                                   "fwd+bwd" if a.mode == "train" else "fwd"),
            "value": round(pairs / dt, 2), "unit": "frame-pairs/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": ("%s %s step, N=256, B=%d per GPU (global %d), ball-query r=2/4/8/16 K=4/8/16/32, "
                                    "kNN K=8%s; synthetic clouds, seeded random-init weights" %
                                    (a.model, ("fwd+bwd training (3 self-supervised losses + Adam, train-mode BN)" if a.model == "raflow"
                                               else "fwd+bwd training (7 cross-modal losses + Adam, train-mode BN)")
                                     if a.mode == "train" else "fwd-only inference (eval-mode BN)", a.batch,
                                     a.batch * world,
                                     ("; %d-frame mini-clips, GRU state handed over detached, optimizer step per frame" % len(clip)
                                      if len(clip) > 1 else "") +
                                     ("; dp%d RCCL grad all-reduce" % world if world > 1 else "") +
                                     ("; RCCL all-reduce of the gradient bucket forced at world size 1" if a.force_allreduce and world == 1 else ""))),
                       "parallelism": "dp%d" % world},
            "roofline": roof, "roofline_isolated": iso, "roofline_hbm": hbm, "cpu_baseline": cpu,
            "cpu_baseline_config1": cpu1,
            # model-level rate against the reference's ALGORITHMIC work (25.52 GFLOP per frame pair forward,
            # x3 for fwd+bwd) -- the build's hoisted first convs execute fewer FLOPs than that
            "algorithmic_model_tflops": round(pairs / dt * 25.52e9 * (3.0 if a.mode == "train" else 1.0) / 1e12, 2),
            # ms per step of three back-to-back timed regions of `steps` steps each (the first one is `value`): rank 0's clock
            "extra": {"ms_per_step_regions": [round(v, 3) for v in spread],
                      # per-launch durations bracketed inside the timed region: launches on four streams overlap, so their sum exceeds
                      # the wall time -- a concurrency artefact, not a roofline figure
                      "contended_per_launch": ({k: contended[k] for k in ("achieved", "frac", "avg_us", "launches")} if contended else None),
                      "host_cores": host_cores,
                      "peak_mem_bytes": peak_mem, "config2_fwd": config2,
                      "config2_fwd_ms_per_step": config2["ms_per_step"] if config2 else None},
            "ranks_seen": ranks_seen, "devices": devices, "backend": backend, "allreduce_ms": allreduce_ms,
            "source_id": _lib.source_id(),
        }
        if a.gemm_table:
            with open(a.gemm_table, "w") as f:
                f.write(gemm_shape_table(prof, "timed region (chains on the side-stream pool: durations are contended)"))
                if iso_prof:
                    f.write("\n" + gemm_shape_table(iso_prof, "3 extra steps with every chain on one stream (the kernel's own rate)"))
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(line) + "\n").encode())
    if world > 1:
        fence()                                             # leave together
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
