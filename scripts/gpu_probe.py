#!/usr/bin/env python
"""GPU-box probe: kernel micro-benchmarks (HIP-event timed, back-to-back launches) dumped as JSON lines.
Usage (through gpurun): python scripts/gpu_probe.py > gpurun_out/probe.jsonl"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytorch_geometric_temporal_amd import ops  # noqa: E402
from pytorch_geometric_temporal_amd.dataset import synthetic as syn  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, warm=10, reps=50):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps   # us


def emit(**kw):
    print(json.dumps(kw), flush=True)


def probe_copy():
    n = 64 * 1024 * 1024
    a, b = torch.empty(n, device=dev), torch.empty(n, device=dev)
    us = timeit(lambda: b.copy_(a))
    emit(probe="torch_copy_256MB", us=us, GBs=2 * 4 * n / us / 1e3)


def _rot_time(g_csr, n, F, pairs, with_t=False):
    """HBM-honest timing: rotate through `pairs` distinct (X, Y) buffer pairs so the 256 MiB Infinity Cache cannot
    keep X resident between launches."""
    Xs = [torch.randn(n, F, device=dev) for _ in range(pairs)]
    Ys = [torch.empty(n, F, device=dev) for _ in range(pairs)]
    i = [0]

    def fn():
        k = i[0] % pairs
        ops.spmm(g_csr, Xs[k], Ys[k])
        i[0] += 1
    return timeit(fn, warm=pairs * 2, reps=pairs * 10)


def identity_graph(n):
    ei = np.stack([np.arange(n), np.arange(n)]).astype(np.int64)
    return ops.DConvGraph(torch.from_numpy(ei).to(dev), None, n)


def probe_spmm_ns():
    """North-star aggregation (N = 200 000, F = 64): the ELLW LDS-window kernel vs the CSR row tiles, in-degree 1 / 8 / 16,
    locality-ordered and uniform-random graphs."""
    from pytorch_geometric_temporal_amd import _lib
    lib = _lib.get_lib()
    n = 200_000
    gid = identity_graph(n)
    nb = ops.spmm_algorithmic_bytes(n, n, 64, False)
    for ellw in (False, True):
        if ellw:
            ops._force_ellw(gid.fwd_o, 32)
        else:
            gid.fwd_o.ellw, gid.fwd_o.halo = None, 0
        us = _rot_time(gid.fwd_o, n, 64, 6)
        emit(probe="spmm_ns_identity_rotating", ellw=ellw, us=us, GBs=nb / us / 1e3,
             note="pure streaming through the kernel (in-degree 1)")
    del gid
    for name, gen in (("local", syn.local_graph), ("uniform", syn.uniform_graph)):
        for deg in (8, 16):
            ei, ew = gen(n, deg, seed=0)
            g = ops.DConvGraph(torch.from_numpy(ei).to(dev), torch.from_numpy(ew).to(dev), n)
            nb = ops.spmm_algorithmic_bytes(n, g.E, 64, False)
            X = torch.randn(n, 64, device=dev)
            Y = torch.empty_like(X)
            variants = [dict(ellw=False, rows=64), dict(ellw=False, rows=32)]
            if name == "local":
                variants += [dict(ellw=True, mode="scale"), dict(ellw=True, mode="vals")]
            for v in variants:
                csr = g.fwd_o
                lib.tune("spmm_tile_rows", v.get("rows", 32))
                lib.tune("spmm_ellw", 1 if v["ellw"] else 0)
                if v.get("mode") == "vals" and csr.ellw is not None and csr.ellw.vals is None:
                    csr.val[0] = csr.val[0] * (1 + 2 ** -20)     # not a function of the source any more: per-slot mode
                    csr.ellw = None
                    ops._force_ellw(csr, 32)
                us_res = timeit(lambda: ops.spmm(csr, X, Y))
                us_rot = _rot_time(csr, n, 64, 6)
                e = csr.ellw
                emit(probe="spmm_ns", graph=name, deg=deg, F=64, measured_halo=csr.halo, E=int(g.E), alg_MB=nb / 1e6,
                     us_resident=us_res, frac_resident=nb / us_res / 1e3 / 8000, us_rotating=us_rot,
                     GBs_rotating=nb / us_rot / 1e3, frac_rotating=nb / us_rot / 1e3 / 8000,
                     layout=None if e is None or not v["ellw"] else dict(tile_rows=e.tile_rows, width=e.width,
                                                                         tiles=e.n_tiles, far=e.far), **v)
            lib.tune("spmm_tile_rows", 32)
            lib.tune("spmm_ellw", 1)
            del g, X, Y


def probe_spmm_batched():
    from pytorch_geometric_temporal_amd import _lib
    lib = _lib.get_lib()
    ei, ew = syn.sensor_graph(207, 1515, seed=0)
    g = ops.DConvGraph(torch.from_numpy(ei).to(dev), torch.from_numpy(ew).to(dev), 207)
    gid = identity_graph(207)
    for B, C in ((64, 66), (256, 66), (1024, 66), (4096, 66), (1024, 64), (16384, 4)):
        F = B * C
        pairs = max(2, int(700e6 // (2 * 207 * F * 4)) + 1)
        pairs = min(pairs, 64)
        nb = ops.spmm_algorithmic_bytes(207, g.E, F, False)
        us_id = _rot_time(gid.fwd_o, 207, F, pairs)
        emit(probe="spmm_nodemajor_identity_rotating", B=B, C=C, us=us_id,
             GBs=ops.spmm_algorithmic_bytes(207, 207, F, False) / us_id / 1e3)
        for unroll, xcd in ((8, 1), (4, 1), (8, 0)):
            lib.tune("spmm_unroll", unroll); lib.tune("spmm_wide_xcd", xcd)
            us = _rot_time(g.fwd_o, 207, F, pairs)
            emit(probe="spmm_metrla_nodemajor_rotating", B=B, C=C, unroll=unroll, xcd=xcd, pairs=pairs, us=us,
                 alg_MB=nb / 1e6, GBs=nb / us / 1e3, frac=nb / us / 1e3 / 8000)
        lib.tune("spmm_unroll", 8); lib.tune("spmm_wide_xcd", 1)


def probe_stack():
    """One DConv diffusion stack (K = 3) at METR-LA shape: LDS-resident one-launch form vs one launch per hop."""
    ei, ew = syn.sensor_graph(207, 1515, seed=0)
    g = ops.DConvGraph(torch.from_numpy(ei).to(dev), torch.from_numpy(ew).to(dev), 207)
    for B, C in ((1024, 66), (256, 66), (1024, 4), (4096, 4)):
        M = 207 * B
        nbuf = max(2, min(8, int(1.2e9 // (5 * M * C * 4))))
        TSs = [torch.randn(5, 1, M, C, device=dev) for _ in range(nbuf)]
        i = [0]

        def fwd_slab():
            ops._slab_fwd(g, TSs[i[0] % nbuf][0, 0], M * C, B, C, 3); i[0] += 1

        def bwd_slab():
            ops._slab_bwd(g, TSs[i[0] % nbuf][0, 0], M * C, B, C, 3, True); i[0] += 1

        def fwd_hops():
            ops._stack_fwd(g, TSs[i[0] % nbuf], 0, 3, 207); i[0] += 1

        def bwd_hops():
            ops._stack_bwd(g, TSs[i[0] % nbuf].view(5, M, C), 3, 207, True); i[0] += 1
        blk = M * C * 4
        for name, fn, nb in (("slab_fwd", fwd_slab, 5 * blk), ("slab_bwd", bwd_slab, 6 * blk),
                             ("hops_fwd", fwd_hops, 10 * blk), ("hops_bwd", bwd_hops, 16 * blk)):
            us = timeit(fn, warm=nbuf, reps=4 * nbuf)
            emit(probe="dconv_stack", form=name, B=B, C=C, us=us, moved_MB=nb / 1e6, GBs=nb / us / 1e3,
                 frac=nb / us / 1e3 / 8000)


def probe_graph():
    """The whole training step (forward, hand-written backward, Adam) captured into ONE hipGraph (torch.cuda.CUDAGraph
    records the C-ABI launches: they go to the stream handed in, allocate nothing and never synchronise) and replayed,
    against the eager Python loop.  Small batches are launch-bound: ~650 launches per step."""
    from bench import Model, masked_mae_loss, FlatGrads, STD, MEAN
    ei, ew = syn.sensor_graph(207, 1515, seed=0)
    ei, ew = torch.from_numpy(ei).to(dev), torch.from_numpy(ew).to(dev)
    for hidden, B in ((64, 64), (64, 256), (64, 1024), (2, 64)):
        torch.manual_seed(0)
        model = Model(hidden).to(dev)
        flat = FlatGrads(model.parameters())
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, capturable=True)
        X = torch.randn(B, 12, 207, 2, device=dev)
        y = torch.randn(B, 12, 207, 2, device=dev)
        loss_buf = torch.zeros((), device=dev)

        def step():
            out = model(X, ei, ew)
            loss = masked_mae_loss(out * STD + MEAN, y * STD + MEAN)
            flat.zero()
            loss.backward()
            opt.step()
            loss_buf.copy_(loss.detach())
        s_ = torch.cuda.Stream()
        s_.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s_):
            for _ in range(3):
                step()
        torch.cuda.current_stream().wait_stream(s_)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        eager = 1e3 * (time.perf_counter() - t0) / 10
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                step()
            torch.cuda.synchronize()
            l0 = float(loss_buf)
            for _ in range(3):
                g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                g.replay()
            torch.cuda.synchronize()
            graph_ms = 1e3 * (time.perf_counter() - t0) / 20
            emit(probe="hipgraph_step", hidden=hidden, B=B, eager_ms=eager, graph_ms=graph_ms, loss_first=l0,
                 loss_after=float(loss_buf), edges_per_s_graph=B * 12 * 1515 / graph_ms * 1e3)
        except Exception as e:
            emit(probe="hipgraph_step", hidden=hidden, B=B, eager_ms=eager, error=repr(e)[:300])
        del model, flat, opt
        torch.cuda.empty_cache()


def probe_models():
    """Forward + backward of the other model families at the BASELINE.json config shapes (synthetic data), next to the
    CPU oracle on a bounded sample of the same call (32 threads)."""
    import time as _t
    from oracle import functional as OF
    from pytorch_geometric_temporal_amd.nn.recurrent import A3TGCN2, TGCN2, EvolveGCNH, GConvGRU
    from pytorch_geometric_temporal_amd.nn.attention import STConv, ChebConvAttention
    torch.set_num_threads(min(os.cpu_count() or 1, 32))

    def gpu_ms(fn, reps=10):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = _t.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return 1e3 * (_t.perf_counter() - t0) / reps

    def cpu_ms(fn, reps=2):
        fn()
        t0 = _t.perf_counter()
        for _ in range(reps):
            fn()
        return 1e3 * (_t.perf_counter() - t0) / reps

    # config 3: PeMS-BAY-shaped A3TGCN2(2, 32, periods=12), B = 64
    ei, ew = syn.sensor_graph(325, 2694, seed=0)
    eid, ewd = torch.from_numpy(ei).to(dev), torch.from_numpy(ew).to(dev)
    for B in (64, 512):
        m = A3TGCN2(2, 32, 12, B).to(dev)
        X = torch.randn(B, 325, 2, 12, device=dev)

        def step():
            m.zero_grad(set_to_none=True)
            m(X, eid, ewd).square().mean().backward()
        ms = gpu_ms(step)
        p = {k: v.detach().cpu() for k, v in m.state_dict().items()}
        Xc = X[:8].cpu()
        cms = cpu_ms(lambda: OF.a3tgcn(Xc, torch.from_numpy(ei), torch.from_numpy(ew), None, p)) * (B / 8)
        emit(probe="model", model="A3TGCN2(2,32,periods=12) PeMS-BAY-shaped 325 nodes / 2694 edges", B=B, gpu_fwd_bwd_ms=ms,
             snapshot_edges_per_s=B * 12 * 2694 / ms * 1e3, cpu_oracle_fwd_only_ms_scaled=cms)
    # config 4: 50 k-node static graph, TGCN2(2, 32)
    ei, ew = syn.local_graph(50_000, 8, seed=0)
    eid, ewd = torch.from_numpy(ei).to(dev), torch.from_numpy(ew).to(dev)
    for B in (8, 64):
        m = TGCN2(2, 32, B).to(dev)
        X, H = torch.randn(B, 50_000, 2, device=dev), torch.randn(B, 50_000, 32, device=dev)

        def step():
            m.zero_grad(set_to_none=True)
            m(X, eid, ewd, H).square().mean().backward()
        ms = gpu_ms(step)
        p = {k: v.detach().cpu() for k, v in m.state_dict().items()}
        Xc, Hc = X[:2].cpu(), H[:2].cpu()
        cms = cpu_ms(lambda: OF.tgcn_cell(Xc, torch.from_numpy(ei), torch.from_numpy(ew), Hc, p), reps=1) * (B / 2)
        emit(probe="model", model="TGCN2(2,32) 50 000 nodes / 400 000 edges", B=B, gpu_fwd_bwd_ms=ms,
             snapshot_edges_per_s=B * 400_000 / ms * 1e3, cpu_oracle_fwd_only_ms_scaled=cms)
    # config 5: dynamic graphs, EvolveGCN-H(129, 8), 61 snapshots with fresh edge lists (graph prep every step)
    graphs = [syn.sensor_graph(129, 129 + int(700 + 20 * s), seed=s) for s in range(61)]
    gd = [(torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)) for a, b in graphs]
    m = EvolveGCNH(129, 8).to(dev)
    Xs = [torch.randn(129, 8, device=dev) for _ in range(61)]

    def epoch():
        m.zero_grad(set_to_none=True)
        m.reinitialize_weight()
        ops.GRAPH_CACHE.clear()
        loss = 0
        for s in range(61):
            loss = loss + m(Xs[s], gd[s][0], gd[s][1]).square().mean()
        loss.backward()
    ms = gpu_ms(epoch, reps=5)
    emit(probe="model", model="EvolveGCNH(129,8) 61 dynamic snapshots (graph prep per snapshot)", gpu_epoch_fwd_bwd_ms=ms,
         snapshot_edges_per_s=sum(g[0].shape[1] for g in graphs) / ms * 1e3)
    # STConv / ChebConvAttention / GConvGRU single calls
    ei, ew = syn.sensor_graph(207, 1515, seed=0)
    eid, ewd = torch.from_numpy(ei).to(dev), torch.from_numpy(ew).to(dev)
    m = STConv(207, 2, 32, 32, kernel_size=3, K=3).to(dev)
    X = torch.randn(64, 12, 207, 2, device=dev)
    emit(probe="model", model="STConv(207,2,32,32,k=3,K=3) B=64 T=12", gpu_fwd_bwd_ms=gpu_ms(
        lambda: (m.zero_grad(set_to_none=True), m(X, eid, ewd).square().mean().backward())))
    m = ChebConvAttention(32, 32, 3, normalization="sym").to(dev)
    X, S = torch.randn(64, 207, 32, device=dev), torch.softmax(torch.randn(64, 207, 207, device=dev), 1)
    emit(probe="model", model="ChebConvAttention(32,32,K=3) B=64 N=207", gpu_fwd_bwd_ms=gpu_ms(
        lambda: (m.zero_grad(set_to_none=True), m(X, eid, S, ewd).square().mean().backward())))
    m = GConvGRU(2, 32, 3).to(dev)
    X, H = torch.randn(207, 2, device=dev), torch.randn(207, 32, device=dev)
    emit(probe="model", model="GConvGRU(2,32,K=3) N=207 one cell step", gpu_fwd_bwd_ms=gpu_ms(
        lambda: (m.zero_grad(set_to_none=True), m(X, eid, ewd, H).square().mean().backward())))


def probe_gemm():
    for M in (13248, 211968):
        for (S, C, N) in ((5, 66, 128), (5, 66, 64), (1, 128, 330)):
            A = torch.randn(S, M, C, device=dev)
            W = torch.randn(S * C, N, device=dev)
            b = torch.randn(N, device=dev)
            out = torch.empty(M, N, device=dev)
            us = timeit(lambda: ops.gemm(A, C, M * C, S, C, W, N, 1, out, N, 0, N, b, M, N), reps=20)
            fl = 2.0 * M * N * S * C
            emit(probe="gemm_nn", M=M, K=S * C, N=N, us=us, TFLOPs=fl / us / 1e6, frac=fl / us / 1e6 / 157.3)
            A2 = torch.cat([A[j] for j in range(S)], 1).contiguous()
            us_t = timeit(lambda: torch.addmm(b, A2, W), reps=20)
            emit(probe="torch_addmm_same_shape", M=M, K=S * C, N=N, us=us_t, TFLOPs=fl / us_t / 1e6)
            G = torch.randn(M, N, device=dev)
            dW = torch.zeros(S * C, N, device=dev)
            db = torch.zeros(N, device=dev)
            us = timeit(lambda: ops.gemm_tn_acc(A, C, M * C, S, C, G, N, dW, N, db, M, N), reps=20)
            emit(probe="gemm_tn", M=M, K=S * C, N=N, us=us, TFLOPs=fl / us / 1e6, frac=fl / us / 1e6 / 157.3)
            Gs = torch.empty(S, M, C, device=dev)
            us = timeit(lambda: ops.gemm(G, N, 0, 1, N, W, 1, N, Gs, C, M * C, C, None, M, S * C), reps=20)
            emit(probe="gemm_nt", M=M, K=N, N=S * C, us=us, TFLOPs=fl / us / 1e6, frac=fl / us / 1e6 / 157.3)
            del A, W, out, A2, G, dW, Gs


def probe_prep():
    for n, e in ((207, 1515), (50_000, 400_000), (200_000, 1_600_000)):
        if n == 207:
            ei, ew = syn.sensor_graph(n, e, seed=0)
        else:
            ei, ew = syn.local_graph(n, e // n, seed=0)
        ei_t, ew_t = torch.from_numpy(ei).to(dev), torch.from_numpy(ew).to(dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g = ops.DConvGraph(ei_t, ew_t, n)
        torch.cuda.synchronize()
        emit(probe="dconv_prep", N=n, E=int(ei.shape[1]), ms=1e3 * (time.perf_counter() - t0))
        del g


def probe_step():
    from bench import Model, masked_mae_loss, FlatGrads, STD, MEAN
    ei, ew = syn.sensor_graph(207, 1515, seed=0)
    ei, ew = torch.from_numpy(ei).to(dev), torch.from_numpy(ew).to(dev)
    for hidden in (64, 2):
        for B in ((64, 256, 1024, 1236, 1900, 2528) if hidden == 64 else (64, 1024, 4096)):
            torch.manual_seed(0)
            model = Model(hidden).to(dev)
            flat = FlatGrads(model.parameters())
            opt = torch.optim.Adam(model.parameters(), lr=1e-3)
            X = torch.randn(B, 12, 207, 2, device=dev)
            y = torch.randn(B, 12, 207, 2, device=dev)

            def step():
                out = model(X, ei, ew)
                loss = masked_mae_loss(out * STD + MEAN, y * STD + MEAN)
                flat.zero()
                loss.backward()
                opt.step()

            def fwd():
                with torch.no_grad():
                    model(X, ei, ew)
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                step()
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / 5
            t0 = time.perf_counter()
            for _ in range(5):
                fwd()
            torch.cuda.synchronize()
            ms_f = 1e3 * (time.perf_counter() - t0) / 5
            emit(probe="dcrnn_train_step_eager", hidden=hidden, B=B, ms=ms, ms_fwd_only=ms_f,
                 snapshot_edges_per_s=B * 12 * 1515 / ms * 1e3, mem_GB=torch.cuda.max_memory_allocated() / 1e9)
            del model, flat, opt, X, y
            torch.cuda.empty_cache()


def probe_shapes():
    """Per-shape GEMM timing inside real DCRNN training steps (B = 1024, hidden 64): which launches dominate."""
    from bench import Model, masked_mae_loss, FlatGrads, STD, MEAN
    from pytorch_geometric_temporal_amd import ops
    ei, ew = syn.sensor_graph(207, 1515, seed=0)
    ei, ew = torch.from_numpy(ei).to(dev), torch.from_numpy(ew).to(dev)
    torch.manual_seed(0)
    model = Model(64).to(dev)
    flat = FlatGrads(model.parameters())
    X = torch.randn(1024, 12, 207, 2, device=dev)
    y = torch.randn(1024, 12, 207, 2, device=dev)

    def step():
        out = model(X, ei, ew)
        loss = masked_mae_loss(out * STD + MEAN, y * STD + MEAN)
        flat.zero()
        loss.backward()
    for _ in range(3):
        step()
    ops.KERNEL_TIMER = ops.KernelTimer()
    step()
    step()
    rows = ops.KERNEL_TIMER.by_tag()
    summ = ops.KERNEL_TIMER.summary()
    ops.KERNEL_TIMER = None
    for r in rows:
        r["TFLOPs"] = r["work_per_launch"] / r["avg_us"] / 1e6 if r["tag"][0].startswith("gemm") else None
        emit(probe="shapes", **r)
    emit(probe="shapes_summary", **{k: v for k, v in summ.items()})


if __name__ == "__main__":
    which = sys.argv[1:] or ["copy", "spmm_ns", "spmm_batched", "gemm", "prep", "step"]
    emit(device=torch.cuda.get_device_name(0), torch=torch.__version__, cpus=os.cpu_count())
    for w in which:
        try:
            globals()["probe_" + w]()
        except Exception as e:  # keep going: one broken probe must not hide the others
            emit(probe=w, error=repr(e))
