#!/usr/bin/env python
"""bench.py — snapshot-edges aggregated per second on the METR-LA-shaped DCRNN training step (BASELINE.json
configs[1]) on N MI355X GPUs of one node, one process per GPU (RCCL), plus the live roofline of the dominant kernel
and the CPU oracle timed on the same host.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over one batch: forward of BatchedDCRNN over B windows x 12 steps, masked-MAE
loss, backward (hand-written BPTT on the transposed operators), one flat gradient all-reduce, Adam.
snapshot-edges/s = sum over processed samples of T * E / wall time (SURVEY.md §8 d); inputs are resident in HBM
("GPU-index-batching", dataset/metr_la.py:180-190) when the timed region starts.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

import bench_line
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from pytorch_geometric_temporal_amd import _lib, dp, ops  # noqa: E402
from pytorch_geometric_temporal_amd.dataset import synthetic as syn  # noqa: E402
from pytorch_geometric_temporal_amd.nn.conv import Linear  # noqa: E402
from pytorch_geometric_temporal_amd.nn.recurrent import BatchedDCRNN  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured float4 copy)
MFMA_F32_PEAK_TFLOPS = 157.3  # v_mfma_f32_32x32x2_f32, exact fp32
MFMA_BF16_PEAK_TFLOPS = 2500.0  # v_mfma_f32_32x32x16_bf16, dense

N_NODES, N_EDGES, SEQ = 207, 1515, 12
PROPAGATES_PER_CELL = 12     # the reference's op count per DCRNN cell step: 6 (K - 1) propagate calls at K = 3
PMC_FILES = [os.path.join(ROOT, "profiles", f) for f in ("r06h_pmc_traffic.json", "r06e_pmc_traffic.json", "r06d_pmc_traffic.json", "r05w_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json",
                                                          "r02_pmc_traffic.json")]
PMC_FILES_TGCN = [os.path.join(ROOT, "profiles", f) for f in ("r06h_tgcn50k_pmc_traffic.json", "r05w_tgcn50k_pmc_traffic.json", "r05_tgcn50k_pmc_traffic.json", "r04_tgcn50k_pmc_traffic.json")]


_T0 = time.time()


def log(msg):
    """Stage marks on stderr (stdout carries the one JSON line): where a slow box spends the run."""
    print(f"[bench {time.time() - _T0:7.1f} s] {msg}", file=sys.stderr, flush=True)


def _safe(fn):
    """Auxiliary figures must never cost the bench line."""
    try:
        return fn()
    except Exception:
        return None


def gemm_operand_bytes(shape):
    """fp32 bytes a product has to move once: A [M, K] + C [M, N] (+ the gate chain's operands and side outputs:
    "NN+zr": H in, H*r out; "NN+h": z, H in, the new state twice out); weight-gradient shapes [M, N, segs, seg_k]:
    A and G once."""
    if isinstance(shape[0], str):
        op, M, N, n_seg, seg_k = shape[0], shape[1], shape[2], shape[3], shape[4]
        b = 4.0 * M * (n_seg * seg_k + N)
        if op == "NN+zr":
            b += 4.0 * M * N                  # H [M, N/2] in, H*r [M, N/2] out
        elif op == "NN+h":
            b += 4.0 * M * N * 4              # z, H in; out0, out1 out
        return b
    M, N, n_seg, seg_k = shape[0], shape[1], shape[2], shape[3]
    return 4.0 * M * (n_seg * seg_k + N)


def pmc_traffic(kind, files=None):
    """HBM bytes per launch of the dominant kernel class from the rocprofv3 --pmc passes of THIS bench command
    (scripts/pmc_bench.sh: FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE, averaged over the kernel's dispatches),
    committed under profiles/ — counters cannot be read from inside the process.  The newest round's file wins.  `kind` may be
    a tuple of kernel entries of one class (forward + backward of the fused cell): their dispatch-weighted mean."""
    for path in (files or PMC_FILES):
        try:
            with open(path) as fh:
                d = json.load(fh)
            if isinstance(kind, (tuple, list)):
                es = [d["kernels"][k] for k in kind]
                n = sum(x["dispatches"] for x in es)
                e = {"dispatches": n, **{f: sum(x[f] * x["dispatches"] for x in es) / n
                                         for f in ("bytes_per_launch", "fetch_bytes_per_launch", "write_bytes_per_launch")}}
            else:
                e = d["kernels"][kind]
            return {"traffic": e["bytes_per_launch"], "traffic_measured_in_this_run": False,
                    "traffic_source": f"BUILDER-BOX PMC, not this run: profiles/{os.path.basename(path)} ({d.get('command', '')}): "
                    f"FETCH_SIZE x2 {e['fetch_bytes_per_launch']:.3e} + WRITE_SIZE {e['write_bytes_per_launch']:.3e} B per launch "
                    f"over {e['dispatches']} dispatches"}
        except Exception:
            continue
    return {"traffic": None}


def split_bf16_shape(tag):
    """True for the products csrc/gemm_bx.hip takes at the benchmark size (pgt_gemm_bx_launch / pgt_gemm_bx_tn_plan):
    they run six bf16 MFMAs per fp32 product on the bf16 matrix pipe."""
    if isinstance(tag[0], str):
        op, M, N, n_seg, seg_k = tag[0], tag[1], tag[2], tag[3], tag[4]
        K = n_seg * seg_k
        if M < 8192 or K > 336:
            return False
        if op in ("NN+zr", "NN+h"):
            return K > 128 and N <= 128
        return (K > 128 and N <= 128) or (K <= 128 and 128 < N <= 320)
    M, N, n_seg, seg_k = tag[0], tag[1], tag[2], tag[3]
    return M >= 16384 and 128 < n_seg * seg_k <= 351 and N <= 128


def kernel_class_rooflines(kernels, shapes):
    """Price every timed kernel class of a step the same way (all of them are HBM-bound — the dense products since they moved
    to the bf16 matrix pipe: six bf16 MFMAs per fp32 product are 2.67x the fp32 MFMA rate, csrc/gemm_bx.hip): ALGORITHMIC
    bytes (operands read once + results written once) / launch time against 8 TB/s.  `kernels` = KernelTimer.summary(),
    `shapes` = KernelTimer.by_tag(); adds achieved_GBs / hbm_frac (and per-shape lines for the products) in place."""
    for kk, v in kernels.items():
        recs = [r for r in shapes if r["tag"][0] == kk]
        if kk in ("gemm", "gemm_tn"):
            tot_b = sum(gemm_operand_bytes(r["tag"][1:]) * r["launches"] for r in recs)
            tot_f = sum(r["work_per_launch"] * r["launches"] for r in recs)
            tot_s = sum(r["total_ms"] for r in recs) * 1e-3
            bx_f = sum(r["work_per_launch"] * r["launches"] for r in recs if split_bf16_shape(r["tag"][1:]))
            bx_s = sum(r["total_ms"] for r in recs if split_bf16_shape(r["tag"][1:])) * 1e-3
            v["algorithmic_bytes_per_launch"] = tot_b / max(v["launches"], 1)
            v["achieved_GBs"] = tot_b / tot_s / 1e9            # time-weighted over the class's shapes
            v["hbm_frac"] = v["achieved_GBs"] / HBM_PEAK_GBS
            v["fp32_product_TFLOPs"] = tot_f / tot_s / 1e12    # 2 M N K of the fp32 product (what the caller asked for)
            # the split-bf16 launches on the pipe they run on: six bf16 MFMAs per fp32 product against 2.5 PFLOP/s
            v["bf16_pipe_frac"] = (6.0 * bx_f / bx_s / 1e12 / MFMA_BF16_PEAK_TFLOPS) if bx_s > 0 else None
            v["by_shape"] = [{"shape": r["tag"][1:], "launches": r["launches"], "avg_us": r["avg_us"],
                              "split_bf16": split_bf16_shape(r["tag"][1:]),
                              "algorithmic_MB": gemm_operand_bytes(r["tag"][1:]) / 1e6,
                              "achieved_GBs": gemm_operand_bytes(r["tag"][1:]) / (r["avg_us"] * 1e-6) / 1e9,
                              "hbm_frac": gemm_operand_bytes(r["tag"][1:]) / (r["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS,
                              "fp32_product_TFLOPs": r["work_per_launch"] / (r["avg_us"] * 1e-6) / 1e12,
                              **({"set_aside_launches": r["set_aside_launches"], "set_aside_ms": r["set_aside_ms"]}
                                 if "set_aside_launches" in r else {})}
                             for r in recs]
        else:
            v["algorithmic_bytes_per_launch"] = v["work_per_launch"]
            v["achieved_GBs"] = v["work_per_launch"] / (v["avg_us"] * 1e-6) / 1e9
            v["hbm_frac"] = v["achieved_GBs"] / HBM_PEAK_GBS
            if kk == "seq64":
                # the one-launch sequence kernels (csrc/seq64.hip): forward = every diffusion term x its weight block (330 -> 192
                # columns per row and step), adjoint = dP x W^T (192 -> 320); six bf16 MFMAs per fp32 product.  Bytes = inputs read
                # once + saved tensors / pre-activation gradients written once (no term crosses HBM between aggregation and product)
                v["by_direction"] = []
                tot_f = tot_s = 0.0
                for r in recs:
                    d, B, T, Nn = r["tag"][1:5]
                    flops = 2.0 * B * T * Nn * (330 * 192 if d == "fwd" else 192 * 320)
                    sec = r["avg_us"] * 1e-6
                    tot_f += flops * r["launches"]
                    tot_s += r["total_ms"] * 1e-3
                    v["by_direction"].append({"direction": d, "launches": r["launches"], "avg_us": r["avg_us"],
                                              "algorithmic_MB": r["work_per_launch"] / 1e6,
                                              "hbm_frac": r["work_per_launch"] / sec / 1e9 / HBM_PEAK_GBS,
                                              # what a plain stream of this launch's read : write mix reaches on this part (lab/hbm_rw_lab,
                                              # fresh windows: write-only 4.4 - 4.7 TB/s, mixed 5.05 - 5.4): the forward only writes
                                              "stream_ceiling_frac": r["work_per_launch"] / sec / 1e9 / (4500.0 if d == "fwd" else 5200.0),
                                              "bf16_pipe_frac": 6.0 * flops / sec / 1e12 / MFMA_BF16_PEAK_TFLOPS})
                v["bf16_pipe_frac"] = 6.0 * tot_f / tot_s / 1e12 / MFMA_BF16_PEAK_TFLOPS if tot_s > 0 else None
    return kernels


def set_aside(roof, kernels):
    """Launches KernelTimer set aside as host stalls (ops.KernelTimer: more than 10x the median of their shape), said on the line."""
    n = sum(v.get("set_aside_launches", 0) for v in kernels.values())
    if n:
        roof["set_aside"] = {"launches": n, "ms": sum(v.get("set_aside_ms", 0.0) for v in kernels.values()),
                             "rule": "event-pair time over 10x the median of the same shape = a host stall inside the pair; not in any average"}


def all_kernel_classes(kernels, profile_steps):
    """The step as a whole: algorithmic bytes of EVERY timed launch (aggregations, stacks, products, weight gradients, gate
    and gate-backward kernels, movers) / their summed time."""
    all_b = sum(v["algorithmic_bytes_per_launch"] * v["launches"] for v in kernels.values())
    all_s = sum(v["total_ms"] for v in kernels.values()) * 1e-3
    return {"achieved_GBs": all_b / all_s / 1e9, "hbm_frac": all_b / all_s / 1e9 / HBM_PEAK_GBS,
            "ms_per_step_in_timed_kernels": 1e3 * all_s / profile_steps,
            "classes": {k: round(v["total_ms"] / profile_steps, 4) for k, v in kernels.items()},
            "classes_ms_frac": {k: [round(v["total_ms"] / profile_steps, 3), round(v["hbm_frac"], 3)] for k, v in kernels.items()},
            "note": "instrumented steps (HIP events around every launch) run slower than the timed ones: fractions are lower bounds"}


MEAN, STD = 54.0, 19.5       # METR-LA-like speed statistics used to de-normalise inside the loss


def masked_mae_loss(y_pred, y_true):
    """examples/indexBatching/DCRNN/utils.py:10-18 (torch plumbing around the path, not part of it)."""
    mask = (y_true != 0).float()
    mask = mask / mask.mean()
    loss = torch.abs(y_pred - y_true) * mask
    loss = torch.where(torch.isnan(loss), torch.zeros_like(loss), loss)
    return loss.mean()


class Model(torch.nn.Module):
    """BatchedDCRNN in its default configuration (contiguous [B, T, N, O] result, as the reference returns it) + the per-node
    read-out of the reference's examples; `dropin`: that read-out as torch.nn.Linear (what swapping the import alone gives:
    BatchedDCRNN's result routes a skinny F.linear to the package's streaming kernels, nn/recurrent/dcrnn.py:_StatesTensor)
    instead of this package's Linear (same parameters, the same kernels)."""

    def __init__(self, hidden, dropin=False, relu=False):
        super().__init__()
        self.rnn = BatchedDCRNN(2, hidden, K=3)
        self.head = None if hidden == 2 else (torch.nn.Linear(hidden, 2) if dropin else Linear(hidden, 2))
        self.relu = relu             # relu between the recurrent layer and the read-out, as the reference's own models have it

    def forward(self, X, ei, ew):
        h = self.rnn(X, ei, ew)
        if self.relu:
            h = torch.relu(h)
        return h if self.head is None else self.head(h)


FlatGrads = dp.FlatGradients   # every gradient in one buffer -> exactly one RCCL all-reduce per step


def make_batches(series, batch, n_batches, seed, device):
    rng = np.random.default_rng(seed)
    T_total = series.shape[0]
    ar = torch.arange(SEQ, device=device)
    out = []
    for _ in range(n_batches):
        idx = torch.from_numpy(rng.integers(0, T_total - 2 * SEQ, size=batch)).to(device)
        out.append((idx[:, None] + ar[None, :], idx[:, None] + SEQ + ar[None, :]))
    return out


def cpu_baseline(hidden, target_seconds=10.0):
    """The CPU oracle (op-for-op the reference: graph prep every conv call, 3 convs per step, gather -> mul ->
    index_add_ -> matmul) on a bounded sample of the same workload, all host cores."""
    from oracle import functional as F
    ei_np, ew_np = syn.sensor_graph(N_NODES, N_EDGES, seed=0, symmetric=False)
    ei, ew = torch.from_numpy(ei_np), torch.from_numpy(ew_np)
    torch.manual_seed(0)
    m = Model(hidden)
    params = {k[len("rnn."):]: v for k, v in m.named_parameters() if k.startswith("rnn.")}
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    Bc = 64
    series = torch.from_numpy(syn.traffic_series(2000, N_NODES, seed=1))

    def one_step(i):
        nonlocal Bc
        g = torch.Generator().manual_seed(i)
        idx = torch.randint(0, 2000 - 2 * SEQ, (Bc,), generator=g)
        ar = torch.arange(SEQ)
        X, y = series[idx[:, None] + ar], series[idx[:, None] + SEQ + ar]
        out = F.batched_dcrnn(X, ei, ew, params)
        if m.head is not None:
            out = torch.nn.functional.linear(out, m.head.weight, m.head.bias)   # the CPU leg stays pure torch
        loss = masked_mae_loss(out * STD + MEAN, y * STD + MEAN)
        opt.zero_grad()
        loss.backward()
        opt.step()

    # thread sweep first (1 / 8 / 32 / 64 / every hardware thread): PyTorch's intra-op pool on these small operands stops
    # scaling - and then collapses - well before the GPU host's hardware threads, so the sweep runs ONE step of a quarter batch
    # per count (bounded even where a step is tens of times slower); the baseline proper then runs at the best count found
    one_step(0)
    sweep = {}
    ncpu = os.cpu_count() or 1
    Bc_full, Bc = Bc, 16
    for n in sorted({1, min(8, ncpu), min(32, ncpu), min(64, ncpu)}):
        torch.set_num_threads(n)
        one_step(1)
        t0 = time.perf_counter()
        one_step(2)
        sweep[str(n)] = Bc * SEQ * N_EDGES / (time.perf_counter() - t0)
        log(f"cpu oracle, {n} thread(s): {sweep[str(n)]:.3e} snapshot-edges/s")
    Bc = Bc_full
    all_threads = None
    if ncpu > 64:
        # every hardware thread (256 on the GPU host): torch's intra-op pool on operands this small may take minutes per step
        # there, so that point runs in a child process under a clock and is recorded as a timeout when it does not return
        import subprocess
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-sweep-worker", str(ncpu), "--hidden", str(hidden)],
                               capture_output=True, text=True, timeout=45)
            all_threads = float(r.stdout.strip().splitlines()[-1])
        except subprocess.TimeoutExpired:
            all_threads = "no result within 45 s (import + two steps of 16 windows)"
        except Exception as e:
            all_threads = repr(e)
        log(f"cpu oracle, {ncpu} threads (child process): {all_threads}")
    cores = int(max(sweep, key=sweep.get))
    torch.set_num_threads(cores)
    t0 = time.perf_counter()
    one_step(1)
    per = time.perf_counter() - t0
    reps = int(max(2, min(200, target_seconds / max(per, 1e-3))))
    t0 = time.perf_counter()
    for i in range(reps):
        one_step(2 + i)
    dt = time.perf_counter() - t0
    return {"value": reps * Bc * SEQ * N_EDGES / dt, "unit": "snapshot-edges/s", "cores": cores, "threads_used": cores,
            "host_threads": ncpu, "kind": "port",
            "thread_sweep_snapshot_edges_per_s": sweep, "thread_sweep_sample": "one step of 16 windows per thread count",
            "all_hardware_threads_snapshot_edges_per_s": all_threads, "host_hardware_threads": ncpu,
            "sample": f"{reps} training steps of the same model on {Bc} windows x {SEQ} steps (oracle/functional.py, "
                      f"fp32, torch.set_num_threads({cores}) = the best of the sweep), {dt:.1f} s"}


def cpu_sweep_point(hidden, threads, Bc=16):
    """One point of the CPU oracle's thread sweep (the same step as cpu_baseline on 16 windows), snapshot-edges/s."""
    from oracle import functional as F
    torch.set_num_threads(threads)
    ei_np, ew_np = syn.sensor_graph(N_NODES, N_EDGES, seed=0, symmetric=False)
    ei, ew = torch.from_numpy(ei_np), torch.from_numpy(ew_np)
    torch.manual_seed(0)
    m = Model(hidden)
    params = {k[len("rnn."):]: v for k, v in m.named_parameters() if k.startswith("rnn.")}
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    series = torch.from_numpy(syn.traffic_series(200, N_NODES, seed=1))
    ar = torch.arange(SEQ)

    def one_step(i):
        idx = torch.randint(0, 200 - 2 * SEQ, (Bc,), generator=torch.Generator().manual_seed(i))
        X, y = series[idx[:, None] + ar], series[idx[:, None] + SEQ + ar]
        out = F.batched_dcrnn(X, ei, ew, params)
        if m.head is not None:
            out = torch.nn.functional.linear(out, m.head.weight, m.head.bias)
        loss = masked_mae_loss(out * STD + MEAN, y * STD + MEAN)
        opt.zero_grad()
        loss.backward()
        opt.step()
    one_step(0)
    t0 = time.perf_counter()
    one_step(1)
    return Bc * SEQ * N_EDGES / (time.perf_counter() - t0)


def cpu_spmm_ns(cores, seconds=2.0):
    """The "optimised CPU" line SURVEY.md §8(d) asks for next to the op-for-op port: the north-star aggregation
    (N = 200 000, F = 64, in-degree 8) as ONE `torch.sparse_csr @ X` on the host cores (no Python per edge, no graph prep
    in the timed region).  Pure torch, bounded to a couple of seconds."""
    import warnings
    n = 200_000
    ei_np, ew_np = syn.local_graph(n, 8, seed=0)
    src, dst, w = torch.from_numpy(ei_np[0]), torch.from_numpy(ei_np[1]), torch.from_numpy(ew_np)
    val = (1.0 / torch.zeros(n).scatter_add_(0, src, w))[src]          # P_o: 1 / deg_out[source]  (dcrnn.py:70-73)
    order = torch.argsort(dst, stable=True)
    rowptr = torch.zeros(n + 1, dtype=torch.int64)
    rowptr[1:] = torch.bincount(dst, minlength=n).cumsum(0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        A = torch.sparse_csr_tensor(rowptr, src[order], val[order], size=(n, n))
    X = torch.randn(n, 64)
    torch.set_num_threads(cores)
    A @ X
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < seconds or reps < 3:
        A @ X
        reps += 1
    us = (time.perf_counter() - t0) / reps * 1e6
    nbytes = ops.spmm_algorithmic_bytes(n, src.numel(), 64, False)
    return {"what": "torch.sparse_csr @ X, N=200k F=64 deg 8 (the north-star aggregation)", "us_per_launch": us,
            "achieved_GBs": nbytes / us / 1e3, "cores": cores, "reps": reps}


def spmm_roofline_ns(device, pairs=6, launches=60, only=None):
    """North-star micro-benchmark: one diffusion-conv aggregation Y = P_o X at N = 200 000, F = 64, in-degree 8 (and 16).
    The launches rotate through `pairs` distinct (X, Y) buffer pairs (6 x 102 MB) so the 256 MiB Infinity Cache
    cannot keep X resident between launches: X really comes from HBM every time."""
    res = {}
    n = 200_000
    for name, gen, deg in (("local", syn.local_graph, 8), ("uniform", syn.uniform_graph, 8), ("local_deg16", syn.local_graph, 16),
                           ("local_hubs_20x2000", syn.hub_graph, 8),
                           ("grid2d_hilbert", lambda n_, d_, seed: syn.grid2d_graph(447, "hilbert", seed), 8),
                           ("grid2d_rowmajor", lambda n_, d_, seed: syn.grid2d_graph(447, "rowmajor", seed), 8),
                           ("grid2d_shuffled", lambda n_, d_, seed: syn.grid2d_graph(447, "shuffled", seed), 8),
                           # an arbitrary numbering AND a skewed in-degree: the shuffled mesh + 20 rows of 2 000 more slots
                           ("grid2d_shuf_hubs_20x2000",
                            lambda n_, d_, seed: syn.hub_graph(n_, d_, seed=seed, base=syn.grid2d_graph(447, "shuffled", seed)), 8)):
        if only is not None and name not in only:
            continue
        n = 447 * 447 if name.startswith("grid2d") else 200_000      # a 447 x 447 mesh: 199 809 nodes
        ei_np, ew_np = gen(n, deg, seed=0)
        g = ops.DConvGraph(torch.from_numpy(ei_np).to(device), torch.from_numpy(ew_np).to(device), n)
        Xs = [torch.randn(n, 64, device=device) for _ in range(pairs)]
        Ys = [torch.empty(n, 64, device=device) for _ in range(pairs)]

        def measure(ellw):
            for i in range(2 * pairs):
                ops.spmm(g.fwd_o, Xs[i % pairs], Ys[i % pairs], ellw=ellw)
            # The launches are captured into ONE hipGraph and replayed: issued from Python a launch costs ~30 us of host
            # time (argument checks + ctypes), which is as long as the kernel itself and would be what gets measured.
            side = torch.cuda.Stream(device=device)
            side.wait_stream(torch.cuda.current_stream(device))
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                with torch.cuda.graph(graph, stream=side):
                    for i in range(launches):
                        ops.spmm(g.fwd_o, Xs[i % pairs], Ys[i % pairs], ellw=ellw)
            torch.cuda.current_stream(device).wait_stream(side)
            for _ in range(25):                                # ~35 ms of launches: clocks out of the idle state and settled (with 5
                graph.replay()                                 # replays the three timed ones still fell by 2 - 4 % from first to last)
            torch.cuda.synchronize()
            times = []
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                graph.replay()
                e1.record()
                torch.cuda.synchronize()
                times.append(1e3 * e0.elapsed_time(e1) / launches)
            del graph
            return sum(times) / len(times), times              # the MEAN of three replays of 60 launches (not the best)

        t_prep = time.perf_counter()
        ops.spmm(g.fwd_o, Xs[0], Ys[0])                         # first use: the layout is chosen and built here
        torch.cuda.synchronize()
        t_prep = time.perf_counter() - t_prep
        us, times = measure(None)
        nbytes = ops.spmm_algorithmic_bytes(n, g.E, 64, False)
        e = g.fwd_o.ellw or None
        res[name] = {"us_per_launch": us, "us_per_launch_replays": times, "algorithmic_MB": nbytes / 1e6,
                     "achieved_GBs": nbytes / us / 1e3, "frac": nbytes / us / 1e3 / HBM_PEAK_GBS, "edges": int(g.E),
                     "in_degree": deg,
                     "kernel": (("spmm_tile_kernel<4,16,32,8> (CSR row tiles, pgt_spmm_csr_f32)" if g.fwd_o.long_rows is None else
                                 f"spmm_tile_kernel + spmm_long_rows_kernel (pgt_spmm_csr_long_f32: {g.fwd_o.long_rows.numel()} rows "
                                 f"longer than {ops.LONG_ROW} slots, one workgroup each; longest {g.fwd_o.max_len})") if e is None else
                                f"spmm_ellw64_kernel<{0 if e.scale is not None else 1}{', EllwCfgC, renumbered' if e.order is not None else ''}> (pgt_spmm_ellw_f32: "
                                f"{e.n_tiles} tiles of {e.tile_rows} rows x {e.width} slots, halo {e.halo}, "
                                f"{'per-source scale table' if e.scale is not None else 'per-slot coefficients'}, "
                                f"{e.far} out-of-window slots)" + ((f" + ellw_hub_combine_kernel: the {e.left_out} rows longer than {ops.ELLW_MAX_SLOTS} slots (longest {g.fwd_o.max_len}) are left out of "
                                 f"the layout, their slots ride with the tiles in {e.hub_split} pieces per row, a second launch adds the partial rows"
                                 if e.hub_col is not None else
                                 f" + spmm_long_rows_kernel (pgt_spmm_csr_rows_f32: the {e.left_out} rows longer than {ops.ELLW_MAX_SLOTS} slots the layout "
                                 f"leaves out, one workgroup each; longest {g.fwd_o.max_len})") if e.left_out else "")),
                     "buffers": f"{pairs} rotating (X,Y) pairs", "launch": f"{launches} launches replayed as one hipGraph, mean of 3 replays"}
        if e is not None and e.order is not None:
            # the library laid the operator out in a numbering of its own (no permutation pass: X / Y rows go through the
            # order inside the kernel); beside it the CSR row tiles on the caller's numbering = what ran before
            us_csr, _ = measure(False)
            res[name].update({"renumbered": True, "layout_prep_s_once_per_graph": t_prep, "csr_row_tiles_us_per_launch": us_csr,
                              "csr_row_tiles_frac": nbytes / us_csr / 1e3 / HBM_PEAK_GBS})
            # the hops of a K = 3 DConv (dcrnn.py:300-321) through the same layouts: T1 = P X, T2 = 2 P T1 - X on P_o and on P_i,
            # each hop reading the previous one in the caller's numbering — there is no permutation pass to include
            def hops(i):
                x, a, b = Xs[i % pairs], Ys[i % pairs], Ys[(i + 1) % pairs]
                for csr in (g.fwd_o, g.fwd_i):
                    ops.spmm(csr, x, a)
                    ops.spmm(csr, a, b, T=x, alpha=2.0, beta=-1.0)
            for i in range(pairs):
                hops(i)
            side = torch.cuda.Stream(device=device)
            side.wait_stream(torch.cuda.current_stream(device))
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                with torch.cuda.graph(graph, stream=side):
                    for i in range(launches // 4):
                        hops(i)
            torch.cuda.current_stream(device).wait_stream(side)
            graph.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            graph.replay()
            e1.record()
            torch.cuda.synchronize()
            per_hop = 1e3 * e0.elapsed_time(e1) / (4 * (launches // 4))
            hop_bytes = (ops.spmm_algorithmic_bytes(n, g.E, 64, False) + ops.spmm_algorithmic_bytes(n, g.E, 64, True)) / 2
            ei_ = g.fwd_i.ellw or None
            res[name]["dconv_K3_hops"] = {"us_per_hop": per_hop, "algorithmic_MB_per_hop": hop_bytes / 1e6,
                                          "frac": hop_bytes / per_hop / 1e3 / HBM_PEAK_GBS,
                                          "P_i_layout": "renumbered" if ei_ is not None and ei_.order is not None else
                                          ("ellw" if ei_ is not None else "csr row tiles"),
                                          "what": "4 hops (2 on P_o, 2 on P_i, two with the Chebyshev epilogue), mean per hop"}
            del graph
        del g, Xs, Ys
    return res


def train_run(device, rank, world, series, n_edges, batch, hidden, steps, warmup, profile_steps=0, dropin=False, graph=False,
              relu=False, torch_adam=False):
    """Build the model on the `n_edges`-edge METR-LA-shaped graph, run one initialisation pass, `warmup` untimed steps,
    then EXACTLY `steps` timed steps bracketed by barrier + synchronize; MAX over ranks.  `dropin`: torch.nn.Linear as the
    read-out (what swapping the import alone gives) instead of this package's Linear.  `graph`: forward + loss + backward
    as one hipGraph and the optimizer update as a second one, the gradient all-reduce between them issued eagerly (a
    collective stays outside the captured work): one host call per phase instead of ~250 launches, for per-GPU batches
    whose step is host-launch bound.  Returns (seconds, final loss, step function, graph tensors)."""
    ei_np, ew_np = syn.sensor_graph(N_NODES, n_edges, seed=0, symmetric=False)
    ei, ew = torch.from_numpy(ei_np).to(device), torch.from_numpy(ew_np).to(device)
    torch.manual_seed(0)
    model = Model(hidden, dropin=dropin, relu=relu).to(device)
    flat = dp.FlatParameters(model.parameters())   # one gradient buffer, one parameter buffer
    # Adam over the flat parameter buffer: the library's elementwise kernel (dp.FlatAdam, the step count on the device: capturable
    # as it is) — or, `torch_adam`, torch.optim.Adam(fused) over the same buffer (96 us per step: one tensor = two workgroups)
    opt_kw = {"capturable": True} if graph else {}
    make_opt = (lambda: flat.optimizer(torch.optim.Adam, lr=1e-3, **opt_kw)) if torch_adam else (lambda: flat.adam(lr=1e-3))
    opt = make_opt()
    n_total = warmup + steps + profile_steps
    batches = make_batches(series, batch, n_total, seed=1000 + rank, device=device)

    def forward_backward(xi, yi):
        X, y = series[xi], series[yi]                       # [B, 12, N, 2] windows gathered from the resident array
        out = model(X, ei, ew)
        loss = masked_mae_loss(out * STD + MEAN, y * STD + MEAN)
        flat.zero()
        loss.backward()
        return loss

    def step(i):
        loss = forward_backward(*batches[i])
        flat.all_reduce_mean(world)                         # ONE RCCL all-reduce over the flat gradient (305 KB at hidden 64)
        opt.step()
        return loss

    # one initialisation pass (not a warmup step): first-use work that does not belong to any step - graph preparation
    # (cached by tensor identity afterwards), code-object load of every kernel, growth of torch's caching allocator to the
    # 12.5 GB of saved activations (0.5-0.7 s the first time, 20 ms afterwards).  Its parameter update is undone.
    snapshot = flat.data.clone()
    first = not getattr(train_run, "_marked", False)      # stage marks for the headline run only (the variants repeat them)
    train_run._marked = True
    if first:
        log("model, optimizer and batches built")
    step(0)
    if first:
        torch.cuda.synchronize()
        log("initialisation pass done")
    flat.data.copy_(snapshot)
    opt = make_opt()
    if graph:
        from pytorch_geometric_temporal_amd.graphed import GraphedStep
        g_fb = GraphedStep(forward_backward, list(batches[0]), warmup=1)
        g_opt = GraphedStep(lambda: opt.step(), [], warmup=1)      # its eager warm-up call moved the parameters ...
        flat.data.copy_(snapshot)                                  # ... back to the start, moments and step count too
        if torch_adam:
            for st in opt.state.values():                          # (in place: the captured update holds their addresses)
                for v in st.values():
                    if isinstance(v, torch.Tensor):
                        v.zero_()
        else:
            opt.reset()

        def step(i):                                               # noqa: F811  (the graphed step replaces the eager one)
            loss = g_fb(*batches[i])
            flat.all_reduce_mean(world)
            g_opt()
            return loss
    del snapshot
    trace = os.environ.get("PGT_BENCH_STEP_TIMES") == "1"     # diagnostic: synchronised wall time of every warmup step
    for i in range(warmup):
        if trace:
            torch.cuda.synchronize()
            ts = time.perf_counter()
        step(i)
        if trace:
            torch.cuda.synchronize()
            print(f"[bench] warmup step {i}: {1e3 * (time.perf_counter() - ts):.2f} ms", file=sys.stderr)
    dt, loss = dp.timed_steps(step, warmup, steps, device)   # barrier + synchronize on both sides, MAX over ranks
    return dt, float(loss.detach()), step, (ei, ew)


AUX_BLOCKS = ("small_batch", "config1_chickenpox", "config1_chickenpox_K2", "config1_chickenpox_K3", "config3_pemsbay_a3tgcn2",
              "config4_50k_tgcn2", "config5_covid_evolvegcnh")


def aux_worker(args):
    """`bench.py --aux-worker` (spawned by the default run): the launch-bound regime and BASELINE configs 1, 3, 4, 5, one JSON
    line per finished block on stdout."""
    import bench_configs as BCfg
    t_start = time.time()
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    assert _lib.get_lib().target == "gfx950"
    cores = min(os.cpu_count() or 1, 32)
    series = torch.from_numpy(syn.traffic_series(34272, N_NODES, seed=1)).to(device)
    ei_np, ew_np = syn.sensor_graph(N_NODES, args.edges, seed=0, symmetric=False)
    ei, ew = torch.from_numpy(ei_np).to(device), torch.from_numpy(ew_np).to(device)
    blocks = {
        "small_batch": lambda: BCfg.small_batch(device, Model, masked_mae_loss, series, ei, ew, args.edges, SEQ, MEAN, STD, cores),
        "config1_chickenpox": lambda: BCfg.chickenpox_epoch(device, cores),
        "config1_chickenpox_K2": lambda: BCfg.chickenpox_epoch(device, cores, K=2),
        "config1_chickenpox_K3": lambda: BCfg.chickenpox_epoch(device, cores, K=3),
        "config3_pemsbay_a3tgcn2": lambda: BCfg.config3_pemsbay(device, cores),
        "config4_50k_tgcn2": lambda: BCfg.config4_50k(device, cores, sys.modules[__name__]),
        "config5_covid_evolvegcnh": lambda: BCfg.covid_epoch(device, cores),
    }
    assert tuple(blocks) == AUX_BLOCKS
    if args.aux_only:
        blocks = {k: v for k, v in blocks.items() if k in args.aux_only.split(",")}
    for name, fn in blocks.items():
        if time.time() - t_start > args.aux_seconds:
            res = {"skipped": f"--aux-seconds spent ({args.aux_seconds:g} s left for the auxiliary process)"}
        else:
            try:
                res = fn()
            except Exception as e:                             # an auxiliary line must never cost the bench line
                res = {"error": repr(e)}
            torch.cuda.synchronize()
        print(json.dumps({name: res}), flush=True)
        log(f"other config {name} done")


def self_launch(argv, n):
    """`python bench.py --gpus N` with no launcher in the environment (the reference starts its N workers from one command too,
    examples/indexBatching/DCRNN/pems_ddp.py:198-207): re-run this script under torch.distributed.run, one rank per GPU, rendezvous
    on 127.0.0.1; the ranks' output (rank 0's ONE JSON line on stdout, stage marks on stderr) passes straight through."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # RCCL on hosts whose driver only has dmabuf IPC
    env.setdefault("OMP_NUM_THREADS", str(max(1, min(8, (os.cpu_count() or 1) // n))))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    log(f"--gpus {n} without a launcher: " + " ".join(cmd[1:8]) + " ...")
    return subprocess.run(cmd, env=env).returncode


def launch_probe(rank, world):
    """PGT_BENCH_LAUNCH_PROBE=1 (tests/test_distributed.py, CPU, gloo): the ranks meet, reduce once and rank 0 prints a line of the
    printed format — everything of the N > 1 path that does not need a GPU."""
    t = torch.tensor([float(rank + 1)])
    if world > 1:
        dist.barrier()
        dist.all_reduce(t)
    if rank == 0:
        bench_line.write_line(bench_line.compact({"metric": "snapshot-edges aggregated/sec", "value": 0.0, "unit": "snapshot-edges/s", "n_gpus": world,
                                                  "steps": 0, "warmup": 0, "ms_per_step": 0.0, "higher_is_better": True, "scaling": "weak",
                                                  "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                                                  "config": {"workload": "launch probe (no GPU work)", "rank_sum": float(t)}}))
    if world > 1:
        dist.destroy_process_group()


def pin_host_threads(local_rank, world):
    """8 ranks x ~250 ctypes launches per step share one host: give every rank its own slice of the cores and keep
    the intra-op pools small (the step is launch-issue bound on the host side, not compute bound)."""
    ncpu = os.cpu_count() or 1
    per = max(1, ncpu // max(world, 1))
    os.environ.setdefault("OMP_NUM_THREADS", str(min(per, 8)))
    torch.set_num_threads(min(per, 8))
    if world > 1 and hasattr(os, "sched_setaffinity"):
        try:
            avail = sorted(os.sched_getaffinity(0))
            per = max(1, len(avail) // world)
            os.sched_setaffinity(0, set(avail[local_rank * per:(local_rank + 1) * per]) or set(avail))
        except OSError:
            pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="dcrnn_metrla", choices=["dcrnn_metrla", "tgcn50k"],
                    help="dcrnn_metrla = BASELINE configs[1] (the headline); tgcn50k = configs[3]: the T = 12 BatchedTGCN training "
                         "step on the 50 000-node graph (bench_tgcn.py), same protocol, --gpus N through the same dp path")
    ap.add_argument("--batch", type=int, default=None, help="windows per GPU per step (weak scaling; default 1024, tgcn50k: 8)")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="strong scaling: total windows per step, split evenly over the ranks (overrides --batch)")
    ap.add_argument("--edges", type=int, default=N_EDGES, help="edges of the METR-LA-shaped graph (1722 = the reference's data)")
    ap.add_argument("--hidden", type=int, default=64, help="DCRNN hidden width (2 = the reference example's model)")
    ap.add_argument("--profile-steps", type=int, default=2, help="extra instrumented steps for the roofline figures")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ns", action="store_true", help="skip the N=200k F=64 aggregation micro-benchmark")
    ap.add_argument("--no-extra", action="store_true", help="skip the drop-in / E=1722 / small-batch / other-config lines")
    ap.add_argument("--graph", action="store_true",
                    help="forward+backward and the update as two hipGraphs per step, the all-reduce between them eager "
                         "(per-GPU batches whose step is host-launch bound, e.g. --global-batch 1024 on 8 GPUs)")
    ap.add_argument("--cpu-sweep-worker", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--aux-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--aux-only", default="", help=argparse.SUPPRESS)      # with --aux-worker: only these blocks (comma-separated)
    ap.add_argument("--aux-seconds", type=float, default=200.0,
                    help="wall-clock budget of the whole run after which the remaining AUXILIARY lines (variants, other configs) are "
                         "skipped and recorded as such: the default run has to finish within minutes on any box")
    args = ap.parse_args()
    if args.aux_worker:                        # child of the default run: the other configurations, one JSON line per block
        aux_worker(args)
        return
    if args.cpu_sweep_worker:                  # child of cpu_baseline's thread sweep: one thread count, CPU only, prints the rate
        print(cpu_sweep_point(args.hidden, args.cpu_sweep_worker))
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:      # plain `python bench.py --gpus N`: start the ranks ourselves
        sys.exit(self_launch(sys.argv[1:], args.gpus))
    t_start = time.time()
    bench_line.capture_stdout()                     # (nothing but the result line reaches this rank's stdout)
    if os.environ.get("PGT_BENCH_STACKS"):          # diagnostic: the Python stack on stderr every N seconds (where a slow box sits)
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["PGT_BENCH_STACKS"]), repeat=True, file=sys.stderr)
    over_budget = lambda: time.time() - t_start > args.aux_seconds      # noqa: E731
    args.batch_given = args.batch is not None
    if args.batch is None:
        args.batch = 1024
    # multi-process GPU work on this pool needs dmabuf IPC (the host driver has no legacy IPC); the launcher's value wins
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    # "nccl" is RCCL on ROCm; no-op for a single process.  PGT_BENCH_BACKEND=gloo is a self-test hook: it lets two ranks share
    # the one GPU of a test box so that the multi-rank code path (barrier, all-reduce, MAX-reduce, rank-0 printing) runs there.
    rank, local_rank, world = dp.init_from_env(os.environ.get("PGT_BENCH_BACKEND", "nccl"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if os.environ.get("PGT_BENCH_LAUNCH_PROBE") == "1":
        launch_probe(rank, world)
        return
    pin_host_threads(local_rank, world)
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    lib = _lib.get_lib()
    assert lib.target == "gfx950"
    if args.config == "tgcn50k":
        import bench_tgcn
        bench_tgcn.main(args, rank, local_rank, world, device, sys.modules[__name__])
        if world > 1:
            dist.destroy_process_group()
        return
    scaling = "weak"
    if args.global_batch > 0:
        assert args.global_batch % world == 0, "--global-batch must be a multiple of the number of ranks"
        args.batch, scaling = args.global_batch // world, "strong"

    series = torch.from_numpy(syn.traffic_series(34272, N_NODES, seed=1)).to(device)   # resident [T, N, 2]

    # north-star micro-benchmark (rank 0 of a single-GPU run only)
    ns = None
    if rank == 0 and world == 1 and not args.no_ns:
        ns = spmm_roofline_ns(device)
        log("north-star aggregation block done")

    dt, final_loss, step, (ei, ew) = train_run(device, rank, world, series, args.edges, args.batch, args.hidden,
                                               args.steps, args.warmup, 0 if args.graph else args.profile_steps,
                                               graph=args.graph)
    log(f"headline: {1e3 * dt / args.steps:.3f} ms/step")
    if args.graph:
        args.profile_steps = 0       # per-launch events cannot be recorded inside a captured step
    n_total = args.warmup + args.steps + args.profile_steps

    # ---- live roofline of the path's kernels: extra instrumented steps, HIP events on the launch stream
    roof, kernels = None, None
    if args.profile_steps > 0:
        # every rank runs the instrumented steps (they contain the gradient all-reduce: a collective); only rank 0
        # records events
        if rank == 0:
            ops.KERNEL_TIMER = ops.KernelTimer()
        for i in range(args.warmup + args.steps, n_total):
            step(i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
    if rank == 0 and args.profile_steps > 0:
        kernels = ops.KERNEL_TIMER.summary()
        shapes = ops.KERNEL_TIMER.by_tag()
        ops.KERNEL_TIMER = None
        kernel_class_rooflines(kernels, shapes)
        dom = max(kernels, key=lambda k: kernels[k]["total_ms"])
        k = kernels[dom]
        names = {"spmm": "spmm_wide_kernel<4> (pgt_spmm_csr_f32)",
                 "stack": "dconv_slab_fwd/bwd_kernel (pgt_dconv_stack_slab(_bwd)_f32): the whole K-hop recursion of one DConv, LDS-resident",
                 "gemm": "gemm_bx_kernel / gemm_bx_sym_kernel (split-bf16 on the bf16 matrix pipe: 330->128 + z|r gates, 330->64 + candidate "
                         "gate, 128->320 and 64->320 feature gradients) + the streaming read-out kernels, behind pgt_gemm_f32 / "
                         "pgt_gemm_gru_zr/h_f32; every launch of the entry points; bytes = operands once + results once",
                 "gemm_tn": "gemm_bx_tn_kernel (pgt_gemm_tn_acc_f32, split-bf16)",
                 "seq64": "dcrnn_seq64_fwd/bwd_kernel (pgt_dcrnn_seq64_f32 / _bwd_f32): all T cell steps of a sample in one workgroup — hops "
                          "gathered out of LDS, split-bf16 products on the terms while they are there, gate chains on the accumulators; "
                          "bytes = inputs once + what the adjoint / the weight gradient needs, written once"}
        roof = {"kernel": names.get(dom, dom), "bound": "hbm", "achieved": k["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": k["hbm_frac"], "traffic": None,
                "algorithmic_bytes_per_launch": k["algorithmic_bytes_per_launch"],
                "avg_us_per_launch": k["avg_us"], "launches_per_step": k["launches"] / args.profile_steps,
                "share_of_step_ms": k["total_ms"] / args.profile_steps}
        set_aside(roof, kernels)
        if dom in ("gemm", "gemm_tn"):
            roof["bf16_pipe_frac"] = k["bf16_pipe_frac"]
            roof["fp32_product_TFLOPs"] = k["fp32_product_TFLOPs"]
        if dom == "seq64":
            roof["bf16_pipe_frac"] = k["bf16_pipe_frac"]
            roof["by_direction"] = k["by_direction"]
            roof["why_hbm"] = ("a fused kernel: its HBM traffic is the saved activations (written once) — at 2 TB/s it is bound by neither "
                               "spec roofline but by the LDS gathers, the per-chunk hand-overs and the stores of the saved stacks inside a "
                               "CU (a fifth of the forward); the forward is a pure write stream: stream_ceiling_frac prices it against "
                               "the 4.5 TB/s a write-only stream reaches on this part (DESIGN.md 3.2)")
        roof.update(pmc_traffic(dom))
        roof["all_kernel_classes"] = all_kernel_classes(kernels, args.profile_steps)
    del step

    # ---- more than one GPU: the exchange on its own, and the same 1 024-window step split over the ranks (strong scaling) beside
    # the weak-scaling headline — every rank runs both (they contain collectives), rank 0 reports
    multi = None
    if world > 1:
        assert dist.is_initialized() and dist.get_world_size() == args.gpus, \
            f"--gpus {args.gpus} but the process group has {dist.get_world_size() if dist.is_initialized() else 0} rank(s)"
        multi = {"world_size": dist.get_world_size(), "backend": dist.get_backend()}
        try:
            n_par = sum(p.numel() for p in Model(args.hidden).parameters())
            buf = torch.zeros(n_par, device=device)
            for _ in range(5):
                dist.all_reduce(buf)
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                dist.all_reduce(buf)
            torch.cuda.synchronize()
            multi["allreduce_us_per_step"] = 1e6 * (time.perf_counter() - t0) / 50
            multi["allreduce_bytes"] = 4 * n_par
            multi["allreduce_what"] = "50 back-to-back all-reduces of the flat gradient buffer, wall clock / 50 (launch + ring latency; nothing overlaps it in the step)"
            del buf
        except Exception as e:
            multi["allreduce_error"] = repr(e)
        if scaling == "weak" and 1024 % world == 0 and not args.no_extra:
            try:
                torch.cuda.empty_cache()
                ds, _, st, _ = train_run(device, rank, world, series, args.edges, 1024 // world, args.hidden, args.steps, args.warmup, graph=True)
                del st
                multi["strong_scaling_global_batch_1024"] = {
                    "batch_per_gpu": 1024 // world, "ms_per_step": 1e3 * ds / args.steps, "graphed": True,
                    "value": 1024 * SEQ * args.edges * args.steps / ds,
                    "what": "the fixed 1 024-window step split over the ranks (forward + backward and the update as hipGraphs, the "
                            "all-reduce between them eager); compare with the 1-GPU line's ms_per_step"}
                log(f"strong scaling, {1024 // world} windows per GPU: {1e3 * ds / args.steps:.3f} ms/step")
            except Exception as e:
                multi["strong_scaling_global_batch_1024"] = {"error": repr(e)}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        log("cpu baseline ...")
        cpu = cpu_baseline(args.hidden)
        log("cpu baseline done")
        try:                                                   # beside roofline_ns_spmm_N200k_F64 (same operator)
            cpu["optimised_spmm"] = cpu_spmm_ns(cpu["cores"])
        except Exception as e:                                 # an auxiliary line must never cost the bench line
            cpu["optimised_spmm"] = {"error": repr(e)}

    def throughput(d, edges, batch, steps):
        return {"value": world * batch * SEQ * edges * steps / d, "ms_per_step": 1e3 * d / steps,
                "edge_messages_per_s": world * batch * SEQ * edges * steps / d * PROPAGATES_PER_CELL}

    variants, extra = None, None
    if world == 1 and not args.no_extra:
        # the same step (a) with torch.nn.Linear as the read-out = what swapping the import alone gives, (b) on the 1 722-edge
        # graph of the reference's data, (c) with the split-bf16 kernels off = every product an exact fp32 fmaf chain,
        # (d) with atomics-free (bitwise reproducible) weight gradients; short runs, same bracketing
        variants = {}

        def variant(name, what, edges=args.edges, **kw):
            if over_budget():
                variants[name] = {"skipped": f"--aux-seconds {args.aux_seconds:g} spent"}
                return
            torch.cuda.empty_cache()
            try:
                best = None
                for _ in range(2):                            # short auxiliary runs on a shared box: the better of two
                    d, _, st, _ = train_run(device, rank, world, series, edges, args.batch, args.hidden, 8, 3, **kw)
                    del st
                    best = d if best is None or d < best else best
                variants[name] = dict(throughput(best, edges, args.batch, 8), what=what + " (8 steps, better of two runs)")
                log(f"variant {name}: {1e3 * best / 8:.3f} ms/step")
            except Exception as e:                            # an auxiliary line must never cost the bench line
                variants[name] = {"error": repr(e)}
            torch.cuda.empty_cache()

        variant("dropin_default", "import swap only: BatchedDCRNN defaults + the user's torch.nn.Linear read-out (its F.linear call on "
                "the returned states runs on the package's streaming kernels: BatchedDCRNN.readout_interception)", dropin=True)
        variant("dropin_relu_readout", "import swap only, with the relu the reference's own models put between the recurrent layer and "
                "the torch.nn.Linear read-out (dcrnn_example.py:27-28, tgcn/metr_la_main.py:43-44): relu(states) keeps the routing, the "
                "relu itself is torch's", dropin=True, relu=True)
        variant("edges_1722", "1 722-edge graph (the reference's METR-LA data)", edges=1722)
        variant("torch_adam", "torch.optim.Adam(fused=True) over the flat parameter buffer instead of dp.FlatAdam (pgt_adam_f32)", torch_adam=True)
        ops.USE_SEQ64 = False
        try:
            variant("per_step_launches", "PGT_SEQ64=0: round 5's path — four launches per cell step forward (two LDS-resident stacks, two "
                    "gate-fused split-bf16 products), six backward — instead of the one-launch sequence kernels (csrc/seq64.hip)")
            lib.tune("gemm_bx", 0)
            try:
                variant("exact_fp32", "PGT_SEQ64=0 + pgt_tune(gemm_bx, 0): every dense product on v_mfma_f32_32x32x2_f32 (bitwise an fmaf chain)")
            finally:
                lib.tune("gemm_bx", 1)
        finally:
            ops.USE_SEQ64 = True
        lib.tune("gemm_bx_tn_pc", 0)
        lib.tune("gemm_bx_sym_pc", 0)
        try:
            variant("wavefronts_all_alike", "pgt_tune(gemm_bx_tn_pc, 0) + (gemm_bx_sym_pc, 0): the weight-gradient and 64 -> 320 products on round 4's "
                    "kernels (every wavefront converts, multiplies and stores) instead of producers / consumers — the same bits, the same box")
        finally:
            lib.tune("gemm_bx_tn_pc", 1)
            lib.tune("gemm_bx_sym_pc", 1)
        keep_det = ops.DETERMINISTIC_WEIGHT_GRADIENTS
        ops.DETERMINISTIC_WEIGHT_GRADIENTS = not keep_det
        try:
            if keep_det:
                variant("atomic_weight_gradients", "PGT_DETERMINISTIC=0: weight gradients through fp32 atomics into dW (the default adds per-slab "
                        "partial sums in a fixed order: bitwise reproducible)")
            else:
                variant("deterministic", "PGT_DETERMINISTIC=1: weight gradients without float atomics (per-slab partial sums added in order)")
        finally:
            ops.DETERMINISTIC_WEIGHT_GRADIENTS = keep_det
        if not args.graph:
            variant("hipgraph_step", "--graph: forward+backward and the update as two hipGraphs, the all-reduce between them eager",
                    graph=True)
        # what 8-way STRONG scaling of the headline could reach: one GPU's share of a 1024-window step is B = 128; its step time
        # (as hipGraphs: that batch is host-launch bound when issued eagerly) against the B = 1024 step is the ceiling of the
        # speed-up, measurable on one GPU
        try:
            if over_budget():
                raise TimeoutError(f"--aux-seconds {args.aux_seconds:g} spent")
            torch.cuda.empty_cache()
            d128 = min(train_run(device, rank, world, series, args.edges, 128, args.hidden, 8, 3, graph=True)[0] for _ in range(2))
            variants["strong_scaling_projection"] = {
                "ms_per_step_B128_graphed": 1e3 * d128 / 8, "ms_per_step_B1024": 1e3 * dt / args.steps,
                "ceiling_of_8way_strong_scaling": (dt / args.steps) / (d128 / 8),
                "what": "ms_per_step(B = 1024) / ms_per_step(B = 128, --graph): the speed-up 8 GPUs could reach on a fixed 1024-window "
                        "step if the all-reduce were free (one GPU's measurement, not a scaling run)"}
        except Exception as e:
            variants["strong_scaling_projection"] = {"error": repr(e)}
        torch.cuda.empty_cache()
        # last: the BLAS library brings its own workspaces into the pool
        BatchedDCRNN.readout_interception = False
        try:
            variant("dropin_blas_readout", "dropin_default with readout_interception = False: torch's own BLAS product for the 64 -> 2 "
                    "read-out over 2.5 M rows", dropin=True)
        finally:
            BatchedDCRNN.readout_interception = True
    if rank == 0 and world == 1 and not args.no_extra:
        # the other configurations run in a CHILD process, one JSON line per finished block: a fault of the GPU runtime inside an
        # auxiliary block (one aborted a builder run of this round) ends the child, never this process — the bench line is printed
        # whatever happens there, with the blocks that did finish and the child's exit status for the rest
        import subprocess
        del series
        torch.cuda.empty_cache()
        budget = max(30.0, args.aux_seconds - (time.time() - t_start)) + 90.0
        extra = {}
        out, err, status = "", "", "not started"
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--aux-worker", "--edges", str(args.edges),
                                "--aux-seconds", str(max(20.0, args.aux_seconds - (time.time() - t_start)))],
                               capture_output=True, text=True, timeout=budget)
            status = f"exit status {r.returncode}"
            out, err = r.stdout, r.stderr
        except subprocess.TimeoutExpired as e:
            status = f"no exit within {budget:.0f} s"
            out = e.stdout.decode() if isinstance(e.stdout, bytes) else (e.stdout or "")
            err = e.stderr.decode() if isinstance(e.stderr, bytes) else (e.stderr or "")
        except Exception as e:                                 # (could not even start it)
            status = repr(e)
        for ln in out.splitlines():
            if ln.startswith("{"):
                try:
                    extra.update(json.loads(ln))
                except ValueError:
                    pass
        for ln in err.splitlines():
            if ln.startswith("[bench"):
                print("  (aux) " + ln, file=sys.stderr)
        for name in AUX_BLOCKS:
            extra.setdefault(name, {"error": f"the auxiliary process ended before this block finished ({status}): "
                                             + " | ".join(l for l in err.splitlines()[-3:] if not l.startswith("[bench"))[-300:]})
        log(f"other configs: child {status}")

    if rank == 0:
        head = throughput(dt, args.edges, args.batch, args.steps)
        line = {
            "metric": "snapshot-edges aggregated/sec",
            "value": head["value"],
            "unit": "snapshot-edges/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"],
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f32 (fp32 operands and results; dense products as three bf16 pieces per operand, six piece products, fp32 "
                     "accumulation on the bf16 matrix pipe: error vs fp64 below the exact-fp32 fmaf chain's; variants.exact_fp32 = that chain)",
            "data": "synthetic",
            "config": {"workload": f"METR-LA-shaped synthetic (207 nodes, {args.edges} edges, 12-step) BatchedDCRNN(2,{args.hidden},K=3)"
                                   + ("" if args.hidden == 2 else "+Linear") + " training step (fwd+bwd+allreduce+Adam)",
                       "batch_per_gpu": args.batch, "global_batch": world * args.batch, "seq_len": SEQ,
                       "parallelism": f"dp{world}", "ranks": (dist.get_world_size() if world > 1 else 1),
                       "backend": (dist.get_backend() if world > 1 else None), "hidden": args.hidden, "K": 3, "init_passes": 1,
                       "graphed": bool(args.graph),
                       "output_layout": "BatchedDCRNN default (contiguous [B,T,N,O], stored by the gate epilogues) + this package's Linear "
                                        "read-out; torch.nn.Linear read-out = variants.dropin_default (routed to the same kernels) / dropin_blas_readout (torch's BLAS)"},
            "edge_messages_per_s": head["edge_messages_per_s"],
            "epoch_time_s_23974_windows": 23974.0 / (world * args.batch) * dt / args.steps,
            "final_loss": final_loss,
            "roofline": roof, "kernels": kernels, "roofline_ns_spmm_N200k_F64": ns, "cpu_baseline": cpu,
            "multi_gpu": multi, "variants": variants, "other_configs": extra,
        }
        partial = args.no_extra or args.no_cpu_baseline or args.profile_steps == 0      # a profiler pass must not overwrite the clocked run's record
        bench_line.emit(line, "bench_partial.json" if partial else "bench_full.json",       # complete record -> gpurun_out/ + stderr; ONE short
                        before=lambda text: log(f"headline: {head['ms_per_step']:.3f} ms/step; printed line {len(text)} bytes"))   # line on stdout, last
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
