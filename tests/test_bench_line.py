"""The printed bench line has to fit a log tail (the driver's parser reads the last few KB of stdout): worst-case records
through bench_line.compact stay under the limit and keep the contract fields."""
import json
import os

import pytest

import bench_line

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LONG = "x" * 400
NUM = 1234567.8912345678


def inflate(v):
    """Every string as long as anyone ever wrote one, every number with all its digits."""
    if isinstance(v, dict):
        return {k: inflate(x) for k, x in v.items()}
    if isinstance(v, list):
        return [inflate(x) for x in v]
    if isinstance(v, str):
        return v + LONG
    if isinstance(v, float):
        return v * 1.0000000123456789
    return v


def worst_case():
    with open(os.path.join(ROOT, "profiles", "r04n_bench.json")) as fh:
        full = json.load(fh)
    full = inflate(full)
    # auxiliary blocks that failed carry whole exception texts
    full["other_configs"]["config5_covid_evolvegcnh"] = {"error": "RuntimeError(" + LONG * 3 + ")"}
    full["variants"]["exact_fp32"] = {"error": LONG * 2}
    full["variants"]["deterministic"] = {"skipped": LONG}
    # more graph kinds, more variants, more batch sizes than any run so far
    for i in range(6):
        full["roofline_ns_spmm_N200k_F64"][f"another_graph_kind_{i}"] = dict(full["roofline_ns_spmm_N200k_F64"]["grid2d_shuffled"])
        full["variants"][f"another_variant_{i}"] = dict(full["variants"]["dropin_default"])
    full["other_configs"]["config4_50k_tgcn2"]["batch_64_reference_default"] = {"batch_per_gpu": 64, "ms_per_step": NUM, "kernels": {"a": LONG}}
    return full


def test_worst_case_line_fits_the_tail():
    text = bench_line.compact(worst_case(), "gpurun_out/bench_full.json")
    assert len(text.encode()) < bench_line.LINE_LIMIT < 8192
    assert "\n" not in text
    line = json.loads(text)
    for k in bench_line.CONTRACT + ("config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["dtype"] == "f32"
    r = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = line["cpu_baseline"]
    for k in ("value", "unit", "cores", "threads_used", "host_threads", "kind", "sample"):
        assert k in c, k
    assert "workload" in line["config"]


def test_real_record_keeps_one_figure_per_block():
    with open(os.path.join(ROOT, "profiles", "r04n_bench.json")) as fh:
        full = json.load(fh)
    text = bench_line.compact(full)
    assert len(text.encode()) < bench_line.LINE_LIMIT
    line = json.loads(text)
    assert line["value"] == pytest.approx(full["value"], rel=1e-6) and line["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-6)
    assert line["roofline"]["frac"] == pytest.approx(full["roofline"]["frac"], rel=1e-3)
    assert set(line["roofline_ns_spmm_N200k_F64"]) == set(full["roofline_ns_spmm_N200k_F64"])
    assert line["roofline_ns_spmm_N200k_F64"]["local"]["frac"] == pytest.approx(full["roofline_ns_spmm_N200k_F64"]["local"]["frac"], rel=1e-3)
    assert set(line["variants_ms_per_step"]) == set(full["variants"])
    assert set(line["other_configs"]) == set(full["other_configs"])
    c4 = line["other_configs"]["config4_50k_tgcn2"]
    assert c4["ms"] == pytest.approx(full["other_configs"]["config4_50k_tgcn2"]["ms_per_step"], rel=1e-3)
    assert c4["roofline"]["frac"] == pytest.approx(full["other_configs"]["config4_50k_tgcn2"]["roofline"]["frac"], rel=1e-3)
    assert "largest_batch_within_10ms" in c4


def test_unforeseen_record_still_gives_a_whole_line():
    full = worst_case()
    full["other_configs"] = {f"block_{i}": {"ms_per_step": NUM, "roofline": {"kernel": LONG, "frac": 0.5}} for i in range(200)}
    text = bench_line.compact(full)
    assert len(text.encode()) < bench_line.LINE_LIMIT
    line = json.loads(text)
    assert "dropped" in line["other_configs"] and line["roofline"] is not None and line["cpu_baseline"] is not None


def test_kernel_timer_sets_a_stalled_launch_aside_and_says_so(monkeypatch):
    """A host stall between the first event and the launch (72 ms inside one of 24 launches of a 119 us product in the round-5
    evidence run) must not become the kernel's average: the launch is set aside by the 10x-the-median-of-its-shape rule, counted
    and reported; fewer than three launches of a shape are never judged."""
    import torch
    from pytorch_geometric_temporal_amd import ops

    clock = {"t": 0.0}

    class FakeEvent:
        def __init__(self, enable_timing=True):
            self.t = None

        def record(self):
            self.t = clock["t"]

        def elapsed_time(self, other):
            return other.t - self.t

    monkeypatch.setattr(torch.cuda, "Event", FakeEvent)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    kt = ops.KernelTimer()

    def kernel(ms):
        def fn():
            clock["t"] += ms
        return fn

    for i in range(24):
        kt.launch("gemm", 100.0, kernel(72.0 if i == 5 else 0.119), tag=("NT", 211968, 320))
    for i in range(2):
        kt.launch("gemm", 10.0, kernel(50.0 if i else 0.1), tag=("NN", 8, 8))       # two launches: no median to judge by
    for i in range(10):
        kt.launch("stack", 7.0, kernel(0.08))
    s = kt.summary()
    assert s["gemm"]["launches"] == 25 and s["gemm"]["set_aside_launches"] == 1 and abs(s["gemm"]["set_aside_ms"] - 72.0) < 1e-9
    assert "set_aside_launches" not in s["stack"] and abs(s["stack"]["avg_us"] - 80.0) < 1e-6
    tags = {tuple(r["tag"]): r for r in kt.by_tag()}
    nt = tags[("gemm", "NT", 211968, 320)]
    assert nt["launches"] == 23 and abs(nt["avg_us"] - 119.0) < 1e-6 and nt["set_aside_launches"] == 1
    assert tags[("gemm", "NN", 8, 8)]["launches"] == 2 and "set_aside_launches" not in tags[("gemm", "NN", 8, 8)]
    # the printed line carries the count
    import bench_line
    line = json.loads(bench_line.compact({"roofline": {"kernel": "k", "frac": 0.4, "set_aside": {"launches": 1, "ms": 72.0, "rule": "x"}}}))
    assert line["roofline"]["set_aside"]["launches"] == 1
