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
