"""The ONE JSON line bench.py prints, kept small enough that a log tail holds all of it.

bench.py measures a great deal (per-shape product tables, thread sweeps, every auxiliary configuration with its own kernel
table); the driver reads the last few KB of stdout, so the printed line carries the contract fields, the roofline of the
dominant kernel class, the CPU baseline and ONE figure per auxiliary block.  The complete record goes to
`gpurun_out/bench_full.json` (and, as a run of `[bench-full]` lines, to stderr).  tests/test_bench_line.py builds a worst-case
record and holds the printed line under LINE_LIMIT bytes.  No torch import: the CPU suite loads this module alone.
"""
import json
import math
import os
import sys

LINE_LIMIT = 6144        # bytes of the printed line (the driver's stdout tail is ~8 KB: keep a wide margin)
ROOT = os.path.dirname(os.path.abspath(__file__))


def _r(x, sig=5):
    """Numbers to `sig` significant digits (the full record keeps every digit)."""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, int):
        return x
    if isinstance(x, float):
        if not math.isfinite(x):
            return None
        if x == 0.0:
            return 0.0
        d = sig - 1 - int(math.floor(math.log10(abs(x))))
        y = round(x, d)
        return int(y) if d <= 0 else y
    return x


def _s(x, n):
    """Strings cut to n characters."""
    if x is None:
        return None
    x = str(x)
    return x if len(x) <= n else x[: n - 1] + "~"


def _pick(d, *names):
    """First of `names` present (and a number) in dict d, rounded."""
    if not isinstance(d, dict):
        return None
    for n in names:
        v = d.get(n)
        if isinstance(v, (int, float)) and not isinstance(v, bool):
            return _r(v)
    return None


def _problem(d):
    """'error' / 'skipped' of an auxiliary block as a short string, else None."""
    if isinstance(d, dict):
        for k in ("error", "skipped"):
            if k in d:
                return {k: _s(d[k], 60)}
    return None


def compact_roofline(roof):
    if not isinstance(roof, dict):
        return None
    out = {"kernel": _s(roof.get("kernel") or roof.get("kernel_class"), 110), "bound": roof.get("bound"),
           "achieved": _r(roof.get("achieved")), "peak": _r(roof.get("peak")), "unit": roof.get("unit"), "frac": _r(roof.get("frac")),
           "traffic": _r(roof.get("traffic"), 6),
           "traffic_measured_in_this_run": bool(roof.get("traffic_measured_in_this_run", False)),
           "traffic_source": _s(roof.get("traffic_source"), 90),
           "algorithmic_bytes_per_launch": _r(roof.get("algorithmic_bytes_per_launch"), 6),
           "avg_us_per_launch": _r(roof.get("avg_us_per_launch")), "launches_per_step": _r(roof.get("launches_per_step")),
           "share_of_step_ms": _r(roof.get("share_of_step_ms"))}
    for k in ("bf16_pipe_frac", "fp32_product_TFLOPs"):
        if roof.get(k) is not None:
            out[k] = _r(roof[k])
    if roof.get("why_hbm"):
        out["why_hbm"] = _s(roof["why_hbm"], 240)
    if isinstance(roof.get("by_direction"), list):
        out["by_direction"] = [{"dir": d.get("direction"), "us": _r(d.get("avg_us")), "hbm_frac": _r(d.get("hbm_frac"), 3),
                                "stream_ceiling_frac": _r(d.get("stream_ceiling_frac"), 3),
                                "bf16_pipe_frac": _r(d.get("bf16_pipe_frac"), 3)} for d in roof["by_direction"][:2]]
    if isinstance(roof.get("set_aside"), dict):
        out["set_aside"] = {"launches": roof["set_aside"].get("launches"), "ms": _r(roof["set_aside"].get("ms"), 4),
                            "rule": "over 10x the median of its shape (host stall inside the event pair)"}
    ak = roof.get("all_kernel_classes")
    if isinstance(ak, dict):
        out["all_kernel_classes"] = {"frac": _r(ak.get("hbm_frac")), "ms_per_step_in_timed_kernels": _r(ak.get("ms_per_step_in_timed_kernels")),
                                     "note": "instrumented steps (events around every launch) run slower than the timed ones: "
                                             "per-class fractions are lower bounds",
                                     "classes_ms_frac": ak.get("classes_ms_frac") or {k: [_r(v, 4)] for k, v in (ak.get("classes") or {}).items()}}
    return out


def compact_cpu(cpu):
    if not isinstance(cpu, dict):
        return None
    used = cpu.get("threads_used", cpu.get("cores"))
    return {"value": _r(cpu.get("value")), "unit": cpu.get("unit"), "cores": used, "threads_used": used,
            "host_threads": cpu.get("host_threads", cpu.get("host_hardware_threads")), "kind": cpu.get("kind"),
            "sample": _s(cpu.get("sample"), 150),
            **({"optimised_spmm_GBs": _pick(cpu.get("optimised_spmm"), "achieved_GBs")} if cpu.get("optimised_spmm") else {})}


def compact_ns(ns):
    """One fraction (+ the launch time) per graph kind of the north-star aggregation block."""
    if not isinstance(ns, dict):
        return None
    out = {}
    for k, v in ns.items():
        if not isinstance(v, dict):
            continue
        e = {"frac": _r(v.get("frac"), 4), "us": _r(v.get("us_per_launch"), 4)}
        if v.get("renumbered"):
            e["renumbered"] = True
        hops = v.get("dconv_K3_hops")
        if isinstance(hops, dict) and hops.get("frac") is not None:
            e["dconv_K3_hop_frac"] = _r(hops["frac"], 4)
        out[_s(k, 24)] = e
    return out


def compact_variants(variants):
    if not isinstance(variants, dict):
        return None
    out = {}
    for k, v in variants.items():
        p = _problem(v)
        if p is not None:
            out[_s(k, 32)] = p
        elif isinstance(v, dict) and "ceiling_of_8way_strong_scaling" in v:
            out[_s(k, 32)] = {"ceiling_of_8way": _r(v["ceiling_of_8way_strong_scaling"], 4), "ms_B128_graphed": _r(v.get("ms_per_step_B128_graphed"), 4)}
        else:
            out[_s(k, 32)] = _pick(v, "ms_per_step")
    return out


def compact_other(extra):
    """One time and (where the block prices a kernel class) one fraction per auxiliary configuration."""
    if not isinstance(extra, dict):
        return None
    out = {}
    for k, v in extra.items():
        p = _problem(v)
        if p is not None:
            out[_s(k, 32)] = p
            continue
        if not isinstance(v, dict):
            continue
        if k == "small_batch":
            out[k] = {hk: {"ms": _pick(hv, "graphed_ms_per_step", "eager_ms_per_step"), "eager_ms": _pick(hv, "eager_ms_per_step"),
                           "cpu_ms": _pick(hv, "cpu_oracle_ms_per_step")}
                      for hk, hv in v.items() if isinstance(hv, dict)}
            continue
        e = {"ms": _pick(v, "ms_per_step", "gpu_graphed_ms_per_epoch", "gpu_graphed_ms", "gpu_eager_ms_per_epoch", "gpu_eager_ms"),
             "eager_ms": _pick(v, "gpu_eager_ms_per_epoch", "gpu_eager_ms"),
             "cpu_ms": _pick(v, "cpu_oracle_ms_per_epoch", "cpu_oracle_ms"),
             "snapshot_edges_per_s": _pick(v, "snapshot_edges_per_s", "snapshot_edges_per_s_graphed", "snapshot_edges_per_s_eager")}
        if isinstance(v.get("cpu_oracle"), dict):
            e["cpu_ms"] = _pick(v["cpu_oracle"], "ms_per_step")
        roof = v.get("roofline")
        if isinstance(roof, dict):
            e["roofline"] = {"kernel": _s(roof.get("kernel") or roof.get("kernel_class"), 40), "frac": _r(roof.get("frac"), 4),
                             "avg_us": _r(roof.get("avg_us_per_launch"), 4), "traffic": _r(roof.get("traffic"), 6),
                             "algorithmic_bytes_per_launch": _r(roof.get("algorithmic_bytes_per_launch"), 6)}
            ak = roof.get("all_kernel_classes")
            if isinstance(ak, dict):
                e["roofline"]["all_classes_frac"] = _r(ak.get("hbm_frac"), 4)
        for name in ("batch_per_gpu",):
            if name in v:
                e[name] = v[name]
        if isinstance(v.get("graphed"), dict) and "ms_per_step" in v["graphed"]:
            e["graphed_ms"] = _r(v["graphed"]["ms_per_step"], 4)      # the same step as two hipGraphs (eager is host-launch bound)
        for name, sub in v.items():                       # nested measurements of the same block at another batch size
            if isinstance(sub, dict) and "batch_per_gpu" in sub and "ms_per_step" in sub:
                e[_s(name, 28)] = {"batch_per_gpu": sub["batch_per_gpu"], "ms": _r(sub["ms_per_step"], 4)}
        out[_s(k, 32)] = {a: b for a, b in e.items() if b is not None}
    return out


CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data")


def compact(full, full_path=None):
    """The printed line from the complete record."""
    line = {k: (_r(full.get(k), 8) if isinstance(full.get(k), float) else full.get(k)) for k in CONTRACT}
    line["dtype"] = _s(str(full.get("dtype", "f32")).split(" ")[0], 8)
    cfg = dict(full.get("config") or {})
    line["config"] = {k: (_s(v, 170) if isinstance(v, str) else v) for k, v in cfg.items() if k != "output_layout"}
    for k in ("edge_messages_per_s", "epoch_time_s_23974_windows", "final_loss"):
        if full.get(k) is not None:
            line[k] = _r(full[k])
    line["roofline"] = compact_roofline(full.get("roofline"))
    line["cpu_baseline"] = compact_cpu(full.get("cpu_baseline"))
    if full.get("roofline_ns_spmm_N200k_F64") is not None:
        line["roofline_ns_spmm_N200k_F64"] = compact_ns(full["roofline_ns_spmm_N200k_F64"])
    if isinstance(full.get("multi_gpu"), dict):
        m = full["multi_gpu"]
        line["multi_gpu"] = {"world_size": m.get("world_size"), "backend": m.get("backend"), "allreduce_us_per_step": _r(m.get("allreduce_us_per_step")),
                             "allreduce_bytes": m.get("allreduce_bytes"),
                             "strong_scaling_global_batch_1024": (_problem(m.get("strong_scaling_global_batch_1024")) or
                                                                  {k: _r(v) if isinstance(v, float) else v
                                                                   for k, v in (m.get("strong_scaling_global_batch_1024") or {}).items() if k != "what"})
                             if m.get("strong_scaling_global_batch_1024") is not None else None,
                             **({"allreduce_error": _s(m["allreduce_error"], 120)} if m.get("allreduce_error") else {})}
    if full.get("variants") is not None:
        line["variants_ms_per_step"] = compact_variants(full["variants"])
    if full.get("other_configs") is not None:
        line["other_configs"] = compact_other(full["other_configs"])
    if full_path:
        line["full_record"] = full_path
    text = json.dumps(line, separators=(",", ":"))
    # a record nobody foresaw must still give a whole line: drop the auxiliary objects, largest first, until it fits
    for k in ("other_configs", "variants_ms_per_step", "roofline_ns_spmm_N200k_F64"):
        if len(text) <= LINE_LIMIT:
            break
        if k in line:
            line[k] = {"dropped": "line over the size limit; see full_record"}
            text = json.dumps(line, separators=(",", ":"))
    return text


def emit(full, name="bench_full.json", before=None):
    """Write the complete record under gpurun_out/ (merged back from a GPU box) and to stderr, then print the compact line on stdout
    as the LAST thing the process writes (`before(text)`: the caller's closing stage mark, ahead of it) — a tail of the merged
    streams ends with the line."""
    path = None
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, name), "w") as fh:
            json.dump(full, fh)
        path = "gpurun_out/" + name
    except OSError:
        pass
    print("[bench-full] " + json.dumps(full), file=sys.stderr, flush=True)
    text = compact(full, path)
    if before is not None:
        before(text)
    sys.stderr.flush()
    write_line(text)
    return text


_REAL_STDOUT = None


def capture_stdout():
    """A rank's stdout carries ONE line (rank 0's result).  Libraries under the script write there too — gloo announces its
    connections on stdout from C++, RCCL its version when NCCL_DEBUG is set — so fd 1 is pointed at stderr for the run and the line
    goes out through a duplicate of the real one (`write_line`)."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def write_line(text):
    sys.stdout.flush()
    if _REAL_STDOUT is None:
        print(text, flush=True)
    else:
        os.write(_REAL_STDOUT, (text + "\n").encode())
