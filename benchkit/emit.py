"""The ONE stdout line of bench.py.

The full record (every sub-measurement, notes, per-launch tables: 20+ KB) goes to `bench_full.json` at the repo root (and to
`gpurun_out/bench_full.json` when that directory exists); stdout gets, as its LAST and only line, a compact strict-JSON object of at
most MAX_LINE bytes with the contract keys (metric, value, unit, n_gpus, steps, warmup, ms_per_step, higher_is_better, scaling,
vs_baseline, dtype, data, config), `roofline`, `cpu_baseline`, the parity figures and an `extra` map name -> [ms_per_step,
views_per_s, executed_frac]. Strict JSON: NaN / +-inf become null (`allow_nan=False` would otherwise raise).
Round 5's driver record had `parsed: null`: its line was 22 KB and carried the token `Infinity`."""
import json
import math
import os
import sys

from benchkit.common import ROOT

MAX_LINE = 4000
FULL_NAME = "bench_full.json"


def strict(o):
    """A copy of `o` that json.dumps(allow_nan=False) accepts: non-finite floats -> None, tuples -> lists, unknown scalars -> float / str."""
    if isinstance(o, float):
        return o if math.isfinite(o) else None
    if isinstance(o, (str, int, bool)) or o is None:
        return o
    if isinstance(o, dict):
        return {str(k): strict(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [strict(v) for v in o]
    try:
        return strict(float(o))
    except Exception:
        return str(o)


def sig(x, n=5):
    """x rounded to n significant digits (None / non-finite -> None): keeps the compact line short without hiding anything the full record has."""
    if x is None or isinstance(x, bool):
        return x
    if isinstance(x, int):
        return x
    try:
        x = float(x)
    except Exception:
        return None
    if not math.isfinite(x):
        return None
    if x == 0.0:
        return 0.0
    return round(x, n - 1 - int(math.floor(math.log10(abs(x)))))


def clip(s, n):
    if s is None:
        return None
    s = str(s)
    return s if len(s) <= n else s[:n - 3] + "..."


def pick(d, keys, text=160):
    """Sub-dict of d with `keys` (missing -> absent), floats rounded, strings clipped."""
    out = {}
    for k in keys:
        if isinstance(d, dict) and k in d:
            v = d[k]
            out[k] = clip(v, text) if isinstance(v, str) else (sig(v) if isinstance(v, float) else v)
    return out


CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
ROOFLINE = ("kernel", "bound", "achieved", "peak", "unit", "frac", "frac_rocprof", "rocprof_source", "traffic", "traffic_algorithmic_bytes",
            "traffic_source", "executed_gflop", "floor_ms", "executed_frac")
CPU = ("value", "unit", "cores", "cpu_model", "kind", "sample")


def extra_row(e):
    """[ms_per_step, views_per_s, executed_frac] of one extra_configs entry (an entry that failed: its error string)."""
    if not isinstance(e, dict):
        return None
    if "error" in e and "ms_per_step" not in e:
        return clip(e["error"], 80)
    return [sig(e.get("ms_per_step"), 4), sig(e.get("views_per_s"), 4), sig((e.get("roofline") or {}).get("executed_frac"), 3)]


def multi_row(r):
    """[ms_per_step, payload_MB, backend, ranks_ok] of one multi_rank sub-record."""
    if not isinstance(r, dict):
        return None
    if "error" in r and "ms_per_step" not in r:
        return clip(r["error"], 100)
    payload = r.get("gradient_bytes_all_reduced_per_step") or ((r.get("all_gather_bytes_per_step") or 0) + (r.get("all_reduce_bytes_per_step") or 0))
    pg = r.get("process_group")
    backend = pg.get("backend") if isinstance(pg, dict) else pg
    row = [sig(r.get("ms_per_step"), 4), sig(payload / 1e6, 4) if payload else None, backend, r.get("ranks_ok")]
    return row + [clip("; ".join(r["errors"]), 80)] if r.get("errors") else row


def compact(full):
    """The driver's line from the full record."""
    c = pick(full, CONTRACT, text=120)
    for k in CONTRACT:                                   # contract keys are always present, null when unknown
        c.setdefault(k, None)
    cfg = full.get("config") or {}
    c["config"] = pick(cfg, ("workload", "scenes_per_gpu", "views_in", "views_out", "feature_grid", "render_grid", "steps_in_flight", "launch",
                             "parallelism", "global_batch"), text=420)
    if "launch" in c["config"]:
        c["config"]["launch"] = clip(c["config"]["launch"], 60)
    if "parallelism" in c["config"]:
        c["config"]["parallelism"] = clip(c["config"]["parallelism"], 60)
    for k in ("ranks_ok", "psnr_vs_oracle_db", "max_abs_err_vs_oracle", "speedup_vs_cpu_baseline", "dry_run", "train", "replicas_identical",
              "views_counted", "mean_loss_all_ranks", "error"):
        if k in full:
            c[k] = sig(full[k]) if isinstance(full[k], float) else (clip(full[k], 200) if isinstance(full[k], str) else full[k])
    pg = full.get("process_group")
    if pg is not None:
        c["process_group"] = pick(pg, ("backend", "world_size", "initialized"), text=40) if isinstance(pg, dict) else pg
    if full.get("errors"):
        c["errors"] = [clip(e, 160) for e in full["errors"][:4]]
    if isinstance(full.get("psnr_to_target_db"), dict):
        c["psnr_abs_diff_vs_oracle_db"] = sig(full["psnr_to_target_db"].get("abs_diff"), 3)
    if isinstance(full.get("roofline"), dict):
        c["roofline"] = pick(full["roofline"], ROOFLINE, text=130)
    if isinstance(full.get("cpu_baseline"), dict):
        c["cpu_baseline"] = pick(full["cpu_baseline"], CPU, text=200)
    if isinstance(full.get("single_stream"), dict):
        c["single_stream_ms"] = sig(full["single_stream"].get("ms_per_step"), 4)
    if isinstance(full.get("repeats"), dict):
        c["repeats"] = pick(full["repeats"], ("regions", "value_min", "value_max"))
    if isinstance(full.get("kernels"), dict):            # HBM-bound hand-written kernels: fraction of the 8 TB/s peak on algorithmic bytes
        c["hbm_kernels_frac"] = {k.replace("_kernel", ""): sig(v.get("frac"), 3) for k, v in full["kernels"].items()
                                 if isinstance(v, dict) and v.get("bound") == "hbm"}
    if isinstance(full.get("extra_configs"), list):
        c["extra"] = {e.get("name", "?"): extra_row(e) for e in full["extra_configs"] if isinstance(e, dict)}
        c["extra_columns"] = ["ms_per_step", "views_per_s", "executed_frac"]
        # the same step with several replays / refinement instances in flight, as the headline runs configs[1] (throughput, not latency)
        fl = {e.get("name", "?"): [sig(e["pipelined"].get("ms_per_step"), 4), sig(e["pipelined"].get("views_per_s"), 4),
                                   sig(e["pipelined"].get("executed_frac"), 3), e["pipelined"].get("depth")]
              for e in full["extra_configs"] if isinstance(e, dict) and isinstance(e.get("pipelined"), dict) and "ms_per_step" in e["pipelined"]}
        if fl:
            c["extra_in_flight"] = fl
        rp = {e.get("name", "?"): [sig(e["hipgraph_replay"].get("ms_per_step"), 4), sig(e["hipgraph_replay"].get("views_per_s"), 4),
                                   sig(e["hipgraph_replay"].get("executed_frac"), 3)]
              for e in full["extra_configs"] if isinstance(e, dict) and isinstance(e.get("hipgraph_replay"), dict) and "ms_per_step" in e["hipgraph_replay"]}
        if rp:                                           # the training steps whose `extra` row is the eager (host-launched) step: the same step as ONE hipGraph
            c["extra_hipgraph_replay"] = rp
    if isinstance(full.get("strong_scaling"), dict):
        c["strong_scaling"] = pick(full["strong_scaling"], ("total_scenes", "scenes_per_gpu", "ms_per_step", "views_per_s", "ranks_ok"))
    if isinstance(full.get("multi_rank"), dict):
        mr = full["multi_rank"]
        c["multi_rank"] = {"error": clip(mr["error"], 160)} if "error" in mr and len(mr) == 1 else {k: multi_row(v) for k, v in mr.items()}
        c["multi_rank_columns"] = ["ms_per_step", "payload_MB", "backend", "ranks_ok"]
    c["full_record"] = FULL_NAME
    return c


def dumps(c):
    """Strict one-line JSON of the compact record, shrunk until it fits MAX_LINE (optional parts go first; the contract keys, roofline and
    cpu_baseline never do)."""
    c = strict(c)
    for drop in (None, "extra_hipgraph_replay", "extra_in_flight", "hbm_kernels_frac", "repeats", "extra_columns", "multi_rank_columns", "errors", "extra", "strong_scaling", "multi_rank"):
        if drop is not None:
            c.pop(drop, None)
        line = json.dumps(c, allow_nan=False, separators=(",", ":"))
        if len(line) <= MAX_LINE:
            return line
    if isinstance(c.get("config"), dict):
        c["config"] = {"workload": clip(c["config"].get("workload"), 200)}
    for k in ("roofline", "cpu_baseline"):
        if isinstance(c.get(k), dict):
            c[k] = {kk: (clip(vv, 60) if isinstance(vv, str) else vv) for kk, vv in c[k].items()}
    line = json.dumps(c, allow_nan=False, separators=(",", ":"))
    if len(line) > MAX_LINE:                              # cannot happen with the key lists above; never print an over-long line
        line = json.dumps({k: c.get(k) for k in CONTRACT}, allow_nan=False, separators=(",", ":"))
    return line


def write_full(full, path=None):
    """The full record as strict indented JSON: to `path`, or to bench_full.json next to bench.py and under gpurun_out/ when present.
    Best effort: a read-only tree must not lose the line."""
    text = json.dumps(strict(full), allow_nan=False, indent=1)
    written = []
    targets = [path] if path else [os.path.join(d, FULL_NAME) for d in (ROOT, os.path.join(ROOT, "gpurun_out")) if os.path.isdir(d)]
    for p in targets:
        try:
            with open(p, "w") as f:
                f.write(text)
            written.append(p)
        except OSError as e:
            print("bench.py: could not write %s: %r" % (p, e), file=sys.stderr)
    return written


def emit(full, path=None):
    """Write the full record (`path`, default bench_full.json beside bench.py), then print the compact line as the last line of stdout."""
    written = write_full(full, path)
    c = compact(full)
    if written:
        c["full_record"] = os.path.basename(written[0]) if not path else path
    line = dumps(c)
    sys.stdout.flush()
    print(line, flush=True)
    return line
