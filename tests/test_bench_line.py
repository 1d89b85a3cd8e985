"""bench.py's stdout contract: ONE compact strict-JSON line under 4 KB as the last line of stdout, the full record in a
side file. Round 5's driver record had `parsed: null` (a 22 KB line carrying the token `Infinity`)."""
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from benchkit import emit  # noqa: E402


def _refuse(tok):
    raise AssertionError("non-strict JSON token %s" % tok)


def _full_record():
    """The shape of a real N = 1 record: contract keys, a roofline with a long note and per-instantiation table, a cpu_baseline with sweeps,
    11 extra configurations of ~2 KB each, non-finite floats where a division had nothing to divide."""
    long = "x" * 3000
    extra = [{"name": "cfg%d" % i, "workload": long, "steps": 10, "ms_per_step": 10.0 + i, "views_per_s": 1e3 / (10.0 + i),
              "roofline": {"executed_gflop": 1e3, "floor_ms": 5.0, "executed_frac": 0.5, "step_over_floor": float("inf"), "launches": {"a": 1}}} for i in range(11)]
    extra.append({"name": "broken", "error": "RuntimeError('" + long + "')"})
    return {"metric": "rendered views/sec (5 views, 128^2 px, 64^3 voxel)", "value": 767.5, "unit": "views/s", "n_gpus": 1, "steps": 100, "warmup": 5,
            "ms_per_step": 6.51, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "ranks_ok": 1,
            "errors": [], "process_group": {"backend": None, "world_size": 1, "initialized": False, "note": long},
            "config": {"workload": long, "scenes_per_gpu": 1, "views_in": 5, "views_out": 5, "rank0_affinity": {"pinned": False, "reason": long}},
            "roofline": {"kernel": "conv_igemm_kernel<BM, BN, waves>", "bound": "mfma", "achieved": 103.7, "peak": 157.3, "unit": "TFLOP/s", "frac": 0.659,
                         "frac_rocprof": float("nan"), "rocprof_source": None, "traffic": 2.9e8, "traffic_algorithmic_bytes": 2.8e8, "traffic_source": "profiles/x.json",
                         "executed_gflop": 682.0, "floor_ms": 4.34, "executed_frac": 0.66, "note": long, "instantiations": {"a<%d>" % i: {"frac": 0.5, "note": long} for i in range(4)}},
            "cpu_baseline": {"value": 7.8, "unit": "views/s", "cores": 16, "cpu_model": "AMD EPYC", "kind": "port", "sample": long, "thread_sweep": {str(i): {"s": 1.0} for i in range(40)}},
            "kernels": {"rotate_fwd_kernel": {"bound": "hbm", "frac": 0.48, "note": long}, "conv x": {"bound": "mfma", "frac": 0.7}},
            "psnr_vs_oracle_db": float("inf"), "max_abs_err_vs_oracle": 4.5e-5, "psnr_to_target_db": {"build": 9.0, "oracle": 9.0, "abs_diff": 0.0},
            "extra_configs": extra, "speedup_vs_cpu_baseline": 98.0, "single_stream": {"ms_per_step": 7.6}}


def test_compact_line_is_short_strict_and_carries_the_contract():
    full = _full_record()
    line = emit.dumps(emit.compact(full))
    assert len(line) < 4096 and "\n" not in line
    d = json.loads(line, parse_constant=_refuse)
    for k in emit.CONTRACT + ("config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["value"] == 767.5 and d["n_gpus"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["psnr_vs_oracle_db"] is None                         # +inf (bit-identical images) is not a JSON number
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["frac"] == 0.659 and r["peak"] == 157.3 and r["frac_rocprof"] is None and r["traffic"] == 2.9e8
    assert "note" not in r and "instantiations" not in r
    c = d["cpu_baseline"]
    assert c["value"] == 7.8 and c["cores"] == 16 and c["kind"] == "port" and len(c["sample"]) <= 200
    assert d["extra"]["cfg3"] == [13.0, round(1e3 / 13.0, 2), 0.5] and isinstance(d["extra"]["broken"], str)
    assert len(d["config"]["workload"]) <= 420


def test_full_record_is_strict_json_and_complete(tmp_path, capsys):
    full = _full_record()
    p = str(tmp_path / "full.json")
    line = emit.emit(full, p)
    out = capsys.readouterr().out.splitlines()
    assert out[-1] == line and json.loads(line)["full_record"] == p
    text = open(p).read()
    back = json.loads(text, parse_constant=_refuse)
    assert back["extra_configs"][0]["roofline"]["step_over_floor"] is None       # inf -> null, nothing else lost
    assert len(back["extra_configs"]) == 12 and len(back["cpu_baseline"]["thread_sweep"]) == 40 and len(back["roofline"]["note"]) == 3000


def test_strict_and_sig_helpers():
    assert emit.strict({"a": (1, float("nan")), 2: float("-inf")}) == {"a": [1, None], "2": None}
    assert emit.sig(123456.789) == 123460.0 and emit.sig(0.00012345678) == 0.00012346 and emit.sig(float("nan")) is None and emit.sig(7) == 7
    assert math.isclose(emit.sig(6.5143219), 6.5143)


def test_oversized_optional_parts_are_dropped_not_the_contract():
    full = _full_record()
    full["extra_configs"] = [{"name": "n" * 60 + str(i), "ms_per_step": 1.0, "views_per_s": 1.0} for i in range(80)]
    line = emit.dumps(emit.compact(full))
    d = json.loads(line, parse_constant=_refuse)
    assert len(line) < 4096 and "extra" not in d and d["roofline"]["frac"] == 0.659 and d["cpu_baseline"]["value"] == 7.8
