# round 5, final evidence refresh on the final code: full GPU suite, default bench line, the 2-rank shared-GPU line with the multi_rank sub-records
cd $GRAFT_REPO_ROOT
timeout 1700 python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/r05_gpu_tests.txt; cat gpurun_out/r05_gpu_tests.txt
python bench.py > gpurun_out/r05_bench_b1.json 2> gpurun_out/r05_bench_b1.err
FORGE_BENCH_ALLOW_SHARED_GPUS=1 timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 --repeats 2 --no-microbench 2> /dev/null | grep -a "^{" > gpurun_out/r05_bench_2ranks_shared_gpu.json
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r05_bench_b1.json") if l.startswith("{")][0])
print("value %.1f ms %.3f single %.2f psnr %.2f cpu %.2f" % (d["value"], d["ms_per_step"], d["single_stream"]["ms_per_step"], d["psnr_vs_oracle_db"], d["cpu_baseline"]["value"]))
r = d["roofline"]; print("frac %.3f frac_rocprof %s frac_of_ceiling %.3f executed_frac %.3f traffic %s" % (r["frac"], r["frac_rocprof"], r["frac_of_ceiling"], r["executed_frac"], r["traffic"]))
for e in d["extra_configs"]:
    print(" ", e.get("name"), round(e.get("ms_per_step", 0), 2), round((e.get("roofline") or {}).get("executed_frac") or 0, 3), {k: round(v["ms_per_step"], 2) for k, v in e.items() if isinstance(v, dict) and "ms_per_step" in v}, e.get("error"))
m = json.load(open("gpurun_out/r05_bench_2ranks_shared_gpu.json"))["multi_rank"]
for k, v in m.items():
    print(" ", k, v.get("ranks_ok"), v.get("errors"), round(v.get("ms_per_step", 0), 1), v.get("ms_per_step_no_sync"), v.get("unsharded_ms_per_step"))
PY
