# round 5, extra evidence: L2->fabric traffic of one 4-scene training step on the final code (the r04 table's successor) and one more default bench line (box-to-box spread)
cd $GRAFT_REPO_ROOT
bash tools/gpu/run_train_pmc_traffic.sh > gpurun_out/r05_train_pmc_traffic_b4.txt 2>&1
head -12 gpurun_out/r05_train_pmc_traffic_b4.txt
python bench.py > gpurun_out/r05_bench_b1_second_box.json 2> /dev/null
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r05_bench_b1_second_box.json") if l.startswith("{")][0])
print("second box: value %.1f, single stream %.2f ms" % (d["value"], d["single_stream"]["ms_per_step"]))
for e in d["extra_configs"]:
    print(" ", e.get("name"), round(e.get("ms_per_step", 0), 2), (e.get("hipgraph_replay") or {}).get("ms_per_step"))
PY
