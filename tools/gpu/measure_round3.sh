# Round-3 evidence set (one MI355X): bench line (4 steps in flight, extra_configs), rocprofv3 kernel statistics of the bench command with one and with
# four steps in flight, one-step timeline, per-stage replay times, PMC summary. Results land under gpurun_out/ and are copied to profiles/ by hand.
set -x
export TMPDIR=/tmp
python bench.py --steps 100 --warmup 5 > gpurun_out/r03_bench_b1.json 2> gpurun_out/r03_bench_b1.err; tail -c 300 gpurun_out/r03_bench_b1.err
python bench.py --steps 10 --warmup 3 --grid 64 --no-cpu-baseline --no-extra > gpurun_out/r03_bench_grid64_b1.json 2> /dev/null
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r03 -o r03 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --pipeline-depth 1 --no-cpu-baseline --no-microbench --no-extra > $GRAFT_REPO_ROOT/gpurun_out/r03_bench_under_rocprof_depth1.json 2> /dev/null
cd $GRAFT_REPO_ROOT
python tools/step_timeline.py gpurun_out/prof_r03/r03_kernel_trace.csv > gpurun_out/r03_step_timeline_b1.txt 2>&1
cp gpurun_out/prof_r03/r03_kernel_stats.csv gpurun_out/r03_rocprofv3_kernel_stats.csv
find gpurun_out/prof_r03 -size +4M -delete
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r03p -o r03p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-microbench --no-extra > $GRAFT_REPO_ROOT/gpurun_out/r03_bench_under_rocprof_depth4.json 2> /dev/null
cd $GRAFT_REPO_ROOT
cp gpurun_out/prof_r03p/r03p_kernel_stats.csv gpurun_out/r03_rocprofv3_kernel_stats_4_in_flight.csv
find gpurun_out/prof_r03p -size +4M -delete
python tools/stage_replay.py 2>&1 | tail -1 > gpurun_out/r03_stage_replay.txt; STAGE_SCENES=8 python tools/stage_replay.py 2>&1 | tail -1 >> gpurun_out/r03_stage_replay.txt; cat gpurun_out/r03_stage_replay.txt
bash tools/pmc_all.sh > gpurun_out/r03_pmc_all.log 2>&1; cp gpurun_out/pmc_summary.json gpurun_out/r03_pmc_summary.json; tail -3 gpurun_out/r03_pmc_all.log
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03_bench_b1.json").read().strip().splitlines()[-1])
print(round(d["value"],1), round(d["ms_per_step"],3), d["single_stream"], d["stages_ms_replay"])
r=d["roofline"]; print({k:r[k] for k in ("frac","executed_frac","floor_ms","step_over_floor","traffic")})
for e in d["extra_configs"]: print(e["name"], round(e.get("ms_per_step",0),2), round(e.get("views_per_s",0),1), e.get("pipelined"), e.get("error"))
print(json.dumps(d["cpu_baseline"])[:700])
PY
for d in 2 3 4 6; do PIPE_DEPTH=$d PIPE_STEPS=120 python tools/pipeline_probe.py 2>&1 | grep "in flight"; done > gpurun_out/r03_pipeline_depth.txt; cat gpurun_out/r03_pipeline_depth.txt
for s in 1 4; do TRAIN_SCENES=$s bash tools/gpu/run_trainprof.sh > gpurun_out/r03_train_step_kernel_stats_b$s.txt 2>&1; head -3 gpurun_out/r03_train_step_kernel_stats_b$s.txt | tail -1; done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
