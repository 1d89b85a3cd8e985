set -x
export TMPDIR=/tmp
python -m pytest tests -m gpu -q 2>&1 | tail -8
for ns in 1 2; do
  FORGE_TRUNK_STREAMS=$ns python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-microbench > gpurun_out/bench_e_ts$ns.json 2> gpurun_out/bench_e_ts$ns.err
  python - <<PY
import json
d=json.load(open("gpurun_out/bench_e_ts$ns.json")); print("streams $ns", round(d["value"],1), round(d["ms_per_step"],3), d["roofline"]["frac"], d["stages_ms"])
PY
done
python bench.py --steps 10 --warmup 3 --scenes 8 --no-cpu-baseline --no-microbench | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('b8', d['value'], d['roofline']['frac'], d['stages_ms'])"
python tools/refine_probe.py 2>&1 | tail -5
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_e -o e --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-microbench > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/step_timeline.py gpurun_out/prof_e/e_kernel_trace.csv > gpurun_out/timeline_e.txt 2>&1
tail -2 gpurun_out/timeline_e.txt
find gpurun_out/prof_e -size +8M -delete
