set -x
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x -k "config3 or trunk or forward_pose3d_golden or gt_pose_5in5out or graphed_forward or batch8" 2>&1 | tail -4
for ns in 1 2 3 5; do
  FORGE_TRUNK_STREAMS=$ns python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-microbench > gpurun_out/bench_ts$ns.json 2> gpurun_out/bench_ts$ns.err
  python - <<PY
import json
d=json.load(open("gpurun_out/bench_ts$ns.json")); print("streams $ns", round(d["value"],1), round(d["ms_per_step"],3), d["stages_ms"])
PY
done
FORGE_TRUNK_STREAMS=5 python bench.py --steps 10 --warmup 3 --scenes 2 --no-cpu-baseline --no-microbench | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('b2 s5', d['value'], d['stages_ms'])"
FORGE_TRUNK_STREAMS=1 python bench.py --steps 10 --warmup 3 --scenes 2 --no-cpu-baseline --no-microbench | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('b2 s1', d['value'], d['stages_ms'])"
python bench.py --steps 5 --warmup 2 --no-microbench 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps(d['cpu_baseline'])[:1500])"
