#!/bin/bash
# round 5, item 1: the joint-step golden test, rocprofv3 kernel traces of the joint step at both grids, the bench line with the joint entries
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
FORGE_TEST_REPORT=1 timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -s -k "joint_training_step" > gpurun_out/r5/joint_test.log 2>&1
tail -40 gpurun_out/r5/joint_test.log
cd /tmp && export TMPDIR=/tmp
for g in 32 64; do
  JOINT_GRID=$g JOINT_STEPS=4 timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r5/prof_joint$g -o j --output-format csv -- python $GRAFT_REPO_ROOT/tools/joint_step_probe.py > $GRAFT_REPO_ROOT/gpurun_out/r5/joint_probe$g.log 2>&1
  tail -1 $GRAFT_REPO_ROOT/gpurun_out/r5/joint_probe$g.log
done
cd $GRAFT_REPO_ROOT
for g in 32 64; do
  f=$(find gpurun_out/r5/prof_joint$g -name "*kernel_stats.csv" | head -1)
  n=joint_step; [ $g = 64 ] && n=joint_step_grid64
  python tools/joint_kernel_share.py $f $n 6 gpurun_out/r5/r05_joint_grid${g}_kernel_share.json gpurun_out/r5/r05_joint_grid${g}_kernel_share.txt | head -12
  cp $f gpurun_out/r5/r05_joint_grid${g}_kernel_stats.csv
done
find gpurun_out/r5 -name "*kernel_trace.csv" -delete
find gpurun_out/r5 -size +3M -delete
timeout 1200 python bench.py --steps 20 --repeats 3 --no-cpu-baseline > gpurun_out/r5/bench_joint.json 2> gpurun_out/r5/bench_joint.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5/bench_joint.json"))
print("value", d["value"], "ms", d["ms_per_step"])
for e in d.get("extra_configs", []):
    print(e.get("name"), e.get("ms_per_step"), (e.get("roofline") or {}).get("executed_frac"), e.get("error"), (e.get("stock_torch") or {}).get("pose_nets_fwd_bwd_ms"), (e.get("stock_torch") or {}).get("gflop"))
PY
tail -5 gpurun_out/r5/bench_joint.err
