cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5
FORGE_TEST_REPORT=1 timeout 1500 python -m pytest tests -m gpu -q -s -x -k "pose_estimators_hip or joint_training_step or joint_mode or joint_finetune or forge_joint_forward or pose3d_predicted" > gpurun_out/r5/pose_tests.log 2>&1
grep -a "passed\|failed\|Error\|assert\|error" gpurun_out/r5/pose_tests.log | tail -20
timeout 1200 python - <<'PY' 2>&1 | grep -v Warn | tail -20
import json, sys, torch
sys.path.insert(0, ".")
import bench
dev = torch.device("cuda:0")
for e in bench.joint_configs(dev, steps=5):
    print(e.get("name"), e.get("ms_per_step"), e.get("error"))
    print("   ", json.dumps({k: e.get(k) for k in ("hipgraph_replay", "pose_networks_on_stock_torch")}))
PY
