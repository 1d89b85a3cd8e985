cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5
FORGE_TEST_REPORT=1 timeout 1500 python -m pytest tests -m gpu -q -s -x -k "conv3d_rows_strided or pose_estimators_hip or joint_training_step or joint_mode or joint_finetune or forge_joint_forward" > gpurun_out/r5/pose_tests.log 2>&1
grep -a "hip/f64\|passed\|failed\|Error\|assert\|error" gpurun_out/r5/pose_tests.log | tail -150
for g in 32 64; do JOINT_GRID=$g JOINT_STEPS=6 timeout 600 python tools/joint_step_probe.py 2>&1 | tail -1; done
