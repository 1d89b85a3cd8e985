cd $GRAFT_REPO_ROOT
FORGE_TEST_REPORT=1 timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -s -k "inference_schedule" 2>&1 | grep -v "^$" | tail -8
TRAIN_SCENES=1 TRAIN_MODE=infer python tools/train_launch_table.py > gpurun_out/r05_infer_launch_table_b1.txt 2>&1
TRAIN_SCENES=1 TRAIN_MODE=infer_pose3d python tools/train_launch_table.py > gpurun_out/r05_infer_pose3d_launch_table_b1.txt 2>&1
head -50 gpurun_out/r05_infer_launch_table_b1.txt
