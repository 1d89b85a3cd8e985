cd $GRAFT_REPO_ROOT
FORGE_TEST_REPORT=1 timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -s -k "pose_estimators or joint" 2>&1 | tail -25
bash tools/gpu/r5_inferjoint.sh 2>&1 | tail -60
