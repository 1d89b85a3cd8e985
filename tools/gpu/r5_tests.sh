cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5
FORGE_TEST_REPORT=1 timeout 1500 python -m pytest tests -m gpu -q -s -x > gpurun_out/r5/gpu_tests_full.log 2>&1
grep -a "forward \|ratio\|passed\|failed\|Error\|assert" gpurun_out/r5/gpu_tests_full.log | grep -av "^  joint grad" | tail -120
