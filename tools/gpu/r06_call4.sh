cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
python -u tools/debug/bn_eval_check.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06/bn_eval_check.txt; cat gpurun_out/r06/bn_eval_check.txt
FORGE_TEST_REPORT=1 python -m pytest tests/test_gpu_bnorm.py tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_gpu_properties.py -m gpu -q --tb=short -s -k "residual_form or remaining_training_stages or hip_convolutions_vs_float64 or rotate" 2>&1 | grep -v "^W2026\|amdgpu.ids" | cut -c1-300 | tail -150 > gpurun_out/r06/gpu_tests_4.txt; tail -100 gpurun_out/r06/gpu_tests_4.txt
echo "--- persistent (product)"; python tools/rotate_limiter_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06/rotate_persistent_timing.txt
echo "--- one-shot grid (rounds 1-5)"; FORGE_AMD_LIB=$GRAFT_REPO_ROOT/tools/debug/libforge_hip_rotate_oneshot.so python tools/rotate_limiter_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06/rotate_oneshot_timing.txt
python tools/rotate_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06/rotate_probe_persistent.txt
FORGE_AMD_LIB=$GRAFT_REPO_ROOT/tools/debug/libforge_hip_rotate_oneshot.so python tools/rotate_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06/rotate_probe_oneshot.txt
python -u tools/debug/bf16x3_throughput_proxy.py > gpurun_out/r06/bf16x3_throughput_proxy.txt 2>&1; cat gpurun_out/r06/bf16x3_throughput_proxy.txt
