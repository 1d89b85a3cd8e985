cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_bnorm.py tests/test_gpu_configs.py tests/test_gpu_parity.py -m gpu -q --tb=short -k "residual_form or remaining_training_stages or hip_convolutions_vs_float64 or rotate_vs_oracle" 2>&1 | grep -v "^W2026\|amdgpu.ids" | cut -c1-400 | tail -150 > gpurun_out/r06/gpu_tests_3.txt; tail -120 gpurun_out/r06/gpu_tests_3.txt
python -u tools/debug/bf16x3_throughput_proxy.py > gpurun_out/r06/bf16x3_throughput_proxy.txt 2>&1; cat gpurun_out/r06/bf16x3_throughput_proxy.txt
