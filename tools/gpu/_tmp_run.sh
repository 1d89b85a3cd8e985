cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x -k "heads or narrow or conv_rgb or training_grad or lines" 2>&1 | tail -2
TRAIN_SCENES=4 bash tools/gpu/run_trainprof_r4.sh r04_train_b4 > /dev/null 2>&1
grep -E "lines16_kernel|per step" gpurun_out/r04_train_b4_kernel_stats.txt; grep "train step" gpurun_out/r04_train_b4.log
