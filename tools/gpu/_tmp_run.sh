python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -k "train or fuse or groups or config3" 2>&1 | tail -6
bash tools/gpu/run_trainprof_r4.sh r04_c_train_b4
