python -m pytest tests/test_gpu_parity.py tests/test_gpu_ddp.py tests/test_gpu_configs.py -x -q -k "train or ddp_two or graphed or config3 or adjoint" 2>&1 | tail -3
bash tools/gpu/run_trainprof_r4.sh r04_l_train_b4
