python -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py tests/test_gpu_winograd.py -x -q -k "train or wgrad or conv or heads or fuse or adjoint" 2>&1 | tail -4
python tools/debug/train_grad_margins.py 2>&1 | tail -2
bash tools/gpu/run_trainprof_r4.sh r04_g_train_b4
