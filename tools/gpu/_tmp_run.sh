python -m pytest tests/test_gpu_parity.py -x -q -k "train or get_feat3D or trunk or conv2d" 2>&1 | tail -4
bash tools/gpu/run_trainprof_r4.sh r04_e_train_b4
