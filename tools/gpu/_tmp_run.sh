python -m pytest tests/test_gpu_bnorm.py -x -q 2>&1 | tail -4
python -m pytest tests/test_gpu_parity.py -x -q -k "train or get_feat3D or refine" 2>&1 | tail -4
bash tools/gpu/run_trainprof_r4.sh r04_b_train_b4
