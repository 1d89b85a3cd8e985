python -m pytest tests/test_gpu_bnorm.py -x -q 2>&1 | tail -3
python -m pytest tests/test_gpu_parity.py tests/test_gpu_ddp.py -x -q -k "train or get_feat3D or trunk or ddp_two" 2>&1 | tail -3
bash tools/gpu/run_trainprof_r4.sh r04_k_train_b4
TRAIN_SCENES=1 bash tools/gpu/run_trainprof_r4.sh r04_k_train_b1
