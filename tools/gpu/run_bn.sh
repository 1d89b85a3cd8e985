cd /root/repo
python -m pytest tests/test_gpu_bnorm.py -q -x 2>&1 | tail -2
python tools/train_step_probe.py 2>&1 | tail -1
TRAIN_SCENES=4 python tools/train_step_probe.py 2>&1 | tail -1
TRAIN_GRID=64 python tools/train_step_probe.py 2>&1 | tail -1
bash tools/gpu/run_trainprof.sh 2>&1 | tail -32
