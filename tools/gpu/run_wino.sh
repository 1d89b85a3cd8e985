cd /root/repo
python -m pytest tests -q -m gpu -x -k "train or autograd or grad or adjoint or fuse_groups_shared or graphed" 2>&1 | tail -4
python tools/train_step_probe.py 2>&1 | tail -1
TRAIN_SCENES=4 python tools/train_step_probe.py 2>&1 | tail -1
TRAIN_GRID=64 python tools/train_step_probe.py 2>&1 | tail -1
