cd /root/repo
python -m pytest tests -q -m gpu -x 2>&1 | tail -8
python tools/train_step_probe.py 2>&1 | tail -2
TRAIN_SCENES=4 python tools/train_step_probe.py 2>&1 | tail -2
TRAIN_GRID=64 python tools/train_step_probe.py 2>&1 | tail -2
python tools/refine_probe.py 2>&1 | tail -4
FORGE_WINOGRAD=0 python tools/refine_probe.py 2>&1 | tail -2
