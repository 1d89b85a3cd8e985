cd $GRAFT_REPO_ROOT
for g in 32 64; do JOINT_GRID=$g JOINT_STEPS=6 timeout 600 python tools/joint_step_probe.py 2>&1 | tail -1; done
TRAIN_SCENES=4 TRAIN_STEPS=5 timeout 600 python tools/train_step_probe.py 2>&1 | tail -1
TRAIN_SCENES=1 TRAIN_STEPS=8 timeout 600 python tools/train_step_probe.py 2>&1 | tail -1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "train_step_harness or graphed_training" 2>&1 | tail -2
