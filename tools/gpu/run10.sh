set -x
python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -m gpu -q -k "render_ab or fuse_groups or forward_pose3d or pose3d_predicted or graphed or chunking" 2>&1 | tail -4
python tools/pose3d_probe.py 2>&1 | grep -v amdgpu | tail -4
POSE3D_SCENES=4 python tools/pose3d_probe.py 2>&1 | grep -v amdgpu | tail -3
TRAIN_GRID=64 TRAIN_SCENES=4 TRAIN_STEPS=2 python tools/train_step_probe.py 2>&1 | tail -2
bash tools/pmc_all.sh 2>&1 | grep -v "^+" | tail -60
