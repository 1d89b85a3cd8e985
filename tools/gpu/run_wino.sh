cd /root/repo
python -m pytest tests/test_gpu_winograd.py tests/test_gpu_configs.py tests/test_gpu_parity.py -q -x 2>&1 | tail -8
python tools/pose3d_probe.py 2>&1 | tail -4
FORGE_WINOGRAD=0 python tools/pose3d_probe.py 2>&1 | tail -4
