cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_winograd.py tests/test_gpu_configs.py tests/test_gpu_parity.py -m gpu -q -x -k "frozen or refine or pipelined or conv_rgb" 2>&1 | tail -5
python tools/refine_probe.py 2>&1 | tail -6
