python -m pytest tests/test_gpu_parity.py -x -q -k "refine or pose" 2>&1 | tail -3
python tools/refine_probe.py 2>&1 | tail -2
