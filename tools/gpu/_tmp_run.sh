python -m pytest tests/test_gpu_properties.py tests/test_gpu_parity.py tests/test_gpu_ddp.py -x -q -k "render or refine or pose or ray_sharded" 2>&1 | tail -3
bash tools/gpu/run_render_bwd_probe.sh 2>&1 | grep -E "grid|backward" | head -20
