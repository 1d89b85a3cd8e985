set -x
FORGE_RENDER_WAVE=1 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -m gpu -q -k "render_analytic or render_vs_oracle or render_golden or render_edge" 2>&1 | tail -3
bash tools/pmc_render.sh 2>&1 | grep -v "^+"
python -m pytest tests/test_gpu_configs.py -m gpu -q -k "config3 or grouped_mse or frozen" 2>&1 | tail -3
python -m pytest tests/test_gpu_parity.py -m gpu -q -k "train_step or training" 2>&1 | tail -3
python tools/train_step_probe.py 2>&1 | tail -3
