set -x
python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -m gpu -q -k "frozen or config3 or refinement or pose_refinement" 2>&1 | tail -4
for s in 1 4; do AB_SCENES=$s python tools/conv_variants.py 2>&1 | grep scenes; done
FORGE_CONV_PRIO=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-microbench | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('b1 prio', d['value'], d['roofline']['frac'], d['stages_ms'])"
FORGE_CONV_PRIO=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-microbench | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('b1 base', d['value'], d['roofline']['frac'], d['stages_ms'])"
FORGE_CONV_PRIO=1 python bench.py --steps 10 --warmup 3 --scenes 8 --no-cpu-baseline --no-microbench | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('b8 prio', d['value'], d['roofline']['frac'])"
python tools/refine_probe.py 2>&1 | tail -2
