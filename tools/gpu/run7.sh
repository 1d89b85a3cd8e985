set -x
AB_VARIANTS="base:FORGE_CONV_ACC2=0;acc2:FORGE_CONV_ACC2=1" python tools/conv_small_ab.py 2>&1 | grep -v amdgpu.ids
AB_SCENES=1 AB_VARIANTS="D:FORGE_CONV_TILE=D,FORGE_CONV_ACC2=0;Dacc2:FORGE_CONV_TILE=D,FORGE_CONV_ACC2=1" python tools/conv_variants.py 2>&1 | grep scenes
AB_SCENES=4 AB_VARIANTS="D:FORGE_CONV_TILE=D,FORGE_CONV_ACC2=0;Dacc2:FORGE_CONV_TILE=D,FORGE_CONV_ACC2=1" python tools/conv_variants.py 2>&1 | grep scenes
python tools/debug/conv_timing.py 2>&1 | grep "M="
python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -k "conv_igemm or fuse_hip or trunk or config3" 2>&1 | tail -12
FORGE_CONV_ACC2=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-microbench | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('b1 acc2', d['value'], d['roofline']['frac'], d['stages_ms'])"
FORGE_CONV_ACC2=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-microbench | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('b1 base', d['value'], d['roofline']['frac'], d['stages_ms'])"
