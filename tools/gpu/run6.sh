set -x
python tools/conv_small_ab.py 2>&1 | grep -v amdgpu.ids
AB_SCENES=1 AB_VARIANTS="D:FORGE_CONV_TILE=D,FORGE_CONV_PF2=0;Dpf2:FORGE_CONV_TILE=D,FORGE_CONV_PF2=1;B:FORGE_CONV_TILE=B,FORGE_CONV_PF2=0;Bpf2:FORGE_CONV_TILE=B,FORGE_CONV_PF2=1" python tools/conv_variants.py 2>&1 | grep scenes
FORGE_CONV_PF2=1 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "conv_igemm or fuse_hip or trunk" 2>&1 | tail -3
FORGE_CONV_PF2=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-microbench | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('b1 pf2', d['value'], d['roofline']['frac'], d['stages_ms'])"
FORGE_CONV_PF2=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-microbench | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('b1 base', d['value'], d['roofline']['frac'], d['stages_ms'])"
bash tools/pmc_render.sh 2>&1 | grep -v "^+" | tail -24
