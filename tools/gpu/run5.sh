set -x
python tools/debug/conv_timing.py 2>&1 | grep "M="
AB_SCENES=1 AB_VARIANTS="D:FORGE_CONV_TILE=D;C:FORGE_CONV_TILE=C;F:FORGE_CONV_TILE=F;A:FORGE_CONV_TILE=A" python tools/conv_variants.py 2>&1 | grep scenes
AB_SCENES=4 AB_VARIANTS="D:FORGE_CONV_TILE=D;C:FORGE_CONV_TILE=C;F:FORGE_CONV_TILE=F;A:FORGE_CONV_TILE=A" python tools/conv_variants.py 2>&1 | grep scenes
python -m pytest tests -m gpu -q -x -k "config3 or every_tile or gt_pose_5in5out or config4 or rotate" 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-microbench | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('b1', d['value'], d['roofline']['frac'], d['stages_ms'])"
