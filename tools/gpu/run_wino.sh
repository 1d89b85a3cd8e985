cd /root/repo
python -m pytest tests/test_gpu_parity.py tests/test_gpu_winograd.py -q -x -k "trunk or encoder or wino or feat3D or forward" 2>&1 | tail -4
for t in 9999 256 128 64; do
echo "FORGE_TRUNK_WINO_MIN=$t"
FORGE_TRUNK_WINO_MIN=$t python bench.py --no-cpu-baseline --no-microbench 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(' b1', round(d['value'],1), round(d['ms_per_step'],3), d['stages_ms']['encoder_resnet'])"
FORGE_TRUNK_WINO_MIN=$t python bench.py --no-cpu-baseline --no-microbench --scenes 8 --steps 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(' b8', round(d['value'],1), round(d['ms_per_step'],3), d['stages_ms']['encoder_resnet'])"
done
