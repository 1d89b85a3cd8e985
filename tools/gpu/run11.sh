set -x
python -m pytest tests -m gpu -q 2>&1 | tail -6
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-microbench | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('b1', d['value'], d['ms_per_step'], d['roofline']['frac'], d['stages_ms'])"
