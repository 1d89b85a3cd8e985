cd /root/repo
python -m pytest tests -q -m gpu 2>&1 | tail -4
python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['stages_ms']); print({k:(round(v.get('ms',v.get('ms_total',0)),4), round(v['achieved'],1)) for k,v in d['kernels'].items()}); print(d['roofline']['instantiations'].keys(), d['psnr_vs_oracle_db'], d['max_abs_err_vs_oracle'])"
