cd $GRAFT_REPO_ROOT
for sens in "" g go; do
  export FORGE_SENS=$sens
  echo "== sens=$sens"
  python bench.py --no-extra --no-cpu-baseline --no-microbench --repeats 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline', round(d['value'],1), 'ms', round(d['ms_per_step'],3), 'single', round(d['single_stream']['ms_per_step'],3))"
done
