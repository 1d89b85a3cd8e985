cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5
timeout 2400 python -m pytest tests/test_gpu_ddp.py -m gpu -q -x > gpurun_out/r5/ddp_tests.log 2>&1
tail -30 gpurun_out/r5/ddp_tests.log
FORGE_BENCH_ALLOW_SHARED_GPUS=1 timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 --repeats 2 --no-microbench > gpurun_out/r5/bench_2ranks_shared.json 2> gpurun_out/r5/bench_2ranks_shared.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5/bench_2ranks_shared.json"))
print(json.dumps(d.get("multi_rank"), indent=1)[:6000])
PY
tail -3 gpurun_out/r5/bench_2ranks_shared.err
