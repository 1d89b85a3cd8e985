set -x
export TMPDIR=/tmp
python bench.py --steps 20 --warmup 5 > gpurun_out/r02_bench_b1.json 2> gpurun_out/r02_bench_b1.err; tail -c 300 gpurun_out/r02_bench_b1.err
python bench.py --steps 10 --warmup 3 --scenes 8 --no-cpu-baseline > gpurun_out/r02_bench_b8.json 2> /dev/null
python bench.py --steps 5 --warmup 2 --grid 64 --no-cpu-baseline > gpurun_out/r02_bench_grid64_b1.json 2> /dev/null
python bench.py --steps 3 --warmup 1 --grid 64 --scenes 4 --no-cpu-baseline > gpurun_out/r02_bench_grid64_b4.json 2> /dev/null
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r02 -o r02 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-microbench > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/step_timeline.py gpurun_out/prof_r02/r02_kernel_trace.csv > gpurun_out/r02_step_timeline_b1.txt 2>&1
cp gpurun_out/prof_r02/r02_kernel_stats.csv gpurun_out/r02_rocprofv3_kernel_stats.csv
find gpurun_out/prof_r02 -size +4M -delete
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r02g -o g64 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --grid 64 --no-cpu-baseline --no-microbench > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
cp gpurun_out/prof_r02g/g64_kernel_stats.csv gpurun_out/r02_rocprofv3_kernel_stats_grid64.csv
find gpurun_out/prof_r02g -size +4M -delete
python - <<'PY'
import json
for f in ("r02_bench_b1","r02_bench_b8","r02_bench_grid64_b1","r02_bench_grid64_b4"):
    d=json.load(open("gpurun_out/%s.json"%f)); print(f, round(d["value"],1), round(d["ms_per_step"],2), round(d["roofline"]["frac"],3), d["stages_ms"])
d=json.load(open("gpurun_out/r02_bench_b1.json")); print(json.dumps(d["cpu_baseline"])[:900]); print(d["kernels"]["rotate_fwd_kernel"], d["kernels"]["render_fwd_kernel"]); print(d.get("psnr_vs_oracle_db"), d.get("psnr_to_target_db"), d.get("speedup_vs_cpu_baseline"))
PY
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
FORGE_BENCH_ALLOW_SHARED_GPUS=1 timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --no-microbench 2> gpurun_out/r02_bench_2ranks_shared_gpu.err | tail -c 600
