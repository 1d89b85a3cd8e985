set -x
export TMPDIR=/tmp
python -m pytest tests/test_gpu_configs.py -m gpu -q -k "config3" 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_b1.json 2> gpurun_out/bench_b1.err; tail -c 600 gpurun_out/bench_b1.err
python bench.py --steps 10 --warmup 3 --scenes 8 --no-cpu-baseline > gpurun_out/bench_b8.json 2> gpurun_out/bench_b8.err; tail -c 300 gpurun_out/bench_b8.err
python bench.py --steps 5 --warmup 2 --grid 64 --no-cpu-baseline > gpurun_out/bench_g64.json 2> gpurun_out/bench_g64.err; tail -c 600 gpurun_out/bench_g64.err
python bench.py --steps 3 --warmup 1 --grid 64 --scenes 4 --no-cpu-baseline --no-microbench > gpurun_out/bench_g64_b4.json 2> gpurun_out/bench_g64_b4.err; tail -c 600 gpurun_out/bench_g64_b4.err
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_b1 -o b1 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-microbench > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
ls gpurun_out/prof_b1 | head
python tools/step_timeline.py $(ls gpurun_out/prof_b1/*kernel_trace.csv | head -1) > gpurun_out/timeline_b1.txt 2>&1
tail -3 gpurun_out/timeline_b1.txt
# keep only the small files
find gpurun_out/prof_b1 -size +8M -delete
