# round 6, GPU call 2: full GPU suite (all failures listed), counter-name list, bf16x3 proxy, rotate limiter timing, render traffic with / without the band order
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/r06/gpu_tests_2.txt; tail -8 gpurun_out/r06/gpu_tests_2.txt
(cd /tmp && export TMPDIR=/tmp && rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/r06/rocprofv3_counters.txt 2>&1); wc -l gpurun_out/r06/rocprofv3_counters.txt
python tools/debug/bf16x3_throughput_proxy.py > gpurun_out/r06/bf16x3_throughput_proxy.txt 2>&1; cat gpurun_out/r06/bf16x3_throughput_proxy.txt
python tools/rotate_limiter_probe.py > gpurun_out/r06/rotate_limiter_timing.txt 2>&1; cat gpurun_out/r06/rotate_limiter_timing.txt
ROTATE_PROBE_D=64 python tools/rotate_limiter_probe.py >> gpurun_out/r06/rotate_limiter_timing.txt 2>&1; tail -8 gpurun_out/r06/rotate_limiter_timing.txt
bash tools/pmc_render.sh > gpurun_out/r06/pmc_render.txt 2>&1; cat gpurun_out/r06/pmc_render.txt
bash tools/pmc_rotate.sh > gpurun_out/r06/pmc_rotate.txt 2>&1; cat gpurun_out/r06/pmc_rotate.txt
