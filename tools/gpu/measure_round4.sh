# the round-4 evidence set in one call (all outputs under gpurun_out/r04_*; copy what is to be judged into profiles/)
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/r04_bench_b1.json 2> gpurun_out/r04_bench_b1.err
tail -c 300 gpurun_out/r04_bench_b1.err
bash tools/gpu/run_inferprof_r4.sh r04_rocprofv3 > gpurun_out/r04_rocprofv3_summary.txt 2>&1
TRAIN_SCENES=4 bash tools/gpu/run_trainprof_r4.sh r04_train_b4 > /dev/null 2>&1
TRAIN_SCENES=1 bash tools/gpu/run_trainprof_r4.sh r04_train_b1 > /dev/null 2>&1
python tools/train_launch_table.py > gpurun_out/r04_train_launch_table_b4.txt 2>&1
bash tools/gpu/run_refineprof.sh > gpurun_out/r04_refine_kernel_stats.txt 2>&1
bash tools/gpu/run_render_bwd_probe.sh > gpurun_out/r04_render_bwd_probe.txt 2>&1
python tools/debug/train_grad_margins.py > gpurun_out/r04_train_grad_margins.txt 2>&1
python tools/cpu_quota_probe.py > gpurun_out/r04_cpu_quota_probe.txt 2>&1
bash tools/pmc_all.sh > gpurun_out/r04_pmc_all.log 2>&1
cp gpurun_out/pmc_summary.json gpurun_out/r04_pmc_summary.json
ls -la gpurun_out/ | grep r04_ | head -40
