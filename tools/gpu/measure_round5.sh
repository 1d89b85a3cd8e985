# the round-5 evidence set in one call (outputs under gpurun_out/r05_*; what is to be judged gets copied into profiles/)
cd $GRAFT_REPO_ROOT
( time python bench.py > gpurun_out/r05_bench_b1.json 2> gpurun_out/r05_bench_b1.err ) 2> gpurun_out/r05_bench_b1.time
tail -c 300 gpurun_out/r05_bench_b1.err; cat gpurun_out/r05_bench_b1.time
bash tools/gpu/run_inferprof_r4.sh r05_rocprofv3 > gpurun_out/r05_rocprofv3_summary.txt 2>&1
TRAIN_SCENES=4 bash tools/gpu/run_trainprof_r4.sh r05_train_b4 > /dev/null 2>&1
TRAIN_SCENES=1 bash tools/gpu/run_trainprof_r4.sh r05_train_b1 > /dev/null 2>&1
TRAIN_MODE=joint python tools/train_launch_table.py > gpurun_out/r05_joint_launch_table.txt 2>&1
python tools/train_launch_table.py > gpurun_out/r05_train_launch_table_b4.txt 2>&1
bash tools/gpu/run_refineprof.sh > gpurun_out/r05_refine_kernel_stats.txt 2>&1
python tools/debug/wgrad_traffic_sensitivity.py > gpurun_out/r05_wgrad_traffic_sensitivity.txt 2>&1
python tools/debug/bf16_split_accuracy.py > gpurun_out/r05_bf16_split_accuracy.txt 2>&1
bash tools/pmc_all.sh > gpurun_out/r05_pmc_all.log 2>&1
cp gpurun_out/pmc_summary.json gpurun_out/r05_pmc_summary.json
python bench.py --steps 3000 --warmup 10 --repeats 1 --no-cpu-baseline --no-extra --no-microbench --min-psnr-db 100 > gpurun_out/r05_soak.json 2> gpurun_out/r05_soak.err; echo "soak exit code $?" > gpurun_out/r05_soak.txt
python - <<'PY' >> gpurun_out/r05_soak.txt
import json
d = json.loads([l for l in open("gpurun_out/r05_soak.json") if l.startswith("{")][0])
print("soak %d steps: %.1f views/s, psnr_vs_oracle %.2f dB (asserted >= 100), max-abs %.2e" % (d["steps"], d["value"], d["psnr_vs_oracle_db"], d["max_abs_err_vs_oracle"]))
PY
cat gpurun_out/r05_soak.txt
ls -la gpurun_out/ | grep r05_ | head -40
