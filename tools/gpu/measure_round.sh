#!/bin/bash
# The evidence set of a round in ONE gpurun call:   bash tools/gpu/measure_round.sh TAG [part ...]
#   TAG    prefix of every output under gpurun_out/ (e.g. r06); what is to be judged gets copied into profiles/ by hand
#   parts  bench   the driver's command (`python bench.py`): compact line -> TAG_bench_line.json, full record -> TAG_bench_full.json, wall time
#          prof    rocprofv3 --kernel-trace --stats of the same command, ONE step in flight (TAG_rocprofv3_kernel_stats.csv + .meta.json: bench.py reads the
#                  newest profiles/r*_rocprofv3_kernel_stats.csv for roofline.frac_rocprof)
#          pmc     tools/pmc_all.sh: FETCH_SIZE / WRITE_SIZE / SQ passes over tools/probe_kernels.py -> TAG_pmc_summary.json (roofline.traffic)
#          train   rocprofv3 kernel statistics + launch table of the training step at 4 and 1 scenes per GPU
#          refine  rocprofv3 kernel statistics of the pose-refinement loop
#          ranks2  `bench.py --gpus 2` on the shared GPU (functional rehearsal of the multi-rank line; collectives on gloo)
#          soak    3000 steps with the last output checked against the oracle (--min-psnr-db 100)
#   default: bench prof pmc
TAG=${1:?usage: measure_round.sh TAG [bench prof pmc train refine ranks2 soak]}
shift
PARTS=${@:-bench prof pmc}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for part in $PARTS; do
case $part in
bench)
  ( time python bench.py --full-record gpurun_out/${TAG}_bench_full.json > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench.err ) 2> gpurun_out/${TAG}_bench.time
  wc -c gpurun_out/${TAG}_bench_line.json; cat gpurun_out/${TAG}_bench_line.json; tail -c 300 gpurun_out/${TAG}_bench.err; cat gpurun_out/${TAG}_bench.time ;;
prof)
  bash tools/gpu/run_inferprof.sh ${TAG}_rocprofv3 > gpurun_out/${TAG}_rocprofv3_summary.txt 2>&1; tail -25 gpurun_out/${TAG}_rocprofv3_summary.txt ;;
pmc)
  bash tools/pmc_all.sh > gpurun_out/${TAG}_pmc_all.log 2>&1; cp gpurun_out/pmc_summary.json gpurun_out/${TAG}_pmc_summary.json; tail -5 gpurun_out/${TAG}_pmc_all.log ;;
train)
  TRAIN_SCENES=4 bash tools/gpu/run_trainprof.sh ${TAG}_train_b4 > /dev/null 2>&1
  TRAIN_SCENES=1 bash tools/gpu/run_trainprof.sh ${TAG}_train_b1 > /dev/null 2>&1
  python tools/train_launch_table.py > gpurun_out/${TAG}_train_launch_table_b4.txt 2>&1; head -4 gpurun_out/${TAG}_train_launch_table_b4.txt ;;
refine)
  bash tools/gpu/run_refineprof.sh > gpurun_out/${TAG}_refine_kernel_stats.txt 2>&1; head -12 gpurun_out/${TAG}_refine_kernel_stats.txt ;;
ranks2)
  FORGE_BENCH_ALLOW_SHARED_GPUS=1 timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 --repeats 2 --no-microbench --full-record gpurun_out/${TAG}_bench_2ranks_shared_gpu_full.json \
      2> /dev/null | tail -1 > gpurun_out/${TAG}_bench_2ranks_shared_gpu.json; cat gpurun_out/${TAG}_bench_2ranks_shared_gpu.json ;;
soak)
  python bench.py --steps 3000 --warmup 10 --repeats 1 --no-cpu-baseline --no-extra --no-microbench --min-psnr-db 100 --full-record gpurun_out/${TAG}_soak_full.json \
      > gpurun_out/${TAG}_soak.json 2> gpurun_out/${TAG}_soak.err; echo "soak exit code $?" | tee gpurun_out/${TAG}_soak.txt; cat gpurun_out/${TAG}_soak.json ;;
*) echo "unknown part $part" ;;
esac
done
ls -la gpurun_out/ | grep " ${TAG}_" | head -40
