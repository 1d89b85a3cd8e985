cd /root/repo
bash tools/gpu/measure_round.sh > gpurun_out/measure_round.log 2>&1
bash tools/pmc_all.sh > gpurun_out/pmc_all.log 2>&1
tail -5 gpurun_out/pmc_all.log
