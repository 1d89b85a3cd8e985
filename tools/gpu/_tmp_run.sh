cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_bnorm.py -m gpu -q -x 2>&1 | tail -2
TRAIN_SCENES=4 bash tools/gpu/run_trainprof_r4.sh r04_train_b4 > /dev/null 2>&1
grep -E "finalize|lines16|small_kernel|per step" gpurun_out/r04_train_b4_kernel_stats.txt
tail -1 gpurun_out/r04_train_b4.log
