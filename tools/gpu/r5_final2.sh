# round 5, evidence refresh after r5-17..r5-19 (inference schedule of the pose estimators, multi-tensor clip, attention kernel): joint-step / joint-inference
# kernel shares and op profile, training-step kernel statistics, then the full GPU suite, the default bench line (which reads the shares) and the 2-rank line
cd $GRAFT_REPO_ROOT
bash tools/gpu/r5_jointprof.sh 2>&1 | tail -4
bash tools/gpu/r5_joint4.sh > gpurun_out/r5_joint4.log 2>&1
bash tools/gpu/r5_inferjoint.sh > gpurun_out/r5_inferjoint.log 2>&1; tail -3 gpurun_out/r5_inferjoint.log
python tools/joint_op_profile.py > gpurun_out/r05_joint_op_profile.txt 2>&1
TRAIN_SCENES=4 bash tools/gpu/run_trainprof_r4.sh r05_train_b4 > /dev/null 2>&1
TRAIN_SCENES=1 bash tools/gpu/run_trainprof_r4.sh r05_train_b1 > /dev/null 2>&1
rm -rf gpurun_out/prof_train
cp gpurun_out/r5/r05_joint_grid*_kernel_share.json gpurun_out/r5/r05_joint_grid*_kernel_share.txt gpurun_out/r05_joint_4scenes_kernel_share.* gpurun_out/r05_joint_inference_kernel_share.* profiles/
bash tools/gpu/r5_final.sh
