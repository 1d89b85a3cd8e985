#!/bin/bash
# rocprofv3 kernel statistics of the joint step at both grids -> per-kernel-name share of stock-torch vs libforge kernels
cd /tmp && export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r5
for g in 32 64 32stock; do
  export JOINT_STOCK=0; gg=$g
  if [ $g = 32stock ]; then export JOINT_STOCK=1; gg=32; fi
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/r5/prof_joint$g
  JOINT_GRID=$gg JOINT_STEPS=4 timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r5/prof_joint$g -o j --output-format csv -- python $GRAFT_REPO_ROOT/tools/joint_step_probe.py > $GRAFT_REPO_ROOT/gpurun_out/r5/joint_probe$g.log 2>&1
  tail -1 $GRAFT_REPO_ROOT/gpurun_out/r5/joint_probe$g.log
done
cd $GRAFT_REPO_ROOT
for g in 32 64 32stock; do
  f=$(find gpurun_out/r5/prof_joint$g -name "*kernel_trace.csv" | head -1)
  n=joint_step; [ $g = 64 ] && n=joint_step_grid64; [ $g = 32stock ] && n=joint_step_pose_networks_on_stock_torch
  python tools/joint_kernel_share.py $f $n 4 gpurun_out/r5/r05_joint_grid${g}_kernel_share.json gpurun_out/r5/r05_joint_grid${g}_kernel_share.txt
  cp $(find gpurun_out/r5/prof_joint$g -name "*kernel_stats.csv" | head -1) gpurun_out/r5/r05_joint_grid${g}_kernel_stats.csv
done
find gpurun_out/r5 -name "*kernel_trace.csv" -delete
find gpurun_out/r5 -size +3M -delete
