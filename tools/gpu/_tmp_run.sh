cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "narrow_transposed" 2>&1 | tail -3
python tools/debug/joint_shard_noise.py 2>&1 | grep -v Warning | tail -40 > gpurun_out/joint_noise_plain.txt
JOINT_DETERMINISTIC=1 python tools/debug/joint_shard_noise.py 2>&1 | grep -v Warning | tail -40 > gpurun_out/joint_noise_det.txt
cat gpurun_out/joint_noise_plain.txt | cut -c1-200
echo ======
cat gpurun_out/joint_noise_det.txt | cut -c1-200
