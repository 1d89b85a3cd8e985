cd /root/repo
bash tools/gpu/measure_round.sh > gpurun_out/measure_round.log 2>&1
{
python tools/pose3d_probe.py 2>&1 | grep -v amdgpu.ids | tail -3
POSE3D_SCENES=4 python tools/pose3d_probe.py 2>&1 | grep -v amdgpu.ids | tail -3
python tools/train_step_probe.py 2>&1 | tail -1
TRAIN_SCENES=4 python tools/train_step_probe.py 2>&1 | tail -1
TRAIN_GRID=64 python tools/train_step_probe.py 2>&1 | tail -1
TRAIN_GRID=64 TRAIN_SCENES=4 python tools/train_step_probe.py 2>&1 | tail -1
TRAIN_GRAPH=1 python tools/train_step_probe.py 2>&1 | tail -1
python tools/refine_probe.py 2>&1 | tail -1
python tools/wino_ab.py 2>&1 | grep -v amdgpu.ids
WINO_SCENES=4 python tools/wino_ab.py 2>&1 | grep -v amdgpu.ids
python tools/wino_gemm_sweep.py 2>&1 | grep -v amdgpu.ids
WINO_SCENES=4 python tools/wino_gemm_sweep.py 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r02_winograd_probes.txt 2>&1
bash tools/gpu/run_trainprof.sh > gpurun_out/r02_train_step_kernel_stats.txt 2>&1
