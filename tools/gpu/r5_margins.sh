cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5
timeout 1500 python tools/debug/train_grad_margins.py > gpurun_out/r5/train_grad_margins.txt 2>&1
grep -v Warn gpurun_out/r5/train_grad_margins.txt | tail -120
