# scratch: the command list of the next `gpurun -- 'bash tools/gpu/_tmp_run.sh'` call (overwritten per experiment; outputs under gpurun_out/)
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x 2>&1 | tail -4
