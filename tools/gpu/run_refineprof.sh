# rocprofv3 kernel statistics of the pose-refinement loop (tools/refine_probe.py, 60 hipGraph-replayed iterations at t = 5)
cd /tmp && export TMPDIR=/tmp
REFINE_ITERS=60 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_refine -o rf --output-format csv -- python $GRAFT_REPO_ROOT/tools/refine_probe.py > $GRAFT_REPO_ROOT/gpurun_out/refineprof.log 2>&1
cd $GRAFT_REPO_ROOT
tail -1 gpurun_out/refineprof.log
python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/prof_refine/rf_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms (60 iterations + warm-up / capture passes):", tot/1e6)
for r in rows[:26]:
    print("%-90s %6s calls %9.2f ms %6.2f%%  avg %8.1f us" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["Percentage"]), float(r["AverageNs"])/1e3))
PY
find gpurun_out/prof_refine -size +2M -delete
