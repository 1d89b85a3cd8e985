cd /tmp && export TMPDIR=/tmp
TRAIN_STEPS=5 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_train -o tr --output-format csv -- python $GRAFT_REPO_ROOT/tools/train_step_probe.py > $GRAFT_REPO_ROOT/gpurun_out/trainprof.log 2>&1
cd $GRAFT_REPO_ROOT
tail -2 gpurun_out/trainprof.log
python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/prof_train/tr_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms over 7 steps + warmup:", tot/1e6)
for r in rows[:28]:
    print("%-90s %6s calls %9.2f ms %6.2f%%  avg %8.1f us" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["Percentage"]), float(r["AverageNs"])/1e3))
PY
find gpurun_out/prof_train -size +2M -delete
