cd /tmp && export TMPDIR=/tmp
for x in 0 1; do
PROBE_PARTS=replay PROBE_X3=$x rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_x3_$x -o p --output-format csv -- python $GRAFT_REPO_ROOT/tools/experiments/bf16x3/bf16x3_probe.py > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
for x in (0, 1):
    f = glob.glob("gpurun_out/prof_x3_%d/**/p_kernel_stats.csv" % x, recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    print("== PROBE_X3 =", x, "total kernel ms", sum(float(r["TotalDurationNs"]) for r in rows) / 1e6)
    for r in rows[:12]:
        print("%-100s %6s calls %9.3f ms avg %8.1f us" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
rm -rf gpurun_out/prof_x3_0 gpurun_out/prof_x3_1
