# training-step evidence: rocprofv3 kernel trace of tools/train_step_probe.py at TRAIN_SCENES (default 4) scenes per GPU,
# kernel statistics + the launch-ordered timeline of one steady-state step + an aggregate by (kernel, grid) of that step
TAG=${1:-train_b4}
export TRAIN_SCENES=${TRAIN_SCENES:-4}
cd /tmp && export TMPDIR=/tmp
TRAIN_STEPS=5 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_train -o tr --output-format csv -- python $GRAFT_REPO_ROOT/tools/train_step_probe.py > $GRAFT_REPO_ROOT/gpurun_out/${TAG}.log 2>&1
cd $GRAFT_REPO_ROOT
tail -2 gpurun_out/${TAG}.log
python - "$TAG" <<'PY'
import csv, sys, collections, glob
tag = sys.argv[1]
out = open("gpurun_out/%s_kernel_stats.txt" % tag, "w")
rows=list(csv.DictReader(open(glob.glob("gpurun_out/prof_train/**/tr_kernel_stats.csv", recursive=True)[0])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms over 7 steps:", tot/1e6, " per step:", tot/7e6, file=out)
for r in rows[:45]:
    print("%-90s %6s calls %9.2f ms %6.2f%%  avg %8.1f us" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["Percentage"]), float(r["AverageNs"])/1e3), file=out)
out.close()
tr = sorted(csv.DictReader(open(glob.glob("gpurun_out/prof_train/**/tr_kernel_trace.csv", recursive=True)[0])), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(tr) if "im2col_nchw_kernel" in r["Kernel_Name"]]
step = tr[marks[-2]:marks[-1]]
def short(n):
    n = n.replace("void ", "").replace("forge::", "")
    return n[:n.index("(")][:70] if "(" in n else n[:70]
agg = collections.OrderedDict()
with open("gpurun_out/%s_step_timeline.txt" % tag, "w") as f:
    t0 = int(step[0]["Start_Timestamp"]); prev = t0
    for r in step:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        wg = int(r.get("Workgroup_Size_X", 0) or 0) or 1
        gx, gy, gz = (int(r.get("Grid_Size_" + a, 1) or 1) for a in "XYZ")
        nwg = (gx // wg) * gy * gz
        f.write("%9.1f %8.1f %6.1f  %-70s %8d %4d %6s %4s\n" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, short(r["Kernel_Name"]), nwg, wg, r.get("LDS_Block_Size", ""), r.get("VGPR_Count", "")))
        prev = max(prev, e)
        k = (short(r["Kernel_Name"]), nwg)
        a = agg.setdefault(k, [0, 0]); a[0] += 1; a[1] += e - s
    f.write("step: %d kernels, span %.3f ms, kernel time %.3f ms\n" % (len(step), (prev - t0) / 1e6, sum(v[1] for v in agg.values()) / 1e6))
with open("gpurun_out/%s_step_by_shape.txt" % tag, "w") as f:
    for (n, g), (c, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        f.write("%9.3f ms %4d x %8.1f us  wgs %8d  %s\n" % (ns / 1e6, c, ns / c / 1e3, g, n))
PY
rm -rf gpurun_out/prof_train
