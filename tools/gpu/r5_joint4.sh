cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_joint4
JOINT_SCENES=4 JOINT_STEPS=4 timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_joint4 -o j --output-format csv -- python $GRAFT_REPO_ROOT/tools/joint_step_probe.py 2>&1 | grep "joint step"
cd $GRAFT_REPO_ROOT
python tools/joint_kernel_share.py $(find gpurun_out/prof_joint4 -name "*kernel_trace.csv" | head -1) joint_step_4_scenes 4 gpurun_out/r05_joint_4scenes_kernel_share.json gpurun_out/r05_joint_4scenes_kernel_share.txt
python - <<'PY'
import csv, glob, collections
tr = sorted(csv.DictReader(open(glob.glob("gpurun_out/prof_joint4/**/*kernel_trace.csv", recursive=True)[0])), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(tr) if "spin_kernel" in r["Kernel_Name"]]
agg = collections.Counter(); cnt = collections.Counter()
def short(n):
    n = n.replace("void ", "").replace("forge::", "")
    return (n[:n.index("(")] if "(" in n else n)[:70]
for r in tr[marks[0]+1:marks[1]]:
    k = short(r["Kernel_Name"]); agg[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); cnt[k] += 1
tot = sum(agg.values())
print("per step %.1f ms kernel time" % (tot / 4e6))
for k, v in agg.most_common(28):
    print("%8.2f ms/step %6d/step  %s" % (v / 4e6, cnt[k] // 4, k))
PY
rm -rf gpurun_out/prof_joint4
