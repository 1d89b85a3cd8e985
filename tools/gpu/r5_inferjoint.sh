# kernel-level breakdown of FORGE inference with predicted poses (bench.py's joint_inference entry), eager, between marker kernels
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_ji
JOINT_INFER_TRACE=8 timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_ji -o j --output-format csv -- python $GRAFT_REPO_ROOT/tools/joint_infer_probe.py 2>&1 | grep "joint inference"
cd $GRAFT_REPO_ROOT
python tools/joint_kernel_share.py $(find gpurun_out/prof_ji -name "*kernel_trace.csv" | head -1) joint_inference 8 gpurun_out/r05_joint_inference_kernel_share.json gpurun_out/r05_joint_inference_kernel_share.txt
python - <<'PY'
import csv, glob, collections
tr = sorted(csv.DictReader(open(glob.glob("gpurun_out/prof_ji/**/*kernel_trace.csv", recursive=True)[0])), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(tr) if "spin_kernel" in r["Kernel_Name"]]
agg = collections.Counter(); cnt = collections.Counter()
def short(n):
    n = n.replace("void ", "").replace("forge::", "")
    return (n[:n.index("(")] if "(" in n else n)[:90]
for r in tr[marks[0]+1:marks[1]]:
    k = short(r["Kernel_Name"]); agg[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); cnt[k] += 1
tot = sum(agg.values())
print("per forward %.2f ms kernel time, %d launches" % (tot / 8e6, sum(cnt.values()) // 8))
for k, v in agg.most_common(40):
    print("%8.3f ms/fwd %6d/fwd  %s" % (v / 8e6, cnt[k] // 8, k))
PY
rm -rf gpurun_out/prof_ji
python tools/joint_infer_probe.py
