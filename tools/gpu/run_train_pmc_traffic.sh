# L2->fabric traffic of every kernel of one GT-pose training step (4 scenes): two rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE) over
# tools/train_step_probe.py, aggregated by (kernel, grid) over the LAST step; bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950: wide loads
# count half, MI355X_MICROARCH.md). Beside each: the kernel's time in a plain --kernel-trace run and the time those bytes need at 5 TB/s.
export TRAIN_SCENES=${TRAIN_SCENES:-4}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  TRAIN_STEPS=2 timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmctr_$c -o p -- python $GRAFT_REPO_ROOT/tools/train_step_probe.py > /dev/null 2>&1
done
TRAIN_STEPS=2 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmctr_T -o p -- python $GRAFT_REPO_ROOT/tools/train_step_probe.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
def short(n):
    n = n.replace("void ", "").replace("forge::", "")
    return (n[:n.index("(")] if "(" in n else n)[:60]
def last_step(rows, key_name):
    rows = sorted(rows, key=lambda r: int(r.get("Start_Timestamp") or r.get("Dispatch_Id") or 0))
    marks = [i for i, r in enumerate(rows) if "im2col_nchw_kernel" in r[key_name]]
    return rows[marks[-1]:] if marks else rows
agg = collections.defaultdict(lambda: [0.0, 0.0, 0.0, 0])
for c, idx in (("FETCH_SIZE", 0), ("WRITE_SIZE", 1)):
    f = glob.glob("gpurun_out/pmctr_%s/**/*counter_collection.csv" % c, recursive=True)
    rows = [r for r in csv.DictReader(open(f[0])) if r["Counter_Name"] == c]
    rows = sorted(rows, key=lambda r: int(r["Dispatch_Id"]))
    marks = [i for i, r in enumerate(rows) if "im2col_nchw_kernel" in r["Kernel_Name"]]
    for r in rows[marks[-1]:]:
        agg[(short(r["Kernel_Name"]), r["Grid_Size"])][idx] += float(r["Counter_Value"])
f = glob.glob("gpurun_out/pmctr_T/**/*kernel_trace.csv", recursive=True)
tr = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(tr) if "im2col_nchw_kernel" in r["Kernel_Name"]]
for r in tr[marks[-1]:]:
    g = str(int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]))
    k = (short(r["Kernel_Name"]), g)
    agg[k][2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg[k][3] += 1
out = []
for (n, g), (fe, wr, us, cnt) in agg.items():
    by = (2 * fe + wr) * 1024
    out.append((by, n, g, us, cnt))
out.sort(reverse=True)
tot = sum(o[0] for o in out)
with open("gpurun_out/r04_train_pmc_traffic_b4.txt", "w") as fo:
    print("L2->fabric bytes of one 4-scene training step: %.1f GB; by (kernel, grid): GB, launches, kernel us, us those bytes need at 5 TB/s, ratio" % (tot / 1e9), file=fo)
    for by, n, g, us, cnt in out[:70]:
        need = by / 5e12 * 1e6
        print("%8.2f GB  %3d x  %9.1f us  %9.1f us  %5.2f  %-60s grid %s" % (by / 1e9, cnt, us, need, need / us if us else 0, n, g), file=fo)
print(open("gpurun_out/r04_train_pmc_traffic_b4.txt").read())
PY
rm -rf gpurun_out/pmctr_FETCH_SIZE gpurun_out/pmctr_WRITE_SIZE gpurun_out/pmctr_T
