cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_rot_$c -o p -- python $GRAFT_REPO_ROOT/tools/rotate_probe.py > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
for c in ["FETCH_SIZE","WRITE_SIZE"]:
    f = glob.glob("gpurun_out/pmc_rot_%s/**/*counter_collection.csv" % c, recursive=True)
    if not f: print(c, "no file"); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "rotate_fwd" in r["Kernel_Name"] or "rotate_bwd_gather" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"][:40], r["Grid_Size"])].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        print(c, k, "launches", len(v), "mean KiB", sum(v)/len(v))
PY
rm -rf gpurun_out/pmc_rot_FETCH_SIZE gpurun_out/pmc_rot_WRITE_SIZE
