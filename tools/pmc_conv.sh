# HBM-side traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes) of the ConvGRU gates (128x128 tile) and state (64x64 tile) launches
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  PROBE_KERNELS=conv timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_conv_$c -o p -- python $GRAFT_REPO_ROOT/tools/probe_kernels.py > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
for c in ["FETCH_SIZE","WRITE_SIZE"]:
    f = glob.glob("gpurun_out/pmc_conv_%s/**/*counter_collection.csv" % c, recursive=True)
    if not f: print(c, "no file"); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "conv_igemm" in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:52]].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        print(c, k, "launches", len(v), "mean KiB", sum(v)/len(v))
PY
rm -rf gpurun_out/pmc_conv_FETCH_SIZE gpurun_out/pmc_conv_WRITE_SIZE
