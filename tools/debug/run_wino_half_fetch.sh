# L2 -> fabric read bytes of the point GEMM in both forms: bash tools/debug/run_wino_half_fetch.sh  (one rocprofv3 --pmc FETCH_SIZE pass)
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_whf -o p -- python $GRAFT_REPO_ROOT/tools/debug/wino_half_fetch.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, sys
sys.path.insert(0, "tools/debug")
f = glob.glob("gpurun_out/pmc_whf/**/*counter_collection.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "conv_igemm_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
vals = [(r["Kernel_Name"][:60], float(r["Counter_Value"])) for r in rows]
cases = (("gates [x|h] -> 256", 128, 128, 256), ("state [x|hr] -> 128", 128, 128, 128), ("gates h half 128 -> 256", 128, 0, 256), ("fusion_conv 128 -> 128", 128, 0, 128))
R = 32 * 16 * 16
i = 0
for name, C1, C2, Cout in cases:
    alg = 4.0 * (16 * R * (C1 + C2) + 16 * 3 * Cout * (C1 + C2)) / 1e6
    for form in ("16 planes", "8 planes"):
        v = vals[i:i + 3]; i += 3
        mb = [2 * x * 1024 / 1e6 for _, x in v]                         # KiB, wide loads counted at half (MI355X_MICROARCH.md)
        print("%-26s %-10s fetched %s MB per launch (operands + weights: %.1f MB) -> x%.2f   [%s]" % (name, form, ", ".join("%.0f" % m for m in mb), alg, mb[-1] / alg, v[0][0][-22:]))
PY
rm -rf gpurun_out/pmc_whf
