# forge_render_fwd at the big-volume shapes: timing + L2->fabric traffic (FETCH_SIZE, separate --pmc pass).
# (Round 2 compared three variants with this script - profiles/r02_render_ab.txt; only the default kernel is in the library now.)
cd /tmp && export TMPDIR=/tmp
python $GRAFT_REPO_ROOT/tools/render_probe.py 2>&1 | grep "render D_r"
RENDER_PROBE_ITERS=3 timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_render -o p -- python $GRAFT_REPO_ROOT/tools/render_probe.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/pmc_render/**/*counter_collection.csv", recursive=True)
if not f:
    print("no counter file")
else:
    rows = [r for r in csv.DictReader(open(f[0])) if "render_fwd" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0)))
    # launch order of tools/render_probe.py: 4 launches (1 warm-up + 3) per case, cases 64x5, 64x28, 128x5, 128x28
    per = len(rows) // 4
    for i, case in enumerate(("D_r=64 V=5", "D_r=64 V=28", "D_r=128 V=5", "D_r=128 V=28")):
        v = [float(r["Counter_Value"]) for r in rows[i * per:(i + 1) * per]]
        if v:
            print("%-13s launches %d  FETCH_SIZE mean %.1f MB (x2 per MI355X_MICROARCH.md: %.1f MB)" % (case, len(v), sum(v) / len(v) / 1024, 2 * sum(v) / len(v) / 1024))
PY
rm -rf gpurun_out/pmc_render
