# forge_render_fwd at the big-volume shapes: timing + L2->fabric traffic (FETCH_SIZE, separate --pmc pass), WITH the product's XCD band order
# (128 image rows) and WITHOUT it (120 rows: 15 tile rows, which forge_render_fwd launches in plain order) - VERDICT r5 item 6.
cd /tmp && export TMPDIR=/tmp
for HR in 128 120; do
RENDER_PROBE_HR=$HR python $GRAFT_REPO_ROOT/tools/render_probe.py 2>&1 | grep "render D_r"
RENDER_PROBE_HR=$HR RENDER_PROBE_ITERS=3 timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_render -o p -- python $GRAFT_REPO_ROOT/tools/render_probe.py > /dev/null 2>&1
(cd $GRAFT_REPO_ROOT && HR=$HR python - <<'PY'
import csv, glob, os
HR = int(os.environ["HR"])
f = glob.glob("gpurun_out/pmc_render/**/*counter_collection.csv", recursive=True)
if not f:
    print("no counter file")
else:
    rows = [r for r in csv.DictReader(open(f[0])) if "render_fwd" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0)))
    # launch order of tools/render_probe.py: 4 launches (1 warm-up + 3) per case, cases 64x5, 64x28, 128x5, 128x28
    per = len(rows) // 4
    for i, (D, V) in enumerate(((64, 5), (64, 28), (128, 5), (128, 28))):
        v = [float(r["Counter_Value"]) for r in rows[i * per:(i + 1) * per]]
        alg = 17 * D ** 3 * 4 + V * 17 * HR * 128 * 4
        if v:
            kib = sum(v) / len(v)
            print("D_r=%-3d V=%-2d rows=%d  launches %d  FETCH_SIZE %.1f MB raw; x2 (the guide's gfx950 correction, calibrated on wide coalesced reads - an UPPER bound for "
                  "16-byte gathers): %.1f MB = %.2fx the %.1f MB algorithmic (volume once + maps)" % (D, V, HR, len(v), kib / 1024, 2 * kib / 1024, 2 * kib * 1024 / alg, alg / 1e6))
PY
)
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_render
done
