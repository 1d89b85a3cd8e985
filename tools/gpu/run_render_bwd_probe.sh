cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_rb -o rb --output-format csv -- python $GRAFT_REPO_ROOT/tools/render_bwd_probe.py > $GRAFT_REPO_ROOT/gpurun_out/render_bwd_probe.log 2>&1
cd $GRAFT_REPO_ROOT
grep backward gpurun_out/render_bwd_probe.log
python - <<'PY'
import csv, glob
tr = sorted(csv.DictReader(open(glob.glob("gpurun_out/prof_rb/**/rb_kernel_trace.csv", recursive=True)[0])), key=lambda r: int(r["Start_Timestamp"]))
seen=set()
for r in tr:
    n = r["Kernel_Name"]
    if "render_bwd" in n:
        k=(n[:60], r["Grid_Size_X"], r["Grid_Size_Z"])
        if k in seen: continue
        seen.add(k)
        print("%8.1f us  grid %s x %s x %s  %s  vgpr %s" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], n[:70], r.get("VGPR_Count")))
PY
rm -rf gpurun_out/prof_rb
