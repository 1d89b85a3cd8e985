# rocprofv3 kernel statistics of the inference bench command, ONE step at a time (a kernel's duration is its own): tag = $1
TAG=${1:-infer_b1}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_inf -o inf --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --repeats 1 --pipeline-depth 1 --no-cpu-baseline --no-oracle-check --no-microbench --no-extra ${BENCH_ARGS:-} > $GRAFT_REPO_ROOT/gpurun_out/${TAG}.json 2> $GRAFT_REPO_ROOT/gpurun_out/${TAG}.err
cd $GRAFT_REPO_ROOT
python - "$TAG" <<'PY'
import csv, sys, glob, json, shutil
tag = sys.argv[1]
f = glob.glob("gpurun_out/prof_inf/**/inf_kernel_stats.csv", recursive=True)[0]
shutil.copy(f, "gpurun_out/%s_kernel_stats.csv" % tag)
rows = list(csv.DictReader(open(f)))
# steps in the trace = launches of a kernel that runs exactly once per step (bench.py reads this sidecar for roofline.frac_rocprof)
steps = [int(r["Calls"]) for r in rows if "rotate_fwd_kernel" in r["Name"]]
json.dump({"steps_traced": steps[0] if steps else None, "command": "python bench.py --steps 20 --warmup 3 --repeats 1 --pipeline-depth 1 --no-cpu-baseline --no-microbench --no-extra",
           "counted_by": "launches of rotate_fwd_kernel (one per step at one scene)"}, open("gpurun_out/%s_kernel_stats.meta.json" % tag, "w"))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms:", tot / 1e6)
for r in rows[:22]:
    print("%-86s %6s calls %9.3f ms %6.2f%%  avg %8.1f us" % (r["Name"][:86], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"]), float(r["AverageNs"]) / 1e3))
try:
    j = json.loads(open("gpurun_out/%s_bench_line.json" % tag.replace("_rocprofv3", "")).read().strip().splitlines()[-1])
    print("bench:", j["value"], "views/s", j["ms_per_step"], "ms/step")
except Exception as e:
    print("bench line unreadable:", e)
PY
rm -rf gpurun_out/prof_inf
