# rotate_fwd_kernel: which unit limits it? rocprofv3 --pmc passes (one counter group per run, kernel trace only) over tools/rotate_limiter_probe.py,
# once per access pattern (identity = source walked in output order; bench = the bench's rotation; copy = mode-0 streaming copy), HBM-resident ring.
# Output: gpurun_out/r06/pmc_rotate.json + .txt (copy into profiles/).
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_rotate
run() { ROTATE_PROBE_CASE=$1 ROTATE_PROBE_ITERS=6 timeout 300 rocprofv3 --pmc $3 --kernel-trace --output-format csv -d $OUT/$1_$2 -o p -- python $GRAFT_REPO_ROOT/tools/rotate_limiter_probe.py > /dev/null 2>&1 || echo "pass $1 $2 failed"; }
for c in copy identity bench; do
  run $c SQ "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"
  run $c TCC "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
  run $c TCC2 "TCC_EA0_RDREQ_DRAM_sum TCC_TAG_STALL_sum TCC_BUSY_avr TCC_EA0_RDREQ_LEVEL_sum"
  run $c TCP "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum"
  run $c TLB "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
  run $c TA "TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum"
  run $c FETCH "FETCH_SIZE"
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import collections, csv, glob, json, os
res = collections.OrderedDict()
for case in ("copy", "identity", "bench"):
    e = {}
    for d in sorted(glob.glob("gpurun_out/pmc_rotate/%s_*" % case)):
        f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
        if not f:
            continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f[0])):
            if "rotate_fwd_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            v = v[len(v) // 2:]                      # the second half of the launches: the ring has been walked once
            e[k] = sum(v) / len(v)
    res[case] = e
def ratio(e, a, b):
    return e[a] / e[b] if e.get(a) is not None and e.get(b) else None
out = {"source": "tools/pmc_rotate.sh: rocprofv3 --pmc passes over tools/rotate_limiter_probe.py (5 volumes 32^3 x 128 ch per launch, ring of sets > Infinity Cache); "
                 "per-launch means; SQ_* in quad-cycles summed over waves", "raw": res, "derived": {}}
for case, e in res.items():
    out["derived"][case] = {
        "wait_any_frac": ratio(e, "SQ_WAIT_ANY", "SQ_WAVE_CYCLES"), "wait_inst_any_frac": ratio(e, "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES"),
        "active_inst_any_frac": ratio(e, "SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES"), "active_valu_frac": ratio(e, "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES"),
        "active_vmem_frac": ratio(e, "SQ_ACTIVE_INST_VMEM", "SQ_WAVE_CYCLES"),
        "l2_hit_rate": (e["TCC_HIT_sum"] / (e["TCC_HIT_sum"] + e["TCC_MISS_sum"])) if e.get("TCC_HIT_sum") is not None and (e.get("TCC_HIT_sum", 0) + e.get("TCC_MISS_sum", 0)) else None,
        "l1_miss_rate (TCP->TCC read requests / TCP cache accesses)": ratio(e, "TCP_TCC_READ_REQ_sum", "TCP_TOTAL_CACHE_ACCESSES_sum"),
        "avg_l2_read_latency_cycles": ratio(e, "TCP_TCC_READ_REQ_LATENCY_sum", "TCP_TCC_READ_REQ_sum"),
        "utcl1_miss_rate": ratio(e, "TCP_UTCL1_TRANSLATION_MISS_sum", "TCP_UTCL1_REQUEST_sum"),
        "hbm_bytes (2 x FETCH_SIZE KiB)": 2 * e["FETCH_SIZE"] * 1024 if e.get("FETCH_SIZE") is not None else None,
        "valu_insts_per_vmem_rd": ratio(e, "SQ_INSTS_VALU", "SQ_INSTS_VMEM_RD")}
os.makedirs("gpurun_out/r06", exist_ok=True)
json.dump(out, open("gpurun_out/r06/pmc_rotate.json", "w"), indent=1)
print(json.dumps(out["derived"], indent=1))
PY
rm -rf gpurun_out/pmc_rotate
