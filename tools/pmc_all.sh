# PMC passes (rounds 2-4) over tools/probe_kernels.py (rotate / ray-march / ConvGRU gates + state launches at the b=1 bench shapes), each counter
# group in its own rocprofv3 run (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE cannot share a pass), summarised into
# gpurun_out/pmc_summary.json (copied to profiles/ by hand; bench.py reads profiles/*pmc_summary.json for roofline.traffic).
cd /tmp && export TMPDIR=/tmp
run() { PROBE_KERNELS=$3 timeout 300 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmcall_$1 -o p -- python $GRAFT_REPO_ROOT/tools/probe_kernels.py > /dev/null 2>&1; }
run FETCH FETCH_SIZE rotate,render,conv
run WRITE WRITE_SIZE rotate,render,conv
run SQ "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT" rotate,render,conv
# the Winograd launches of the inference fusion in their own runs (the point-GEMM launch shares its kernel name with the direct launches)
run WFETCH FETCH_SIZE wino
run WWRITE WRITE_SIZE wino
run WSQ "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT" wino
# the ray-march backward (round 4): both launches, with and without camera gradients
run BFETCH FETCH_SIZE render_bwd
run BWRITE WRITE_SIZE render_bwd
run BSQ "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" render_bwd
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, json, collections
def load(tag):
    f = glob.glob("gpurun_out/pmcall_%s/**/*counter_collection.csv" % tag, recursive=True)
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    if not f: return out
    for r in csv.DictReader(open(f[0])):
        out[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out
F, W, S = load("FETCH"), load("WRITE"), load("SQ")
WF, WW, WS = load("WFETCH"), load("WWRITE"), load("WSQ")
BF, BW, BS = load("BFETCH"), load("BWRITE"), load("BSQ")
mean = lambda v: sum(v) / len(v) if v else None
def pick(d, sub):
    for k in d:
        if sub in k: return k
    return None
summary = {"source": "tools/pmc_all.sh: separate rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ+GRBM) over tools/probe_kernels.py, 5 launches "
                     "per kernel at the one-scene bench shapes; FETCH_SIZE / WRITE_SIZE are KiB; hbm_bytes_corrected = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 "
                     "(gfx950 counts wide loads at half, MI355X_MICROARCH.md - these are L2->fabric bytes, Infinity-Cache hits included)"}
alg = {"rotate_fwd_kernel": 5 * 128 * 32 ** 3 * 4 * 2, "render_fwd_kernel": 17 * 64 ** 3 * 4 + 5 * 17 * 128 * 128 * 4,
       "conv_igemm_kernel<128, 128": 74280000.0, "conv_igemm_kernel<64, 64": 87100000.0}
names = {"rotate_fwd_kernel": "rotate_fwd_kernel", "render_fwd_kernel": "render_fwd_kernel<4>",
         "conv_igemm_kernel<128, 128": "conv_igemm_kernel<128, 128, 8> (ConvGRU gates, M=32768 N=256 K=6912)",
         "conv_igemm_kernel<64, 64": "conv_igemm_kernel<64, 64, 4> (ConvGRU state, M=32768 N=128 K=6912)"}
R_ = 32 * 16 * 16
wino = {"wino_input_kernel": ("wino_input_kernel (h -> V_h, 32^3 x 128 channels)", 4.0 * (32 ** 3 * 128 + 16 * R_ * 128)),
        "conv_igemm_kernel<": ("conv_igemm_kernel winograd gates point GEMMs (16 x [8192 x 768] x [768 x 256], one launch; row stage of the inverse "
                               "transform in the epilogue: 8 planes written)", 4.0 * (16 * R_ * 256 + 16 * 3 * 256 * 256 + 8 * R_ * 256)),
        "wino_output_kernel": ("wino_output_kernel<GRU gates, HALF> (Mm8 -> z, h*r)", 4.0 * (8 * R_ * 256 + 3 * 32 ** 3 * 128))}
Vb, Hb, Sb, Cb, Db = 10, 128, 64, 16, 64
rbwd = {"render_bwd_rays_kernel<4, false>": ("render_bwd_rays_kernel<4, false> (10 views x 128^2 rays x 64 samples of one 64^3 volume: march + per-sample scalars)",
                                              4.0 * (17 * Db ** 3 + Vb * Hb * Hb * (Cb + 1) + 2 * Vb * Hb * Hb * Sb)),
        "render_bwd_rays_kernel<4, true>": ("render_bwd_rays_kernel<4, true> (the same + camera gradients: six position-gradient scalars per sample parked by pass A, read by pass C)",
                                             4.0 * (17 * Db ** 3 + Vb * Hb * Hb * (Cb + 1) + 2 * Vb * Hb * Hb * Sb + 2 * 6 * Vb * Hb * Hb * Sb)),
        "render_bwd_voxels_kernel": ("render_bwd_voxels_kernel<4> (voxel-parallel gather of 10 views into one 64^3 x 17 gradient volume)",
                                     4.0 * (2 * Vb * Hb * Hb * Sb + Vb * Hb * Hb * Cb + 17 * Db ** 3))}
jobs = ([(sub, label, alg[sub], F, W, S) for sub, label in names.items()] + [(sub, lab, ab, WF, WW, WS) for sub, (lab, ab) in wino.items()]
        + [(sub, lab, ab, BF, BW, BS) for sub, (lab, ab) in rbwd.items()])
for sub, label, abytes, F, W, S in jobs:
    e = {}
    kf, kw, ks = pick(F, sub), pick(W, sub), pick(S, sub)
    if kf: e["fetch_kib_raw"] = mean(F[kf]["FETCH_SIZE"])
    if kw: e["write_kib"] = mean(W[kw]["WRITE_SIZE"])
    if "fetch_kib_raw" in e and "write_kib" in e:
        e["hbm_bytes_corrected"] = (2 * e["fetch_kib_raw"] + e["write_kib"]) * 1024
    e["algorithmic_bytes"] = abytes
    if ks:
        c = {k: mean(v) for k, v in S[ks].items()}
        e.update({"mfma_busy_cycles": c.get("SQ_VALU_MFMA_BUSY_CYCLES"), "grbm_gui_active": c.get("GRBM_GUI_ACTIVE"), "sq_wave_cycles": c.get("SQ_WAVE_CYCLES"),
                  "sq_wait_any": c.get("SQ_WAIT_ANY"), "lds_bank_conflict": c.get("SQ_LDS_BANK_CONFLICT")})
        if c.get("SQ_WAIT_INST_ANY") is not None and c.get("SQ_WAVE_CYCLES"):
            e.update({"wait_inst_any_frac": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], "active_inst_any_frac": (c.get("SQ_ACTIVE_INST_ANY") or 0.0) / c["SQ_WAVE_CYCLES"]})
        if c.get("SQ_VALU_MFMA_BUSY_CYCLES") and c.get("GRBM_GUI_ACTIVE"):
            e["mfma_util_in_kernel"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024)
        if c.get("SQ_WAIT_ANY") and c.get("SQ_WAVE_CYCLES"):
            e["wait_any_frac"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
    summary[label] = e
json.dump(summary, open("gpurun_out/pmc_summary.json", "w"), indent=1)
print(json.dumps(summary, indent=1))
PY
rm -rf gpurun_out/pmcall_FETCH gpurun_out/pmcall_WRITE gpurun_out/pmcall_SQ gpurun_out/pmcall_WFETCH gpurun_out/pmcall_WWRITE gpurun_out/pmcall_WSQ gpurun_out/pmcall_BFETCH gpurun_out/pmcall_BWRITE gpurun_out/pmcall_BSQ
