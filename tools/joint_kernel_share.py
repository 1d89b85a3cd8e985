#!/usr/bin/env python
"""Split the kernel time of a rocprofv3 `--kernel-trace --stats` run of tools/joint_step_probe.py into libforge_hip.so kernels and
stock-torch kernels (MIOpen / rocBLAS / hipBLASLt / ATen element-wise, reductions, optimizer), by kernel NAME:

    python tools/joint_kernel_share.py <..._kernel_trace.csv> <workload name> <timed steps> <out.json> [out.txt]

Only the launches BETWEEN the probe's two marker kernels (torch.cuda._sleep -> `spin_kernel`, before and after the timed steps) count: the warm-up
steps run MIOpen's solver search (hundreds of `naive_conv_*` benchmark launches on a cold find-db), which is not part of a training step.
Writes the JSON bench.py reads (joint_stock_share) and a text table for profiles/."""
import csv
import json
import re
import sys

FORGE = re.compile(r"forge::|(adam_small|attention_fwd|affine_act_bwd|bn_apply_bwd|bn_apply_fwd|bn_finalize|bn_from_totals|bn_reduce_bwd|bn_stats|colsum_flat|conv_direct|conv_igemm|"
                   r"conv_splitk_epilogue|conv_wgrad|gru_gates|gru_state|im2col_nchw|maxpool2d_nhwc|pack_cameras|pose_chain|pose_xf|render_bwd|render_fwd|"
                   r"resize_bilinear|rotate_bwd|rotate_fwd|sse_groups|transpose_kernel|wino_dw|wino_dy|wino_input|wino_output|wino_weight)[a-z_0-9]*_?kernel|"
                   r"^(void )?(conv_igemm|conv_wgrad|wino_|render_|rotate_|bn_|gru_)")


def family(name):
    n = name
    if "Cijk_" in n or "rocblas" in n.lower() or "hipblaslt" in n.lower():
        return "GEMM library (rocBLAS / hipBLASLt)"
    if "miopen" in n.lower() or "igemm_" in n or "naive_conv" in n or "gridwise" in n.lower() or "Conv" in n and "at::" not in n:
        return "MIOpen convolution"
    if "batch_norm" in n or "BatchNorm" in n:
        return "ATen batch norm"
    if "multi_tensor" in n or "adam" in n.lower() or "FusedAdam" in n:
        return "ATen optimizer / clip"
    return "ATen element-wise / reduce / copy"


def main():
    path, workload, steps, out_json = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    trace = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(trace) if "spin_kernel" in r["Kernel_Name"]]
    if len(marks) != 2:
        raise SystemExit("expected the probe's two marker kernels in the trace, found %d" % len(marks))
    agg = {}
    for r in trace[marks[0] + 1:marks[1]]:
        a = agg.setdefault(r["Kernel_Name"], {"Name": r["Kernel_Name"], "Calls": 0, "TotalDurationNs": 0.0})
        a["Calls"] += 1
        a["TotalDurationNs"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    rows = list(agg.values())
    span_ms = (int(trace[marks[1]]["Start_Timestamp"]) - int(trace[marks[0]]["End_Timestamp"])) / 1e6
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    forge = [r for r in rows if FORGE.search(r["Name"])]
    stock = [r for r in rows if not FORGE.search(r["Name"])]
    f_ns, s_ns = sum(float(r["TotalDurationNs"]) for r in forge), sum(float(r["TotalDurationNs"]) for r in stock)
    fam = {}
    for r in stock:
        fam[family(r["Name"])] = fam.get(family(r["Name"]), 0.0) + float(r["TotalDurationNs"])
    res = {"workload": workload, "source_csv": path.split("/")[-1], "timed_steps": steps, "wall_ms_per_step": span_ms / steps, "kernel_ms_per_step": tot / steps / 1e6,
           "forge_share": f_ns / tot, "stock_share": s_ns / tot, "stock_ms_per_step": s_ns / steps / 1e6, "forge_ms_per_step": f_ns / steps / 1e6,
           "stock_families_ms_per_step": {k: v / steps / 1e6 for k, v in sorted(fam.items(), key=lambda kv: -kv[1])},
           "top_stock_kernels": [{"name": r["Name"][:100], "calls": int(r["Calls"]), "ms_per_step": float(r["TotalDurationNs"]) / steps / 1e6}
                                 for r in sorted(stock, key=lambda r: -float(r["TotalDurationNs"]))[:12]],
           "top_forge_kernels": [{"name": r["Name"][:100], "calls": int(r["Calls"]), "ms_per_step": float(r["TotalDurationNs"]) / steps / 1e6}
                                 for r in sorted(forge, key=lambda r: -float(r["TotalDurationNs"]))[:12]]}
    json.dump(res, open(out_json, "w"), indent=1)
    lines = ["%s: %.2f ms of kernel time per step, %.2f ms wall per step, over the %d timed steps between the probe's markers (rocprofv3 --kernel-trace)"
             % (workload, res["kernel_ms_per_step"], res["wall_ms_per_step"], steps),
             "  libforge_hip.so kernels  %7.2f ms  %5.1f %%" % (res["forge_ms_per_step"], 100 * res["forge_share"]),
             "  stock-torch kernels      %7.2f ms  %5.1f %%" % (res["stock_ms_per_step"], 100 * res["stock_share"])]
    lines += ["    %-42s %7.2f ms" % (k, v) for k, v in res["stock_families_ms_per_step"].items()]
    lines.append("  largest stock-torch kernels:")
    lines += ["    %-100s %5d calls %7.3f ms/step" % (k["name"], k["calls"], k["ms_per_step"]) for k in res["top_stock_kernels"]]
    lines.append("  largest libforge kernels:")
    lines += ["    %-100s %5d calls %7.3f ms/step" % (k["name"], k["calls"], k["ms_per_step"]) for k in res["top_forge_kernels"]]
    txt = "\n".join(lines)
    print(txt)
    if len(sys.argv) > 5:
        open(sys.argv[5], "w").write(txt + "\n")


if __name__ == "__main__":
    main()
