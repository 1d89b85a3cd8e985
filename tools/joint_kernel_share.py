#!/usr/bin/env python
"""Split the kernel time of a rocprofv3 `--kernel-trace --stats` run of tools/joint_step_probe.py into libforge_hip.so kernels and
stock-torch kernels (MIOpen / rocBLAS / hipBLASLt / ATen element-wise, reductions, optimizer), by kernel NAME:

    python tools/joint_kernel_share.py <..._kernel_stats.csv> <workload name> <steps incl. warm-up> <out.json> [out.txt]

Writes the JSON bench.py reads (joint_stock_share) and a text table for profiles/."""
import csv
import json
import re
import sys

FORGE = re.compile(r"(adam_small|affine_act_bwd|bn_apply_bwd|bn_apply_fwd|bn_finalize|bn_from_totals|bn_reduce_bwd|bn_stats|colsum_flat|conv_direct|conv_igemm|"
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
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    forge = [r for r in rows if FORGE.search(r["Name"])]
    stock = [r for r in rows if not FORGE.search(r["Name"])]
    f_ns, s_ns = sum(float(r["TotalDurationNs"]) for r in forge), sum(float(r["TotalDurationNs"]) for r in stock)
    fam = {}
    for r in stock:
        fam[family(r["Name"])] = fam.get(family(r["Name"]), 0.0) + float(r["TotalDurationNs"])
    res = {"workload": workload, "source_csv": path.split("/")[-1], "steps_in_trace": steps, "kernel_ms_per_step": tot / steps / 1e6,
           "forge_share": f_ns / tot, "stock_share": s_ns / tot, "stock_ms_per_step": s_ns / steps / 1e6, "forge_ms_per_step": f_ns / steps / 1e6,
           "stock_families_ms_per_step": {k: v / steps / 1e6 for k, v in sorted(fam.items(), key=lambda kv: -kv[1])},
           "top_stock_kernels": [{"name": r["Name"][:100], "calls": int(r["Calls"]), "ms_per_step": float(r["TotalDurationNs"]) / steps / 1e6}
                                 for r in sorted(stock, key=lambda r: -float(r["TotalDurationNs"]))[:12]],
           "top_forge_kernels": [{"name": r["Name"][:100], "calls": int(r["Calls"]), "ms_per_step": float(r["TotalDurationNs"]) / steps / 1e6}
                                 for r in sorted(forge, key=lambda r: -float(r["TotalDurationNs"]))[:12]]}
    json.dump(res, open(out_json, "w"), indent=1)
    lines = ["%s: %.2f ms of kernel time per step over %d steps (rocprofv3 --kernel-trace --stats)" % (workload, res["kernel_ms_per_step"], steps),
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
