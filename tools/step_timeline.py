#!/usr/bin/env python
"""Timeline of ONE step from a rocprofv3 --kernel-trace CSV: every kernel of the last complete step (delimited by the step's first
kernel, default forge::im2col_nchw_kernel) with start offset, duration, gap to the previous kernel, grid / workgroup size, LDS and
register counts. Shows where a launch-bound stage (ResNet trunk at one scene) spends its time: kernel bodies vs inter-kernel gaps.

    python tools/step_timeline.py <kernel_trace.csv> [first-kernel-substring]
"""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
first = sys.argv[2] if len(sys.argv) > 2 else "im2col_nchw_kernel"
marks = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
if len(marks) < 3:
    raise SystemExit("need at least 3 steps in the trace")
lo, hi = marks[-2], marks[-1]
step = rows[lo:hi]
t0 = int(step[0]["Start_Timestamp"])
prev_end = t0
tot = gaps = 0


def short(n):
    n = n.replace("void ", "").replace("forge::", "")
    return n[:n.index("(")][:58] if "(" in n else n[:58]


print("%8s %8s %7s  %-58s %9s %6s %6s %5s" % ("t_us", "dur_us", "gap_us", "kernel", "grid", "wg", "lds", "vgpr"))
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    g = s - prev_end
    tot += e - s
    gaps += max(g, 0)
    grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)
    wg = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 0)) or 0)
    print("%8.1f %8.1f %7.1f  %-58s %9d %6d %6s %5s" % ((s - t0) / 1e3, (e - s) / 1e3, g / 1e3, short(r["Kernel_Name"]), grid // max(wg, 1), wg,
                                                     r.get("LDS_Block_Size", ""), r.get("VGPR_Count", r.get("Arch_VGPR_Count", ""))))
    prev_end = max(prev_end, e)
span = prev_end - t0
print("step: %d kernels, span %.3f ms, kernel time %.3f ms, gaps %.3f ms" % (len(step), span / 1e6, tot / 1e6, gaps / 1e6))
