#!/usr/bin/env python
"""Idle time between kernels in the steady-state tail of a rocprofv3 kernel trace: span vs sum of durations, largest gaps."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
rows = sorted(rows, key=lambda r: int(r["Start_Timestamp"]))
rows = rows[int(len(rows) * (1 - frac)):]
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
print("launches %d  span %.2f ms  kernel time %.2f ms  idle %.2f ms (%.1f%%)" % (len(rows), span / 1e6, busy / 1e6, (span - busy) / 1e6, 100.0 * (span - busy) / span))
gaps = []
for a, b in zip(rows, rows[1:]):
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    gaps.append((g, a["Kernel_Name"][:60], b["Kernel_Name"][:60]))
gaps.sort(reverse=True)
for g, a, b in gaps[:12]:
    print("%8.1f us  after %-60s before %s" % (g / 1e3, a, b))
import collections
small = collections.Counter()
for r in rows:
    if "forge::" not in r["Kernel_Name"]:
        small[r["Kernel_Name"][:70]] += 1
print("non-forge kernels in the tail:", sum(small.values()))
for k, v in small.most_common(12):
    print("%5d  %s" % (v, k))
