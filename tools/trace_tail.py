#!/usr/bin/env python
"""Aggregate the LAST fraction of a rocprofv3 kernel trace CSV by kernel name (steady-state steps, past warm-up / solver search)."""
import collections
import csv
import sys

path, frac = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1 / 3
rows = list(csv.DictReader(open(path)))
rows = rows[int(len(rows) * (1 - frac)):]
agg, cnt = collections.Counter(), collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:90]
    agg[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    cnt[k] += 1
print("tail %.0f%% of the trace: %d launches, %.1f ms of kernel time" % (frac * 100, len(rows), sum(agg.values()) / 1e6))
for k, v in agg.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 25):
    print("%9.2f ms %6d  %s" % (v / 1e6, cnt[k], k))
