#!/usr/bin/env python3
"""tools/probes/link_timeline.py <kernel_trace.csv> [t0_ms] [len_ms] — every kernel of a stretch of a rocprofv3 kernel trace as one line (start, duration, name):
what the device did, and when the link kernels (k_copy_gather*) did not run"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
link = [r for r in rows if r["Kernel_Name"].startswith("k_copy_gather")]
# the longest pass: the stretch of link kernels without a 2 ms gap that has the most of them
best, cur = [], [link[0]]
for r in link[1:]:
    if int(r["Start_Timestamp"]) - int(cur[-1]["End_Timestamp"]) > 2_000_000:
        if len(cur) > len(best): best = cur
        cur = [r]
    else: cur.append(r)
if len(cur) > len(best): best = cur
t_begin = int(best[0]["Start_Timestamp"])
t0 = t_begin + int(1e6 * float(sys.argv[2] if len(sys.argv) > 2 else 6.0))
t1 = t0 + int(1e6 * float(sys.argv[3] if len(sys.argv) > 3 else 1.5))
print("pass of %d link kernels, %.2f ms; showing %.2f .. %.2f ms" % (len(best), (int(best[-1]["End_Timestamp"]) - t_begin) / 1e6, (t0 - t_begin) / 1e6, (t1 - t_begin) / 1e6))
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e < t0 or s > t1: continue
    print("%9.1f us  +%7.1f us  %s" % ((s - t_begin) / 1e3, (e - s) / 1e3, r["Kernel_Name"].split("(")[0][:40]))
