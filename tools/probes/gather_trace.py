#!/usr/bin/env python3
"""tools/probes/gather_trace.py <kernel_trace.csv> — the gather kernels of a rocprofv3 kernel trace of tools/probes/chain_custom.py: durations by kernel, and
how much of the time at least one of them was on the device (the link's duty cycle as the shader cores see it) inside the busy stretches of the run"""
import csv, sys
import numpy as np
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].startswith("k_copy_gather")]
byname = {}
for r in rows:
    byname.setdefault(r["Kernel_Name"].split("(")[0], []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for k, v in sorted(byname.items()):
    d = np.array([e - s for s, e in v]) / 1e3
    print("%-18s n %5d  duration us: median %.1f mean %.1f p90 %.1f max %.1f  total %.1f ms" % (k, len(d), np.median(d), d.mean(), np.percentile(d, 90), d.max(), d.sum() / 1e3))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
# passes = stretches without a gap of more than 2 ms between link kernels
passes, cur = [], [iv[0]]
for a in iv[1:]:
    if a[0] - max(e for _, e in cur) > 2_000_000: passes.append(cur); cur = [a]
    else: cur.append(a)
passes.append(cur)
for p in passes:
    if len(p) < 50: continue
    span = max(e for _, e in p) - p[0][0]
    busy, end = 0, p[0][0]
    for s, e in p:
        if e <= end: continue
        busy += e - max(s, end); end = e
    print("pass of %5d link kernels: %.2f ms, at least one on the device %.2f of the time" % (len(p), span / 1e6, busy / span))
