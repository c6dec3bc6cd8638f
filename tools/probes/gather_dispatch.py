#!/usr/bin/env python3
"""tools/probes/gather_dispatch.py <dir with rocprofv3 csv (--hip-runtime-trace --kernel-trace)> — for the gather kernels: host launch call -> kernel start,
kernel duration, and what ran on the device in between"""
import csv, glob, sys
import numpy as np
d = sys.argv[1]
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
ht = glob.glob(d + "/**/*hip_api_trace.csv", recursive=True)[0]
api = {}
for r in csv.DictReader(open(ht)):
    if "LaunchKernel" in r["Function"]:
        api[r["Correlation_Id"]] = (int(r["Start_Timestamp"]), int(r["End_Timestamp"]))
rows = list(csv.DictReader(open(kt)))
for name in ("k_copy_gather(", "k_copy_gather_w", "k_rx_1ss_lds", "k_set_flag"):
    lat, dur, call = [], [], []
    for r in rows:
        if not r["Kernel_Name"].startswith(name): continue
        a = api.get(r["Correlation_Id"])
        if a is None: continue
        lat.append((int(r["Start_Timestamp"]) - a[1]) / 1e3); dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3); call.append((a[1] - a[0]) / 1e3)
    if not lat: continue
    lat, dur, call = np.array(lat), np.array(dur), np.array(call)
    print("%-18s n %5d | launch call us: median %.1f | call end -> kernel start us: median %.1f p75 %.1f p90 %.1f max %.1f | duration us: median %.1f p90 %.1f" %
          (name, len(lat), np.median(call), np.median(lat), np.percentile(lat, 75), np.percentile(lat, 90), lat.max(), np.median(dur), np.percentile(dur, 90)))
