#!/usr/bin/env python3
"""tools/probes/gather_trace.py <kernel_trace.csv> — the gather kernels of a rocprofv3 kernel trace of tools/probes/chain_custom.py: durations, the gaps
between consecutive ones, and how busy their stream was"""
import csv, sys
import numpy as np
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].startswith("k_copy_gather")]
names = sorted(set(r["Kernel_Name"].split("(")[0] for r in rows))
s = np.array([int(r["Start_Timestamp"]) for r in rows]); e = np.array([int(r["End_Timestamp"]) for r in rows])
o = np.argsort(s); s, e = s[o], e[o]
d = (e - s) / 1e3
gap = (s[1:] - e[:-1]) / 1e3
per = (s[1:] - s[:-1]) / 1e3
busy = per < 200            # consecutive gathers of one pass (not the pauses between passes)
print("kernels", names, "n", len(d))
print("duration us: median %.1f mean %.1f p90 %.1f max %.1f" % (np.median(d), d.mean(), np.percentile(d, 90), d.max()))
print("start-to-start us (inside a pass): median %.1f  ->  stream busy %.2f" % (np.median(per[busy]), d[1:][busy].sum() / per[busy].sum()))
print("gap end-to-start us: median %.1f p10 %.1f" % (np.median(gap[busy]), np.percentile(gap[busy], 10)))
