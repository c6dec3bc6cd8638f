#!/usr/bin/env python3
"""stdin: stderr of a run with C8P_DET_TLOG=1 / C8P_RAWCHAIN_TLOG=1 -> median, p90 and count per lap label"""
import re
import sys
from collections import defaultdict

import numpy as np

laps = defaultdict(list)
for ln in sys.stdin:
    m = re.match(r"\[c8p_detect\] (.*?)\s+([0-9.]+) us", ln)
    if m:
        laps[m.group(1).strip()].append(float(m.group(2)))
for k, v in laps.items():
    print("%-34s median %8.1f us   p90 %8.1f us   n %d" % (k, np.median(v), np.percentile(v, 90), len(v)))
