#!/usr/bin/env python3
"""tools/probes/host_cost.py — host-side cost of one c8p_rx submit, and the step time with the per-kernel HIP events off / sampled /
on every launch (c8p_cfg.profile = 0 / 4 / 1), alternated in one process.  Run on the GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
torch.cuda.init(); dev = torch.device("cuda", 0)
from c8p_cases import cfg2_workload_on_device
from gr_ieee80211_amd import engine
pk, iq_dev, chan_dev, expected, _keep = cfg2_workload_on_device(4096, 30.0, 0, torch, dev, 0)
eng = engine.Engine(device=0, profile=0)
for _ in range(5): eng.rx(pk, iq_dev, chan_dev)
ts = []
for _ in range(20):
    eng.sync()
    t = time.perf_counter(); eng.rx(pk, iq_dev, chan_dev); ts.append(time.perf_counter() - t)
print("enqueue-only host time per c8p_rx call (ms): min %.3f median %.3f" % (1e3 * min(ts), 1e3 * sorted(ts)[10]))
eng.close()
for rep in range(3):
    for prof in (0, 4, 1):
        eng = engine.Engine(device=0, profile=prof)
        for _ in range(5): eng.rx(pk, iq_dev, chan_dev)
        eng.sync()
        t = time.perf_counter()
        for _ in range(60): eng.rx(pk, iq_dev, chan_dev)
        eng.sync()
        print("profile=%d: %.4f ms/step" % (prof, 1e3 * (time.perf_counter() - t) / 60))
        eng.close()
