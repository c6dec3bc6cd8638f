#!/usr/bin/env python3
"""tools/probes/submit_cost.py — host time of one c8p_rx call (enqueue only: device-resident arenas, results not fetched) by batch size, the batch
struct built once; and of the c8p_fetch_gather behind it.  What the block fronts' `submit` / `enqueue_downloads` phases pay per batch."""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from __graft_entry__ import load_package
load_package()
import torch
from gr_ieee80211_amd import abi, engine
from gr_ieee80211_amd.workloads import workload_on_device
dev = torch.device("cuda:0")
pk, iq_dev, chan_dev, exp, keep = workload_on_device("cfg2", 4096, 30.0, 0, torch, dev, 0)
for n in (256, 512, 1024, 4096):
    eng = engine.Engine(device=0)
    b = eng._batch(pk[:n], iq_dev, chan_dev)
    for _ in range(8):
        eng._chk(eng.L.c8p_rx(eng.ctx, C.byref(b)))
    eng.sync()
    ts = []
    for _ in range(40):
        eng.sync()
        t = time.perf_counter(); eng.L.c8p_rx(eng.ctx, C.byref(b)); ts.append(time.perf_counter() - t)
    # three in flight, as the fronts keep them: the call that takes a slot finds its stream idle or not
    tp = []
    for _ in range(40):
        t = time.perf_counter(); eng.L.c8p_rx(eng.ctx, C.byref(b)); tp.append(time.perf_counter() - t)
        time.sleep(0.0015)
    eng.sync()
    print(json.dumps({"n_pkt": n, "c8p_rx_call_us_median": round(1e6 * float(np.median(ts)), 1), "min": round(1e6 * min(ts), 1),
                      "per_packet_us": round(1e6 * float(np.median(ts)) / n, 3), "paced_call_us_median": round(1e6 * float(np.median(tp)), 1)}), flush=True)
    eng.close()
