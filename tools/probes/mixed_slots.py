#!/usr/bin/env python3
"""tools/probes/mixed_slots.py — the mixed-format batch (bench.py sensitivity.mixed_formats) by number of engine slots and by packet
order inside the batch: packets/s and trellis steps/s"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from __graft_entry__ import load_package
load_package()
import torch
from gr_ieee80211_amd import engine
from gr_ieee80211_amd.workloads import workload_on_device
dev = torch.device("cuda:0")
for kind in ("random1", "cfg2"):
    pk, A, C_, exp, keep = workload_on_device(kind, 4096, 30.0, 0, torch, dev, 0)
    orders = {"arrival": pk, "longest_first": pk[np.argsort(-pk["n_items"].astype(np.int64), kind="stable")]}
    for oname, pko in orders.items():
        pko = np.ascontiguousarray(pko)
        for slots in (3, 4, 6, 8):
            os.environ["C8P_SLOTS"] = str(slots)
            eng = engine.Engine(device=0)
            eng.rx(pko, A, C_); eng.sync()
            tr = eng.tags()["trellis"].astype(np.int64)
            t = time.perf_counter()
            while time.perf_counter() - t < 0.1:
                eng.rx(pko, A, C_)
            eng.sync()
            k = 300
            t = time.perf_counter()
            for _ in range(k):
                eng.rx(pko, A, C_)
            eng.sync()
            dt = time.perf_counter() - t
            eng.close()
            print(json.dumps({"workload": kind, "order": oname, "slots": slots, "packets_per_s": round(4096 * k / dt), "G_trellis_steps_per_s": round(float(tr.sum()) * k / dt / 1e9, 2),
                              "ms_per_step": round(1e3 * dt / k, 4)}), flush=True)
