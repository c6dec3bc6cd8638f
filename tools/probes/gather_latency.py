#!/usr/bin/env python3
"""tools/probes/gather_latency.py — what one c8p_copy_gather costs the caller and how long its items take, by size and direction:
host call time of gather / mark / wait right after, for pinned host -> device (the block mirrors' input) and device -> pinned host
(demod's LLR output), inline piece tables (16 pieces).  C8P_COPY_CUS in the environment selects the copy stream's CU mask."""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from __graft_entry__ import load_package
load_package()
import torch
from gr_ieee80211_amd import abi, engine

e = engine.Engine(device=0)
L = e.L
host = torch.empty(1 << 24, dtype=torch.float32).pin_memory()          # 64 MB
host.normal_()
dev = torch.zeros(1 << 24, dtype=torch.float32, device="cuda:0")
res = []
for direction in ("h2d", "d2h"):
    for kb in (64, 256, 512, 1024, 2048, 4096):
        n_f = kb * 256                       # floats
        npc = 16
        per = n_f // npc
        t_g, t_m, t_w = [], [], []
        for rep in range(30):
            base = (rep % 8) * n_f
            if direction == "h2d":
                pcs = [abi.Piece(host.data_ptr() + 4 * (base + i * per), base + i * per, per, 1, abi.FMT_F32, 1.0, 0) for i in range(npc)]
                dst = dev.data_ptr()
            else:
                pcs = [abi.Piece(dev.data_ptr() + 4 * (base + i * per), base + i * per, per, 1, abi.FMT_F32, 1.0, 0) for i in range(npc)]
                dst = host.data_ptr()
            arr = (abi.Piece * npc)(*pcs)
            t0 = time.perf_counter()
            L.c8p_copy_gather(e.ctx, C.c_void_p(dst), arr, npc)
            t1 = time.perf_counter()
            m = L.c8p_copy_mark(e.ctx)
            t2 = time.perf_counter()
            L.c8p_copy_wait_mark(e.ctx, m)
            t3 = time.perf_counter()
            if rep >= 5:
                t_g.append(t1 - t0); t_m.append(t2 - t1); t_w.append(t3 - t2)
        r = {"dir": direction, "KB": kb, "gather_call_us": round(1e6 * float(np.median(t_g)), 1), "mark_us": round(1e6 * float(np.median(t_m)), 1),
             "wait_us": round(1e6 * float(np.median(t_w)), 1)}
        r["total_us"] = round(r["gather_call_us"] + r["mark_us"] + r["wait_us"], 1)
        r["GBps_incl_latency"] = round(kb * 1024 / (r["total_us"] * 1e-6) / 1e9, 1)
        res.append(r)
        print(json.dumps(r), flush=True)
# the wait for an already finished mark
m = L.c8p_copy_mark(e.ctx); L.c8p_copy_wait_mark(e.ctx, m)
t0 = time.perf_counter()
for _ in range(100):
    L.c8p_copy_wait_mark(e.ctx, m)
print(json.dumps({"wait_on_finished_mark_us": round(1e4 * (time.perf_counter() - t0), 2)}))
e.close()
