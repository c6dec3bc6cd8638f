#!/usr/bin/env python3
"""tools/probes/subbatch_regions.py — what filling and draining the engine's three-slot pipeline costs a SHORT timed region (the driver's
`--steps 20`), and what submitting the same 4096-packet step as 2 / 4 smaller c8p_rx calls would change: K steps between two syncs, K =
20 and 200, the step as 1 x 4096, 2 x 2048, 4 x 1024 packets.  Run on the GPU box."""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
torch.zeros(1, device="cuda")
from c8p_testlib import abi  # noqa
from gr_ieee80211_amd import engine
from gr_ieee80211_amd.workloads import workload_on_device

dev = torch.device("cuda", 0)
pk, A, C_, exp, keep = workload_on_device("cfg2", 4096, 30.0, 0, torch, dev, 0)
for slots in (3, 4, 6):
    os.environ["C8P_SLOTS"] = str(slots)
    eng = engine.Engine(device=0)
    for parts in (1, 2, 4):
        n = len(pk) // parts
        subs = [np.ascontiguousarray(pk[i * n:(i + 1) * n]) for i in range(parts)]
        for _ in range(30):
            for s in subs:
                eng.rx(s, A, C_)
        eng.sync()
        for K in (20, 200):
            ts = []
            for rep in range(40 if K == 20 else 8):
                eng.sync(); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(K):
                    for s in subs:
                        eng.rx(s, A, C_)
                eng.sync()
                ts.append(time.perf_counter() - t0)
            t = float(np.median(ts))
            print(json.dumps({"slots": slots, "calls_per_step": parts, "K": K, "ms_per_step": round(1e3 * t / K, 4), "packets_per_s": round(4096 * K / t)}), flush=True)
    eng.close()
