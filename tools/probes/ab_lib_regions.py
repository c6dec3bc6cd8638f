#!/usr/bin/env python3
"""tools/probes/ab_lib_regions.py libA.so libB.so ... — the headline step (4096 cfg-2 packets, c8p_rx, 3 slots) on several builds of the engine
library, alternated in one process tree (one subprocess per measurement): K = 20 and K = 200 steps between syncs.  AB_KIND = the workload (cfg2;
random1: mixed formats, rates and lengths).  Run on the GPU box."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import os, sys, time, json
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import torch
torch.zeros(1, device="cuda")
from __graft_entry__ import load_package
load_package()
from gr_ieee80211_amd import engine
from gr_ieee80211_amd.workloads import workload_on_device
dev = torch.device("cuda", 0)
kind = os.environ.get("AB_KIND", "cfg2")
pk, A, C_, exp, keep = workload_on_device(kind, 4096, 30.0, 0, torch, dev, 0)
eng = engine.Engine(device=0)
for _ in range(60): eng.rx(pk, A, C_)
eng.sync()
res = eng.results(want_llr=False)
ok = sum(1 for r, e in zip(res, exp) if r["pdus"] == [e])
out = {"lib": os.path.basename(engine.LIB_PATH), "kind": kind, "pdus_ok": ok}
for K, reps in ((20, 60), (200, 10)):
    ts = []
    for rep in range(reps):
        eng.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K): eng.rx(pk, A, C_)
        eng.sync()
        ts.append(time.perf_counter() - t0)
    out["K%%d_ms_per_step" %% K] = round(1e3 * float(np.median(ts)) / K, 4)
print(json.dumps(out))
''' % (ROOT, ROOT)
libs = sys.argv[1:]
for rnd in range(3):
    for lib in libs:
        env = dict(os.environ, C8P_LIB=os.path.abspath(lib))
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        print(r.stdout.strip().split("\n")[-1] if r.returncode == 0 else "FAILED: " + r.stderr[-400:], flush=True)
