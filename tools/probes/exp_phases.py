#!/usr/bin/env python3
"""tools/probes/exp_phases.py — where does the pipelined step go?  Saturated (software-pipelined over the engine's slots) throughput of
c8p_rx, of c8p_demod alone and of c8p_decode alone on the cfg-2 batch, resident in HBM.  If rx = demod + decode the two halves
compete for the same resource; a decode-only time well above its VALU bound says the decoder does not fill the machine by itself."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from c8p_cases import cfg2_workload_on_device
from gr_ieee80211_amd import engine, abi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = 100
dev = torch.device("cuda", 0)
pk, iq_dev, chan_dev, expected, _keep = cfg2_workload_on_device(n, 30.0, 0, torch, dev, 0)
eng = engine.Engine(device=0, profile=0)
eng.demod(pk, iq_dev, chan_dev); eng.sync()
tags = eng.tags()
llr = eng.fetch(abi.BUF_LLR, np.float32)
llr_d = torch.from_numpy(llr).to(dev)
llr_dev = engine.DevArray(llr_d.data_ptr(), llr.size, llr_d)

def timed(f):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.2: f()
    eng.sync()
    best = 1e9
    for r in range(3):
        t0 = time.perf_counter()
        for _ in range(steps): f()
        eng.sync()
        best = min(best, (time.perf_counter() - t0) / steps * 1e3)
    return best
r = timed(lambda: eng.rx(pk, iq_dev, chan_dev))
d = timed(lambda: eng.demod(pk, iq_dev, chan_dev))
v = timed(lambda: eng.decode(tags, llr_dev))
print("slots %s batch %d: rx %.4f ms  demod-only %.4f ms  decode-only %.4f ms  (demod + decode = %.4f)" % (os.environ.get("C8P_SLOTS", "default"), n, r, d, v, d + v))
