#!/usr/bin/env python3
"""tools/probes/rawchain_pcie.py — the host-fed raw chain (c8p_rawchain_run over segments in pinned host memory) for runs of 12 and 40 segments:
the steady-state figure of a 12-segment run is a window of 8 completions that arrive in bunches (55-70 GB/s "over PCIe"), 36 completions
and the whole run say 45-49.  Run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.getcwd())
from __graft_entry__ import load_package
load_package()
import numpy as np, torch
from gr_ieee80211_amd import blocks
from gr_ieee80211_amd.workloads import raw_stream_unit
unit, _ = raw_stream_unit(0)
reps = (32 << 20) // len(unit)
pin = torch.from_numpy(np.tile(unit, reps).view(np.float32)).pin_memory()
n = pin.numel() // 2
h = blocks.RawChain(device=0, reserve_seg_items=n, reserve_seg_events=2 * reps + reps // 4 + 64, reserve_seg_packets=reps + 64, reserve_pkt_items=len(unit), reserve_host=True)
for segs in (12, 40, 12, 40):
    hs = [(pin.data_ptr(), n)] * segs
    h.run(hs[:3], host=True)
    t0 = time.perf_counter()
    st = h.run(hs, host=True)
    dt = time.perf_counter() - t0
    print("segments %2d: wall %.1f ms = %.2f ms / segment = %.1f GB/s whole run; steady %d segments in %.2f ms = %.2f ms each = %.1f GB/s; pdus %d" % (
        segs, 1e3 * dt, 1e3 * dt / segs, 8.0 * n * segs / dt / 1e9, st["steady_segments"], 1e3 * st["steady_s"], 1e3 * st["steady_s"] / st["steady_segments"],
        8.0 * n * st["steady_segments"] / st["steady_s"] / 1e9, st["pdus"]))
h.close()
