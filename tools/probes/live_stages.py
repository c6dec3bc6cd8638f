#!/usr/bin/env python3
"""tools/probes/live_stages.py — the live ingest's per-round stage durations (c8p_rawchain_timeline), host-fed against device-fed:
does the bulk PCIe traffic slow the stages' own small transfers?"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
import torch  # noqa: E402
from gr_ieee80211_amd import blocks  # noqa: E402
from gr_ieee80211_amd.workloads import raw_stream_unit  # noqa: E402

dev = torch.device("cuda:0")
unit, _ = raw_stream_unit(0)
reps = (32 << 20) // len(unit)
pin = torch.from_numpy(np.tile(unit, reps).view(np.float32)).pin_memory()
devt = pin.to(dev)
torch.cuda.synchronize()
n = pin.numel() // 2
K, items, rounds = (int(sys.argv[1]) if len(sys.argv) > 1 else 64), (int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 19), 30
for feed in ("device", "host", "device", "host"):
    base = (pin if feed == "host" else devt).data_ptr()
    per = K * (items // len(unit) + 2)
    h = blocks.RawChain(device=0, streams=K, stream_round_items=items, reserve_seg_events=2 * per + per // 4, reserve_seg_packets=per, reserve_pkt_items=len(unit))
    h.live(None)
    off = [(s * 7919 * len(unit)) % (n - items * (rounds + 3)) for s in range(K)]
    t0 = time.perf_counter()
    for r in range(rounds):
        for s in range(K):
            h.push(s, (base + 8 * (off[s] + r * items), items), host=(feed == "host"))
        h.round(wait=False)
    st = h.drain()
    dt = time.perf_counter() - t0
    tl = h.timeline()[5:]
    med = lambda f: round(1e3 * float(np.median([f(m) for m in tl])), 3)
    print(json.dumps({"feed": feed, "ms_per_round": round(1e3 * dt / rounds, 3), "detect_ms": med(lambda m: m["det_end"] - m["det_begin"]),
                      "signal_ms": med(lambda m: m["sig_end"] - m["sig_begin"]), "rx_ms": med(lambda m: m["rx_done"] - m["rx_submit"]),
                      "queue_det_to_sig_ms": med(lambda m: m["sig_begin"] - m["det_end"]), "queue_sig_to_rx_ms": med(lambda m: m["rx_submit"] - m["sig_end"]),
                      "pdus": st["pdus"], "threads": {k: {f: round(v, 3) for f, v in st[k].items()} for k in ("detect", "signal", "rx")}}))
    h.close()
