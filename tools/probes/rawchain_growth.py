#!/usr/bin/env python3
"""tools/probes/rawchain_growth.py — which arenas still grow inside a raw-chain run after c8p_rawchain_create reserved them
(C8P_GROW_LOG=1 prints every growth): bench.py's leg in small."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["C8P_GROW_LOG"] = "1"
from __graft_entry__ import load_package  # noqa: E402

load_package()
import torch  # noqa: E402
from gr_ieee80211_amd import blocks, engine  # noqa: E402
from gr_ieee80211_amd.workloads import raw_stream_unit  # noqa: E402

dev = torch.device("cuda:0")
unit, _ = raw_stream_unit(0)
reps = (32 << 20) // len(unit)
x = torch.from_numpy(unit.view(np.float32)).to(dev).repeat(reps)
n = x.numel() // 2
arr = engine.DevArray(x.data_ptr(), n, x)
print("create", file=sys.stderr)
h = blocks.RawChain(device=0, reserve_seg_items=n, reserve_seg_events=2 * reps + reps // 4 + 64, reserve_seg_packets=reps + 64, reserve_pkt_items=len(unit), reserve_host=True)
for k in range(3):
    print("run %d" % k, file=sys.stderr)
    t = time.perf_counter()
    st = h.run([(arr.ptr, n)] * 8)
    print("run %d: %.2f ms, growths %d, events %d packets %d" % (k, 1e3 * (time.perf_counter() - t), st["arena_growths"], st["events"], st["packets"]), file=sys.stderr)
h.close()
