#!/usr/bin/env python3
"""tools/probes/phase_account.py — where a k_rx_1ss_lds wave's time goes, by phase.  Needs C8P_LIB = a library built with
-DC8P_EXPERIMENTS -DC8P_EXP_PHASE=0: every wave leaves its s_memtime cycles — in all, demodulating symbols, in the block loop — in
C8P_BUF_DECSTAT.  Runs the bench's batch (4096 x cfg 2) alone (one slot: the exclusive launch) and three batches in flight (the bench's step)."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from __graft_entry__ import load_package
load_package()
from gr_ieee80211_amd import engine
from gr_ieee80211_amd.workloads import workload_on_device
dev = torch.device("cuda", 0)
pk, iq_dev, chan_dev, expected, _keep = workload_on_device("cfg2", 4096, 30.0, 0, torch, dev, 0)
out = {}
for slots in ("1", "3"):
    os.environ["C8P_SLOTS"] = slots
    eng = engine.Engine(device=0, profile=0)
    for i in range(40):
        eng.rx(pk, iq_dev, chan_dev)
    eng.sync()
    # the last submit's statistics (with 3 slots: a launch that shared the GPU with its neighbours)
    for i in range(12):
        eng.rx(pk, iq_dev, chan_dev)
    eng.sync()
    _, allc, packed = eng.decstat()
    allc = allc.astype(np.float64); packed = packed.view(np.uint32)
    dem = (packed >> 16).astype(np.float64) * 64.0; blk = (packed & 0xffff).astype(np.float64) * 64.0
    rest = allc - dem - blk
    out["slots_" + slots] = {"waves": int(len(allc)), "cycles_per_wave_median": float(np.median(allc)),
                             "demod_share": float(dem.sum() / allc.sum()), "block_loop_share": float(blk.sum() / allc.sum()), "rest_share": float(rest.sum() / allc.sum()),
                             "demod_cycles_per_symbol_median": float(np.median(dem) / 46.0), "block_loop_cycles_per_trellis_step_median": float(np.median(blk) / 11960.0),
                             "rest_cycles_median": float(np.median(rest))}
    eng.close()
print(json.dumps(out))
