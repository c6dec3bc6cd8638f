#!/usr/bin/env python3
"""tools/probes/exp_finclock.py — where a k_finish wave spends its life (needs a library built with -DC8P_EXPERIMENTS -DC8P_EXP_FINCLOCK, loaded through
C8P_LIB: every wave then leaves its shader-clock deltas per phase in c8p_pkt_out instead of the results).  One 4096-packet cfg-2 batch at a time
(C8P_SLOTS=1).  Prints mean cycles per phase.
Build: source tools/hipflags.sh; $C8P_HIPCC -DC8P_EXPERIMENTS -DC8P_EXP_FINCLOCK -shared -o /tmp/libc8p_fin.so gr-ieee80211_amd/csrc/c8p_engine.hip"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["C8P_SLOTS"] = "1"
from c8p_cases import cfg2_workload_on_device
from gr_ieee80211_amd import engine
dev = torch.device("cuda", 0)
pk, iq_dev, chan_dev, expected, _keep = cfg2_workload_on_device(4096, 30.0, 0, torch, dev, 0)
eng = engine.Engine(device=0)
for i in range(20):
    eng.rx(pk, iq_dev, chan_dev)
eng.sync()
o = eng.outs()
tile, bits, desc, pdu = o["n_pdu"].astype(np.int64), o["n_crc_fail"].astype(np.int64), o["pdu_bytes"].astype(np.int64), o["rsv"].astype(np.int64)
t0 = o["bits_off"].astype(np.int64)
tot = tile + bits + desc + pdu
print("cycles per wave (mean / p10 / p90), s_memtime ticks at 100 MHz x ... as reported:")
for name, v in (("traceback tiles", tile), ("bits from snapshots", bits), ("descramble + table", desc), ("A-MPDU walk + CRC + PDU", pdu), ("total", tot)):
    print("  %-26s %9.0f %9.0f %9.0f" % (name, v.mean(), np.percentile(v, 10), np.percentile(v, 90)))
print("per CU and cycle during the tile phase (7 waves per CU, 7 tiles of 16 KB each): %.1f B" % (7 * 7 * 16384 / tile.mean() * (tile.mean() / tot.mean())))
