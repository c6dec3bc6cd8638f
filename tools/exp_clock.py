#!/usr/bin/env python3
"""tools/exp_clock.py [packets] — per-wave shader-cycle account of the lean Viterbi kernel (needs a library built with -DC8P_EXPERIMENTS -DC8P_EXP_CLOCK:
every decoder wave then leaves its s_memtime delta and its HW_ID / XCC_ID in C8P_BUF_DECSTAT instead of the statistics; C8P_LIB=<that build>).
Prints cycles per trellis step per wave, how the waves were spread over the SIMDs, and in which order the waves of a SIMD finish.
Results on MI355X: profiles/r2b/decoder_wave_cycles.txt."""
import sys, os, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from c8p_cases import cfg2_workload_on_device
from gr_ieee80211_amd import engine
dev = torch.device("cuda", 0)
pk, iq_dev, chan_dev, expected, _keep = cfg2_workload_on_device(int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 30.0, 0, torch, dev, 0)
eng = engine.Engine(device=0, profile=1, keep_llr=True)     # the C8P_EXP_CLOCK instrumentation sits in k_rx_1ss (the arena path)
for i in range(30):
    eng.rx(pk, iq_dev, chan_dev); eng.sync()
eng.profile_reset()
eng.rx(pk, iq_dev, chan_dev); eng.sync()
_, cyc, hw = eng.decstat()
hw = hw.view(np.uint32)
kt = eng.kernel_times() if hasattr(eng, "kernel_times") else None
print("kernel times", kt)
cyc = cyc.astype(np.float64)
print("cycles per wave: min %.0f med %.0f max %.0f ; per step med %.1f max %.1f" % (cyc.min(), np.median(cyc), cyc.max(), np.median(cyc) / 11960, cyc.max() / 11960))
# HW_ID gfx9: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ... (gfx940: xcc in XCC_ID reg)
simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
xcc = (hw >> 16) & 15
key = xcc * 1024 + se * 64 + sh * 32 + cu * 4 + simd
import collections
cnt = collections.Counter(key.tolist())
print("distinct (se,sh,cu,simd) seen", len(cnt), "waves per key histogram", collections.Counter(cnt.values()))

percu = collections.Counter((key // 4).tolist())
print("CUs used", len(percu), "waves per CU histogram", sorted(collections.Counter(percu.values()).items()))
# cycles vs waves on the same SIMD
bysimd = collections.defaultdict(list)
for k, c in zip(key.tolist(), cyc.tolist()): bysimd[k].append(c)
for nw in sorted(set(len(v) for v in bysimd.values())):
    cs = [np.mean(v) for v in bysimd.values() if len(v) == nw]
    print("SIMDs with %d waves: %d, mean cycles/step %.1f" % (nw, len(cs), np.mean(cs) / 11960))
W = max(len(v) for v in bysimd.values())
srt = np.array([sorted(v) for v in bysimd.values() if len(v) == W])
print("per-SIMD sorted wave lifetimes (mean of 1st..4th to finish) cycles/step:", (srt.mean(axis=0) / 11960).round(1))
fin = srt[:, -1] / 11960
print("SIMD finish time cycles/step: min %.1f p10 %.1f med %.1f p90 %.1f max %.1f" % (fin.min(), np.percentile(fin, 10), np.median(fin), np.percentile(fin, 90), fin.max()))
xs = np.array([k // 1024 for k in bysimd.keys() if len(bysimd[k]) == W])
for x in sorted(set(xs.tolist())):
    print("  xcc %d: SIMD finish mean %.1f max %.1f" % (x, fin[xs == x].mean(), fin[xs == x].max()))
