#!/usr/bin/env python3
"""tools/probes/chain_quick.py [fused 0/1] [chunk] [nout] [max_batch] — one configuration of the demod -> decode block chain on the engine, 4 timed
passes, host phases; C8P_CHAIN_TLOG=1 prints where the two scheduler threads were"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
from __graft_entry__ import load_package
load_package()
from gr_ieee80211_amd import blocks
import bench_blocks as bb
fused = bool(int(sys.argv[1])) if len(sys.argv) > 1 else True
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 16
nout = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 18
mb = int(sys.argv[4]) if len(sys.argv) > 4 else 256
n = 4096
pk, iq, chan, exp = bb.host_batch(n)
dem = blocks.Block(blocks.DEMOD, max_batch=mb, fused=fused)
dec = blocks.Block(blocks.DECODE, max_batch=mb, fused=fused)
ts = []
n_in = 0
for rep in range(5):
    n_in = blocks.feed_demod(dem, pk, iq, chan, base=rep * n_in)
    t0 = time.perf_counter(); blocks.run_chain(dem, dec, chunk=chunk, nout=nout); ts.append(time.perf_counter() - t0)
    if rep == 0:
        dem.host_phase_reset(); dec.host_phase_reset()
dt = float(np.median(ts[1:]))
ok = sum(1 for (ln, m), e in zip(dec.messages()[-n:], exp) if m == e)
tot = 4 * n
print(json.dumps({"fused": fused, "chunk": chunk, "nout": nout, "max_batch": mb, "packets_per_s": round(n / dt), "wall_us": round(1e6 * dt / n, 3), "pdus_ok": ok,
                  "demod": {k: round(v / tot, 3) for k, v in dem.host_phase_us().items()},
                  "decode": {k: round(v / tot, 3) for k, v in dec.host_phase_us().items()}}))
dem.close(); dec.close()
