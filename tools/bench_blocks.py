#!/usr/bin/env python3
"""tools/bench_blocks.py — throughput of the drop-in BLOCK chain (demod -> decode mirrors of host/c8p_blocks.cc on the HIP
engine), fed the way the GNU Radio scheduler feeds blocks: tagged host item streams in bounded chunks.  Unlike c8p_rx the
block boundary forces the LLR stream through host memory (demod's output buffer, decode's input buffer), so this is the
number a flowgraph sees, not the engine's.  Run on the GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from c8p_cases import cfg2_batch
from gr_ieee80211_amd import blocks

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
b, exp = cfg2_batch(n, snr_db=30.0, seed=1)
pk, iq, chan = b.arrays()
for max_batch in (64, 256, 1024):
    dem = blocks.Block(blocks.DEMOD, max_batch=max_batch)
    blocks.feed_demod(dem, pk, iq, chan)
    t0 = time.perf_counter()
    dem.run(chunk=1 << 16, nout=1 << 18)
    t1 = time.perf_counter()
    llr, tags = dem.output(), dem.tags()
    dec = blocks.Block(blocks.DECODE, max_batch=max_batch)
    dec.push_items(0, llr)
    for off, key, val in tags:
        dec.push_tag(off, key, val)
    t2 = time.perf_counter()
    dec.run(chunk=1 << 18, nout=1 << 16)
    t3 = time.perf_counter()
    ok = sum(1 for (ln, m), e in zip(dec.messages(), exp) if m == e)
    print({"packets": n, "max_batch": max_batch, "demod_block_packets_per_s": round(n / (t1 - t0)), "decode_block_packets_per_s": round(n / (t3 - t2)),
           "chain_packets_per_s": round(n / ((t1 - t0) + (t3 - t2))), "pdus_ok": ok})
    dem.close(); dec.close()
