#!/usr/bin/env python3
"""tools/probes/chain_custom.py [max_batch] [io] [fused|plain|both|default] — the demod -> decode chain in bench.py's custom-buffer configuration alone (one
max_batch), for A/B runs under C8P_GATHER_DEPTH / C8P_GATHER_STATS / C8P_COPY_CUS and for kernel traces; prints the chain figures and the demod
thread's phases"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
from __graft_entry__ import load_package
load_package()
from gr_ieee80211_amd import blocks, hostbind
import bench_blocks as bb
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 512
io = sys.argv[2] if len(sys.argv) > 2 else "custom"
mode = sys.argv[3] if len(sys.argv) > 3 else "both"
passes = int(os.environ.get("PASSES", "8"))
n = int(os.environ.get("N_PKT", "4096"))
pk, iq, chan, exp = bb.host_batch(n)
chunk_in, nout_llr = bb.IO_SIZES[io]
pin = hostbind.cpu_pair_sharing_l3()
if pin:
    os.environ["C8P_CHAIN_PIN"] = "%d,%d" % tuple(pin)
out = {"n": n, "depth": os.environ.get("C8P_GATHER_DEPTH"), "max_batch": mb, "io": io, "pin": pin}
# default: nothing said about fused — the demod front decides from who reads its port (batching_cfg::fused_auto: fused, with lean tags
# unless C8P_FUSED_TAGS=full); "fused" forces it (the reference's full tag set on the edge)
for fused in ((False, True) if mode == "both" else ((None,) if mode == "default" else (mode == "fused",))):
    dem = blocks.Block(blocks.DEMOD, max_batch=mb, fused=fused)
    dec = blocks.Block(blocks.DECODE, max_batch=mb, fused=fused)
    ts, n_in = [], 0
    for rep in range(1 + passes):
        n_in = blocks.feed_demod(dem, pk, iq, chan, base=rep * n_in)
        t0 = time.perf_counter(); blocks.run_chain(dem, dec, chunk=chunk_in, nout=nout_llr); ts.append(time.perf_counter() - t0)
        if rep == 0:
            dem.host_phase_reset(); dec.host_phase_reset()
    ok = sum(1 for (ln, m), e in zip(dec.messages()[-n:], exp) if m == e)
    k = "default" if fused is None else ("fused" if fused else "plain")
    if fused is None:
        out["default_is_fused"] = dem.fused_state(); out["default_tags_per_packet"] = round(dem.tags_emitted() / (n * (1 + passes)), 2)
    out[k] = round(n / float(np.median(ts[1:]))); out[k + "_best"] = round(n / min(ts[1:])); out[k + "_worst"] = round(n / max(ts[1:])); out[k + "_ok"] = ok
    out[k + "_pass_ms"] = [round(1e3 * t, 2) for t in ts]
    out[k + "_demod"] = {q: round(v / (n * passes), 3) for q, v in dem.host_phase_us().items() if v / (n * passes) >= 0.02}
    out[k + "_decode_sum"] = round(sum(dec.host_phase_us().values()) / (n * passes), 2)
    dem.close(); dec.close()
print(json.dumps(out))
