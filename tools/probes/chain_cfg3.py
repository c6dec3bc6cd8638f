#!/usr/bin/env python3
"""tools/probes/chain_cfg3.py [max_batch] — BASELINE config 3 (HT MCS15, two antennas) through the demod2 -> decode block mirrors on pinned buffers:
two ports, two pieces per packet in every gather"""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
from __graft_entry__ import load_package
load_package()
import torch
from gr_ieee80211_amd import blocks, hostbind
from gr_ieee80211_amd.workloads import workload_on_device
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n = 2048
dev = torch.device("cuda:0")
pk, iq_dev, chan_dev, exp, keep = workload_on_device("cfg3", n, 30.0, 0, torch, dev, 0)
iq = np.zeros(iq_dev.n, np.complex64); chan = np.zeros(chan_dev.n, np.complex64)
torch.cuda.synchronize(dev)
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemcpy(ctypes.c_void_p(iq.ctypes.data), ctypes.c_void_p(iq_dev.ptr), ctypes.c_size_t(iq.nbytes), ctypes.c_int(2))
hip.hipMemcpy(ctypes.c_void_p(chan.ctypes.data), ctypes.c_void_p(chan_dev.ptr), ctypes.c_size_t(chan.nbytes), ctypes.c_int(2))
pin = hostbind.cpu_pair_sharing_l3()
if pin:
    os.environ["C8P_CHAIN_PIN"] = "%d,%d" % tuple(pin)
out = {"n": n, "max_batch": mb, "bytes_per_packet_down": int(8 * iq_dev.n / n)}
for fused in (False, True):
    dem = blocks.Block(blocks.DEMOD2, max_batch=mb, fused=fused)
    dec = blocks.Block(blocks.DECODE, max_batch=mb, fused=fused)
    ts, n_in = [], 0
    for rep in range(6):
        n_in = blocks.feed_demod(dem, pk, iq, chan, base=rep * n_in)
        t0 = time.perf_counter(); blocks.run_chain(dem, dec, chunk=1 << 20, nout=1 << 21); ts.append(time.perf_counter() - t0)
        if rep == 0:
            dem.host_phase_reset(); dec.host_phase_reset()
    ok = sum(1 for (ln, m), e in zip(dec.messages()[-n:], exp) if m == e)
    k = "fused" if fused else "plain"
    out[k] = round(n / float(np.median(ts[1:]))); out[k + "_ok"] = ok
    out[k + "_demod"] = {q: round(v / (n * 5), 3) for q, v in dem.host_phase_us().items() if v / (n * 5) >= 0.02}
    dem.close(); dec.close()
print(json.dumps(out))
