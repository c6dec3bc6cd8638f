#!/usr/bin/env python3
"""tools/soak.py — many full-size batches of distinct packets (c8p_tx on the device) at several SNRs through c8p_rx:
every published PDU must be the transmitted MPDU; a sample of every batch is compared with the oracle packet by packet
(status, tags, PDUs — including frames that fail the FCS)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
torch.zeros(1, device="cuda")
from c8p_cases import cfg2_workload_on_device
from c8p_testlib import abi, oracle
from gr_ieee80211_amd import engine
import ctypes

dev = torch.device("cuda", 0)
eng = engine.Engine(device=0)
hip = ctypes.CDLL("libamdhip64.so")
o = oracle()
tot = okc = 0
t0 = time.time()
for k, snr in enumerate([30.0, 26.0, 23.0, 21.0, 30.0, 24.0, 22.0, 20.0]):
    pk, iq_dev, chan_dev, exp, keep = cfg2_workload_on_device(4096, snr, 1000003 * (k + 1), torch, dev, 0)
    eng.rx(pk, iq_dev, chan_dev)
    res = eng.results(want_llr=False)
    n_ok = 0
    for r, e in zip(res, exp):
        assert r["pdus"] in ([], [e]), "a published PDU differs from the transmitted MPDU"
        n_ok += r["pdus"] == [e]
    # oracle on a sample (first 48 packets): identical PDUs and status even when the FCS fails
    iq = np.zeros(iq_dev.n, np.complex64); chan = np.zeros(chan_dev.n, np.complex64)
    torch.cuda.synchronize()
    hip.hipMemcpy(ctypes.c_void_p(iq.ctypes.data), ctypes.c_void_p(iq_dev.ptr), ctypes.c_size_t(iq.nbytes), ctypes.c_int(2))
    hip.hipMemcpy(ctypes.c_void_p(chan.ctypes.data), ctypes.c_void_p(chan_dev.ptr), ctypes.c_size_t(chan.nbytes), ctypes.c_int(2))
    for p in range(48):
        t, llr, mu = o.demod(pk, iq, chan, p)
        out, pdus, sb, ub = o.decode(t, llr, mu)
        assert res[p]["tags"]["status"] == t["status"] and res[p]["pdus"] == pdus, (k, p)
    tot += len(exp); okc += n_ok
    print("batch %d  snr %.0f dB: %d packets, %d FCS ok (%.1f%%), oracle sample identical" % (k, snr, len(exp), n_ok, 100.0 * n_ok / len(exp)), flush=True)
    del keep
print("soak ok: %d packets, %d PDUs, %.1f s" % (tot, okc, time.time() - t0))
