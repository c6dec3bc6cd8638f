#!/usr/bin/env python3
"""tools/soak_rawchain.py [streams] [packets per stream] — host/c8p_rawchain.cc against the oracle's detect -> signal (+ scan rule) ->
demod + decode on random streams (every fixture format incl. two-stream ones and the NDP on two antennas, carrier offsets of +-150 kHz,
6-35 dB, gaps from 150 items): events, kept packets, PDUs and FCS-failure counts must agree.  Run on the GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from c8p_cases import dewindow
from c8p_testlib import H22, oracle, oracle_rx, through_channel
from gr_ieee80211_amd import blocks
from test_detect_frontend import SWEEP, scan_rule, sig_events

n_streams = int(sys.argv[1]) if len(sys.argv) > 1 else 40
per = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = torch.device("cuda:0")
o = oracle()
tot = {"streams": 0, "items": 0, "events": 0, "kept": 0, "pdus": 0, "fcs_fail": 0, "mismatch": 0}
t0 = time.time()
for n_ant in (1, 2):
    names = [k for k in SWEEP if (n_ant == 2 or int(SWEEP[k]["nss"]) == 1)]
    segs, hosts = [], []
    for sd in range(n_streams // 2):
        rng = np.random.default_rng(1000 * n_ant + sd)
        parts = [[], []]
        for k in range(per):
            d = SWEEP[names[int(rng.integers(len(names)))]]
            nss = int(d["nss"])
            Hc = None if n_ant == 1 else (H22 if nss == 2 else np.array([[1.0], [0.4 - 0.7j]], np.complex64))
            rx = np.atleast_2d(through_channel(np.atleast_2d(dewindow(d["tx"]))[: (nss if n_ant == 2 else 1)], Hc, float(rng.uniform(6.0, 35.0)), rng,
                                               lead=int(rng.choice([150, 260, 400, 900, 2500]))))
            w = 2 * np.pi * float(rng.uniform(-150e3, 150e3)) / 20e6
            rx = (rx * np.exp(1j * w * np.arange(rx.shape[1]))).astype(np.complex64)
            parts[0].append(rx[0]); parts[1].append(rx[1] if n_ant == 2 else rx[0])
        a0, a1 = np.concatenate(parts[0]), np.concatenate(parts[1])
        hosts.append((a0, a1))
        arena = np.concatenate([a0, a1]) if n_ant == 2 else a0
        segs.append((torch.from_numpy(arena.view(np.float32).copy()).to(dev), len(a0)))
    torch.cuda.synchronize()
    got = []
    blocks.rawchain_run([(t.data_ptr(), n, n if n_ant == 2 else 0) for t, n in segs], on_segment=got.append)
    for (a0, a1), g in zip(hosts, got):
        n = len(a0)
        arena = np.concatenate([a0, a1]) if n_ant == 2 else a0
        ev = o.detect(a0)[0]
        s = sig_events(ev, n, n_ant=n_ant)
        s["n_avail"] = np.minimum(s["n_avail"], 224 + 109280)
        if n_ant == 2:
            s["raw_off"][:, 1] = n + s["raw_off"][:, 0]
        pk, iq, chan = o.signal_events(s, arena)
        kept, code = scan_rule(s, pk)
        ok = g["n_events"] == len(ev) and len(g["pkts"]) == len(kept)
        if ok:
            for f in ("lsig_mcs", "lsig_len", "lsig_nsamp", "n_items"):
                ok = ok and np.array_equal(g["pkts"][f], pk[kept][f])
        ref = oracle_rx(pk[kept], iq, chan) if ok else []
        for r, oo in zip(ref, g["outs"]):
            rec = g["pdu_arena"][int(oo["pdu_off"]):int(oo["pdu_off"]) + int(oo["pdu_bytes"])].tobytes()
            want = b"".join(len(p).to_bytes(2, "little") + p for p in r["pdus"])
            if int(oo["n_pdu"]) != len(r["pdus"]) or int(oo["n_crc_fail"]) != int(r["out"]["n_crc_fail"]) or rec != want:
                ok = False
            tot["pdus"] += len(r["pdus"]); tot["fcs_fail"] += int(r["out"]["n_crc_fail"])
        tot["streams"] += 1; tot["items"] += n; tot["events"] += len(ev); tot["kept"] += len(kept); tot["mismatch"] += 0 if ok else 1
print("raw-chain soak: %(streams)d streams, %(items)d items, %(events)d sync events, %(kept)d packets kept, %(pdus)d PDUs, %(fcs_fail)d FCS failures, "
      "%(mismatch)d streams that differ from the oracle's chain" % tot, "(%.0f s)" % (time.time() - t0))
sys.exit(1 if tot["mismatch"] else 0)
