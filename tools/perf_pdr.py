#!/usr/bin/env python3
"""tools/perf_pdr.py — packet delivery ratio against SNR, the statistical test the reference runs with
tools/performance/perf_siso.py (100 packets x MCS x SNR 0..30 dB, signal amplitude 0.1875, Gaussian noise of
sqrt(amp^2 / 10^(SNR/10)), CRC-pass counters), here with every stage on the MI355X:

    c8p_tx (100 packets per MCS, 500-byte payload MPDUs, multiplier 12, 1200-item gaps, the toolbox's symbol windowing)
      -> AWGN (torch, on the device) -> c8p_detect -> c8p_signal -> c8p_rx -> FCS-correct PDUs per MCS

Prints, per format, one row per SNR with the delivered-packet counts per MCS — the table perf_siso.py prints.
    python tools/perf_pdr.py [--packets 100] [--snr 0 31 1] [--no-window]"""
import argparse
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(fmt, n_mcs, packets, snrs, torch, dev, window=True, seed=1, nss=1):
    from c8p_cases import ampdu_vht
    from c8p_testlib import abi
    from gr_ieee80211_amd import engine
    from test_detect_frontend import consume_filter
    rng = np.random.default_rng(seed)
    body = bytes(rng.integers(0, 256, size=24 + 8 + 28 + 500, dtype=np.uint8))        # MAC + LLC + IP/UDP headers + 500-byte payload
    mpdu = body + zlib.crc32(body).to_bytes(4, "little")
    ps = ampdu_vht(mpdu) if fmt == abi.F_VHT else mpdu
    n = n_mcs * packets
    pk = np.zeros(n, abi.TX_PKT_DT)
    pk["psdu_off"], pk["len"], pk["format"], pk["scr_seed"], pk["gap"], pk["nss"] = 0, len(ps), fmt, 93, 1200, nss
    pk["mult"] = 12.0 * (np.sqrt(2.0) if nss == 2 else 1.0)        # tools/pktGenExample.py:187,210
    pk["mcs"] = np.repeat(np.arange(n_mcs), packets)
    pk["flags"] = 0 if window else abi.TX_NO_WINDOW
    e_tx = engine.Engine(device=0)
    e_tx.tx(pk, np.frombuffer(ps, np.uint8), tail_gap=1200)
    e_tx.sync()
    tx = torch.from_numpy(e_tx.fetch(abi.BUF_TX_IQ, np.complex64)).to(dev)
    e_tx.close()
    if nss == 2:                                                   # flat 2x2 channel (SURVEY.md §8d cfg 3), antenna streams back to back
        from c8p_testlib import H22
        h = torch.tensor(H22, device=dev)
        half = tx.numel() // 2
        t0_, t1_ = tx[:half], tx[half:]
        tx = torch.cat([h[0, 0] * t0_ + h[0, 1] * t1_, h[1, 0] * t0_ + h[1, 1] * t1_])
    clean = torch.view_as_real(tx).contiguous()
    ntot = clean.shape[0] // nss                                   # items per antenna
    want = bytes([fmt, len(mpdu) % 256, len(mpdu) // 256]) + mpdu
    mcs_tag = 8 if (fmt == abi.F_HT and nss == 2) else 0           # HT two-stream rates are MCS 8-15
    g = torch.Generator(device=dev)
    g.manual_seed(13579)
    rows = []
    e_det, e_sig, e_rx = engine.Engine(device=0), engine.Engine(device=0), engine.Engine(device=0)
    for snr in snrs:
        sigma = float(np.sqrt(0.1875 ** 2 / (10.0 ** (snr / 10.0))))
        noisy = (clean + torch.randn(ntot * nss, 2, generator=g, device=dev, dtype=torch.float32) * (sigma / np.sqrt(2))).contiguous()
        x = engine.DevArray(noisy.data_ptr(), ntot * nss, noisy)
        ev = e_det.detect(engine.DevArray(noisy.data_ptr(), ntot, noisy))     # the detector listens on antenna 0
        counts = [0] * n_mcs
        if len(ev):
            s = np.zeros(len(ev), abi.SIG_IN_DT)
            s["raw_off"][:, 0] = ev["index"]
            s["raw_off"][:, 1] = ntot + ev["index"]
            s["n_avail"] = ntot - ev["index"]
            s["cfo_rad"], s["snr"], s["rssi"], s["n_ant"] = ev["rad"], ev["snr"], ev["rssi"], nss
            e_sig.signal(s, x)
            e_sig.sync()
            gpk = e_sig.fetch(abi.BUF_SIG_PKT, abi.PKT_IN_DT)
            keep = consume_filter(ev, gpk)
            if keep:
                pi, bi = e_sig.buffer(abi.BUF_SIG_IQ)
                pc, bc = e_sig.buffer(abi.BUF_SIG_CHAN)
                e_rx.rx(gpk[keep], engine.DevArray(pi, bi // 8), engine.DevArray(pc, bc // 8))
                for r in e_rx.results(want_llr=False):
                    for p in r["pdus"]:
                        if p[:-1] == want and 0 <= p[-1] - mcs_tag < n_mcs:
                            counts[p[-1] - mcs_tag] += 1
        rows.append(counts)
    for e in (e_det, e_sig, e_rx):
        e.close()
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--packets", type=int, default=100)
    ap.add_argument("--snr", type=int, nargs=3, default=[0, 31, 1])
    ap.add_argument("--nss", type=int, default=1, help="2: two spatial streams through a flat 2x2 channel into signal2 / demod2 (perf_sumimo.py)")
    ap.add_argument("--no-window", action="store_true", help="clean OFDM symbols instead of the toolbox's half-amplitude junction samples")
    a = ap.parse_args()
    import torch
    torch.cuda.init()
    dev = torch.device("cuda", 0)
    from c8p_testlib import abi
    snrs = list(range(*a.snr))
    t0 = time.time()
    for name, fmt, n_mcs in (("legacy", abi.F_L, 8), ("ht", abi.F_HT, 8), ("vht", abi.F_VHT, 9)):
        if a.nss == 2 and fmt == abi.F_L:
            continue
        rows = run(fmt, n_mcs, a.packets, snrs, torch, dev, window=not a.no_window, nss=a.nss)
        print("%s: delivered of %d per MCS 0..%d" % (name, a.packets, n_mcs - 1))
        for snr, r in zip(snrs, rows):
            print("  %2d dB  %s" % (snr, " ".join("%3d" % c for c in r)))
    print("%.1f s" % (time.time() - t0))


if __name__ == "__main__":
    main()
