"""Shared helpers for tests / bench / smoke: fixtures -> boundary inputs, oracle and engine drivers."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")

from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
abi = pkg.abi

SYNC = 184            # L-LTF starts at sample 160 of a packet; signal_impl expects the sync flag at LTF + 24
CONSUMED = 224        # signal_impl.cc:154 — items consumed before the first item handed to demod
PAD = 320             # signal_impl.cc:194-202


def bench_legs():
    """tools/bench_legs.py as a module (bench.py's legs: sharded_stream, usable_cores, ...)"""
    tools = os.path.join(ROOT, "tools")
    if tools not in sys.path:
        sys.path.insert(0, tools)
    import bench_legs as m
    return m


def oracle():
    from oracle import oracle_py
    return oracle_py


def load_npz(name):
    z = np.load(os.path.join(GOLD, name))
    groups = {}
    for k in z.files:
        if "/" in k:
            g, f = k.split("/", 1)
            groups.setdefault(g, {})[f] = z[k]
        else:
            groups[k] = z[k]
    return groups


from gr_ieee80211_amd.workloads import awgn, through_channel  # noqa: E402,F401


def through_multipath(tx, taps, snr_db=None, rng=None, sig_amp=0.1875):
    """tx: [nss][N] -> rx [n_rx][N + PAD + 160]; taps[r][s] = list of (delay in samples, complex gain): a frequency-selective
    channel whose echoes stay inside the guard interval, so every subcarrier sees its own H[k]."""
    tx = np.atleast_2d(tx)
    nss, n = tx.shape
    n_rx = len(taps)
    rx = np.zeros((n_rx, n + PAD + 160), np.complex64)
    for r in range(n_rx):
        for s_ in range(nss):
            for d, g in taps[r][s_]:
                rx[r, d:d + n] += (np.complex64(g) * tx[s_]).astype(np.complex64)
    if snr_db is not None:
        sigma = np.sqrt(sig_amp ** 2 / (10 ** (snr_db / 10.0)))
        rx = rx + awgn(rx.shape, sigma, rng)
    return rx.astype(np.complex64)


class Batch:
    """Accumulates packets at the signal->demod boundary into the arenas of c8p_batch_in."""

    def __init__(self):
        self.pk, self.iq, self.chan = [], [], []
        self.n_iq = 0

    def add_boundary(self, iq_ports, chan, lsig_mcs, lsig_len, lsig_nsamp, mupos=0, seq=0, cfo=0.0, snr=30.0, rssi=1.0):
        rec = np.zeros(1, abi.PKT_IN_DT)[0]
        n_items = len(iq_ports[0])
        for a, x in enumerate(iq_ports):
            assert len(x) == n_items
            rec["iq_off"][a] = self.n_iq
            self.iq.append(np.ascontiguousarray(x, np.complex64))
            self.n_iq += n_items
        rec["n_items"] = n_items
        rec["chan_off"] = 64 * len(self.chan)
        rec["lsig_mcs"], rec["lsig_len"], rec["lsig_nsamp"] = lsig_mcs, lsig_len, lsig_nsamp
        rec["seq"], rec["cfo"], rec["snr"], rec["rssi"] = seq, cfo, snr, rssi
        rec["n_ant"], rec["mupos"] = len(iq_ports), mupos
        self.chan.append(np.ascontiguousarray(chan, np.complex64))
        self.pk.append(rec)
        return len(self.pk) - 1

    def add_rx(self, rx, n_ant=1, mupos=0, lead=0):
        """rx: [n_rx][N] received stream; runs the signal front-end (oracle restatement of signal_impl,
        cfo 0) at the sync index to get the tags, then slices the boundary items."""
        s = lead + SYNC
        ok, h, mcs, ln, nsamp, _, _ = oracle().signal(rx[0][s:s + CONSUMED])
        if not ok:
            return -1
        start = s + CONSUMED
        n_items = nsamp + PAD
        ports = []
        for a in range(n_ant):
            x = np.zeros(n_items, np.complex64)
            avail = rx[a][start:start + n_items]
            x[:len(avail)] = avail
            ports.append(x)
        return self.add_boundary(ports, h, mcs, ln, nsamp, mupos=mupos, seq=len(self.pk) + 1)

    def arrays(self):
        pk = np.array(self.pk, abi.PKT_IN_DT) if self.pk else np.zeros(0, abi.PKT_IN_DT)
        iq = np.concatenate(self.iq) if self.iq else np.zeros(0, np.complex64)
        chan = np.concatenate(self.chan) if self.chan else np.zeros(0, np.complex64)
        return pk, iq, chan


def oracle_rx(pk, iq, chan):
    """Full oracle pass; returns per packet dict(tags, llr, mu, out, pdus, sbits, bits)."""
    o = oracle()
    res = []
    for p in range(len(pk)):
        tags, llr, mu = o.demod(pk, iq, chan, p)
        out, pdus, sb, ub = o.decode(tags, llr, mu)
        res.append(dict(tags=tags, llr=llr, mu=mu, out=out, pdus=pdus, sbits=sb, bits=ub))
    return res


H22 = np.array([[1, 0.3j], [0.2, 0.9]], np.complex64)   # SURVEY §8d cfg 3 fixture channel
