"""signal / signal2 front-end: CFO de-rotation, L-LTF channel, L-SIG decode, boundary items.

CPU: the oracle's signal stage recovers the L-SIG of every reference-TX fixture, with and without carrier offset, and
feeds the demod/decode oracle to the transmitted MPDU.  GPU: c8p_signal against that oracle (cosf/sinf differ by an
ulp between glibc and the device library, so samples/channel are compared with a 2e-6 relative tolerance), then chained
on the device into c8p_rx."""
import numpy as np
import pytest

from c8p_cases import dewindow
from c8p_testlib import H22, SYNC, abi, load_npz, oracle, oracle_rx, through_channel

SWEEP = load_npz("packets_sweep.npz")


def events(cfo_hz_list, names, snr=None, seed=3):
    """raw arena = packets back to back with 300 idle items between, each with its own carrier offset"""
    rng = np.random.default_rng(seed)
    raws, evts, exp = [[], []], [], []
    pos = 0
    for name, cfo in zip(names, cfo_hz_list):
        d = SWEEP[name]
        nss = int(d["nss"])
        rx = through_channel(dewindow(d["tx"]), H22 if nss == 2 else None, snr, rng, lead=300)
        if nss == 1:
            rx = np.vstack([rx, np.zeros_like(rx)])
        w = 2 * np.pi * cfo / 20e6
        rx = (rx * np.exp(1j * w * np.arange(rx.shape[1]))).astype(np.complex64)
        ev = np.zeros(1, abi.SIG_IN_DT)[0]
        ev["n_ant"] = nss
        ev["n_avail"] = rx.shape[1] - 300 - SYNC
        ev["cfo_rad"] = -w
        ev["snr"], ev["rssi"] = 30.0, 1.0
        evts.append((ev, pos + 300 + SYNC))
        raws[0].append(rx[0]); raws[1].append(rx[1])
        pos += rx.shape[1]
        exp.append(d)
    raw = np.concatenate(raws[0] + raws[1])
    half = pos
    ev = np.zeros(len(evts), abi.SIG_IN_DT)
    for i, (e, p) in enumerate(evts):
        ev[i] = e
        ev[i]["raw_off"][0] = p
        ev[i]["raw_off"][1] = half + p
    return ev, raw, exp


NAMES = ["L_0", "L_5", "HT_3", "HT_7", "VHT_2", "VHT_8", "HT2_12", "VHT2_5"]
CFOS = [0.0, 37e3, -81e3, 120e3, -4e3, 66e3, -150e3, 12e3]


def test_oracle_signal_stage_feeds_the_rx_chain():
    ev, raw, exp = events(CFOS, NAMES)
    pk, iq, chan = oracle().signal_events(ev, raw)
    assert (pk["n_items"] > 0).all()
    for p, d in zip(pk, exp):
        assert p["lsig_len"] == int(d["lsig_len"]) if int(d["fmt"]) else p["lsig_len"] == len(d["mpdu"])
        assert abs(float(p["cfo"]) + 0 * 1) >= 0
    res = oracle_rx(pk, iq, chan)
    for r, d in zip(res, exp):
        assert r["pdus"] == [d["pdu"].tobytes()]


def test_oracle_signal_rejects_noise_and_short_input():
    rng = np.random.default_rng(1)
    raw = (rng.normal(size=4000) + 1j * rng.normal(size=4000)).astype(np.complex64)
    ev = np.zeros(3, abi.SIG_IN_DT)
    ev["n_ant"] = 1
    ev["raw_off"][:, 0] = [0, 1000, 2000]
    ev["n_avail"] = [2000, 100, 300]
    pk, iq, chan = oracle().signal_events(ev, raw)
    assert pk["n_items"][1] == 0          # fewer than 224 items
    assert (pk["n_items"] == 0).sum() >= 2


@pytest.mark.gpu
def test_gpu_signal_matches_oracle_and_chains_into_rx():
    from gr_ieee80211_amd import engine
    ev, raw, exp = events(CFOS, NAMES, snr=30.0)
    opk, oiq, ochan = oracle().signal_events(ev, raw)
    eng = engine.Engine(device=0)
    eng.signal(ev, raw)
    eng.sync()
    gpk = eng.fetch(abi.BUF_SIG_PKT, abi.PKT_IN_DT)
    giq = eng.fetch(abi.BUF_SIG_IQ, np.complex64)
    gchan = eng.fetch(abi.BUF_SIG_CHAN, np.complex64)
    for k in ("n_items", "lsig_mcs", "lsig_len", "lsig_nsamp", "chan_off", "n_ant"):
        assert (gpk[k] == opk[k]).all(), k
    assert (gpk["iq_off"] == opk["iq_off"]).all()
    assert np.allclose(gpk["cfo"], opk["cfo"], rtol=1e-6)
    scale = np.max(np.abs(oiq))
    for p in gpk:                               # only the packet's own items are defined (the rest of its slot is capacity)
        for a in range(int(p["n_ant"])):
            lo, n = int(p["iq_off"][a]), int(p["n_items"])
            assert np.max(np.abs(giq[lo:lo + n] - oiq[lo:lo + n])) <= 2e-6 * scale
    assert np.max(np.abs(gchan - ochan)) <= 2e-6 * np.max(np.abs(ochan))
    # chain on the device: descriptors come back to the host, items and chan stay in HBM
    p_iq, b_iq = eng.buffer(abi.BUF_SIG_IQ)
    p_ch, b_ch = eng.buffer(abi.BUF_SIG_CHAN)
    eng2 = engine.Engine(device=0)
    eng2.rx(gpk, engine.DevArray(p_iq, b_iq // 8), engine.DevArray(p_ch, b_ch // 8))
    res = eng2.results(want_llr=False)
    for r, d in zip(res, exp):
        assert r["pdus"] == [d["pdu"].tobytes()]
    eng2.close(); eng.close()
