"""Transmit chain, one spatial stream (SURVEY.md §8f-4 "TX-lite").

CPU: the oracle's restatement of the reference TX reproduces the reference toolbox's own output — IQ and punctured coded
bits of all 31 single-stream fixtures (legacy MCS 0-7, HT MCS 0-7, VHT MCS 0-8, BASELINE cfg 1 / 2 / 4) — and closes the
loop through the detect / signal / demod / decode oracle.
GPU: c8p_tx against that oracle (coded bits identical, IQ within 1e-6 of the packet's peak: the IFFT runs in fp64 on both
sides, only the summation order differs), then a device-resident loopback c8p_tx -> AWGN -> c8p_detect -> c8p_signal ->
c8p_rx whose PDUs must be the transmitted MPDUs."""
import os
import zlib

import numpy as np
import pytest

from c8p_cases import ampdu_vht
from c8p_testlib import abi, load_npz, oracle, oracle_rx

FIX, FIX2 = {}, {}
for _fn in ("packets_sweep.npz", "packets_cfg.npz"):
    for _name, _d in load_npz(_fn).items():
        if int(_d["nss"]) == 1 and len(_d["mpdu"]):
            FIX[_name] = _d
        elif int(_d["nss"]) == 2:
            FIX2[_name] = _d                   # HT MCS 8-15, VHT 2 streams MCS 0-8, the VHT NDP, BASELINE cfg 3


def psdu_of(fmt, mpdu):
    return ampdu_vht(mpdu) if fmt == abi.F_VHT else bytes(mpdu)


def mpdu_with_fcs(rng, n):
    body = bytes(rng.integers(0, 256, size=n - 4, dtype=np.uint8))
    return body + zlib.crc32(body).to_bytes(4, "little")


@pytest.mark.parametrize("name", sorted(FIX))
def test_oracle_tx_reproduces_the_reference_toolbox(name):
    d = FIX[name]
    fmt, mcs = int(d["fmt"]), int(d["mcs"])
    ps = psdu_of(fmt, d["mpdu"].tobytes())
    if fmt == abi.F_VHT:
        assert len(ps) == int(d["psdu_len"])
    iq, coded = oracle().tx_packet(fmt, mcs, ps)           # seed 93, multiplier 12: tools/phy80211.py:40, pktGenExample.py:187
    tx = d["tx"][0]
    assert len(iq) == len(tx)
    assert np.array_equal(coded, d["coded"])
    assert np.max(np.abs(iq - tx)) <= 1e-6 * np.max(np.abs(tx))


@pytest.mark.parametrize("name", sorted(FIX2))
def test_oracle_tx_two_streams_reproduces_the_reference_toolbox(name):
    d = FIX2[name]
    fmt, mcs = int(d["fmt"]), int(d["mcs"])
    mp = d["mpdu"].tobytes()
    ps = (ampdu_vht(mp) if len(mp) else b"") if fmt == abi.F_VHT else mp
    iq, coded = oracle().tx_packet2(fmt, mcs % 8 if fmt == abi.F_HT else mcs, 2, ps, mult=12.0 * np.sqrt(2))   # gen_golden.py: MULT * sqrt(2)
    assert iq.shape == d["tx"].shape
    assert np.array_equal(coded, d["coded"])
    assert np.max(np.abs(iq - d["tx"])) <= 1e-6 * np.max(np.abs(d["tx"]))


def random_packets(rng, n, lens=(14, 1600)):
    specs = []
    for _ in range(n):
        fmt = int(rng.integers(0, 3))
        mcs = int(rng.integers(0, 8 if fmt < 2 else 9))
        ln = int(rng.integers(*lens))
        mpdu = mpdu_with_fcs(rng, ln)
        specs.append((fmt, mcs, mpdu, psdu_of(fmt, mpdu)))
    return specs


def tx_batch(specs, gap=700, seed=93, mult=12.0, nss=1):
    pk = np.zeros(len(specs), abi.TX_PKT_DT)
    arena, off = [np.zeros(4, np.uint8)], 4
    for i, (fmt, mcs, mpdu, ps) in enumerate(specs):
        pk[i]["psdu_off"], pk[i]["len"], pk[i]["format"], pk[i]["mcs"] = off, len(ps), fmt, mcs
        pk[i]["scr_seed"], pk[i]["mult"], pk[i]["gap"], pk[i]["nss"] = seed, mult, gap, nss
        arena.append(np.frombuffer(ps, np.uint8))
        off += len(ps)
    return pk, np.concatenate(arena)


def test_oracle_loopback_tx_to_pdus():
    from test_detect_frontend import consume_filter, sig_events
    rng = np.random.default_rng(21)
    o = oracle()
    specs = random_packets(rng, 10, lens=(20, 400))
    parts = []
    for fmt, mcs, mpdu, ps in specs:
        iq, _ = o.tx_packet(fmt, mcs, ps, mult=0.3)
        parts += [np.zeros(600, np.complex64), iq]
    x = np.concatenate(parts + [np.zeros(600, np.complex64)])
    sigma = 0.3 * 0.0156 / 10 ** (30 / 20)                 # ~30 dB below the packet power (|x|^2 ~ mult^2 / 64^2 per tone set)
    x = (x + sigma * (rng.normal(size=len(x)) + 1j * rng.normal(size=len(x)))).astype(np.complex64)
    ev, _, _, _ = o.detect(x)
    spk, siq, schan = o.signal_events(sig_events(ev, len(x)), x)
    keep = consume_filter(ev, spk)
    assert len(keep) == len(specs)
    res = oracle_rx(spk[keep], siq, schan)
    for r, (fmt, mcs, mpdu, ps) in zip(res, specs):
        assert r["pdus"] == [bytes([fmt, len(mpdu) % 256, len(mpdu) // 256]) + mpdu + bytes([mcs])]


@pytest.mark.gpu
def test_gpu_tx_matches_oracle():
    from gr_ieee80211_amd import engine
    rng = np.random.default_rng(5)
    specs = [(int(d["fmt"]), int(d["mcs"]), d["mpdu"].tobytes(), psdu_of(int(d["fmt"]), d["mpdu"].tobytes())) for d in FIX.values()]
    for fmt in range(3):                                   # awkward lengths: minimum, odd, block edges, maximum
        for mcs in (0, 7, 8 if fmt == 2 else 5):
            for ln in (1, 2, 3, 4, 5, 13, 14, 15, 63, 64, 65, 1023, 4092, 4095):
                if fmt == 2:
                    ln = min(ln + (-ln) % 4, 4092)         # an A-MPDU is a multiple of 4 bytes
                specs.append((fmt, mcs, b"", bytes(rng.integers(0, 256, size=ln, dtype=np.uint8))))
    pk, arena = tx_batch(specs, gap=0)
    pk["scr_seed"] = [93 if i < len(FIX) else int(rng.integers(1, 128)) for i in range(len(pk))]
    eng = engine.Engine(device=0)
    eng.tx(pk, arena)
    eng.sync()
    iq = eng.fetch(abi.BUF_TX_IQ, np.complex64)
    coded = eng.fetch(abi.BUF_TX_CODED, np.uint8)
    desc = eng.fetch(abi.BUF_TX_DESC, abi.TX_DESC_DT)
    eng.close()
    o = oracle()
    for i, (fmt, mcs, mpdu, ps) in enumerate(specs):
        oiq, ocoded = o.tx_packet(fmt, mcs, ps, scr_seed=int(pk[i]["scr_seed"]))
        d = desc[i]
        assert int(d["n_samp"]) == len(oiq) and int(d["n_coded"]) == len(ocoded), (i, fmt, mcs, len(ps))
        got = iq[int(d["iq_off"]):int(d["iq_off"]) + len(oiq)]
        assert np.array_equal(coded[int(d["coded_off"]):int(d["coded_off"]) + len(ocoded)], ocoded), (i, fmt, mcs, len(ps))
        assert np.max(np.abs(got - oiq)) <= 1e-6 * np.max(np.abs(oiq)), (i, fmt, mcs, len(ps))
    for name, d, i in zip(FIX, FIX.values(), range(len(FIX))):      # and the toolbox's own output, directly
        got = iq[int(desc[i]["iq_off"]):int(desc[i]["iq_off"]) + int(desc[i]["n_samp"])]
        assert np.max(np.abs(got - d["tx"][0])) <= 1e-6 * np.max(np.abs(d["tx"][0])), name


@pytest.mark.gpu
def test_gpu_loopback_tx_noise_detect_signal_rx():
    """everything on the device: c8p_tx -> AWGN (torch) -> c8p_detect -> c8p_signal -> c8p_rx; PDUs = transmitted MPDUs"""
    import torch
    from gr_ieee80211_amd import engine
    from test_detect_frontend import consume_filter, sig_events
    rng = np.random.default_rng(33)
    specs = random_packets(rng, 200, lens=(30, 1500))
    pk, arena = tx_batch(specs, gap=500, mult=0.3)
    eng = engine.Engine(device=0)
    eng.tx(pk, arena, tail_gap=600)
    eng.sync()
    p_iq, nbytes = eng.buffer(abi.BUF_TX_IQ)
    n = nbytes // 8
    clean = torch.from_numpy(eng.fetch(abi.BUF_TX_IQ, np.complex64).view(np.float32)).cuda()
    torch.manual_seed(1)
    noisy = clean + 1.0e-4 * torch.randn_like(clean)      # ~33 dB SNR
    x = engine.DevArray(noisy.data_ptr(), n, noisy)
    e2 = engine.Engine(device=0)
    ev = e2.detect(x)
    e3 = engine.Engine(device=0)
    e3.signal(sig_events(ev, n), x)
    e3.sync()
    gpk = e3.fetch(abi.BUF_SIG_PKT, abi.PKT_IN_DT)
    keep = consume_filter(ev, gpk)
    assert len(keep) == len(specs)
    pi, bi = e3.buffer(abi.BUF_SIG_IQ)
    pc, bc = e3.buffer(abi.BUF_SIG_CHAN)
    e4 = engine.Engine(device=0)
    e4.rx(gpk[keep], engine.DevArray(pi, bi // 8), engine.DevArray(pc, bc // 8))
    res = e4.results(want_llr=False)
    ok = 0
    for r, (fmt, mcs, mpdu, ps) in zip(res, specs):
        want = bytes([fmt, len(mpdu) % 256, len(mpdu) // 256]) + mpdu + bytes([mcs])
        assert r["pdus"] in ([], [want])
        ok += r["pdus"] == [want]
    assert ok >= 0.97 * len(specs), ok
    for e in (eng, e2, e3, e4):
        e.close()


@pytest.mark.gpu
def test_gpu_new_entry_points_reject_bad_input():
    """c8p_tx / c8p_detect / c8p_signal: negative return code + message, and the ctx stays usable"""
    from gr_ieee80211_amd import engine
    eng = engine.Engine(device=0)
    rng = np.random.default_rng(2)
    pk, arena = tx_batch(random_packets(rng, 3, lens=(20, 60)))
    bad = pk.copy(); bad["mcs"][1] = 9
    with pytest.raises(RuntimeError, match="unsupported tx"):
        eng.tx(bad, arena)
    bad = pk.copy(); bad["format"][0] = 7
    with pytest.raises(RuntimeError, match="unsupported tx"):
        eng.tx(bad, arena)
    bad = pk.copy(); bad["scr_seed"][2] = 0
    with pytest.raises(RuntimeError, match="seed"):
        eng.tx(bad, arena)
    bad = pk.copy(); bad["psdu_off"][2] = len(arena)
    with pytest.raises(RuntimeError, match="outside"):
        eng.tx(bad, arena)
    bad = pk.copy(); bad["len"][0] = 4096
    with pytest.raises(RuntimeError, match="unsupported tx"):
        eng.tx(bad, arena)
    eng.tx(pk, arena)                                     # still usable
    eng.sync()
    assert len(eng.fetch(abi.BUF_TX_DESC, abi.TX_DESC_DT)) == 3
    x = np.zeros(1000, np.complex64)
    with pytest.raises(RuntimeError, match="bad detect"):
        eng.detect(x, n_hist=2000)
    ev = np.zeros(1, abi.SIG_IN_DT); ev["n_ant"] = 1; ev["raw_off"][0, 0] = 900; ev["n_avail"] = 500
    with pytest.raises(RuntimeError, match="outside"):
        eng.signal(ev, x)
    ev["n_ant"] = 3; ev["n_avail"] = 50
    with pytest.raises(RuntimeError, match="n_ant"):
        eng.signal(ev, x)
    assert len(eng.detect(x)) == 0
    eng.close()


@pytest.mark.gpu
def test_gpu_tx_two_streams_matches_oracle_and_toolbox():
    from gr_ieee80211_amd import engine
    rng = np.random.default_rng(9)
    specs = []
    for d in FIX2.values():
        fmt, mcs, mp = int(d["fmt"]), int(d["mcs"]), d["mpdu"].tobytes()
        specs.append((fmt, mcs % 8 if fmt == abi.F_HT else mcs, mp, (ampdu_vht(mp) if len(mp) else b"") if fmt == abi.F_VHT else mp))
    for fmt in (1, 2):
        for mcs in (0, 3, 7):
            for ln in (4, 20, 61, 1000, 4092):
                specs.append((fmt, mcs, b"", bytes(rng.integers(0, 256, size=ln + ((-ln) % 4 if fmt == 2 else 0), dtype=np.uint8))))
    mult = np.float32(12.0 * np.sqrt(2))
    pk, arena = tx_batch(specs, gap=0, mult=mult, nss=2)
    eng = engine.Engine(device=0)
    eng.tx(pk, arena)
    eng.sync()
    iq = eng.fetch(abi.BUF_TX_IQ, np.complex64)
    coded = eng.fetch(abi.BUF_TX_CODED, np.uint8)
    desc = eng.fetch(abi.BUF_TX_DESC, abi.TX_DESC_DT)
    eng.close()
    half = len(iq) // 2
    o = oracle()
    for i, (fmt, mcs, mpdu, ps) in enumerate(specs):
        oiq, ocoded = o.tx_packet2(fmt, mcs, 2, ps, mult=float(mult))
        d = desc[i]
        assert int(d["n_samp"]) == oiq.shape[1] and int(d["n_coded"]) == len(ocoded), (i, fmt, mcs, len(ps))
        assert np.array_equal(coded[int(d["coded_off"]):int(d["coded_off"]) + len(ocoded)], ocoded), (i, fmt, mcs, len(ps))
        for a in range(2):
            got = iq[a * half + int(d["iq_off"]):a * half + int(d["iq_off"]) + oiq.shape[1]]
            assert np.max(np.abs(got - oiq[a])) <= 1e-6 * np.max(np.abs(oiq)), (i, a, fmt, mcs, len(ps))
    for (name, d), i in zip(FIX2.items(), range(len(FIX2))):
        for a in range(2):
            got = iq[a * half + int(desc[i]["iq_off"]):a * half + int(desc[i]["iq_off"]) + int(desc[i]["n_samp"])]
            assert np.max(np.abs(got - d["tx"][a])) <= 1e-6 * np.max(np.abs(d["tx"])), (name, a)


@pytest.mark.gpu
def test_gpu_loopback_two_streams_through_a_2x2_channel():
    """c8p_tx (2 chains) -> flat 2x2 channel + AWGN (torch) -> c8p_detect on antenna 0 -> c8p_signal (2 antennas) -> c8p_rx"""
    import torch
    from c8p_testlib import H22
    from gr_ieee80211_amd import engine
    from test_detect_frontend import consume_filter
    rng = np.random.default_rng(44)
    specs = []
    for _ in range(60):
        fmt = int(rng.integers(1, 3))
        mcs = int(rng.integers(0, 8 if fmt == 1 else 9))
        mpdu = mpdu_with_fcs(rng, int(rng.integers(30, 1500)))
        specs.append((fmt, mcs, mpdu, psdu_of(fmt, mpdu)))
    pk, arena = tx_batch(specs, gap=500, mult=0.3 * np.sqrt(2), nss=2)
    pk["flags"] = abi.TX_NO_WINDOW
    eng = engine.Engine(device=0)
    eng.tx(pk, arena, tail_gap=600)
    eng.sync()
    tx = torch.from_numpy(eng.fetch(abi.BUF_TX_IQ, np.complex64)).cuda()
    n = tx.numel() // 2
    t0, t1 = tx[:n], tx[n:]
    H = torch.tensor(H22).cuda()
    torch.manual_seed(3)
    rx = torch.stack([H[0, 0] * t0 + H[0, 1] * t1, H[1, 0] * t0 + H[1, 1] * t1])
    rx = (torch.view_as_real(rx) + 8.0e-5 * torch.randn(2, n, 2, device="cuda")).contiguous()     # antenna 0 | antenna 1
    x = engine.DevArray(rx.data_ptr(), 2 * n, rx)
    e2 = engine.Engine(device=0)
    ev = e2.detect(engine.DevArray(rx.data_ptr(), n, rx))
    s = np.zeros(len(ev), abi.SIG_IN_DT)
    s["raw_off"][:, 0] = ev["index"]; s["raw_off"][:, 1] = n + ev["index"]
    s["n_avail"] = n - ev["index"]
    s["cfo_rad"], s["snr"], s["rssi"], s["n_ant"] = ev["rad"], ev["snr"], ev["rssi"], 2
    e3 = engine.Engine(device=0)
    e3.signal(s, x)
    e3.sync()
    gpk = e3.fetch(abi.BUF_SIG_PKT, abi.PKT_IN_DT)
    keep = consume_filter(ev, gpk)
    assert len(keep) == len(specs)
    pi, bi = e3.buffer(abi.BUF_SIG_IQ)
    pc, bc = e3.buffer(abi.BUF_SIG_CHAN)
    e4 = engine.Engine(device=0)
    e4.rx(gpk[keep], engine.DevArray(pi, bi // 8), engine.DevArray(pc, bc // 8))
    res = e4.results(want_llr=False)
    ok = 0
    for r, (fmt, mcs, mpdu, ps) in zip(res, specs):
        want = bytes([fmt, len(mpdu) % 256, len(mpdu) // 256]) + mpdu + bytes([mcs + 8 if fmt == 1 else mcs])
        assert r["pdus"] in ([], [want])
        ok += r["pdus"] == [want]
    assert ok >= 0.9 * len(specs), ok
    for e in (eng, e2, e3, e4):
        e.close()


# ---------------------------------------------------------------------------------------------- 2-user MU-MIMO (C8P_F_VHT_MU)
FIXMU = load_npz("packets_mu_tx.npz")
H_MU = np.array([[1 + 0.2j, 0.5 - 0.4j], [0.3 + 0.6j, -0.9 + 0.1j]], np.complex64)     # SURVEY.md §8d cfg 5: rows = stations


def zf_q(h):
    """steering matrix that nulls the other user's stream at each single-antenna station (columns normalised), per bin"""
    q = np.linalg.inv(np.asarray(h, np.complex128))
    q = q / np.linalg.norm(q, axis=0, keepdims=True)
    return np.tile(q.reshape(1, 4), (64, 1)).astype(np.complex64)


@pytest.mark.parametrize("name", sorted(FIXMU))
def test_oracle_tx_mu_reproduces_the_reference_toolbox(name):
    d = FIXMU[name]
    iq, c0, c1 = oracle().tx_packet_mu(d["mcs"], (d["ampdu0"].tobytes(), d["ampdu1"].tobytes()), int(d["gid"]), d["bfq"])
    assert iq.shape == d["tx"].shape and iq.shape[1] == 320 + 80 * (7 + int(d["nsym"]))
    assert np.array_equal(c0, d["coded0"]) and np.array_equal(c1, d["coded1"])
    assert np.max(np.abs(iq - d["tx"])) <= 1e-6 * np.max(np.abs(d["tx"]))


NDBPS_VHT = (26, 52, 78, 104, 156, 208, 234, 260, 312)


def mu_specs(rng, n, lens=(30, 700)):
    specs = []
    while len(specs) < n:
        m = [mpdu_with_fcs(rng, int(rng.integers(*lens))) for _ in range(2)]
        mcs = [int(rng.integers(0, 9)), int(rng.integers(0, 9))]
        a = [ampdu_vht(x) for x in m]
        nsym = max(-(-(8 * len(a[u]) + 22) // NDBPS_VHT[mcs[u]]) for u in range(2))
        if nsym * max(NDBPS_VHT[mcs[0]], NDBPS_VHT[mcs[1]]) <= 32782:      # both users are padded to the longer one (decoder trellis limit)
            specs.append((mcs, m, a, int(rng.integers(1, 63))))
    return specs


def test_oracle_mu_loopback_each_station_decodes_its_own_user():
    from c8p_testlib import Batch, through_channel
    rng = np.random.default_rng(77)
    q = zf_q(H_MU)
    for mcs, mpdus, ampdus, gid in mu_specs(rng, 3, lens=(30, 300)):
        iq, _, _ = oracle().tx_packet_mu(mcs, ampdus, gid, q)
        for u in range(2):
            b = Batch()
            assert b.add_rx(through_channel(iq, H_MU[u:u + 1]), n_ant=1, mupos=u) == 0
            r = oracle_rx(*b.arrays())[0]
            assert r["tags"]["sumu"] == 1 and r["tags"]["mcs"] == mcs[u]
            assert r["pdus"] == [bytes([2, len(mpdus[u]) % 256, len(mpdus[u]) // 256]) + mpdus[u] + bytes([mcs[u]])]


def tx_batch_mu(entries, gap=300, mult=12.0 * np.sqrt(2.0), flags=0):
    """entries: ("su", fmt, mcs, nss, psdu) or ("mu", [mcs0, mcs1], [ampdu0, ampdu1], gid, bfq[64][4])"""
    pk = np.zeros(len(entries), abi.TX_PKT_DT)
    mu, qs = [], []
    arena, off = [np.zeros(4, np.uint8)], 4

    def put(b):
        nonlocal off
        arena.append(np.frombuffer(bytes(b), np.uint8))
        off += len(b)
        return off - len(b)
    for i, e in enumerate(entries):
        pk[i]["scr_seed"], pk[i]["mult"], pk[i]["gap"], pk[i]["flags"] = 93, mult, gap, flags
        if e[0] == "su":
            _, fmt, mcs, nss, ps = e
            pk[i]["psdu_off"], pk[i]["len"], pk[i]["format"], pk[i]["mcs"], pk[i]["nss"] = put(ps), len(ps), fmt, mcs, nss
        else:
            _, mcs, ampdus, gid, bfq = e
            pk[i]["psdu_off"], pk[i]["len"], pk[i]["format"], pk[i]["mcs"], pk[i]["nss"] = put(ampdus[0]), len(ampdus[0]), abi.F_VHT_MU, mcs[0], 2
            rec = np.zeros(1, abi.TX_MU_DT)[0]
            rec["psdu_off"], rec["len"], rec["mcs"], rec["group_id"], rec["q_off"] = put(ampdus[1]), len(ampdus[1]), mcs[1], gid, 256 * len(qs)
            mu.append(rec)
            qs.append(np.ascontiguousarray(bfq, np.complex64).reshape(256))
    return pk, np.concatenate(arena), (np.array(mu, abi.TX_MU_DT) if mu else None), (np.concatenate(qs) if qs else None)


@pytest.mark.gpu
def test_gpu_tx_mu_matches_oracle_and_toolbox():
    from gr_ieee80211_amd import engine
    rng = np.random.default_rng(9)
    entries = []
    for name in sorted(FIXMU):
        d = FIXMU[name]
        entries.append(("mu", [int(x) for x in d["mcs"]], [d["ampdu0"].tobytes(), d["ampdu1"].tobytes()], int(d["gid"]), d["bfq"]))
    for mcs, mpdus, ampdus, gid in mu_specs(rng, 12, lens=(20, 1500)):         # MU packets between single-user ones of every kind
        q = ((rng.normal(size=(64, 4)) + 1j * rng.normal(size=(64, 4))) / 2).astype(np.complex64)
        entries.append(("mu", mcs, ampdus, gid, q))
        fmt = int(rng.integers(0, 3)); nss = 1 if fmt == 0 else int(rng.integers(1, 3))
        mp = mpdu_with_fcs(rng, int(rng.integers(20, 600)))
        entries.append(("su", fmt, int(rng.integers(0, 8)), nss, psdu_of(fmt, mp)))
    pk, arena, mu, bfq = tx_batch_mu(entries)
    eng = engine.Engine(device=0)
    eng.tx(pk, arena, mu=mu, bfq=bfq, tail_gap=100)
    eng.sync()
    iq = eng.fetch(abi.BUF_TX_IQ, np.complex64)
    coded = eng.fetch(abi.BUF_TX_CODED, np.uint8)
    desc = eng.fetch(abi.BUF_TX_DESC, abi.TX_DESC_DT)
    eng.close()
    stride = len(iq) // 2
    o = oracle()
    for i, e in enumerate(entries):
        d = desc[i]
        o0, n = int(d["iq_off"]), int(d["n_samp"])
        if e[0] == "mu":
            oiq, c0, c1 = o.tx_packet_mu(e[1], e[2], e[3], e[4])
            assert n == oiq.shape[1] and int(d["n_coded"]) == len(c0) and int(d["n_coded1"]) == len(c1), i
            assert np.array_equal(coded[int(d["coded_off"]):int(d["coded_off"]) + len(c0)], c0), i
            assert np.array_equal(coded[int(d["coded_off1"]):int(d["coded_off1"]) + len(c1)], c1), i
            got = np.stack([iq[o0:o0 + n], iq[stride + o0:stride + o0 + n]])
        else:
            oiq, c0 = o.tx_packet2(e[1], e[2], e[3], e[4], mult=12.0 * np.sqrt(2.0))
            assert n == oiq.shape[1] and np.array_equal(coded[int(d["coded_off"]):int(d["coded_off"]) + len(c0)], c0), i
            got = np.stack([iq[o0:o0 + n], iq[stride + o0:stride + o0 + n]])[:e[3]]
        assert np.max(np.abs(got - oiq)) <= 1e-6 * np.max(np.abs(oiq)), (i, e[0])
    for i, name in enumerate(sorted(FIXMU)):                                      # the toolbox's own samples, directly
        o0, n = int(desc[i]["iq_off"]), int(desc[i]["n_samp"])
        got = np.stack([iq[o0:o0 + n], iq[stride + o0:stride + o0 + n]])
        assert np.max(np.abs(got - FIXMU[name]["tx"])) <= 1e-6 * np.max(np.abs(FIXMU[name]["tx"])), name


@pytest.mark.gpu
def test_gpu_mu_plan_and_error_paths():
    import ctypes as C
    from gr_ieee80211_amd import engine
    L = engine.load_library()
    plans = (abi.TxPlan * 2)()
    assert L.c8p_tx_plan_mu(0, 44, 7, 304, plans) == 1
    assert plans[0].nsym == plans[1].nsym == int(FIXMU["mu1"]["nsym"]) and plans[0].legacy_len == int(FIXMU["mu1"]["lsig_len"])
    assert plans[0].n_samp == FIXMU["mu1"]["tx"].shape[1] and (plans[0].ndbps, plans[1].ndbps) == (26, 260)
    assert L.c8p_tx_plan_mu(9, 44, 7, 304, plans) == 0 and L.c8p_tx_plan_mu(0, 0, 7, 304, plans) == 0
    assert L.c8p_tx_plan_mu(0, 400, 8, 40, plans) == 0      # 124 symbols x 312 bits: beyond the decoder's 32782-step trellis
    assert L.c8p_tx_plan_mu(0, 336, 8, 40, plans) == 1 and plans[1].nsym * plans[1].ndbps <= 32782
    d = FIXMU["mu0"]
    e = ("mu", [3, 3], [d["ampdu0"].tobytes(), d["ampdu1"].tobytes()], 2, d["bfq"])
    pk, arena, mu, bfq = tx_batch_mu([e])
    eng = engine.Engine(device=0)
    with pytest.raises(RuntimeError, match="MU-MIMO packet without"):
        eng.tx(pk, arena)                                   # MU packet without its second user
    with pytest.raises(RuntimeError, match="steering matrix outside"):
        eng.tx(pk, arena, mu=mu, bfq=bfq[:200])
    bad = mu.copy(); bad["group_id"] = 0
    with pytest.raises(RuntimeError, match="group id"):
        eng.tx(pk, arena, mu=bad, bfq=bfq)
    bad = mu.copy(); bad["psdu_off"] = len(arena)
    with pytest.raises(RuntimeError, match="psdu outside"):
        eng.tx(pk, arena, mu=bad, bfq=bfq)
    bad = mu.copy(); bad["mcs"] = 9
    with pytest.raises(RuntimeError, match="unsupported MU-MIMO"):
        eng.tx(pk, arena, mu=bad, bfq=bfq)
    eng.tx(pk, arena, mu=mu, bfq=bfq)                       # and the context is still usable
    eng.sync()
    assert len(eng.fetch(abi.BUF_TX_DESC, abi.TX_DESC_DT)) == 1
    eng.close()


@pytest.mark.gpu
def test_gpu_loopback_mu_mimo_two_stations():
    """c8p_tx (MU-MIMO, zero-forcing Q) -> each station's 2x1 flat channel + AWGN (torch) -> c8p_detect -> c8p_signal ->
    c8p_rx with that station's user position: every station publishes exactly its own user's MPDUs"""
    import torch
    from gr_ieee80211_amd import engine
    from test_detect_frontend import consume_filter
    rng = np.random.default_rng(52)
    specs = mu_specs(rng, 48, lens=(30, 900))
    q = zf_q(H_MU)
    pk, arena, mu, bfq = tx_batch_mu([("mu", m, a, g, q) for m, _, a, g in specs], gap=500, mult=0.3 * np.sqrt(2), flags=abi.TX_NO_WINDOW)
    mu["q_off"] = 0                                        # one steering matrix for all packets
    eng = engine.Engine(device=0)
    eng.tx(pk, arena, mu=mu, bfq=bfq[:256], tail_gap=600)
    eng.sync()
    tx = torch.from_numpy(eng.fetch(abi.BUF_TX_IQ, np.complex64)).cuda()
    n = tx.numel() // 2
    t0, t1 = tx[:n], tx[n:]
    torch.manual_seed(4)
    engs = [eng]
    for u in range(2):
        rx = complex(H_MU[u, 0]) * t0 + complex(H_MU[u, 1]) * t1
        rx = (torch.view_as_real(rx) + 6.0e-5 * torch.randn(n, 2, device="cuda")).contiguous()
        x = engine.DevArray(rx.data_ptr(), n, rx)
        e2, e3, e4 = engine.Engine(device=0), engine.Engine(device=0), engine.Engine(device=0)
        engs += [e2, e3, e4]
        ev = e2.detect(x)
        s = np.zeros(len(ev), abi.SIG_IN_DT)
        s["raw_off"][:, 0] = ev["index"]
        s["n_avail"] = n - ev["index"]
        s["cfo_rad"], s["snr"], s["rssi"], s["n_ant"] = ev["rad"], ev["snr"], ev["rssi"], 1
        e3.signal(s, x)
        e3.sync()
        gpk = e3.fetch(abi.BUF_SIG_PKT, abi.PKT_IN_DT)
        keep = consume_filter(ev, gpk)
        assert len(keep) == len(specs), (u, len(keep))
        gpk = gpk[keep].copy()
        gpk["mupos"] = u
        pi, bi = e3.buffer(abi.BUF_SIG_IQ)
        pc, bc = e3.buffer(abi.BUF_SIG_CHAN)
        e4.rx(gpk, engine.DevArray(pi, bi // 8), engine.DevArray(pc, bc // 8))
        res = e4.results(want_llr=False)
        ok = 0
        for r, (mcs, mpdus, ampdus, gid) in zip(res, specs):
            want = bytes([2, len(mpdus[u]) % 256, len(mpdus[u]) // 256]) + mpdus[u] + bytes([mcs[u]])
            assert r["tags"]["sumu"] == 1 and r["pdus"] in ([], [want])
            ok += r["pdus"] == [want]
        assert ok >= 0.9 * len(specs), (u, ok)
    for e in engs:
        e.close()


@pytest.mark.gpu
def test_gpu_delivery_ratio_against_snr_has_the_waterfall_shape():
    """tools/perf_pdr.py (the reference's tools/performance/perf_siso.py on the engine): per MCS the delivered count rises
    with the SNR, nothing comes through at 0 dB, everything at 30 dB, and a denser constellation never beats a sparser one
    by more than noise"""
    import sys
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import perf_pdr
    torch.cuda.init()
    dev = torch.device("cuda", 0)
    snrs = [0, 6, 12, 18, 24, 30]
    for fmt, n_mcs in ((abi.F_L, 8), (abi.F_HT, 8), (abi.F_VHT, 9)):
        rows = np.array(perf_pdr.run(fmt, n_mcs, 25, snrs, torch, dev))
        assert rows[0].sum() == 0 and (rows[-1] >= 24).all(), (fmt, rows)
        assert (np.diff(rows, axis=0) >= -2).all(), (fmt, rows)              # rising in SNR
        assert (rows[:, :-1] - rows[:, 1:] >= -3).all(), (fmt, rows)         # falling in MCS
