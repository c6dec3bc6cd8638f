"""K independent raw streams on one GPU (SURVEY.md §8f-1 / f-2 widened to "many independent 20 MHz packet streams"):
c8p_detect_multi (one k_detect_scan launch over the tiles of every stream), c8p_signal's scan rule per stream, and the live ingest
of host/c8p_rawchain (push / round / drain) — against c8p_detect per stream, and against the oracle's chain per stream
(trigger_impl.cc:59-117, sync_impl.cc:62-192, signal_impl.cc:68-204 restated in oracle/)."""
import numpy as np
import pytest

from c8p_cases import dewindow
from c8p_testlib import abi, oracle, oracle_rx, through_channel
from test_detect_frontend import CFOS, NAMES, SWEEP, burst_stream, near_tie, scan_rule, sig_events, stream


def random_stream(seed, n_pkt, snr=(6.0, 35.0)):
    """one antenna: n_pkt packets of random one-stream formats / MCS, carrier offsets of +-150 kHz, random SNR, gaps from 150 items up"""
    rng = np.random.default_rng(seed)
    names = [k for k in SWEEP if int(SWEEP[k]["nss"]) == 1]
    parts = []
    for _ in range(n_pkt):
        d = SWEEP[names[int(rng.integers(len(names)))]]
        rx = through_channel(np.atleast_2d(dewindow(d["tx"]))[:1], None, float(rng.uniform(*snr)), rng, lead=int(rng.choice([150, 260, 400, 900, 2500])))[0]
        w = 2 * np.pi * float(rng.uniform(-150e3, 150e3)) / 20e6
        parts.append((rx * np.exp(1j * w * np.arange(len(rx)))).astype(np.complex64))
    parts.append((0.004 * (rng.normal(size=700) + 1j * rng.normal(size=700))).astype(np.complex64))
    return np.concatenate(parts)


def random_stream2(seed, n_pkt, snr=(10.0, 35.0)):
    """two antennas: packets of every fixture format incl. the two-stream ones and the NDP through a 2x2 (or 2x1) flat channel"""
    from c8p_testlib import H22
    rng = np.random.default_rng(seed)
    names = list(SWEEP)
    parts = [[], []]
    for _ in range(n_pkt):
        d = SWEEP[names[int(rng.integers(len(names)))]]
        nss = int(d["nss"])
        Hc = H22 if nss == 2 else np.array([[1.0], [0.4 - 0.7j]], np.complex64)
        rx = np.atleast_2d(through_channel(np.atleast_2d(dewindow(d["tx"]))[:nss], Hc, float(rng.uniform(*snr)), rng, lead=int(rng.choice([150, 400, 900, 2500]))))
        w = 2 * np.pi * float(rng.uniform(-100e3, 100e3)) / 20e6
        rx = (rx * np.exp(1j * w * np.arange(rx.shape[1]))).astype(np.complex64)
        parts[0].append(rx[0]); parts[1].append(rx[1])
    tail = (0.004 * (rng.normal(size=(2, 700)) + 1j * rng.normal(size=(2, 700)))).astype(np.complex64)
    return np.concatenate(parts[0] + [tail[0]]), np.concatenate(parts[1] + [tail[1]])


def lay_out(xs, rng, pad=(0, 5)):
    """the streams in one arena at offsets that are not multiples of anything (padding of random length between them)"""
    offs, parts, pos = [], [], 0
    for x in xs:
        g = int(rng.integers(pad[0], pad[1] + 1))
        parts.append((rng.normal(size=g) + 1j * rng.normal(size=g)).astype(np.complex64))     # loud garbage: must never be looked at
        pos += g
        offs.append(pos)
        parts.append(x)
        pos += len(x)
    return np.concatenate(parts), offs


@pytest.mark.gpu
@pytest.mark.parametrize("tickets", ["0", "1", "2"])
def test_detect_multi_equals_detect_per_stream_and_the_oracle(tickets, monkeypatch):
    """seven streams of very different lengths (a long one with plateaus that span tiles, short ones, one shorter than a tile, an empty
    one, silence) in one arena, one c8p_detect_multi call: per stream the events and trigger bytes of c8p_detect on that stream alone —
    bit for bit, they run the same arithmetic — and the oracle's events.  With blockIdx numbering, ticket numbering, and the repeat."""
    from gr_ieee80211_amd import engine
    monkeypatch.setenv("C8P_DET_TICKETS", tickets)
    rng = np.random.default_rng(5)
    xs = [burst_stream(31, n_burst=40), stream(NAMES, CFOS, seed=71)[0], random_stream(72, 12), stream(NAMES[:1], CFOS[:1], seed=73)[0][:1500],
          np.zeros(0, np.complex64), np.zeros(5000, np.complex64), random_stream(74, 25)]
    arena, offs = lay_out(xs, rng)
    eng = engine.Engine(device=0)
    evs, out = eng.detect_multi(arena, [dict(off=o, n=len(x)) for o, x in zip(offs, xs)])
    trig_all = eng.fetch(abi.BUF_DET_TRIG, np.uint8)
    assert eng.counter(abi.CNT_DET_TICKET_SCANS) == (0 if tickets == "0" else 1) and eng.counter(abi.CNT_DET_GIVEUPS) == (1 if tickets == "2" else 0)
    o = oracle()
    total = 0
    for s, x in enumerate(xs):
        e1, st1, c1 = eng.detect(x, stream=True) if len(x) else (np.zeros(0, abi.SYNC_EVT_DT), abi.DetState(), 0)
        t1 = eng.fetch(abi.BUF_DET_TRIG, np.uint8) if len(x) else np.zeros(0, np.uint8)
        assert np.array_equal(evs[s], e1), s
        assert out[s]["commit"] == c1 and out[s]["trig_off"] % 8 == 0, (s, out[s]["commit"], c1)
        assert bytes(out[s]["state"]) == bytes(st1), s
        assert np.array_equal(trig_all[out[s]["trig_off"]:out[s]["trig_off"] + len(x)], t1), s
        if len(x):
            ev = o.detect(x)[0]
            assert np.array_equal(evs[s]["index"], ev["index"]) and np.allclose(evs[s]["rad"], ev["rad"], rtol=0, atol=5e-6), s
            total += len(ev)
    assert total >= 60
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2])
def test_detect_multi_streaming_in_ragged_increments_equals_one_shot_per_stream(seed):
    """five streams advanced together by c8p_detect_multi in increments of random size per stream and call (also none at all), each
    stream continuing from ITS commit point and state with DET_HIST items of history: per stream the union of the events and the
    stitched trigger bytes equal one c8p_detect over the whole stream — plateaus, countdowns and search windows that straddle the calls
    of one stream never see another stream's items or state"""
    from gr_ieee80211_amd import engine
    rng = np.random.default_rng(100 + seed)
    xs = [burst_stream(40 + seed, n_burst=30), random_stream(50 + seed, 20), stream(NAMES, CFOS, seed=60 + seed)[0], burst_stream(70 + seed, n_burst=10),
          random_stream(80 + seed, 8)]
    K = len(xs)
    eng = engine.Engine(device=0)
    one = [eng.detect(x) for x in xs]
    one_trig = []
    for x in xs:
        eng.detect(x)
        one_trig.append(eng.fetch(abi.BUF_DET_TRIG, np.uint8))
    end, commit, state = [0] * K, [0] * K, [None] * K
    got = [[] for _ in range(K)]
    gtrig = [np.zeros(len(x), np.uint8) for x in xs]
    ncall = 0
    while any(e < len(x) for e, x in zip(end, xs)):
        parts, streams, h0s, pos = [], [], [], 0
        for s, x in enumerate(xs):
            step = int(rng.choice([0, 97, 700, 960, 961, 1984, 1985, 5000, 12000]))
            end[s] = min(len(x), end[s] + step)
            h0 = max(0, commit[s] - abi.DET_HIST)
            g = int(rng.integers(0, 4))
            parts.append(np.full(g, 3 + 3j, np.complex64)); pos += g
            streams.append(dict(off=pos, n=end[s] - h0, n_hist=commit[s] - h0, state=state[s]))
            parts.append(x[h0:end[s]]); pos += end[s] - h0
            h0s.append(h0)
        arena = np.concatenate(parts + [np.zeros(1, np.complex64)])
        evs, out = eng.detect_multi(arena, streams)
        tb = eng.fetch(abi.BUF_DET_TRIG, np.uint8)
        for s in range(K):
            new_commit = h0s[s] + out[s]["commit"]
            assert commit[s] <= new_commit <= end[s], (s, commit[s], new_commit, end[s])
            gtrig[s][commit[s]:new_commit] = tb[out[s]["trig_off"]:out[s]["trig_off"] + new_commit - commit[s]]
            for k in evs[s]:
                got[s].append((int(k["index"]) + h0s[s], float(k["rad"]), float(k["snr"]), float(k["rssi"])))
            commit[s] = new_commit
            if streams[s]["n"] > streams[s]["n_hist"] or state[s] is not None:
                state[s] = out[s]["state"]
        ncall += 1
    o = oracle()
    for s, x in enumerate(xs):
        assert [g[0] for g in got[s]] == [int(k["index"]) for k in one[s]], s
        for j, f in enumerate(("rad", "snr", "rssi")):
            assert np.array_equal(np.array([g[1 + j] for g in got[s]], np.float32), one[s][f]), (s, f)
        c = commit[s]
        assert c >= len(x) - 400
        assert np.array_equal(gtrig[s][:c] & 1, one_trig[s][:c] & 1)
        ac = o.detect(x)[1]
        bad = np.nonzero((gtrig[s][:c] & 2) != (one_trig[s][:c] & 2))[0]
        assert len(bad) <= 3 and all(near_tie(ac, i) for i in bad), (s, bad[:10])
    print("seed %d: %d calls" % (seed, ncall))
    eng.close()


@pytest.mark.gpu
def test_signal_scan_rule_per_stream():
    """c8p_signal_in.n_streams = 3: the flags of three streams in one call (one arena), the scan rule applied inside each stream from
    that stream's position: per stream the descriptors and codes of a single-stream call over that stream's flags; kept packets back
    to back in stream order; flags out of stream order are refused"""
    from gr_ieee80211_amd import engine
    eng = engine.Engine(device=0)
    rng = np.random.default_rng(9)
    xs = [np.concatenate([burst_stream(13, n_burst=10), stream(NAMES, CFOS, seed=21)[0]]), stream(NAMES[::-1], CFOS, seed=22)[0], random_stream(23, 15)]
    arena, offs = lay_out(xs, rng, pad=(3, 40))
    scan0 = [0, 3000, 0]
    evts, single = [], []
    for s, x in enumerate(xs):
        ev = eng.detect(x)
        q = sig_events(ev, len(x))
        q["n_avail"] = np.minimum(q["n_avail"], 224 + 109280)
        if s == 2:
            q["n_avail"][-3:] = np.minimum(q["n_avail"][-3:], 300)       # the last packets of this stream wait for more items
        pk1 = eng.signal(q, x, consume=True, scan_from=scan0[s])
        single.append((q, pk1, int(eng.scan_next[0]), eng.fetch(abi.BUF_SIG_IQ, np.complex64)))
        g = q.copy()
        g["raw_off"][:, 0] += offs[s]
        g["stream"] = s
        evts.append(g)
    allq = np.concatenate(evts)
    pk = eng.signal(allq, arena, consume=True, n_streams=3, scan_from=[offs[s] + scan0[s] for s in range(3)])
    iq = eng.fetch(abi.BUF_SIG_IQ, np.complex64)
    nxt = eng.scan_next.copy()
    k, used = 0, 0
    for s, (q, pk1, nx, iq1) in enumerate(single):
        mine = pk[k:k + len(q)]
        for f in ("n_items", "lsig_mcs", "lsig_len", "lsig_nsamp", "cfo", "snr", "rssi"):
            assert np.array_equal(mine[f], pk1[f]), (s, f)
        assert np.array_equal(mine["rsv"][:, 0], pk1["rsv"][:, 0]), s
        assert int(nxt[s]) == offs[s] + nx, s
        kept = np.nonzero(mine["n_items"])[0]
        assert len(kept) >= 5
        for i in kept:
            n = int(mine[i]["n_items"])
            assert int(mine[i]["iq_off"][0]) == used
            assert np.array_equal(iq[used:used + n], iq1[int(pk1[i]["iq_off"][0]):int(pk1[i]["iq_off"][0]) + n]), (s, i)
            used += n
        k += len(q)
    assert np.isin(pk[k - 3:]["rsv"][:, 0], (abi.SIG_NEED_MORE, abi.SIG_PENDING)).any()      # (stream 2's last flags wait for more items)
    with pytest.raises(RuntimeError, match="grouped by stream"):
        eng.signal(allq[::-1].copy(), arena, consume=True, n_streams=3)
    eng.close()


def oracle_chain(o, x):
    """the oracle's chain over one whole stream: (kept descriptors, their sync indices, per-packet oracle results)"""
    ev = o.detect(x)[0]
    s = sig_events(ev, len(x))
    s["n_avail"] = np.minimum(s["n_avail"], 224 + 109280)
    pk, iq, chan = o.signal_events(s, x)
    kept, code = scan_rule(s, pk)
    return pk[kept], [int(ev[i]["index"]) for i in kept], oracle_rx(pk[kept], iq, chan), len(ev)


def to_sc16(x, scale, bits=16):
    """what the radio sent: int16 (int8) I/Q pairs, and the complex64 stream a host-side conversion (UHD's: one float32 multiply per
    component) makes of them — the stream the reference's first block sees"""
    top = (1 << (bits - 1)) - 1
    q = np.clip(np.rint(x.view(np.float32) / np.float32(scale)), -top - 1, top).astype(np.int16 if bits == 16 else np.int8)
    return q, (q.astype(np.float32) * np.float32(scale)).view(np.complex64)


@pytest.mark.gpu
@pytest.mark.parametrize("feed", ["host", "device", "host-sc16", "device-sc16", "host-sc8", "host-nomask", "host-sc16-nomask"])
def test_rawchain_live_eight_streams_in_ragged_increments_equal_the_oracle_chain_per_stream(feed, monkeypatch):
    """K = 8 random streams (every one-stream format / MCS, carrier offsets of +-150 kHz, SNRs from 6 to 35 dB so that many fail
    their FCS, gaps from 150 items up so that some preambles are missed or merged; one stream is silence, one starts late) pushed in
    increments of random size per stream and round — rounds end inside preambles, inside packets, inside plateaus — through the live
    ingest of host/c8p_rawchain (one c8p_detect_multi, one c8p_signal, one c8p_rx per round) and through the oracle's detect -> signal
    (+ scan rule) -> demod + decode over each whole stream: per stream, in order, the same kept packets (sync index, L-SIG fields), the
    same PDUs and FCS-failure counts.  -sc16: the streams arrive as int16 I/Q pairs (C8P_MEM_SC16, 4 bytes per item over the link) and
    are converted by the gather kernel; the oracle runs over the host-side conversion of the same pairs."""
    import torch
    from gr_ieee80211_amd import blocks
    dev = torch.device("cuda:0")
    o = oracle()
    rng = np.random.default_rng(77)
    xs = [random_stream(200 + s, n) for s, n in enumerate([60, 45, 70, 30, 55, 40, 25])]
    xs.insert(3, (0.004 * (rng.normal(size=90000) + 1j * rng.normal(size=90000))).astype(np.complex64))
    K = len(xs)
    late = 5                                            # this stream delivers nothing during the first rounds
    if feed.endswith("-nomask"):                        # the copy stream on all CUs, non-blocking (C8P_COPY_CUS=0): the path before the mask
        monkeypatch.setenv("C8P_COPY_CUS", "0")
        feed = feed[:-len("-nomask")]
    bits = 16 if feed.endswith("-sc16") else (8 if feed.endswith("-sc8") else 0)
    sc16 = bits or False                                # (what RawChain.push takes: False / 16 / 8)
    feed = feed.split("-")[0]
    if bits == 8:               # eight bits: every radio's gain set to its own peak, as an AGC would
        xs = [(x / np.float32(max(float(np.abs(x.view(np.float32)).max()), 1e-9))).astype(np.complex64) for x in xs]
    scale = np.float32(1.02 * max(float(np.abs(x.view(np.float32)).max()) for x in xs) / ((1 << ((bits or 16) - 1)) - 1))
    if sc16:
        qs, xs = zip(*[to_sc16(x, scale, bits) for x in xs])
        ts = [torch.from_numpy(q.copy()) for q in qs]
    else:
        ts = [torch.from_numpy(x.view(np.float32).copy()) for x in xs]
    ts = [t.pin_memory() if feed == "host" else t.to(dev) for t in ts]
    torch.cuda.synchronize()
    got = []
    h = blocks.RawChain(streams=K, stream_round_items=40000, reserve_seg_packets=64, reserve_pkt_items=6000, sc16_scale=float(scale), sc8_scale=float(scale))
    h.live(got.append)
    pos = [0] * K
    r = 0
    while any(p < len(x) for p, x in zip(pos, xs)):
        for s, x in enumerate(xs):
            n = int(rng.choice([0, 137, 2500, 9000, 20000, 39999]))
            if s == late and r < 4:
                n = 0
            n = min(n, len(x) - pos[s])
            if n:
                h.push(s, (ts[s].data_ptr() + (bits // 4 if sc16 else 8) * pos[s], n), host=("pinned" if sc16 else True) if feed == "host" else False, sc16=sc16)
                pos[s] += n
        h.round(wait=(feed == "host"))             # device feed: c8p_rawchain_round_async (the detect stage waits for the round's copies)
        r += 1
    st = h.drain()
    h.close()
    assert st["segments"] == r == len(got) and st["items"] == sum(len(x) for x in xs) and st["det_giveups"] == 0
    per = [dict(pk=[], idx=[], outs=[], recs=[]) for _ in range(K)]
    for g in got:
        assert np.all(np.diff(g["pkt_stream"].astype(np.int64)) >= 0)
        for i in range(len(g["pkts"])):
            P = per[int(g["pkt_stream"][i])]
            oo = g["outs"][i]
            P["pk"].append(g["pkts"][i]); P["idx"].append(int(g["pkt_index"][i])); P["outs"].append(oo)
            P["recs"].append(g["pdu_arena"][int(oo["pdu_off"]):int(oo["pdu_off"]) + int(oo["pdu_bytes"])].tobytes())
    n_pdu = n_fail = n_ev = 0
    for s, x in enumerate(xs):
        pk, idx, ref, nev = oracle_chain(o, x)
        n_ev += nev
        P = per[s]
        assert P["idx"] == idx, s
        for f in ("lsig_mcs", "lsig_len", "lsig_nsamp", "n_items", "n_ant"):
            assert [int(p[f]) for p in P["pk"]] == [int(v) for v in pk[f]], (s, f)
        assert np.allclose([float(p["cfo"]) for p in P["pk"]], pk["cfo"], rtol=0, atol=0.5)
        for i, (rr, oo, rec) in enumerate(zip(ref, P["outs"], P["recs"])):
            assert int(oo["n_pdu"]) == len(rr["pdus"]) and int(oo["n_crc_fail"]) == int(rr["out"]["n_crc_fail"]), (s, i)
            for p in rr["pdus"]:
                assert int.from_bytes(rec[:2], "little") == len(p) and rec[2:2 + len(p)] == p, (s, i)
                rec = rec[2 + len(p):]
            n_pdu += len(rr["pdus"]); n_fail += int(rr["out"]["n_crc_fail"])
    assert st["pdus"] == n_pdu >= (100 if bits == 8 else 150) and n_fail >= (10 if bits == 8 else 30) and st["events"] == n_ev, (n_pdu, n_fail, n_ev, st)


@pytest.mark.gpu
def test_rawchain_handle_reserved_once_never_allocates_again():
    """a handle with reserved arenas: no device / pinned allocation inside any run (c8p_rawchain_stats.arena_growths, from the engine's
    own counters), the first run included; the same results from every run; per-stage clocks and marks are there"""
    import torch
    from gr_ieee80211_amd import blocks
    dev = torch.device("cuda:0")
    xs = [stream(NAMES, CFOS, snr=33.0, seed=300 + i)[0] for i in range(6)]
    ts = [torch.from_numpy(x.view(np.float32).copy()).to(dev) for x in xs]
    pins = [torch.from_numpy(x.view(np.float32).copy()).pin_memory() for x in xs]
    torch.cuda.synchronize()
    h = blocks.RawChain(reserve_seg_items=max(len(x) for x in xs), reserve_seg_packets=16, reserve_pkt_items=2000, reserve_host=True)
    first = None
    for rep in range(3):
        got = []
        host = rep == 2
        st = h.run([(t.data_ptr(), len(x)) for t, x in zip(pins if host else ts, xs)], on_segment=got.append, host=host)
        assert st["arena_growths"] == 0, (rep, st)
        assert st["segments"] == 6 and st["pdus"] == 6 * len(NAMES)
        sig = [(g["segment"], g["pkts"]["lsig_len"].tolist(), g["outs"]["n_pdu"].tolist(), g["pdu_arena"][:int(g["outs"]["pdu_off"][-1])].tobytes()) for g in got]
        first = first or sig
        assert sig == first
        assert len(h.timeline()) == 6 and st["rx"]["busy"] > 0 and st["steady_segments"] == 5
    h.close()


@pytest.mark.gpu
@pytest.mark.parametrize("slots", ["1", "2"])
def test_rawchain_host_segments_with_fewer_engine_slots(slots, monkeypatch):
    """C8P_SLOTS = 1 / 2 on the engine: host segments (the signal stage reads the detect context's per-slot copy of the items,
    C8P_BUF_DET_RAW) and device segments deliver what three slots deliver — the detect stage does not reuse a slot before
    c8p_signal is through with it, the rx stage collects before it submits when only one submit stays resident"""
    import torch
    from gr_ieee80211_amd import blocks
    dev = torch.device("cuda:0")
    xs = [stream(NAMES, CFOS, snr=33.0, seed=600 + i)[0] for i in range(9)]
    pins = [torch.from_numpy(x.view(np.float32).copy()).pin_memory() for x in xs]
    devs = [p.to(dev) for p in pins]
    torch.cuda.synchronize()
    ref = []
    blocks.rawchain_run([(t.data_ptr(), len(x)) for t, x in zip(devs, xs)], on_segment=ref.append)
    monkeypatch.setenv("C8P_SLOTS", slots)
    for host in (True, False):
        got = []
        st = blocks.rawchain_run([(t.data_ptr(), len(x)) for t, x in zip(pins if host else devs, xs)], on_segment=got.append, host=host)
        assert st["segments"] == 9 and st["pdus"] == 9 * len(NAMES)
        for a, b in zip(got, ref):
            assert np.array_equal(a["outs"]["n_pdu"], b["outs"]["n_pdu"]) and np.array_equal(a["pkts"]["lsig_len"], b["pkts"]["lsig_len"])
            for oa, ob in zip(a["outs"], b["outs"]):
                assert np.array_equal(a["pdu_arena"][int(oa["pdu_off"]):int(oa["pdu_off"]) + int(oa["pdu_bytes"])],
                                      b["pdu_arena"][int(ob["pdu_off"]):int(ob["pdu_off"]) + int(ob["pdu_bytes"])])


def live_two_antennas_equal_one_pass(engine_lib, n_pkt, dev=None, sc16=False):
    """three two-antenna streams pushed in ragged increments (c8p_rawchain_push2) against ONE pass per stream through
    c8p_rawchain_run with ant1_off (rx2.grc's chain, itself compared with the oracle's chain elsewhere): per stream the same packets
    (sync index, L-SIG fields, both ports) and PDU bytes, in order.  sc16: both antennas arrive as int16 I/Q pairs (C8P_MEM_SC16); the
    one pass runs over their host-side conversion"""
    from gr_ieee80211_amd import blocks
    rng = np.random.default_rng(31)
    xs = [random_stream2(400 + s, n_pkt) for s in range(3)]
    scale, qs = 0.0, None
    if sc16:
        scale = np.float32(1.02 * max(float(np.abs(a.view(np.float32)).max()) for x in xs for a in x) / 32767.0)
        conv = [[to_sc16(a, scale) for a in x] for x in xs]
        qs = [[c[0] for c in cs] for cs in conv]
        xs = [tuple(np.ascontiguousarray(c[1]) for c in cs) for cs in conv]
        if dev is not None:                      # the device reads the pairs itself: pinned memory
            import torch
            qs = [[torch.from_numpy(q.copy()).pin_memory() for q in qq] for qq in qs]
    want = []
    for a0, a1 in xs:
        both = np.concatenate([a0, a1])
        got = []
        if dev is None:
            blocks.rawchain_run([(both.ctypes.data, len(a0), len(a0))], engine_lib=engine_lib, on_segment=got.append)
        else:
            import torch
            t = torch.from_numpy(both.view(np.float32).copy()).to(dev)
            torch.cuda.synchronize()
            blocks.rawchain_run([(t.data_ptr(), len(a0), len(a0))], engine_lib=engine_lib, on_segment=got.append)
        g = got[0]
        want.append(([int(i) for i in g["pkt_index"]], g["pkts"], [g["pdu_arena"][int(o["pdu_off"]):int(o["pdu_off"]) + int(o["pdu_bytes"])].tobytes() for o in g["outs"]]))
    got = []
    h = blocks.RawChain(engine_lib=engine_lib, streams=3, stream_round_items=30000, stream_ant=2, sc16_scale=float(scale))
    h.live(got.append)
    pos = [0] * 3
    while any(p < len(x[0]) for p, x in zip(pos, xs)):
        for s, (a0, a1) in enumerate(xs):
            n = min(int(rng.choice([0, 137, 2500, 9000, 29999])), len(a0) - pos[s])
            if n and sc16:
                p0, p1 = ((q.data_ptr() if dev is not None else q.ctypes.data) + 4 * pos[s] for q in qs[s])
                h.push2(s, (p0, n), (p1, n), host="pinned", sc16=True)
                pos[s] += n
            elif n:
                h.push2(s, a0[pos[s]:pos[s] + n], a1[pos[s]:pos[s] + n])
                pos[s] += n
        h.round()
    with pytest.raises(RuntimeError, match="two antennas"):
        h.push(0, xs[0][0][:10])
    st = h.drain()
    h.close()
    per = [([], [], []) for _ in range(3)]
    for g in got:
        for i in range(len(g["pkts"])):
            P = per[int(g["pkt_stream"][i])]
            o = g["outs"][i]
            P[0].append(int(g["pkt_index"][i])); P[1].append(g["pkts"][i])
            P[2].append(g["pdu_arena"][int(o["pdu_off"]):int(o["pdu_off"]) + int(o["pdu_bytes"])].tobytes())
    n_pdu = 0
    for s in range(3):
        assert per[s][0] == want[s][0], s
        for f in ("lsig_mcs", "lsig_len", "lsig_nsamp", "n_items", "n_ant", "cfo"):
            assert [p[f] for p in per[s][1]] == [p[f] for p in want[s][1]], (s, f)
        assert per[s][2] == want[s][2], s
        assert all(int(p["n_ant"]) == 2 for p in per[s][1])
        n_pdu += sum(1 for r in want[s][2] if r)
    assert st["pdus"] >= n_pdu >= 3 * n_pkt // 2
    return st


@pytest.mark.gpu
def test_rawchain_live_two_antenna_streams_equal_one_pass_per_stream():
    import torch
    from gr_ieee80211_amd import blocks
    live_two_antennas_equal_one_pass(blocks.ENGINE_LIB, 30, torch.device("cuda:0"))
    live_two_antennas_equal_one_pass(blocks.ENGINE_LIB, 20, torch.device("cuda:0"), sc16=True)


@pytest.mark.gpu
@pytest.mark.parametrize("cus", ["default", "0", "8"])
def test_copy_gather_converts_sc16_pieces_exactly_at_any_alignment(cus, monkeypatch):
    """c8p_copy_gather, C8P_FMT_SC16: pieces that start and end at any 4-byte item of pinned host memory, into any 8-byte item of the
    arena, mixed with complex64 pieces in one call — every destination item = ((float)i * scale, (float)q * scale) bit for bit, nothing
    outside the pieces is written; pageable or 2-byte aligned sources are refused.  cus: C8P_COPY_CUS — the copy stream confined to 4 CUs
    of every XCD (default), unconfined (0: a plain non-blocking stream), or to one CU per XCD"""
    import ctypes as C
    import torch
    from gr_ieee80211_amd import engine
    if cus != "default":
        monkeypatch.setenv("C8P_COPY_CUS", cus)
    rng = np.random.default_rng(5)
    e = engine.Engine(device=0)
    L = e.L
    n_src = 1 << 16
    q = torch.from_numpy(rng.integers(-32768, 32768, size=2 * n_src, dtype=np.int16)).pin_memory()
    f = torch.from_numpy((rng.normal(size=2 * 4096)).astype(np.float32)).pin_memory()
    scale = np.float32(3.1e-5)
    n_dst = 1 << 17
    dst = torch.full((2 * n_dst,), -7.0, dtype=torch.float32, device="cuda:0")
    want = np.full(2 * n_dst, -7.0, np.float32)
    pcs, pos = [], 3
    qn = q.numpy()
    for i in range(40):
        a = int(rng.integers(0, n_src - 3000))
        n = int(rng.choice([1, 2, 3, 4, 5, 7, 64, 1000, 2999]))
        pcs.append(abi.Piece(q.data_ptr() + 4 * a, pos, n, 1, abi.FMT_SC16, float(scale), 0))
        want[2 * pos:2 * (pos + n)] = qn[2 * a:2 * (a + n)].astype(np.float32) * scale
        pos += n + int(rng.integers(0, 4))
        if i % 5 == 2:                               # an sc8 piece (2 bytes per item, any 2-byte alignment)
            a8 = int(rng.integers(0, 2 * n_src - 3000))
            m = int(rng.choice([1, 3, 8, 9, 17, 500, 2999]))
            pcs.append(abi.Piece(q.data_ptr() + 2 * a8, pos, m, 1, abi.FMT_SC8, float(scale), 0))
            want[2 * pos:2 * (pos + m)] = qn.view(np.int8)[2 * a8:2 * (a8 + m)].astype(np.float32) * scale
            pos += m + int(rng.integers(0, 4))
        if i % 9 == 4:                               # a complex64 piece between them
            m = int(rng.integers(1, 300))
            pcs.append(abi.Piece(f.data_ptr() + 8 * 5, pos, m, 1, abi.FMT_C32, 0.0, 0))
            want[2 * pos:2 * (pos + m)] = f.numpy()[10:10 + 2 * m]
            pos += m + 1
    arr = (abi.Piece * len(pcs))(*pcs)
    assert L.c8p_copy_gather(e.ctx, C.c_void_p(dst.data_ptr()), arr, len(pcs)) == 0, L.c8p_last_error(e.ctx)
    assert L.c8p_copy_wait(e.ctx) == 0
    assert np.array_equal(dst.cpu().numpy().view(np.uint32), want.view(np.uint32))
    odd = (abi.Piece * 1)(abi.Piece(q.data_ptr() + 2, 0, 8, 1, abi.FMT_SC16, float(scale), 0))
    assert L.c8p_copy_gather(e.ctx, C.c_void_p(dst.data_ptr()), odd, 1) == -22
    odd8 = (abi.Piece * 1)(abi.Piece(q.data_ptr() + 1, 0, 8, 1, abi.FMT_SC8, float(scale), 0))
    assert L.c8p_copy_gather(e.ctx, C.c_void_p(dst.data_ptr()), odd8, 1) == -22
    pageable = np.zeros(64, np.int16)
    pg = (abi.Piece * 1)(abi.Piece(pageable.ctypes.data, 0, 8, 0, abi.FMT_SC16, float(scale), 0))
    assert L.c8p_copy_gather(e.ctx, C.c_void_p(dst.data_ptr()), pg, 1) == -14          # C8P_ENOTDEV: the one refusal answered with c8p_copy_async
    bad = (abi.Piece * 1)(abi.Piece(q.data_ptr(), 0, 8, 1, 7, float(scale), 0))
    assert L.c8p_copy_gather(e.ctx, C.c_void_p(dst.data_ptr()), bad, 1) == -22
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n_pieces", [7, 300])
def test_copy_gather_moves_f32_pieces_both_ways_exactly(n_pieces):
    """C8P_FMT_F32 pieces (the LLR stream between the demod and decode mirrors): 4-byte items from any 4-byte aligned address of pinned
    host memory into any float of a device arena — and, with source and destination swapped, from the device into pinned host memory
    (how demod's LLR items reach a pinned scheduler buffer).  7 pieces travel in the launch's arguments, 300 through the table; nothing
    outside the pieces is written; an F32 piece in a call with 8-byte-item pieces is refused; c8p_copy_fence orders a following submit."""
    import ctypes as C
    import torch
    from gr_ieee80211_amd import engine
    rng = np.random.default_rng(17 + n_pieces)
    e = engine.Engine(device=0)
    L = e.L
    n_src = 1 << 18
    src = torch.from_numpy(rng.normal(size=n_src).astype(np.float32)).pin_memory()
    n_dst = 1 << 19
    dst = torch.full((n_dst,), -7.0, dtype=torch.float32, device="cuda:0")
    want = np.full(n_dst, -7.0, np.float32)
    pcs, pos, spans = [], 1, []
    for i in range(n_pieces):
        a = int(rng.integers(0, n_src - 3000))
        n = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 64, 1000, 2999])) if n_pieces < 100 else int(rng.choice([1, 2, 3, 5, 9, 33, 500]))
        pcs.append(abi.Piece(src.data_ptr() + 4 * a, pos, n, 1, abi.FMT_F32, 1.0, 0))
        want[pos:pos + n] = src.numpy()[a:a + n]
        spans.append((pos, n))
        pos += n + int(rng.integers(0, 5))
    arr = (abi.Piece * len(pcs))(*pcs)
    assert L.c8p_copy_gather(e.ctx, C.c_void_p(dst.data_ptr()), arr, len(pcs)) == 0, L.c8p_last_error(e.ctx)
    assert L.c8p_copy_wait(e.ctx) == 0
    assert np.array_equal(dst.cpu().numpy().view(np.uint32), want.view(np.uint32))
    # the other way: the same spans of the device arena into a pinned host buffer, packed back to back from an odd float on
    back = torch.full((n_dst,), -3.0, dtype=torch.float32).pin_memory()
    wantb = np.full(n_dst, -3.0, np.float32)
    pcs2, o = [], 3
    for (p0, n) in spans:
        pcs2.append(abi.Piece(dst.data_ptr() + 4 * p0, o, n, 1, abi.FMT_F32, 1.0, 0))
        wantb[o:o + n] = want[p0:p0 + n]
        o += n
    arr2 = (abi.Piece * len(pcs2))(*pcs2)
    assert L.c8p_copy_gather(e.ctx, C.c_void_p(back.data_ptr()), arr2, len(pcs2)) == 0, L.c8p_last_error(e.ctx)
    mark = L.c8p_copy_mark(e.ctx)
    assert mark > 0 and L.c8p_copy_wait_mark(e.ctx, mark) == 0
    assert np.array_equal(back.numpy().view(np.uint32), wantb.view(np.uint32))
    mixed = (abi.Piece * 2)(abi.Piece(src.data_ptr(), 0, 8, 1, abi.FMT_F32, 1.0, 0), abi.Piece(src.data_ptr(), 64, 8, 1, abi.FMT_C32, 0.0, 0))
    assert L.c8p_copy_gather(e.ctx, C.c_void_p(dst.data_ptr()), mixed, 2) == -22
    odd = (abi.Piece * 1)(abi.Piece(src.data_ptr() + 2, 0, 8, 1, abi.FMT_F32, 1.0, 0))
    assert L.c8p_copy_gather(e.ctx, C.c_void_p(dst.data_ptr()), odd, 1) == -22
    assert L.c8p_copy_fence(e.ctx) == 0
    e.close()


def unordered_gathers_and_mark_done(L, ctx, alloc_dst, read_dst, pinned):
    """24 C8P_PIECE_UNORDERED gathers (they alternate between the copy stream and its twin) into disjoint stretches of one arena, one mark
    behind them: c8p_copy_mark_done goes from "not yet" (or straight) to 1 and never back, c8p_copy_wait_mark then returns at once, and
    every stretch has arrived; a mark that does not exist is refused"""
    import ctypes as C
    rng = np.random.default_rng(23)
    n_each, k = 50_000, 24
    src = pinned(n_each * k)
    src[:] = rng.normal(size=n_each * k).astype(np.float32)
    dst = alloc_dst(n_each * k + 16)
    for i in range(k):
        pc = (abi.Piece * 1)(abi.Piece(src.ctypes.data + 4 * n_each * i, 5 + n_each * i, n_each, 1, abi.FMT_F32, 1.0, abi.PIECE_UNORDERED))
        assert L.c8p_copy_gather(ctx, C.c_void_p(dst), pc, 1) == 0
    mark = L.c8p_copy_mark(ctx)
    assert mark > 0
    assert L.c8p_copy_mark_done(ctx, mark + 1) == -22 and L.c8p_copy_mark_done(ctx, 0) == -22
    seen_done = False
    for _ in range(2_000_000):
        d = L.c8p_copy_mark_done(ctx, mark)
        assert d in (0, 1) and not (seen_done and d == 0)
        if d == 1:
            seen_done = True
            break
    assert seen_done
    assert L.c8p_copy_mark_done(ctx, mark) == 1 and L.c8p_copy_wait_mark(ctx, mark) == 0
    got = read_dst(dst, n_each * k + 16)
    assert np.array_equal(got[5:5 + n_each * k].view(np.uint32), src.view(np.uint32))
    assert L.c8p_copy_fence(ctx) == 0 and L.c8p_copy_wait(ctx) == 0


def test_unordered_gathers_and_mark_done_on_the_fake_engine():
    import ctypes as C, os
    from gr_ieee80211_amd import blocks
    L = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "fake_c8p", "libc8p_fake.so"))
    L.c8p_create.restype = C.c_void_p
    L.c8p_copy_mark.restype = C.c_uint64
    L.c8p_copy_mark.argtypes = [C.c_void_p]
    L.c8p_copy_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
    L.c8p_copy_mark_done.argtypes = [C.c_void_p, C.c_uint64]
    L.c8p_copy_wait_mark.argtypes = [C.c_void_p, C.c_uint64]
    L.c8p_copy_fence.argtypes = [C.c_void_p]
    L.c8p_copy_wait.argtypes = [C.c_void_p]
    L.c8p_destroy.argtypes = [C.c_void_p]
    ctx = L.c8p_create(None)
    keep = []
    def alloc(n):
        a = np.full(n, -1.0, np.float32); keep.append(a); return a.ctypes.data
    unordered_gathers_and_mark_done(L, ctx, alloc, lambda p, n: keep[-1], lambda n: np.zeros(n, np.float32))
    L.c8p_destroy(ctx)


@pytest.mark.gpu
def test_unordered_gathers_and_mark_done_on_the_engine():
    import torch
    from gr_ieee80211_amd import engine
    e = engine.Engine(device=0)
    box = {}
    def alloc(n):
        box["dst"] = torch.full((n,), -1.0, dtype=torch.float32, device="cuda:0"); torch.cuda.synchronize(); return box["dst"].data_ptr()
    def pinned(n):
        box["src"] = torch.empty(n, dtype=torch.float32).pin_memory(); return box["src"].numpy()
    unordered_gathers_and_mark_done(e.L, e.ctx, alloc, lambda p, n: box["dst"].cpu().numpy(), pinned)
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cus", ["default", "0"])
def test_copy_spans_moves_exactly_the_spans(cus, monkeypatch):
    """c8p_copy_spans (the carry of every live stream's tail into the next round's arena): spans of any length (none, one item, across
    chunk boundaries) at any item offsets of source and destination — the spans arrive, nothing else is touched; behind a gather into
    the spans' source on the copy stream (the carry runs on a stream of its own, ordered by an event) and in front of a mark"""
    import ctypes as C
    import torch
    from gr_ieee80211_amd import engine
    if cus != "default":
        monkeypatch.setenv("C8P_COPY_CUS", cus)
    rng = np.random.default_rng(9)
    e = engine.Engine(device=0)
    L = e.L
    n = 1 << 18
    host = torch.from_numpy(rng.normal(size=2 * n).astype(np.float32)).pin_memory()
    src = torch.zeros(2 * n, dtype=torch.float32, device="cuda:0")
    dst = torch.full((2 * n,), -3.0, dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    want = np.full(2 * n, -3.0, np.float32)
    sh = host.numpy()
    fill = (abi.Piece * 1)(abi.Piece(host.data_ptr(), 0, n, 1, abi.FMT_C32, 0.0, 0))      # the source arrives through the copy stream ...
    assert L.c8p_copy_gather(e.ctx, C.c_void_p(src.data_ptr()), fill, 1) == 0, L.c8p_last_error(e.ctx)
    spans, pos = [], 1
    for m in [0, 1, 2, 63, 64, 65, 511, 512, 513, 1023, 1025, 4097, 30000, 7, 0, 110000]:
        a = int(rng.integers(0, n - m))
        spans.append(abi.Span(pos, a, m))
        want[2 * pos:2 * (pos + m)] = sh[2 * a:2 * (a + m)]
        pos += m + int(rng.integers(0, 3))
    arr = (abi.Span * len(spans))(*spans)
    assert L.c8p_copy_spans(e.ctx, C.c_void_p(dst.data_ptr()), C.c_void_p(src.data_ptr()), arr, len(spans)) == 0, L.c8p_last_error(e.ctx)
    mark = L.c8p_copy_mark(e.ctx)                                                            # ... and "every copy so far" covers the spans
    assert mark > 0 and L.c8p_copy_wait_mark(e.ctx, mark) == 0
    assert np.array_equal(dst.cpu().numpy().view(np.uint32), want.view(np.uint32))
    e.close()


@pytest.mark.gpu
def test_rawchain_live_rounds_are_delivered_without_waiting_for_the_next():
    from gr_ieee80211_amd import blocks
    from test_rawchain_cpu import live_rounds_are_delivered_without_waiting_for_the_next
    live_rounds_are_delivered_without_waiting_for_the_next(blocks.ENGINE_LIB)
