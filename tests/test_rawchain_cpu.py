"""host/c8p_rawchain.cc on the CPU: the pipeline driver (three stage threads, hand-over queues, the arena counter, the carry of the
detection state and of the signal block's scan position in stream mode) against tests/fake_c8p — the C ABI on top of the oracle,
test-only — so that the threading and bookkeeping are exercised without a GPU (and under the sanitizers, tools/sanitize_cpu.sh).
The GPU tests of the same driver are in tests/test_detect_frontend.py."""
import os
import subprocess

import numpy as np
import pytest

from c8p_testlib import ROOT
from gr_ieee80211_amd import blocks
from test_detect_frontend import CFOS, NAMES, SWEEP, burst_stream, stream

FAKE = os.path.join(ROOT, "tests", "fake_c8p", "libc8p_fake.so")


@pytest.fixture(scope="module", autouse=True)
def built():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "gr-ieee80211_amd", "host")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "fake_c8p")])


def with_tail(x, seed=0):
    """(the L-SIG duration of the last packet may run a few items past the end of what stream() returns: idle air behind it)"""
    r = np.random.default_rng(seed)
    return np.concatenate([x, 0.004 * (r.normal(size=600) + 1j * r.normal(size=600))]).astype(np.complex64)


def pdus_of(g):
    out = []
    for o in g["outs"]:
        rec = g["pdu_arena"][int(o["pdu_off"]):int(o["pdu_off"]) + int(o["pdu_bytes"])].tobytes()
        for _ in range(int(o["n_pdu"])):
            n = int.from_bytes(rec[:2], "little")
            out.append(rec[2:2 + n])
            rec = rec[2 + n:]
    return out


def test_rawchain_run_delivers_every_segment_in_order():
    order = [NAMES, NAMES[::-1], NAMES[2:6], NAMES[:3], NAMES[4:], NAMES[1:7]]
    xs = [with_tail(stream(nm, CFOS[:len(nm)], snr=33.0, seed=70 + i)[0], i) for i, nm in enumerate(order)]
    xs.insert(3, (0.004 * (np.random.default_rng(1).normal(size=20000) + 1j * np.random.default_rng(2).normal(size=20000))).astype(np.complex64))
    order.insert(3, [])
    got = []
    st = blocks.rawchain_run([(x.ctypes.data, len(x)) for x in xs], engine_lib=FAKE, on_segment=got.append)
    assert st["segments"] == len(xs) and [g["segment"] for g in got] == list(range(len(xs)))
    for g, nm in zip(got, order):
        assert pdus_of(g) == [SWEEP[k]["pdu"].tobytes() for k in nm]
        assert len(g["pkts"]) == len(nm) <= g["n_events"]
    assert st["pdus"] == sum(len(nm) for nm in order) == st["packets"] <= st["events"] and st["items"] == sum(len(x) for x in xs)


@pytest.mark.parametrize("seg_items", [3000, 20011, 1 << 22])
def test_rawchain_stream_carries_detection_state_and_scan_position_across_cuts(seg_items):
    """one capture (packets, then bursts that fail the L-SIG check or fall inside packets, then packets) cut into segments shorter
    than a packet / of an odd length / longer than the capture: the same packets and PDUs as one pass over the capture"""
    x = with_tail(np.concatenate([stream(NAMES, CFOS, snr=33.0, seed=81)[0], burst_stream(13, n_burst=12), stream(NAMES[::-1], CFOS, snr=33.0, seed=82)[0]]))
    one, cut = [], []
    blocks.rawchain_run([(x.ctypes.data, len(x))], engine_lib=FAKE, on_segment=one.append)
    st = blocks.rawchain_run([(x.ctypes.data, len(x))], engine_lib=FAKE, on_segment=cut.append, stream_items=seg_items)
    assert st["segments"] == (len(x) + seg_items - 1) // seg_items == len(cut)
    want = pdus_of(one[0])
    assert len(want) >= 2 * len(NAMES) - 1
    assert [p for g in cut for p in pdus_of(g)] == want
    ref = one[0]["pkts"]
    allp = np.concatenate([g["pkts"] for g in cut])
    assert len(allp) == len(ref)
    for f in ("lsig_mcs", "lsig_len", "lsig_nsamp", "n_items", "cfo"):
        assert np.array_equal(allp[f], ref[f]), f


def test_rawchain_reports_errors_instead_of_hanging():
    import ctypes as C
    L = blocks.lib()
    x = with_tail(stream(NAMES[:2], CFOS[:2], seed=90)[0])
    seg = (blocks._RawSeg * 1)(blocks._RawSeg(x.ctypes.data, len(x), 0, 1, 0))
    st, err = blocks._RawStats(), C.create_string_buffer(256)
    rc = L.c8p_rawchain_run_once(os.path.join(ROOT, "does", "not", "exist.so").encode(), 0, seg, 1, None, None, C.byref(st), err, len(err))
    assert rc != 0 and b"exist.so" in err.value
    with pytest.raises(RuntimeError, match="exist.so"):
        blocks.RawChain(engine_lib=os.path.join(ROOT, "does", "not", "exist.so"))
    cfg = blocks._RawCfg(4, 0, FAKE.encode())                                             # a caller built against another struct layout
    assert not L.c8p_rawchain_create(C.byref(cfg), err, len(err)) and b"struct_size" in err.value
    bad = (blocks._RawSeg * 1)(blocks._RawSeg(x.ctypes.data, len(x), 7, 0, 0))           # host items with a second antenna: refused
    rc = L.c8p_rawchain_run_once(FAKE.encode(), 0, bad, 1, None, None, C.byref(st), err, len(err))
    assert rc != 0 and b"host segments" in err.value
    rc = L.c8p_rawchain_stream_once(FAKE.encode(), 0, x.ctypes.data, len(x), 0, 16, None, None, C.byref(st), err, len(err))   # segments shorter than the history
    assert rc != 0
    h = blocks.RawChain(engine_lib=FAKE)                                                  # no live streams configured
    with pytest.raises(RuntimeError, match="without live streams"):
        h.push(0, x)
    h.close()


def test_rawchain_handle_is_reused_and_its_stats_say_where_the_time_went():
    """one handle, several runs: the contexts outlive a run; the stats carry per-stage clocks, the first-segment and steady-state
    figures, and the timeline has one mark per segment in stage order"""
    xs = [with_tail(stream(NAMES[:4], CFOS[:4], snr=33.0, seed=40 + i)[0], i) for i in range(9)]
    segs = [(x.ctypes.data, len(x)) for x in xs]
    h = blocks.RawChain(engine_lib=FAKE, reserve_seg_items=max(len(x) for x in xs), reserve_seg_packets=8, reserve_pkt_items=5000)
    for rep in range(2):
        got = []
        st = h.run(segs, on_segment=got.append)
        assert st["segments"] == 9 and st["pdus"] == 36 and [g["segment"] for g in got] == list(range(9))
        assert st["steady_segments"] == 5 and st["steady_pdus"] == 20 and 0 < st["steady_s"] < st["seconds"]
        assert 0 < st["first_segment_s"] <= st["seconds"] and st["arena_growths"] == 0 and st["det_giveups"] == 0
        for k in ("detect", "signal", "rx"):
            assert st[k]["busy"] > 0 and st[k]["busy"] + st[k]["wait_in"] + st[k]["wait_out"] <= st["seconds"] * 1.01
        tl = h.timeline()
        assert len(tl) == 9
        for m in tl:
            assert 0 <= m["det_begin"] <= m["det_end"] <= m["sig_begin"] <= m["sig_end"] <= m["rx_submit"] <= m["rx_done"] <= st["seconds"]
        assert all(a["rx_done"] <= b["rx_done"] for a, b in zip(tl, tl[1:]))
        for g in got:
            assert np.array_equal(g["pkt_stream"], np.zeros(4, np.uint32)) and np.all(np.diff(g["pkt_index"].astype(np.int64)) > 0)
    h.close()


def chain_pdus(x):
    """one pass of the chain over a whole stream (one-shot handle, fake backend): (pdus, sync-flag indices of the kept packets)"""
    got = []
    blocks.rawchain_run([(x.ctypes.data, len(x))], engine_lib=FAKE, on_segment=got.append)
    return pdus_of(got[0]), [int(i) for i in got[0]["pkt_index"]]


@pytest.mark.parametrize("chunks", ["even", "ragged", "ragged-short-carry"])
def test_rawchain_live_streams_pushed_in_increments_equal_one_pass_per_stream(chunks, monkeypatch):
    """K = 4 independent streams fed in increments (even: 6000 items per round; ragged: random sizes incl. empty ones, so that rounds
    end inside preambles, inside packets and inside idle air): per stream, in order, the packets and PDUs of one pass over that stream.
    short-carry: only the last 2600 items of a stream go into the next round (the packets here are at most 1600 + 224 items long), so
    the arenas' origins move along the streams as they do after the first 110 000 items of a real one"""
    short = chunks.endswith("short-carry")
    if short:       # (no random bursts then: one of them may carry a valid L-SIG that claims a packet longer than this carry)
        monkeypatch.setenv("C8P_RAWCHAIN_KEEP", "2600")
        chunks = "ragged"
    rng = np.random.default_rng(7 if chunks == "even" else 8)
    orders = [NAMES, NAMES[::-1], NAMES[2:7], []]
    xs = []
    for i, nm in enumerate(orders):
        parts = [stream(nm, CFOS[:len(nm)], snr=33.0, seed=120 + i)[0]] if nm else []
        if i == 1 and not short:
            parts.insert(0, burst_stream(21, n_burst=6))
        if i == 2:
            parts.append(stream(NAMES[:3], CFOS[:3], snr=33.0, seed=131)[0])
        parts.append(0.004 * (rng.normal(size=3000) + 1j * rng.normal(size=3000)))
        xs.append(np.concatenate(parts).astype(np.complex64))
    want = [chain_pdus(x) for x in xs]
    got = []
    h = blocks.RawChain(engine_lib=FAKE, streams=4, stream_round_items=20000)
    h.live(got.append)
    pos = [0] * 4
    while any(p < len(x) for p, x in zip(pos, xs)):
        for s, x in enumerate(xs):
            n = 6000 if chunks == "even" else int(rng.choice([0, 0, 137, 2500, 9000, 19999]))
            n = min(n, len(x) - pos[s])
            if n:
                h.push(s, x[pos[s]:pos[s] + n])
                pos[s] += n
        h.round(wait=(chunks != "even"))          # (even: c8p_rawchain_round_async, the detect stage waits for the copies)
    st = h.drain()
    assert [g["segment"] for g in got] == list(range(len(got))) and st["segments"] == len(got) and st["items"] == sum(len(x) for x in xs)
    per = [([], []) for _ in range(4)]
    for g in got:
        assert np.all(np.diff(g["pkt_stream"].astype(np.int64)) >= 0)
        recs = pdus_of(g)
        assert len(recs) == int(g["outs"]["n_pdu"].sum())
        k = 0
        for i in range(len(g["pkts"])):
            s = int(g["pkt_stream"][i])
            per[s][1].append(int(g["pkt_index"][i]))
            for _ in range(int(g["outs"]["n_pdu"][i])):
                per[s][0].append(recs[k]); k += 1
    for s in range(4):
        assert per[s][0] == want[s][0], s
        assert per[s][1] == want[s][1], s
    assert sum(len(w[0]) for w in want) == st["pdus"] >= 20
    # the handle starts another session from the constructor state
    got2 = []
    h.live(got2.append)
    half = len(xs[0]) // 2
    h.push(0, xs[0][:half]); h.round()
    h.push(0, xs[0][half:]); h.round()
    st2 = h.drain()
    assert st2["pdus"] == len(want[0][0]) and [p for g in got2 for p in pdus_of(g)] == want[0][0]
    h.close()


def test_rawchain_pushv_is_many_pushes():
    xs = [with_tail(stream(NAMES[:3], CFOS[:3], snr=33.0, seed=700 + i)[0]) for i in range(3)]
    ref = [chain_pdus(x)[0] for x in xs]
    got = []
    h = blocks.RawChain(engine_lib=FAKE, streams=3, stream_round_items=30000)
    h.live(got.append)
    h.pushv([0, 1, 2], [x.ctypes.data for x in xs], [len(x) for x in xs])
    h.round()
    st = h.drain()
    h.close()
    assert st["pdus"] == sum(len(r) for r in ref) == 9
    by = {s: [] for s in range(3)}
    for g in got:
        recs = pdus_of(g)
        k = 0
        for i in range(len(g["pkts"])):
            for _ in range(int(g["outs"]["n_pdu"][i])):
                by[int(g["pkt_stream"][i])].append(recs[k]); k += 1
    assert [by[s] for s in range(3)] == ref


@pytest.mark.parametrize("bits", [16, 8])
def test_rawchain_live_sc16_items_are_the_converted_stream(bits):
    """pushes flagged C8P_MEM_SC16: int16 I/Q pairs in ragged increments (so the pieces start at any 4-byte item) = one pass over the
    complex64 stream a host-side conversion makes of the same pairs; the handle's scale is c8p_rawchain_cfg.sc16_scale"""
    from test_multistream import to_sc16
    rng = np.random.default_rng(31)
    xs = [with_tail(stream(nm, CFOS[:len(nm)], snr=33.0, seed=900 + i)[0], seed=i) for i, nm in enumerate([NAMES, NAMES[::-1][:6], NAMES[3:8]])]
    scale = np.float32(1.02 * max(float(np.abs(x.view(np.float32)).max()) for x in xs) / ((1 << (bits - 1)) - 1))
    qs, xf = zip(*[to_sc16(x, scale, bits) for x in xs])
    want = [chain_pdus(np.ascontiguousarray(x)) for x in xf]
    got = []
    h = blocks.RawChain(engine_lib=FAKE, streams=3, stream_round_items=20000, sc16_scale=float(scale), sc8_scale=float(scale))
    h.live(got.append)
    pos = [0] * 3
    while any(p < len(x) for p, x in zip(pos, xf)):
        for s, q in enumerate(qs):
            n = min(int(rng.choice([0, 1, 137, 2501, 9000, 19999])), len(xf[s]) - pos[s])
            if n:
                h.push(s, q[2 * pos[s]:2 * (pos[s] + n)], host="pinned", sc16=bits)
                pos[s] += n
        h.round()
    st = h.drain()
    h.close()
    per = [([], []) for _ in range(3)]
    for g in got:
        recs = pdus_of(g)
        k = 0
        for i in range(len(g["pkts"])):
            s = int(g["pkt_stream"][i])
            per[s][1].append(int(g["pkt_index"][i]))
            for _ in range(int(g["outs"]["n_pdu"][i])):
                per[s][0].append(recs[k]); k += 1
    for s in range(3):
        assert per[s][0] == want[s][0] and per[s][1] == want[s][1], s
    assert st["pdus"] == sum(len(w[0]) for w in want) >= (15 if bits == 16 else 8) and st["items"] == sum(len(x) for x in xf)


def live_rounds_are_delivered_without_waiting_for_the_next(engine_lib, timeout_s=5.0):
    """a live session at a radio's pace: every round's callback fires on its own — not when the next round comes by (the rx stage collects
    a round when it has finished OR the next one arrives, whichever comes first)"""
    import time
    x = with_tail(stream(NAMES, CFOS[:len(NAMES)], snr=33.0, seed=77)[0])
    got = []
    h = blocks.RawChain(engine_lib=engine_lib, streams=1, stream_round_items=20000)
    h.live(got.append)
    step = 6000
    for r, pos in enumerate(range(0, len(x), step)):
        h.push(0, x[pos:pos + step])
        h.round()
        t_end = time.time() + timeout_s
        while len(got) < r + 1:                      # nothing else is pushed until this round has been delivered
            assert time.time() < t_end, "round %d was not delivered before the next one was pushed" % r
            time.sleep(0.001)
    st = h.drain()
    h.close()
    assert st["segments"] == len(got) and [p for g in got for p in pdus_of(g)] == chain_pdus(x)[0]


def test_rawchain_live_rounds_are_delivered_without_waiting_for_the_next():
    live_rounds_are_delivered_without_waiting_for_the_next(FAKE)


def test_rawchain_live_refuses_more_than_a_round_holds():
    h = blocks.RawChain(engine_lib=FAKE, streams=2, stream_round_items=4096)
    x = np.zeros(5000, np.complex64)
    h.push(1, x[:4096])
    with pytest.raises(RuntimeError, match="stream_round_items"):
        for _ in range(40):
            h.push(1, x)
    with pytest.raises(RuntimeError, match="no such stream"):
        h.push(2, x[:10])
    h.drain()
    h.close()


@pytest.mark.parametrize("slots", ["1", "2"])
def test_rawchain_with_fewer_engine_slots(slots, monkeypatch):
    """C8P_SLOTS = 1 / 2 (a documented switch; tools/profile_gpu.sh sets 1): the rx stage collects a segment before it submits the next
    when only one submit stays resident, and the detect stage holds at most `slots` segments between c8p_detect and the end of
    c8p_signal (the engine's copy of host items lives in a slot): the same PDUs as with three slots, device and host segments"""
    xs = [with_tail(stream(NAMES[:5], CFOS[:5], snr=33.0, seed=500 + i)[0], i) for i in range(7)]
    segs = [(x.ctypes.data, len(x)) for x in xs]
    ref = []
    blocks.rawchain_run(segs, engine_lib=FAKE, on_segment=ref.append)
    monkeypatch.setenv("C8P_SLOTS", slots)
    for host in (False, True):
        got = []
        st = blocks.rawchain_run(segs, engine_lib=FAKE, on_segment=got.append, host=host)
        assert st["segments"] == 7 and st["pdus"] == 35
        assert [pdus_of(g) for g in got] == [pdus_of(g) for g in ref]


def test_rawchain_live_two_antenna_streams_on_the_test_engine():
    """rx2.grc's chain per live stream (c8p_rawchain_cfg.stream_ant = 2, c8p_rawchain_push2): see test_multistream"""
    from test_multistream import live_two_antennas_equal_one_pass
    live_two_antennas_equal_one_pass(FAKE, 8)
    live_two_antennas_equal_one_pass(FAKE, 6, sc16=True)


def test_rawchain_live_session_reports_a_failing_stage_and_does_not_hang(monkeypatch):
    """the third detection pass fails inside the engine: the caller's next round() (or the drain) reports it, nothing blocks, and the
    handle starts a clean session afterwards"""
    monkeypatch.setenv("C8P_FAKE_FAIL_DETECT_MULTI_AT", "3")
    x = with_tail(stream(NAMES, CFOS, snr=33.0, seed=77)[0])
    h = blocks.RawChain(engine_lib=FAKE, streams=2, stream_round_items=8192)
    got = []
    h.live(got.append)
    raised = None
    try:
        for r in range(12):
            h.push(0, x[4000 * r:4000 * r + 4000])
            h.round()
        h.drain()
    except RuntimeError as e:
        raised = str(e)
    assert raised and "c8p_detect_multi" in raised, raised
    try:
        h.drain()
    except RuntimeError:
        pass
    monkeypatch.delenv("C8P_FAKE_FAIL_DETECT_MULTI_AT")
    got2 = []
    h.live(got2.append)
    h.push(1, x[:8000]); h.round()
    st = h.drain()
    assert st["segments"] == 1 and len(got2) == 1
    h.close()
