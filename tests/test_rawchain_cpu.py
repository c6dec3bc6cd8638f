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
    rc = L.c8p_rawchain_run(os.path.join(ROOT, "does", "not", "exist.so").encode(), 0, seg, 1, None, None, C.byref(st), err, len(err))
    assert rc != 0 and b"exist.so" in err.value
    bad = (blocks._RawSeg * 1)(blocks._RawSeg(x.ctypes.data, len(x), 7, 0, 0))           # host items with a second antenna: refused
    rc = L.c8p_rawchain_run(FAKE.encode(), 0, bad, 1, None, None, C.byref(st), err, len(err))
    assert rc != 0 and b"host segments" in err.value
    rc = L.c8p_rawchain_stream(FAKE.encode(), 0, x.ctypes.data, len(x), 0, 16, None, None, C.byref(st), err, len(err))   # segments shorter than the history
    assert rc != 0
