"""Host-side block mirror (gr::ieee80211::demod / demod2 / decode general_work contract).

CPU: the C++ batching fronts run against tests/fake_c8p (the C ABI on top of the oracle — test-only) so that the
stream cutting, batching, tag replay, PDU layout and debug text are checked without a GPU.
GPU (-m gpu): the same streams through the real libc8p_rx.so, demod chained into decode like rx.grc.
"""
import os
import subprocess

import numpy as np
import pytest

from c8p_cases import sweep_batch
from c8p_testlib import ROOT, abi, oracle_rx
from gr_ieee80211_amd import blocks

FAKE = os.path.join(ROOT, "tests", "fake_c8p", "libc8p_fake.so")


@pytest.fixture(autouse=True)
def gather_at_any_size(monkeypatch):
    """the fronts gather out of a pinned buffer only when it is megabytes large (C8P_GATHER_MIN_BUFFER): here always, so that the
    small streams of these tests go through the gather path whenever a test asks for pinned buffers"""
    monkeypatch.setenv("C8P_GATHER_MIN_BUFFER", "0")


@pytest.fixture(scope="module", autouse=True)
def built():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "gr-ieee80211_amd", "host")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "fake_c8p")])


def split(b, n_ant):
    """the SISO flowgraph (rx.grc) and the 2-antenna one (rx2.grc) are different streams"""
    pk, iq, chan = b.arrays()
    sel = pk[(pk["n_ant"] == n_ant) & (pk["mupos"] == 0)]     # one demod block = one MU user position
    return sel, iq, chan


def chain(engine_lib, pk, iq, chan, n_ant, chunk, nout, max_batch, mupos=0):
    dem = blocks.Block(blocks.DEMOD if n_ant == 1 else blocks.DEMOD2, mupos=mupos, engine_lib=engine_lib, max_batch=max_batch)
    blocks.feed_demod(dem, pk, iq, chan)
    dem.run(chunk=chunk, nout=nout)
    llr, tags = dem.output(), dem.tags()
    dec = blocks.Block(blocks.DECODE, ifdebug=True, engine_lib=engine_lib, max_batch=max_batch)
    dec.push_items(0, llr)
    for off, key, val in tags:
        dec.push_tag(off, key, val)
    dec.run(chunk=chunk, nout=nout)
    res = (llr, tags, dec.messages(), dec.debug_lines())
    dem.close(); dec.close()
    return res


def check_against_oracle(res, pk, iq, chan):
    llr, tags, msgs, dbg = res
    ref = oracle_rx(pk, iq, chan)
    exp_llr = np.concatenate([r["llr"] if r["tags"]["status"] == 0 else np.zeros(0, np.float32) for r in ref] + [np.zeros(0, np.float32)])
    assert len(llr) == len(exp_llr)
    assert np.max(np.abs(llr - exp_llr), initial=0) <= 1e-5
    # tags sit on the first item of every accepted packet, with the reference's keys
    starts, pos = [], 0
    for r in ref:
        if r["tags"]["status"] == 0:
            starts.append(pos)
            pos += len(r["llr"])
    by_off = {}
    for off, key, val in tags:
        by_off.setdefault(off, {})[key] = val
    assert sorted(by_off) == starts
    k = 0
    for r in ref:
        t = r["tags"]
        if t["status"] != 0:
            continue
        d = by_off[starts[k]]; k += 1
        for key in ("format", "mcs", "len", "cr", "ampdu", "trellis", "total"):
            assert d[key] == int(t[key]), key
        assert ("sssnr0" in d) == (t["format"] == abi.F_VHT)
        assert ("mu2x1chan" in d) == bool(t["is_ndp"])
        assert {"cfo", "snr", "rssi"} <= set(d)
    exp_msgs = [(len(x), x) for r in ref for x in r["pdus"]]
    assert msgs == exp_msgs
    n_ok = sum(1 for r in ref for x in r["pdus"] if r["tags"]["trellis"] > 0)
    assert sum("crc32 correct" in s for s in dbg) == n_ok
    if dbg:
        assert dbg[-1].startswith("ieee80211 decode, ")
        assert ",0:" in dbg[-1] and ",cfo:" in dbg[-1]          # the fields tools/performance/perf_siso.py:105-118 scrapes
    return len(exp_msgs)


@pytest.mark.parametrize("n_ant", [1, 2])
@pytest.mark.parametrize("chunk,nout,max_batch", [(4096, 8192, 256), (333, 500, 5), (80, 97, 1)])
def test_block_mirror_with_fake_backend(n_ant, chunk, nout, max_batch):
    b, _ = sweep_batch(snr_db=28.0, seed=5)
    pk, iq, chan = split(b, n_ant)
    res = chain(FAKE, pk, iq, chan, n_ant, chunk, nout, max_batch)
    assert check_against_oracle(res, pk, iq, chan) >= (20 if n_ant == 1 else 10)


def test_missing_library_fails_loudly():
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        blocks.Block(blocks.DEMOD, engine_lib="/nonexistent/libc8p_rx.so")


@pytest.mark.gpu
@pytest.mark.parametrize("n_ant", [1, 2])
def test_block_mirror_on_gpu(n_ant):
    b, _ = sweep_batch(snr_db=28.0, seed=6)
    pk, iq, chan = split(b, n_ant)
    res = chain(None, pk, iq, chan, n_ant, 1000, 4096, 16)
    assert check_against_oracle(res, pk, iq, chan) >= 10


# ---------------------------------------------------------------------------------------------- signal / signal2 mirror
def _signal_case(n_ant, seed=9):
    """raw stream(s) + the sync block's flagged items (oracle detection on antenna 0), plus two flags the reference
    would treat specially: one inside a packet that is being copied (never seen) and one on noise (L-SIG check fails,
    80 items skipped)."""
    from c8p_cases import dewindow
    from c8p_testlib import H22, load_npz, oracle, through_channel
    sw = load_npz("packets_sweep.npz")
    names = ["L_0", "HT_3", "VHT_2", "L_5", "VHT_8"] if n_ant == 1 else ["HT2_12", "VHT2_5", "HT2_9"]
    rng = np.random.default_rng(seed)
    parts, exp = [], []
    for k, name in enumerate(names):
        d = sw[name]
        rx = through_channel(dewindow(d["tx"]), H22 if n_ant == 2 else None, 27.0, rng, lead=int(rng.integers(500, 1200)))
        w = 2 * np.pi * (30e3 * (k - 2)) / 20e6
        parts.append((rx * np.exp(1j * w * np.arange(rx.shape[1]))).astype(np.complex64))
        exp.append(d)
    raw = np.concatenate(parts, axis=1)
    o = oracle()
    ev, _, _, _ = o.detect(raw[0])
    extra = np.zeros(2, abi.SYNC_EVT_DT)
    extra[0]["index"] = int(ev[1]["index"]) + 700          # inside the second packet's copy region
    extra[1]["index"] = int(ev[2]["index"]) - 450          # idle air before the third packet
    extra["snr"], extra["rssi"] = 10.0, 0.5
    ev = np.sort(np.concatenate([ev, extra]), order="index")
    return raw, ev, exp


def _expected_signal_output(raw, ev, n_ant):
    from c8p_testlib import oracle
    from test_detect_frontend import consume_filter
    o = oracle()
    s = np.zeros(len(ev), abi.SIG_IN_DT)
    n = raw.shape[1]
    arena = np.concatenate([raw[a] for a in range(n_ant)])
    for i, e in enumerate(ev):
        s[i]["raw_off"][0] = int(e["index"]); s[i]["raw_off"][1] = n + int(e["index"]) if n_ant == 2 else 0
        s[i]["n_avail"] = n - int(e["index"])
        s[i]["cfo_rad"], s[i]["snr"], s[i]["rssi"], s[i]["n_ant"] = e["rad"], e["snr"], e["rssi"], n_ant
    pk, iq, chan = o.signal_events(s, arena)
    keep = consume_filter(ev, pk)
    return pk, iq, chan, keep


def _run_signal(engine_lib, raw, ev, n_ant, chunk, nout, max_batch):
    sig = blocks.Block(blocks.SIGNAL if n_ant == 1 else blocks.SIGNAL2, engine_lib=engine_lib, max_batch=max_batch)
    blocks.feed_signal(sig, ev, [raw[a] for a in range(n_ant)])
    sig.run(chunk=chunk, nout=nout)
    outs = [sig.output(a) for a in range(n_ant)]
    tags = sig.tags()
    sig.close()
    return outs, tags


def _check_signal(outs, tags, raw, ev, n_ant, tol):
    pk, iq, chan, keep = _expected_signal_output(raw, ev, n_ant)
    assert len(keep) >= 3
    pos = 0
    by_off = {}
    for off, key, val in tags:
        by_off.setdefault(off, {})[key] = val
    starts = []
    for seq, i in enumerate(keep, 1):
        p = pk[i]
        n = int(p["n_items"])
        starts.append(pos)
        d = by_off[pos]
        assert d["seq"] == seq and d["mcs"] == int(p["lsig_mcs"]) and d["len"] == int(p["lsig_len"]) and d["nsamp"] == int(p["lsig_nsamp"])
        assert abs(d["cfo"] - float(p["cfo"])) <= 1e-3 * max(1.0, abs(float(p["cfo"]))) and d["snr"] == float(p["snr"])
        c = chan[64 * i:64 * i + 64]
        assert np.max(np.abs(d["chan"] - c)) <= tol * max(np.max(np.abs(c)), 1e-9)
        for a in range(n_ant):
            want = iq[int(p["iq_off"][a]):int(p["iq_off"][a]) + n]
            got = outs[a][pos:pos + n]
            assert np.max(np.abs(got - want)) <= tol * np.max(np.abs(want)), (i, a)
        pos += n
    assert sorted(by_off) == starts
    assert all(len(o) == pos for o in outs)
    return pk[keep]


@pytest.mark.parametrize("n_ant", [1, 2])
@pytest.mark.parametrize("chunk,nout,max_batch", [(4096, 8192, 256), (1000, 777, 2), (160, 300, 1)])
def test_signal_mirror_with_fake_backend(n_ant, chunk, nout, max_batch):
    raw, ev, exp = _signal_case(n_ant)
    outs, tags = _run_signal(FAKE, raw, ev, n_ant, chunk, nout, max_batch)
    _check_signal(outs, tags, raw, ev, n_ant, 0.0)


def test_block_chain_raw_stream_to_pdus_with_fake_backend():
    """signal -> demod -> decode, the rx.grc chain downstream of the sync block, on the block mirrors"""
    raw, ev, exp = _signal_case(1)
    sig = blocks.Block(blocks.SIGNAL, engine_lib=FAKE, max_batch=3)
    blocks.feed_signal(sig, ev, [raw[0]])
    sig.run(chunk=2000, nout=3000)
    dem = blocks.Block(blocks.DEMOD, engine_lib=FAKE, max_batch=2)
    dem.push_items(0, sig.output(0))
    for off, key, val in sig.tags():
        dem.push_tag(off, key, val)
    dem.run(chunk=1500, nout=4096)
    dec = blocks.Block(blocks.DECODE, engine_lib=FAKE, max_batch=4)
    dec.push_items(0, dem.output())
    for off, key, val in dem.tags():
        dec.push_tag(off, key, val)
    dec.run(chunk=3000, nout=100)
    assert [m[1] for m in dec.messages()] == [d["pdu"].tobytes() for d in exp]
    for b in (sig, dem, dec):
        b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n_ant", [1, 2])
def test_signal_mirror_on_gpu(n_ant):
    raw, ev, exp = _signal_case(n_ant, seed=10)
    outs, tags = _run_signal(None, raw, ev, n_ant, 3000, 4096, 4)
    _check_signal(outs, tags, raw, ev, n_ant, 2e-6)


# ---------------------------------------------------------------------------------------------- detect mirror + whole chain
def _rx_chain_on_mirrors(engine_lib, raw, chunk, items_per_detect):
    """rx.grc on the block mirrors: detect (presiso + trigger + sync) -> signal -> demod -> decode"""
    det = blocks.Block(blocks.DETECT, engine_lib=engine_lib, max_batch=items_per_detect)
    det.push_items(0, raw)
    det.run(chunk=chunk, nout=chunk)
    sync_bytes, sync_tags = det.output(), det.tags()
    sig = blocks.Block(blocks.SIGNAL, engine_lib=engine_lib, max_batch=4)
    sig.push_items(0, sync_bytes)
    sig.push_items(1, raw[:len(sync_bytes)])
    for off, key, val in sync_tags:
        sig.push_tag(off, key, val)
    sig.run(chunk=chunk, nout=4096)
    dem = blocks.Block(blocks.DEMOD, engine_lib=engine_lib, max_batch=4)
    dem.push_items(0, sig.output(0))
    for off, key, val in sig.tags():
        dem.push_tag(off, key, val)
    dem.run(chunk=2048, nout=4096)
    dec = blocks.Block(blocks.DECODE, engine_lib=engine_lib, max_batch=8)
    dec.push_items(0, dem.output())
    for off, key, val in dem.tags():
        dec.push_tag(off, key, val)
    dec.run(chunk=4096, nout=64)
    msgs = [m[1] for m in dec.messages()]
    for b in (det, sig, dem, dec):
        b.close()
    return sync_bytes, sync_tags, msgs


def _detect_case(seed=4):
    from test_detect_frontend import NAMES, CFOS, stream
    x, starts, exp = stream(NAMES[:5], CFOS[:5], seed=seed)
    return np.concatenate([x, np.zeros(600, np.complex64)]), exp      # idle tail: the last packet's windows complete


@pytest.mark.parametrize("chunk,items", [(8192, 1 << 20), (1500, 4000), (700, 900)])
def test_detect_mirror_and_full_chain_with_fake_backend(chunk, items):
    from c8p_testlib import oracle
    raw, exp = _detect_case()
    sync_bytes, sync_tags, msgs = _rx_chain_on_mirrors(FAKE, raw, chunk, items)
    ev, _, _, _ = oracle().detect(raw)
    flagged = np.nonzero(sync_bytes)[0]
    assert list(flagged) == [int(e["index"]) for e in ev if int(e["index"]) < len(sync_bytes)]
    by_off = {}
    for off, key, val in sync_tags:
        by_off.setdefault(off, {})[key] = val
    assert sorted(by_off) == list(flagged)
    for e in ev[:len(flagged)]:
        d = by_off[int(e["index"])]
        assert d["rad"] == float(e["rad"]) and d["snr"] == float(e["snr"]) and d["rssi"] == float(e["rssi"])
    assert msgs == [d["pdu"].tobytes() for d, w in exp]


@pytest.mark.gpu
@pytest.mark.parametrize("chunk,items", [(8192, 1 << 20), (1500, 4000)])
def test_full_chain_on_gpu_mirrors(chunk, items):
    """raw IQ -> detect -> signal -> demod -> decode -> PDUs, every block a front of the HIP engine"""
    raw, exp = _detect_case(seed=6)
    sync_bytes, sync_tags, msgs = _rx_chain_on_mirrors(None, raw, chunk, items)
    assert msgs == [d["pdu"].tobytes() for d, w in exp]


# ---------------------------------------------------------------------------------------------- scheduler jitter
@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_blocks_under_random_scheduler_chunks_with_fake_backend(seed):
    """every general_work() call gets a random number of input items (sometimes none, sometimes more on one port) and a
    random output space, as a loaded scheduler hands them out: the item streams, tags and PDUs must not depend on it"""
    rng = np.random.default_rng(seed)
    n_ant = 1 + seed % 2
    # demod(2) -> decode
    b, _ = sweep_batch(snr_db=28.0, seed=5)
    pk, iq, chan = split(b, n_ant)
    dem = blocks.Block(blocks.DEMOD if n_ant == 1 else blocks.DEMOD2, engine_lib=FAKE, max_batch=int(rng.integers(1, 9)))
    blocks.feed_demod(dem, pk, iq, chan)
    dem.set_jitter(1000 + seed)
    dem.run(chunk=int(rng.integers(100, 3000)), nout=int(rng.integers(50, 5000)))
    llr, tags = dem.output(), dem.tags()
    dec = blocks.Block(blocks.DECODE, ifdebug=True, engine_lib=FAKE, max_batch=int(rng.integers(1, 9)))
    dec.push_items(0, llr)
    for off, key, val in tags:
        dec.push_tag(off, key, val)
    dec.set_jitter(2000 + seed)
    dec.run(chunk=int(rng.integers(200, 4000)), nout=int(rng.integers(1, 300)))
    res = (llr, tags, dec.messages(), dec.debug_lines())
    dem.close(); dec.close()
    assert check_against_oracle(res, pk, iq, chan) >= 10
    # signal(2)
    raw, ev, exp = _signal_case(n_ant)
    sig = blocks.Block(blocks.SIGNAL if n_ant == 1 else blocks.SIGNAL2, engine_lib=FAKE, max_batch=int(rng.integers(1, 5)))
    blocks.feed_signal(sig, ev, [raw[a] for a in range(n_ant)])
    sig.set_jitter(3000 + seed)
    sig.run(chunk=int(rng.integers(100, 3000)), nout=int(rng.integers(100, 3000)))
    outs, stags = [sig.output(a) for a in range(n_ant)], sig.tags()
    sig.close()
    _check_signal(outs, stags, raw, ev, n_ant, 0.0)


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_detect_mirror_under_random_scheduler_chunks_with_fake_backend(seed):
    raw, exp = _detect_case()
    ref_bytes, ref_tags, ref_msgs = _rx_chain_on_mirrors(FAKE, raw, 8192, 1 << 20)
    rng = np.random.default_rng(seed)
    det = blocks.Block(blocks.DETECT, engine_lib=FAKE, max_batch=int(rng.integers(700, 5000)))
    det.push_items(0, raw)
    det.set_jitter(seed)
    c = int(rng.integers(300, 3000))
    det.run(chunk=c, nout=c)
    got_bytes, got_tags = det.output(), det.tags()
    det.close()
    assert np.array_equal(got_bytes, ref_bytes)
    assert [(o, k) for o, k, v in got_tags] == [(o, k) for o, k, v in ref_tags]
    for (o, k, v), (o2, k2, v2) in zip(got_tags, ref_tags):
        assert abs(v - v2) <= 1e-6 * max(1.0, abs(v2)), (o, k)


# ---------------------------------------------------------------------------------------------- fused mode, threaded chain
def _chain_threads(engine_lib, pk, iq, chan, n_ant, max_batch, fused, pinned_io=None, defer_consume=None, poison=False, chunk=3000, nout=5000):
    dem = blocks.Block(blocks.DEMOD if n_ant == 1 else blocks.DEMOD2, engine_lib=engine_lib, max_batch=max_batch, fused=fused,
                       pinned_io=pinned_io, defer_consume=defer_consume)
    dec = blocks.Block(blocks.DECODE, ifdebug=True, engine_lib=engine_lib, max_batch=max_batch, fused=fused, pinned_io=pinned_io,
                       defer_consume=defer_consume)
    if poison:
        dem.set_poison(True); dec.set_poison(True)
    blocks.feed_demod(dem, pk, iq, chan)
    blocks.run_chain(dem, dec, chunk=chunk, nout=nout)
    res = (dec.messages(), dec.debug_lines())
    dem.close(); dec.close()
    return res


# ---------------------------------------------------------------------------------------------- custom (pinned) scheduler buffers
@pytest.mark.parametrize("n_ant", [1, 2])
@pytest.mark.parametrize("pinned_io,defer", [(False, True), (True, False), (True, True)])
@pytest.mark.parametrize("chunk,nout,max_batch", [(4096, 8192, 256), (333, 500, 5), (80, 97, 1)])
def test_block_mirror_with_and_without_pinned_buffers(n_ant, pinned_io, defer, chunk, nout, max_batch):
    """The mirrors with the scheduler's buffers allocated through their buffer type (pinned: the packets are gathered out of the input
    buffer by the device, demod's LLR items are written into the output buffer by the device; the items are consumed one call later, or in
    the same call after a wait) and with ordinary buffers (staged through pinned copies): the same streams, tags, PDUs, debug lines.
    The harness OVERWRITES every consumed item at once and the test engine's gathers run only when something orders or awaits them
    (tests/fake_c8p: as asynchronous as the device), so an item let go of too early or a submit without c8p_copy_fence shows."""
    b, _ = sweep_batch(snr_db=28.0, seed=5)
    pk, iq, chan = split(b, n_ant)
    dem = blocks.Block(blocks.DEMOD if n_ant == 1 else blocks.DEMOD2, engine_lib=FAKE, max_batch=max_batch, pinned_io=pinned_io, defer_consume=defer)
    assert dem.input_pinned(0) == pinned_io and dem.output_pinned() == pinned_io
    dem.set_poison(True)
    blocks.feed_demod(dem, pk, iq, chan)
    dem.run(chunk=chunk, nout=nout)
    llr, tags = dem.output(), dem.tags()
    dec = blocks.Block(blocks.DECODE, ifdebug=True, engine_lib=FAKE, max_batch=max_batch, pinned_io=pinned_io, defer_consume=defer)
    assert dec.input_pinned(0) == pinned_io
    dec.set_poison(True)
    dec.push_items(0, llr)
    for off, key, val in tags:
        dec.push_tag(off, key, val)
    dec.run(chunk=chunk, nout=nout)
    res = (llr, tags, dec.messages(), dec.debug_lines())
    dem.close(); dec.close()
    assert check_against_oracle(res, pk, iq, chan) >= (20 if n_ant == 1 else 10)


@pytest.mark.parametrize("n_ant", [1, 2])
@pytest.mark.parametrize("every", [1, 2, 5])
def test_deferred_consume_survives_a_buffer_that_relocates_its_items(n_ant, every):
    """A single-mapped GNU Radio buffer MOVES unread items to its start when a reader is blocked at its end.  With deferred consume a front
    has gathers under way on exactly such items; the buffer announces the move (c8p::buffer_will_move, gnuradio/lib/c8p_pinned_buffer.h)
    and the front waits for them first.  Here the harness moves the unread items of the pinned input buffers to other memory before every
    1st / 2nd / 5th call and overwrites the old memory; the test engine runs a gather only when something waits for it."""
    b, _ = sweep_batch(snr_db=28.0, seed=5)
    pk, iq, chan = split(b, n_ant)
    dem = blocks.Block(blocks.DEMOD if n_ant == 1 else blocks.DEMOD2, engine_lib=FAKE, max_batch=5, pinned_io=True, defer_consume=True)
    dem.set_poison(True); dem.set_relocate(every)
    blocks.feed_demod(dem, pk, iq, chan)
    dem.run(chunk=700, nout=3000)
    llr, tags = dem.output(), dem.tags()
    dec = blocks.Block(blocks.DECODE, ifdebug=True, engine_lib=FAKE, max_batch=5, pinned_io=True, defer_consume=True)
    dec.set_poison(True); dec.set_relocate(every)
    dec.push_items(0, llr)
    for off, key, val in tags:
        dec.push_tag(off, key, val)
    dec.run(chunk=2500, nout=3000)
    res = (llr, tags, dec.messages(), dec.debug_lines())
    dem.close(); dec.close()
    assert check_against_oracle(res, pk, iq, chan) >= (20 if n_ant == 1 else 10)


def moved_output_run(engine_lib, n_ant, every, nout, max_batch=5):
    b, _ = sweep_batch(snr_db=28.0, seed=5)
    pk, iq, chan = split(b, n_ant)
    dem = blocks.Block(blocks.DEMOD if n_ant == 1 else blocks.DEMOD2, engine_lib=engine_lib, max_batch=max_batch, pinned_io=True, defer_consume=True, fused=False)
    assert dem.output_pinned()
    dem.set_poison(True); dem.set_move_out(every)
    blocks.feed_demod(dem, pk, iq, chan)
    dem.run(chunk=700, nout=nout)
    llr, tags, moves = dem.output(), dem.tags(), dem.moves_done()
    dec = blocks.Block(blocks.DECODE, ifdebug=True, engine_lib=engine_lib, max_batch=max_batch)
    dec.push_items(0, llr)
    for off, key, val in tags:
        dec.push_tag(off, key, val)
    dec.run(chunk=2500, nout=3000)
    res = (llr, tags, dec.messages(), dec.debug_lines())
    dem.close(); dec.close()
    assert moves >= 3
    assert check_against_oracle(res, pk, iq, chan) >= (20 if n_ant == 1 else 10)


@pytest.mark.parametrize("n_ant", [1, 2])
@pytest.mark.parametrize("every,nout", [(1, 3000), (2, 3000), (3, 700), (5, 40000)])
def test_items_placed_ahead_survive_a_buffer_that_moves_its_write_position(n_ant, every, nout):
    """demod writes LLR items into a pinned output buffer AHEAD of produce(); GNU Radio's single-mapped buffers move the write
    position from their blocked callbacks (d_write_index += gap, or back to the start) without asking the writer, so the items ahead would
    stay behind and the next call would report whatever lies at the new position.  The buffer brackets its moves with
    c8p::buffer_move_scope: the front waits for the launches that write those items, they are saved, and put back where the write
    position is afterwards.  Here the harness moves the output position before every 1st / 2nd / 3rd / 5th call and fills the rest of
    the buffer with NaNs; items, tags, PDUs as the oracle's."""
    moved_output_run(FAKE, n_ant, every, nout)


def test_a_buffer_that_moves_without_saying_so_is_caught(monkeypatch):
    """the same moves WITHOUT the scope (C8P_TEST_UNBRACKETED_MOVE=1: the harness just changes the position): the front refuses to report
    items from a place it did not put them — an error, not garbage LLRs with correct tags"""
    monkeypatch.setenv("C8P_TEST_UNBRACKETED_MOVE", "1")
    b, _ = sweep_batch(snr_db=28.0, seed=5)
    pk, iq, chan = split(b, 1)
    dem = blocks.Block(blocks.DEMOD, engine_lib=FAKE, max_batch=5, pinned_io=True, defer_consume=True, fused=False)
    dem.set_move_out(1)
    blocks.feed_demod(dem, pk, iq, chan)
    with pytest.raises(RuntimeError, match="output space moved"):
        dem.run(chunk=700, nout=3000)
    dem.close()


@pytest.mark.gpu
@pytest.mark.parametrize("every,nout", [(1, 3000), (3, 40000)])
def test_items_placed_ahead_survive_a_moving_buffer_on_the_engine(every, nout):
    moved_output_run(None, 1, every, nout, max_batch=16)


@pytest.mark.parametrize("defer", [False, True])
@pytest.mark.parametrize("max_batch", [3, 64])
def test_arenas_that_grow_while_gathers_fill_them(defer, max_batch, monkeypatch):
    """A batch's device arena grows while the call's gathers into it are under way (C8P_ARENA_MIN makes it start at 4 KB): what has been cut
    so far must have ARRIVED before the old arena's contents move to the new one — the gathers are C8P_PIECE_UNORDERED, they run on either
    of two copy streams and the move is ordered behind one of them only (the test engine leaves them pending past a c8p_copy_async)."""
    monkeypatch.setenv("C8P_ARENA_MIN", "4096")
    b, _ = sweep_batch(snr_db=28.0, seed=5)
    pk, iq, chan = split(b, 1)
    ref = chain(FAKE, pk, iq, chan, 1, 4096, 8192, 256)
    for chunk, nout in ((1500, 3000), (20000, 60000)):
        msgs, dbg = _chain_threads(FAKE, pk, iq, chan, 1, max_batch, False, pinned_io=True, defer_consume=defer, poison=True, chunk=chunk, nout=nout)
        assert msgs == ref[2] and dbg == ref[3]


@pytest.mark.parametrize("handle", ["1", "0"])
def test_plain_chain_with_and_without_the_llr_handle(handle, monkeypatch):
    """batching_cfg::llr_handle (plain mode on pinned buffers): demod names the device copy of a batch's LLR arena in a "c8p_llr" tag and
    decode decodes the packets out of it, skipping their items in the stream — which are written all the same (the captured stream
    equals the oracle's LLRs either way); without the handle decode gathers the items out of the edge.  Same PDUs, same debug lines."""
    monkeypatch.setenv("C8P_LLR_HANDLE", handle)
    b, _ = sweep_batch(snr_db=28.0, seed=5)
    pk, iq, chan = split(b, 1)
    res = chain(FAKE, pk, iq, chan, 1, 4096, 8192, 5)
    assert check_against_oracle(res, pk, iq, chan) >= 20
    assert any(k == "c8p_llr" for _, k, _ in res[1]) == (handle == "1")
    ref = chain(FAKE, pk, iq, chan, 1, 4096, 8192, 256)
    msgs, dbg = _chain_threads(FAKE, pk, iq, chan, 1, 7, False, pinned_io=True, defer_consume=True, poison=True, chunk=3000, nout=5000)
    assert msgs == ref[2] and dbg == ref[3]


def test_small_pinned_buffers_are_staged_not_gathered(monkeypatch):
    """below C8P_GATHER_MIN_BUFFER (2 MB of input buffer) a front copies the call's window into pinned staging itself even though the
    buffer is pinned — a gather's ~18 us of latency per call is more than that copy; same results either way"""
    monkeypatch.delenv("C8P_GATHER_MIN_BUFFER", raising=False)
    b, _ = sweep_batch(snr_db=28.0, seed=5)
    pk, iq, chan = split(b, 1)
    ref = chain(FAKE, pk, iq, chan, 1, 4096, 8192, 256)
    for chunk, nout in ((3000, 5000), (1 << 19, 1 << 21)):
        dem = blocks.Block(blocks.DEMOD, engine_lib=FAKE, max_batch=7, fused=False, pinned_io=True)
        dec = blocks.Block(blocks.DECODE, ifdebug=True, engine_lib=FAKE, max_batch=7, fused=False, pinned_io=True)
        blocks.feed_demod(dem, pk, iq, chan)
        blocks.run_chain(dem, dec, chunk=chunk, nout=nout)
        gathered = dem.host_phase_us()["gather_enqueue"] > 0
        assert gathered == (chunk >= 1 << 19)
        assert dec.messages() == ref[2] and dec.debug_lines() == ref[3]
        dem.close(); dec.close()


def test_llr_handles_survive_the_demod_that_made_them():
    """decode holds the batch whose handle packets it is working on; the demod mirror that made it may be destroyed meanwhile: its
    registry entries go with it (later packets fall back to the items in the stream), the arena copy decode still holds is freed
    through the engine's cell without the context (c8p_dev_free(NULL, p)) when decode lets go of it."""
    b, _ = sweep_batch(snr_db=28.0, seed=5)
    pk, iq, chan = split(b, 1)
    ref = chain(FAKE, pk, iq, chan, 1, 4096, 8192, 256)
    dem = blocks.Block(blocks.DEMOD, engine_lib=FAKE, max_batch=6)
    blocks.feed_demod(dem, pk, iq, chan)
    dem.run(chunk=4096, nout=8192)
    llr, tags = dem.output(), dem.tags()
    assert any(k == "c8p_llr" for _, k, _ in tags)
    dec = blocks.Block(blocks.DECODE, ifdebug=True, engine_lib=FAKE, max_batch=4)
    half = (len(llr) // 2) & ~1
    dec.push_items(0, llr[:half])
    for off, key, val in tags:
        dec.push_tag(off, key, val)
    dec.run(chunk=4096, nout=8192, flush_at_end=False)
    dem.close()                                                # the demod goes while decode sits on one of its batches
    dec.push_items(0, llr[half:])
    dec.run(chunk=4096, nout=8192)
    assert dec.messages() == ref[2] and dec.debug_lines() == ref[3]
    dec.close()


def pinned_buffers_under_random_scheduler_chunks(engine_lib, seed, defer, relocate=0):
    rng = np.random.default_rng(40 + seed)
    n_ant = 1 + seed % 2
    b, _ = sweep_batch(snr_db=28.0, seed=5)
    pk, iq, chan = split(b, n_ant)
    dem = blocks.Block(blocks.DEMOD if n_ant == 1 else blocks.DEMOD2, engine_lib=engine_lib, max_batch=int(rng.integers(1, 9)), pinned_io=True, defer_consume=defer)
    dem.set_poison(True); dem.set_relocate(relocate)
    blocks.feed_demod(dem, pk, iq, chan)
    dem.set_jitter(5000 + seed)
    dem.run(chunk=int(rng.integers(100, 3000)), nout=int(rng.integers(50, 5000)))
    llr, tags = dem.output(), dem.tags()
    dec = blocks.Block(blocks.DECODE, ifdebug=True, engine_lib=engine_lib, max_batch=int(rng.integers(1, 9)), pinned_io=True, defer_consume=defer)
    dec.set_poison(True); dec.set_relocate(relocate)
    dec.push_items(0, llr)
    for off, key, val in tags:
        dec.push_tag(off, key, val)
    dec.set_jitter(6000 + seed)
    dec.run(chunk=int(rng.integers(200, 4000)), nout=int(rng.integers(1, 300)))
    res = (llr, tags, dec.messages(), dec.debug_lines())
    dem.close(); dec.close()
    assert check_against_oracle(res, pk, iq, chan) >= 10


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
@pytest.mark.parametrize("defer", [False, True])
def test_pinned_buffers_under_random_scheduler_chunks(seed, defer):
    """pinned buffers + a scheduler that hands out random item counts (sometimes fewer than the block has already looked at, sometimes
    none): consumed items are overwritten at once; with deferred consume also with the unread items moved before every third call"""
    pinned_buffers_under_random_scheduler_chunks(FAKE, seed, defer)
    if defer:
        pinned_buffers_under_random_scheduler_chunks(FAKE, seed, defer, relocate=3)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_pinned_buffers_under_random_scheduler_chunks_on_the_engine(seed, monkeypatch):
    """the same on the device: random item counts per call, consumed items overwritten at once, unread items moved before every third call,
    arenas that start at 4 KB; one and two antennas"""
    monkeypatch.setenv("C8P_ARENA_MIN", "4096")
    pinned_buffers_under_random_scheduler_chunks(None, seed, True, relocate=3)
    pinned_buffers_under_random_scheduler_chunks(None, seed, False)


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("pinned_io,defer", [(False, True), (True, False), (True, True)])
@pytest.mark.parametrize("chunk,nout", [(3000, 5000), (700, 40000), (1 << 16, 1 << 18)])
def test_chain_on_two_threads_with_and_without_pinned_buffers(fused, pinned_io, defer, chunk, nout):
    """demod -> decode on two threads through ONE edge buffer that decode reads while demod writes it; consumed items (and recycled
    segments of the edge) are overwritten at once"""
    b, _ = sweep_batch(snr_db=28.0, seed=5)
    pk, iq, chan = split(b, 1)
    ref = chain(FAKE, pk, iq, chan, 1, 4096, 8192, 256)
    msgs, dbg = _chain_threads(FAKE, pk, iq, chan, 1, 7, fused, pinned_io=pinned_io, defer_consume=defer, poison=True, chunk=chunk, nout=nout)
    assert msgs == ref[2] and len(msgs) >= 10
    assert dbg == ref[3]


@pytest.mark.parametrize("n_ant", [1, 2])
@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("max_batch", [3, 64])
def test_chain_on_two_threads_with_fake_backend(n_ant, fused, max_batch):
    """demod -> decode each on its own thread (the scheduler's thread-per-block), plain and fused (batching_cfg::fused: demod
    runs c8p_rx, the LLR items are not written, decode publishes what waits behind the "c8p_fused" handle): same PDUs in the
    same order, same debug lines as the two-pass run"""
    b, _ = sweep_batch(snr_db=28.0, seed=5)
    pk, iq, chan = split(b, n_ant)
    ref = chain(FAKE, pk, iq, chan, n_ant, 4096, 8192, 256)
    msgs, dbg = _chain_threads(FAKE, pk, iq, chan, n_ant, max_batch, fused)
    assert msgs == ref[2] and len(msgs) >= 10
    assert dbg == ref[3]


def test_fused_demod_emits_the_reference_tags_plus_a_handle():
    b, _ = sweep_batch(snr_db=28.0, seed=5)
    pk, iq, chan = split(b, 1)
    dem = blocks.Block(blocks.DEMOD, engine_lib=FAKE, max_batch=7, fused=True)
    blocks.feed_demod(dem, pk, iq, chan)
    dem.run(chunk=4096, nout=8192)
    llr, tags = dem.output(), dem.tags()
    dem.close()
    plain = chain(FAKE, pk, iq, chan, 1, 4096, 8192, 256)
    assert len(llr) == len(plain[0])                                   # same item counts (values unwritten)
    strip = [(o, k) for o, k, _ in tags if k != "c8p_fused"]
    assert strip == [(o, k) for o, k, _ in plain[1] if k != "c8p_llr"]       # (plain mode on pinned buffers adds its own handle tag)
    handles = [v for _, k, v in tags if k == "c8p_fused"]
    assert len(handles) == len({o for o, _, _ in tags}) and len(set(handles)) == len(handles)


def test_decode_steps_over_a_tag_set_that_announces_no_items():
    """a tag set with total <= 0 must not trap general_work (the loop re-read the same tags forever)"""
    dec = blocks.Block(blocks.DECODE, engine_lib=FAKE, max_batch=4)
    dec.push_items(0, np.zeros(100, np.float32))
    for k, v in (("format", 0), ("mcs", 0), ("len", 10), ("cr", 0), ("ampdu", 0), ("trellis", 102), ("total", 0)):
        dec.push_tag(10, k, v)
    dec.run(chunk=64, nout=64)
    assert dec.messages() == []
    dec.close()


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [False, True])
def test_chain_on_two_threads_on_gpu(fused):
    """the HIP engine behind both mirrors, each on its own thread; pipelined submits (max_batch 8 over ~25 packets), plain and
    fused: PDUs and debug lines equal the oracle-checked two-pass run"""
    b, _ = sweep_batch(snr_db=28.0, seed=6)
    pk, iq, chan = split(b, 1)
    ref = chain(None, pk, iq, chan, 1, 1000, 4096, 16)
    assert check_against_oracle(ref, pk, iq, chan) >= 10
    msgs, dbg = _chain_threads(None, pk, iq, chan, 1, 8, fused)
    assert msgs == ref[2]
    assert dbg == ref[3]


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [False, True])
def test_asynchronous_paths_of_the_fronts_on_the_engine(fused, monkeypatch):
    """What the fake-engine tests above model, on the device itself: arenas that start at 4 KB and grow while gathers fill them, consumed
    items overwritten at once (deferred consume), scatters written ahead, batches of 3 / 7 / 64 in flight, small and large windows — PDUs
    and debug lines equal the oracle-checked two-pass run; then each mirror alone with its input buffers RELOCATED before every second
    call (announced through c8p::buffer_will_move): the LLR stream, the tags and the PDUs are the same again."""
    monkeypatch.setenv("C8P_ARENA_MIN", "4096")
    b, _ = sweep_batch(snr_db=28.0, seed=6)
    pk, iq, chan = split(b, 1)
    ref = chain(None, pk, iq, chan, 1, 1000, 4096, 16)
    assert check_against_oracle(ref, pk, iq, chan) >= 10
    for chunk, nout, max_batch in ((1500, 3000, 3), (20000, 60000, 64), (4096, 8192, 7)):
        msgs, dbg = _chain_threads(None, pk, iq, chan, 1, max_batch, fused, pinned_io=True, defer_consume=True, poison=True, chunk=chunk, nout=nout)
        assert msgs == ref[2] and dbg == ref[3], (chunk, nout, max_batch)
    if fused:
        return
    dem = blocks.Block(blocks.DEMOD, max_batch=5, pinned_io=True, defer_consume=True)
    dem.set_poison(True); dem.set_relocate(2)
    blocks.feed_demod(dem, pk, iq, chan)
    dem.run(chunk=700, nout=3000)
    llr, tags = dem.output(), dem.tags()
    dec = blocks.Block(blocks.DECODE, ifdebug=True, max_batch=5, pinned_io=True, defer_consume=True)
    dec.set_poison(True); dec.set_relocate(2)
    dec.push_items(0, llr)
    for off, key, val in tags:
        dec.push_tag(off, key, val)
    dec.run(chunk=2500, nout=3000)
    res = (llr, tags, dec.messages(), dec.debug_lines())
    dem.close(); dec.close()
    assert check_against_oracle(res, pk, iq, chan) >= 10


# ---------------------------------------------------------------------------------------------- a live scheduler (no dry calls, no flush)
def test_a_burst_comes_out_without_further_input_and_without_flush():
    """The GNU Radio scheduler never calls a block that has no input, and demod's input is bursty (signal emits items only during
    packets).  A burst of fewer packets than max_batch must still come out — LLR items + tags from demod, PDUs from decode — within
    a few max_wait_us, although no further item ever arrives and nobody calls flush(): the fronts' latency timer submits and
    collects, demod asks the scheduler for a call through block_io::wake() and its forecast() then needs no input."""
    b, _ = sweep_batch(snr_db=None, seed=2)
    pk, iq, chan = split(b, 1)
    pk = pk[:5]
    ref = oracle_rx(pk, iq, chan)
    n_llr = sum(len(r["llr"]) for r in ref if r["tags"]["status"] == 0)
    want = [(len(x), x) for r in ref for x in r["pdus"]]
    assert n_llr > 0 and len(want) >= 4
    dem = blocks.Block(blocks.DEMOD, engine_lib=FAKE, max_batch=256)
    blocks.feed_demod(dem, pk, iq, chan)
    dem.run_live(chunk=1024, nout=4096, linger_ms=200)            # no dry call is ever made, nothing is flushed
    llr, tags = dem.output(), dem.tags()
    assert len(llr) == n_llr                                       # everything came out although max_batch was never reached
    dec = blocks.Block(blocks.DECODE, ifdebug=True, engine_lib=FAKE, max_batch=256)
    dec.push_items(0, llr)
    for off, key, val in tags:
        dec.push_tag(off, key, val)
    dec.run_live(chunk=1024, nout=4096, linger_ms=200)
    assert dec.messages() == want
    ph = dem.host_phase_us()
    assert set(ph) == {"cut_and_stage", "submit", "enqueue_downloads", "staging_alloc", "wait", "fetch", "replay", "drain", "gather_enqueue", "gather_wait"} and ph["cut_and_stage"] >= 0
    dem.close(); dec.close()


def test_live_scheduler_equals_the_flushed_run_on_a_long_stream():
    """the same stream through run_live (calls only with input, timer-driven tail) and through run (dry calls + flush): same items, tags, PDUs"""
    b, _ = sweep_batch(snr_db=25.0, seed=4)
    pk, iq, chan = split(b, 1)
    outs = []
    for live in (False, True):
        dem = blocks.Block(blocks.DEMOD, engine_lib=FAKE, max_batch=8)
        blocks.feed_demod(dem, pk, iq, chan)
        (dem.run_live(chunk=3000, nout=5000, linger_ms=100) if live else dem.run(chunk=3000, nout=5000))
        llr, tags = dem.output(), dem.tags()
        dec = blocks.Block(blocks.DECODE, ifdebug=True, engine_lib=FAKE, max_batch=8)
        dec.push_items(0, llr)
        for off, key, val in tags:
            dec.push_tag(off, key, val)
        (dec.run_live(chunk=3000, nout=5000, linger_ms=100) if live else dec.run(chunk=3000, nout=5000))
        # (the handle tags name a batch by a process-wide counter: not part of the comparison)
        outs.append((llr, [(o, k, v if not isinstance(v, np.ndarray) else v.tobytes()) for o, k, v in tags if k != "c8p_llr"], dec.messages()))
        dem.close(); dec.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and outs[0][1] == outs[1][1] and outs[0][2] == outs[1][2]


@pytest.mark.parametrize("fused", [False, True])
def test_chain_with_a_single_engine_slot(fused, monkeypatch):
    """C8P_SLOTS=1 (documented; tools/profile_gpu.sh uses it): the engine then keeps ONE submit's result arenas, so a front must collect
    batch i before it submits batch i+1 instead of after — same PDUs as with the default three slots"""
    b, _ = sweep_batch(snr_db=None, seed=6)
    pk, iq, chan = split(b, 1)
    res = []
    for slots in ("1", "3"):
        monkeypatch.setenv("C8P_SLOTS", slots)
        dem = blocks.Block(blocks.DEMOD, engine_lib=FAKE, max_batch=4, fused=fused)
        dec = blocks.Block(blocks.DECODE, ifdebug=True, engine_lib=FAKE, max_batch=4, fused=fused)
        blocks.feed_demod(dem, pk, iq, chan)
        blocks.run_chain(dem, dec, chunk=3000, nout=8192)
        res.append(dec.messages())
        dem.close(); dec.close()
    assert res[0] == res[1] and len(res[0]) >= 20


@pytest.mark.gpu
def test_chain_with_a_single_engine_slot_on_gpu(monkeypatch):
    monkeypatch.setenv("C8P_SLOTS", "1")
    b, _ = sweep_batch(snr_db=None, seed=6)
    pk, iq, chan = split(b, 1)
    want = [(len(x), x) for r in oracle_rx(pk, iq, chan) for x in r["pdus"]]
    for fused in (False, True):
        dem = blocks.Block(blocks.DEMOD, max_batch=4, fused=fused)
        dec = blocks.Block(blocks.DECODE, ifdebug=True, max_batch=4, fused=fused)
        blocks.feed_demod(dem, pk, iq, chan)
        blocks.run_chain(dem, dec, chunk=3000, nout=8192)
        assert dec.messages() == want, fused
        dem.close(); dec.close()


@pytest.mark.parametrize("kind", ["demod", "demod_fused", "decode"])
def test_a_second_burst_after_an_idle_gap_comes_out_too(kind):
    """two bursts separated by far more than max_wait_us, no dry call, no flush: the latency timer drains the first burst, goes to
    sleep without a deadline, and must be woken by the second burst's first general_work call (round 3 remembered "something was
    pending" on the scheduler's side only: the partial batch of the second burst then stayed in the front until a third one came)"""
    import time
    b, _ = sweep_batch(snr_db=None, seed=2)
    pk, iq, chan = split(b, 1)
    bursts = [pk[:3], pk[3:7]]
    ref = [oracle_rx(q, iq, chan) for q in bursts]
    if kind == "decode":
        dec = blocks.Block(blocks.DECODE, ifdebug=True, engine_lib=FAKE, max_batch=256)
        pos, want = 0, []
        for r in ref:
            for q in r:
                if q["tags"]["status"] != 0:
                    continue
                t = q["tags"]
                for key in abi.TAG_KEYS:
                    dec.push_tag(pos, key, int(t[key]))
                for key in ("cfo", "snr", "rssi"):
                    dec.push_tag(pos, key, float(t[key]))
                dec.push_tag(pos, "seq", int(t["seq"]))
                dec.push_items(0, q["llr"])
                pos += len(q["llr"])
                want += [(len(x), x) for x in q["pdus"]]
            dec.run_live(chunk=2048, nout=4096, linger_ms=60)
            assert dec.messages() == want                    # this burst's PDUs are out before the next burst arrives
            time.sleep(0.05)
        assert len(want) >= 5
        dec.close()
        return
    dem = blocks.Block(blocks.DEMOD, engine_lib=FAKE, max_batch=256, fused=(kind == "demod_fused"))
    pushed = n_llr = n_tagged = 0
    for q, r in zip(bursts, ref):
        pushed += blocks.feed_demod(dem, q, iq, chan, base=pushed)
        dem.run_live(chunk=1024, nout=4096, linger_ms=60)
        n_llr += sum(len(x["llr"]) for x in r if x["tags"]["status"] == 0)
        n_tagged += sum(1 for x in r if x["tags"]["status"] == 0)
        assert len(dem.output()) == n_llr                       # all of this burst's items are out although nothing else arrived
        assert sum(1 for _, k, _ in dem.tags() if k == "trellis") == n_tagged
        time.sleep(0.05)
    assert n_tagged >= 5
    dem.close()


def test_a_burst_comes_out_in_fused_mode_without_further_input():
    """the latency timer in fused mode (demod runs c8p_rx, decode publishes behind the handle tag): a 3-packet burst, no dry call, no flush"""
    b, _ = sweep_batch(snr_db=None, seed=2)
    pk, iq, chan = split(b, 1)
    pk = pk[:3]
    want = [(len(x), x) for r in oracle_rx(pk, iq, chan) for x in r["pdus"]]
    dem = blocks.Block(blocks.DEMOD, engine_lib=FAKE, max_batch=256, fused=True)
    blocks.feed_demod(dem, pk, iq, chan)
    dem.run_live(chunk=2048, nout=4096, linger_ms=200)
    llr, tags = dem.output(), dem.tags()
    assert len(llr) > 0                                    # the items exist; their values are not written (c8p_blocks.h, batching_cfg::fused)
    assert sum(1 for _, k, _ in tags if k == "c8p_fused") == sum(1 for r in oracle_rx(pk, iq, chan) if r["tags"]["status"] == 0)
    dec = blocks.Block(blocks.DECODE, ifdebug=True, engine_lib=FAKE, max_batch=256, fused=True)
    dec.push_items(0, llr)
    for off, key, val in tags:
        dec.push_tag(off, key, val)
    dec.run_live(chunk=2048, nout=4096, linger_ms=100)
    assert dec.messages() == want
    dem.close(); dec.close()


# ------------------------------------------------------------------------------------------------ rx_chain
def rx_chain_streams(seed, K):
    """K equally long raw streams (the block's ports advance together): different packet sequences, idle air behind the shorter ones"""
    from test_detect_frontend import CFOS, NAMES, SWEEP, stream
    order = [NAMES, NAMES[::-1], NAMES[2:7], NAMES[1:4]][:K]
    xs = [stream(nm, CFOS[:len(nm)], snr=33.0, seed=seed + i)[0] for i, nm in enumerate(order)]
    n = max(len(x) for x in xs) + 900
    rng = np.random.default_rng(seed)
    xs = [np.concatenate([x, 0.004 * (rng.normal(size=n - len(x)) + 1j * rng.normal(size=n - len(x)))]).astype(np.complex64) for x in xs]
    return xs, [[SWEEP[k]["pdu"].tobytes() for k in nm] for nm in order]


def run_rx_chain(engine_lib, seed, K=3, per_round=7000, jitter=0, bits=0):
    """bits = 16 / 8: the ports carry int16 / int8 I/Q pairs (what a UHD source with cpu_format sc16 / sc8 emits); 33 dB packets survive
    either quantisation, so the PDUs are the fixtures' all the same"""
    xs, want = rx_chain_streams(seed, K)
    if bits:
        top = (1 << (bits - 1)) - 1
        scale = np.float32(1.02 * max(float(np.abs(x.view(np.float32)).max()) for x in xs) / top)
        xs = [np.clip(np.rint(x.view(np.float32) / scale), -top - 1, top).astype(np.int16 if bits == 16 else np.int8).reshape(-1, 2) for x in xs]
        b = blocks.Block(blocks.RX_CHAIN(K), engine_lib=engine_lib, max_batch=per_round, item_format=abi.FMT_SC16 if bits == 16 else abi.FMT_SC8, scale=float(scale))
    else:
        b = blocks.Block(blocks.RX_CHAIN(K), engine_lib=engine_lib, max_batch=per_round)
    for s, x in enumerate(xs):
        b.push_items(s, x)
    if jitter:
        b.set_jitter(jitter)
    b.run(chunk=2500, nout=4096)
    msgs, who = b.messages(), b.message_streams()
    b.close()
    got = {s: [] for s in range(K)}
    for (n, blob), s in zip(msgs, who):
        assert n == len(blob)
        got[s].append(blob)
    for s in range(K):
        assert got[s] == want[s], s
    return len(msgs)


def test_rx_chain_block_on_the_test_engine():
    """rx_chain (rx.grc's receive chain for K radios as ONE block, on host/c8p_rawchain's live ingest) through the scheduler harness, with
    tests/fake_c8p as the engine: per input port, in order, the PDUs decode would publish for that radio (blob = format, len, MPDU, mcs)"""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "fake_c8p")])
    fake = os.path.join(ROOT, "tests", "fake_c8p", "libc8p_fake.so")
    assert run_rx_chain(fake, 400) >= 15
    assert run_rx_chain(fake, 410, K=2, per_round=3000, jitter=7) >= 10         # ragged general_work calls, rounds that end inside packets
    assert run_rx_chain(fake, 400, K=2, bits=16) >= 10                          # std::complex<short> ports
    assert run_rx_chain(fake, 400, K=2, per_round=5000, bits=8) >= 10           # std::complex<char> ports


def test_rx_chain_block_publishes_the_rounds_it_has_when_the_source_pauses():
    """a LIVE scheduler (no flush; general_work only while there is input or the block asks for none): after the input is used up the
    PDUs of every round that was closed are published all the same — forecast() asks for no input while PDUs wait and the rawchain's
    callback wakes the scheduler; only the open round's packets stay behind"""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "fake_c8p")])
    fake = os.path.join(ROOT, "tests", "fake_c8p", "libc8p_fake.so")
    xs, want = rx_chain_streams(440, 2)
    b = blocks.Block(blocks.RX_CHAIN(2), engine_lib=fake, max_batch=3000, flush_when_idle=False)
    for s, x in enumerate(xs):
        b.push_items(s, x)
    b.run_live(chunk=2500, nout=4096, linger_ms=300)
    msgs, who = b.messages(), b.message_streams()
    b.close()
    got = {0: [], 1: []}
    for (n, blob), s in zip(msgs, who):
        got[s].append(blob)
    for s in range(2):
        assert got[s] == want[s][:len(got[s])] and len(got[s]) >= len(want[s]) - 2, (s, len(got[s]), len(want[s]))
    assert len(msgs) >= 8


@pytest.mark.gpu
def test_rx_chain_block_on_the_engine():
    from gr_ieee80211_amd.blocks import ENGINE_LIB
    assert run_rx_chain(ENGINE_LIB, 420, K=4, per_round=9000) >= 20
    assert run_rx_chain(ENGINE_LIB, 430, K=3, per_round=4000, jitter=11) >= 15
    assert run_rx_chain(ENGINE_LIB, 420, K=3, bits=16) >= 15
    assert run_rx_chain(ENGINE_LIB, 420, K=3, per_round=5000, jitter=5, bits=8) >= 15


def ref_tags_per_packet(pk, iq, chan):
    """per packet of the batch: does demod emit it (status OK)?  From a plain demod run on the test engine."""
    dem = blocks.Block(blocks.DEMOD, engine_lib=FAKE, max_batch=64, fused=False)
    blocks.feed_demod(dem, pk, iq, chan)
    dem.run(chunk=4096, nout=8192)
    offs = sorted({o for o, k, _ in dem.tags()})
    dem.close()
    return [True] * len(offs)


def test_fused_is_the_default_when_demod_feeds_only_decode(monkeypatch):
    """batching_cfg::fused_auto: with nothing said (no fused argument, no C8P_FUSED in the environment) a demod front whose LLR
    port is read by the decode mirror only — what rx.grc wires, and what the chain harness tells it through
    block_io::output_feeds_only_decode — runs fused; run by itself (its reader unknown) it stays plain and writes its items; C8P_FUSED=0
    keeps it plain in the chain too; the PDUs and debug lines are the same in all of them."""
    b, _ = sweep_batch(snr_db=28.0, seed=5)
    pk, iq, chan = split(b, 1)
    want = None
    monkeypatch.delenv("C8P_FUSED_TAGS", raising=False)
    for env, expect, lean in ((None, True, True), ("0", False, False), ("auto", True, True), ("1", True, False), ("auto+full", True, False)):
        monkeypatch.delenv("C8P_FUSED_TAGS", raising=False)
        if env is None:
            monkeypatch.delenv("C8P_FUSED", raising=False)
        elif env == "auto+full":
            monkeypatch.setenv("C8P_FUSED", "auto"); monkeypatch.setenv("C8P_FUSED_TAGS", "full")
        else:
            monkeypatch.setenv("C8P_FUSED", env)
        dem = blocks.Block(blocks.DEMOD, engine_lib=FAKE, max_batch=7)
        dec = blocks.Block(blocks.DECODE, ifdebug=True, engine_lib=FAKE, max_batch=7)
        assert dem.fused_state() is None                       # nothing decided before the first call
        blocks.feed_demod(dem, pk, iq, chan)
        blocks.run_chain(dem, dec, chunk=3000, nout=5000)
        assert dem.fused_state() is expect, (env, dem.fused_state())
        got = (dec.messages(), dec.debug_lines())
        n_tags = dem.tags_emitted()
        dem.close(); dec.close()
        want = want or got
        assert got == want, env                                # PDUs AND the debug lines (cfo, snr, rssi, sssnr, per-MCS counts: the tag record)
        assert len(got[0]) >= 10
        # lean tags (batching_cfg::fused_lean_tags): fused by fused_auto = ONE tag per emitted packet, the handle; every other mode writes
        # the reference's eleven to fourteen (+ a handle)
        n_emitted = sum(1 for r in ref_tags_per_packet(pk, iq, chan) if r)
        assert (n_tags == n_emitted) if lean else (n_tags >= 11 * n_emitted), (env, n_tags, n_emitted)
    monkeypatch.delenv("C8P_FUSED", raising=False)
    dem = blocks.Block(blocks.DEMOD, engine_lib=FAKE, max_batch=7)      # by itself: who reads the port is unknown -> plain, items written
    blocks.feed_demod(dem, pk, iq, chan)
    dem.run(chunk=4096, nout=8192)
    assert dem.fused_state() is False
    assert not any(k == "c8p_fused" for _, k, _ in dem.tags())
    dem.close()


def test_safe_mode_uses_nothing_but_the_reference_blocks_own_scheduler_calls(monkeypatch):
    """C8P_GR_SAFE=1 (batching_cfg::safe_mode; the GNU Radio adapters' fallback for an installation they have never met, and what
    -DC8P_GR_STOCK_API_ONLY compiles in): ordinary buffers on every port, the demod -> decode edge in plain mode whoever reads it — every
    item value written, the reference's eleven to fourteen tags per packet, no handle tags — items consumed in the call that cut them, no
    timer thread.  Other switches do not override it.  PDUs, debug lines, LLR items and tags as the oracle's."""
    monkeypatch.setenv("C8P_GR_SAFE", "1")
    monkeypatch.setenv("C8P_FUSED", "1"); monkeypatch.setenv("C8P_PINNED_IO", "1")           # (safe mode wins)
    b, _ = sweep_batch(snr_db=28.0, seed=5)
    pk, iq, chan = split(b, 1)
    # the chain on two threads, nothing said about fused / pinned: the harness tells demod that decode is its only reader (fused_auto would fuse)
    dem = blocks.Block(blocks.DEMOD, engine_lib=FAKE, max_batch=7)
    dec = blocks.Block(blocks.DECODE, ifdebug=True, engine_lib=FAKE, max_batch=7)
    assert not dem.input_pinned(0) and not dem.output_pinned() and not dec.input_pinned(0)
    blocks.feed_demod(dem, pk, iq, chan)
    blocks.run_chain(dem, dec, chunk=3000, nout=5000)
    assert dem.fused_state() is False
    chain_out = (dec.messages(), dec.debug_lines())
    n_tags = dem.tags_emitted()
    dem.close(); dec.close()
    n_emitted = sum(1 for r in ref_tags_per_packet(pk, iq, chan) if r)
    assert n_tags >= 10 * n_emitted                   # (a legacy packet has ten: format mcs len cr ampdu trellis total cfo snr rssi)
    # the same blocks one after the other: the items and tags of the edge are the reference's, nothing else on it
    res = chain(FAKE, pk, iq, chan, 1, 3000, 5000, 7)
    assert check_against_oracle(res, pk, iq, chan) >= 20
    assert not any(k.startswith("c8p_") for _, k, _ in res[1])
    assert (res[2], res[3]) == chain_out


def test_a_buffer_move_never_waits_for_a_busy_front_without_a_bound(monkeypatch):
    """c8p::buffer_move_scope takes hold of a front with a bounded try-lock: a scheduler may call a buffer's blocked callback while holding a
    lock the front's RUNNING general_work needs (add_item_tag takes the buffer's mutex in GNU Radio) — waiting for the front there would
    deadlock.  Here the test engine's submits sleep 30 ms under the front's lock while another thread asks to move a buffer: with a 2 ms
    bound the scope gives up (ok() == false: the callback returns false, the scheduler comes back later), with a generous one it gets
    through between two calls; the run itself is undisturbed."""
    import threading
    import time
    monkeypatch.setenv("C8P_FAKE_SUBMIT_SLEEP_US", "30000")
    b, _ = sweep_batch(snr_db=28.0, seed=5)
    pk, iq, chan = split(b, 1)
    dem = blocks.Block(blocks.DEMOD, engine_lib=FAKE, max_batch=3, fused=False)
    blocks.feed_demod(dem, pk, iq, chan)
    L = blocks.lib()
    seen = {"short": [], "long": []}
    stop = threading.Event()

    def mover():
        while not stop.is_set():
            t = time.perf_counter()
            seen["short"].append((L.c8ph_try_move(2000), time.perf_counter() - t))
            time.sleep(0.002)
        seen["long"].append(L.c8ph_try_move(2000000))
    th = threading.Thread(target=mover)
    th.start()
    try:
        dem.run(chunk=700, nout=3000)                   # ~12 submits of 30 ms each: the front's lock is held most of the time
    finally:
        stop.set(); th.join()
    llr, tags = dem.output(), dem.tags()
    dem.close()
    oks = [ok for ok, _ in seen["short"]]
    assert 0 in oks, "the front was never busy when asked"                      # it gave up at least once ...
    assert max(dt for _, dt in seen["short"]) < 0.2                              # ... and never waited much beyond its bound
    assert seen["long"] == [1]                                                   # with time it gets through
    ref = oracle_rx(pk, iq, chan)
    assert len(llr) == sum(len(r["llr"]) for r in ref if r["tags"]["status"] == 0)
