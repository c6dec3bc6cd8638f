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
