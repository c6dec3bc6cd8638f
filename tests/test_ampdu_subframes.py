"""VHT A-MPDUs with several subframes through packetAssemble (decode_impl.cc:325-424).

The reference walks the delimiters with ONE length variable that is never reset: every delimiter's length bits are OR-ed
into it (:346-351), and after a subframe with a correct FCS the same variable is advanced by 4 for the PDU length (:414)
before the next delimiter is OR-ed in.  So a second subframe is only found where its length survives that OR; otherwise the
walk continues with a wrong length (FCS failures, then overrun).  Delimiter CRC and signature are not checked.  The walk is
restated literally in Python here, independently of the C oracle, and oracle and engine must both agree with it."""
import zlib

import numpy as np
import pytest

from c8p_cases import ampdu_vht_multi
from c8p_testlib import Batch, abi, oracle, oracle_rx, through_channel


def mpdu(rng, n):
    body = bytes(rng.integers(0, 256, size=n - 4, dtype=np.uint8))
    return body + zlib.crc32(body).to_bytes(4, "little")


def literal_walk(bits, T, mcs):
    """decode_impl.cc:331-424 line by line on the descrambled bits; returns (PDUs, FCS failures)"""
    pdus, fails = [], 0
    procd = 16
    if procd >= T:
        return pdus, fails
    bp, tmp_len = 16, 0
    while True:
        procd += 32
        if procd > T:
            break
        eof = int(bits[bp])
        tmp_len |= int(bits[bp + 2]) << 12
        tmp_len |= int(bits[bp + 3]) << 13
        for i in range(12):
            tmp_len |= int(bits[bp + 4 + i]) << i
        padded = (tmp_len // 4 + (tmp_len % 4 != 0)) * 32
        procd += padded
        if procd > T:
            break
        bp += 32
        body = bytes(int(sum(int(bits[bp + 8 * i + j]) << j for j in range(8))) for i in range(tmp_len))
        bp += padded
        if zlib.crc32(body) != 0x2144DF1C:
            fails += 1
        else:
            pdus.append(bytes([2, tmp_len % 256, tmp_len // 256]) + body + bytes([mcs]))
            tmp_len += 4
        if eof:
            break
    return pdus, fails


CASES = [   # (mcs, MPDU lengths)
    (3, [60, 100]),             # 64 | 100 == 100: the second subframe is found
    (3, [100, 60]),             # 104 | 60 = 124: the second one is parsed with the wrong length
    (5, [28, 60, 124]),         # 32 | 60 = 60, 64 | 124 = 124: all three found
    (7, [200, 200, 200]),       # 204 | 200 = 204: wrong length from the second on
    (0, [24]),                  # single subframe, EOF set: stops after it
    (6, [508, 1020, 252]),      # 512 | 1020 = 1020, 1024 | 252 = 1276: third lost
    (4, [36, 36]),              # 40 | 36 = 44
    (2, [12, 28, 60, 124, 252]),
]


def make(case, rng):
    mcs, lens = case
    ms = [mpdu(rng, n) for n in lens]
    return mcs, ms, ampdu_vht_multi(ms)


@pytest.mark.parametrize("ci", range(len(CASES)))
def test_oracle_walks_subframes_like_the_literal_restatement(ci):
    rng = np.random.default_rng(600 + ci)
    mcs, ms, a = make(CASES[ci], rng)
    iq, _ = oracle().tx_packet(abi.F_VHT, mcs, a)
    b = Batch()
    assert b.add_rx(through_channel(iq), n_ant=1) == 0
    r = oracle_rx(*b.arrays())[0]
    want, fails = literal_walk(r["bits"], int(r["tags"]["trellis"]), mcs)
    assert r["pdus"] == want and int(r["out"]["n_crc_fail"]) == fails
    first = bytes([2, len(ms[0]) % 256, len(ms[0]) // 256]) + ms[0] + bytes([mcs])
    assert want[0] == first
    if CASES[ci][1] == [60, 100]:
        assert len(want) == 2 and want[1][3:-1] == ms[1]
    if CASES[ci][1] == [100, 60]:
        assert len(want) == 1            # (124 bytes from there overrun the trellis: the walk stops)


@pytest.mark.gpu
def test_gpu_subframe_walk_equals_oracle():
    from gr_ieee80211_amd import engine
    rng = np.random.default_rng(77)
    b = Batch()
    metas = []
    for rep in range(4):
        for ci in range(len(CASES)):
            mcs, ms, a = make(CASES[ci], rng)
            iq, _ = oracle().tx_packet(abi.F_VHT, mcs, a)
            snr = None if rep == 0 else (30.0 if rep < 3 else 14.0)       # the last round: FCS failures inside the walk
            assert b.add_rx(through_channel(iq, snr_db=snr, rng=rng), n_ant=1) == len(metas)
            metas.append(mcs)
    pk, iq, chan = b.arrays()
    ref = oracle_rx(pk, iq, chan)
    eng = engine.Engine(device=0, keep_bits=True)
    eng.rx(pk, iq, chan)
    got = eng.results(want_llr=False, want_bits=True)
    eng.close()
    multi = 0
    for g, r, mcs in zip(got, ref, metas):
        assert g["pdus"] == r["pdus"]
        assert int(g["out"]["n_crc_fail"]) == int(r["out"]["n_crc_fail"])
        assert np.array_equal(g["bits"], r["bits"])
        want, fails = literal_walk(r["bits"], int(r["tags"]["trellis"]), mcs)
        assert g["pdus"] == want and int(g["out"]["n_crc_fail"]) == fails
        multi += len(want) > 1
    assert multi >= 6
