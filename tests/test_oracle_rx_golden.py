"""The oracle's RECEIVE side against the reference toolbox's (tests/golden/rx_side.npz, made by gen_golden.py --rx-side from
/root/reference/tools/phy80211.py:842-1016 and phy80211header.py:1055-1175 — imported and run, only data is committed).

The toolbox's receiver is Python in double precision and is NOT the C++ blocks' algorithm everywhere; what each comparison can and
cannot say is in its docstring and in NOTES/parity.md ("Python RX against the oracle")."""
import os

import numpy as np
import pytest

from c8p_testlib import ROOT, abi
from oracle import oracle_py as o

G = np.load(os.path.join(ROOT, "tests", "golden", "rx_side.npz"))
CASES = sorted({k.split("/")[0] for k in G.files if "/" in k})


def case(name):
    return {k.split("/")[1]: G[k] for k in G.files if k.startswith(name + "/")}


def test_fixture_is_complete():
    assert len(CASES) == int(G["n_cases"]) == 24


@pytest.mark.parametrize("name", CASES)
def test_detection_and_cfo_against_the_toolbox(name):
    """c8p_oracle_detect.c (presiso + trigger_impl + sync_impl restated) against procSisoRx's STF trigger / LTF sync / CFO on the same
    items.  Two estimators of the same quantities, not one algorithm: the toolbox thresholds the STF autocorrelation at 0.4 over 16-item
    windows and puts the LTF start in the middle of the 80 % shoulders of a 64-item autocorrelation; the C++ (and the oracle) use 0.3
    over presiso's moving averages and the maximum of the LTF autocorrelation.  So: exactly one packet found; both LTF positions inside
    the long training field's 32-item guard interval (any FFT window that starts there is valid); the CFO estimates within 2e-4 rad /
    item of each other (640 Hz: the toolbox's own estimates scatter by that much around the truth at 35 dB) and of the truth."""
    c = case(name)
    x, lead = c["x"], int(c["lead"])
    ev, _, _, _ = o.detect(x)
    assert len(ev) == 1
    ltf_start = lead + 160                                  # L-STF is 160 items
    assert ltf_start <= int(ev[0]["index"]) <= ltf_start + 32
    assert ltf_start <= int(c["ltf_index"]) <= ltf_start + 32
    rad_py = float(c["rx_cfo_hz"]) * 2 * np.pi / 20e6
    rad_true = -float(c["cfo_hz"]) * 2 * np.pi / 20e6       # both report the rotation that UNDOES the offset
    assert abs(float(ev[0]["rad"]) - rad_py) <= 2e-4
    assert abs(float(ev[0]["rad"]) - rad_true) <= 2e-4 and abs(rad_py - rad_true) <= 2e-4


@pytest.mark.parametrize("name", CASES)
def test_signal_front_end_against_the_toolbox(name):
    """c8po_signal_event (signal_impl.cc:108-162 restated: de-rotation, two LTF FFTs, H = (F1 + F2) / (2 L), L-SIG demodulation,
    Viterbi(24), signalCheckLegacy) handed the toolbox's LTF position and CFO, so that both transform the SAME three 64-item windows
    (the block reads them at +8, +72, +152 from its sync item; the toolbox at +0, +64, +144 from its LTF index): the legacy channel
    estimate equals the toolbox's rxChanL on all 52 tones (float32 against double: 2e-5 relative to the largest tone; the block
    counts the de-rotation from its sync item, 8 items earlier: a known constant phase), the 24 decoded L-SIG bits are identical, and
    so are rate / length.  The L-SIG LLRs are NOT the same numbers by design: the block multiplies the equalised tones by the
    conjugate of its four pilots' mean phase (cloud80211phy.cc:609-627), the toolbox does not track the pilots; they agree in sign on
    every coded bit and within 0.3 in value: the toolbox's transmitter halves the samples at every symbol edge (phy80211header.py:
    894-899), which leaves ~20 dB of self-interference on every tone whatever the AWGN, so a phase taken from four pilots moves by
    up to ~0.2 rad."""
    c = case(name)
    x = c["x"]
    ltf = int(c["ltf_index"])
    rad = np.float32(float(c["rx_cfo_hz"]) * 2 * np.pi / 20e6)
    ev = np.zeros(1, abi.SIG_IN_DT)
    ev["raw_off"][0, 0] = ltf - 8
    ev["n_avail"] = len(x) - (ltf - 8)
    ev["cfo_rad"], ev["snr"], ev["rssi"], ev["n_ant"] = rad, 35.0, 1.0, 1
    pk, iq, chan = o.signal_events(ev, x)
    assert pk[0]["n_items"] > 0, "the oracle's L-SIG check failed on a packet the toolbox accepts"
    h = chan[:64].astype(np.complex128) * np.exp(-1j * 8.0 * float(rad))      # back to the toolbox's phase origin
    py = c["chan_l"]                                          # fft-shifted: index j is bin j - 32
    tones = [k for k in range(-26, 27) if k != 0]
    d = np.array([h[k % 64] - py[k + 32] for k in tones])
    assert np.max(np.abs(d)) <= 2e-5 * np.max(np.abs(py))
    # L-SIG: bits, rate, length
    ok, h2, mcs, ln, nsamp, llr, bits = False, None, 0, 0, 0, None, None
    de = (x[ltf - 8:ltf - 8 + 224].astype(np.complex64) * np.exp(1j * np.arange(224, dtype=np.float32) * rad).astype(np.complex64)).astype(np.complex64)
    ok, h2, mcs, ln, nsamp, llr, bits = o.signal(de)
    assert ok and bool(c["lsig_check"])
    assert np.array_equal(bits, c["lsig_bits"])
    assert pk[0]["lsig_len"] == ln and pk[0]["lsig_mcs"] == mcs
    exp_len = int(c["mpdu_len"]) if int(c["fmt"]) == 0 else ln
    assert ln == exp_len
    coded = c["lsig_llr_coded"]
    assert np.all(np.sign(llr) == np.sign(coded))
    assert np.max(np.abs(llr - coded)) <= 0.3


@pytest.mark.parametrize("name", [n for n in CASES if "_L" not in n])
def test_ht_sig_and_vht_sig_a_bits_against_the_toolbox(name):
    """The oracle's demod (demod_impl.cc DEMOD_S_SYNC ... restated) classifies the packet and decodes HT-SIG / VHT-SIG-A from the same
    two symbols the toolbox's __procRxNonLegacySigDemod takes: the 48 decoded bits of the accepted format are identical, the check
    agrees, and the parsed MCS / length are the transmitted ones.  At CFO 0 only: procSisoRx restarts its de-rotation phase at the
    first item behind the L-SIG (phy80211.py:949 `sigSamples = [... cos(tmpCfoRadStep*i) ...]`, i from 0) while its channel estimate
    carries the phase of 208 items earlier, so with +-50 kHz its own HT-SIG / VHT-SIG-A checks fail (3.3 rad off) — the fixture records
    that — while the oracle, like the C++ block, keeps one phase origin per packet and decodes all of them."""
    c = case(name)
    x = c["x"]
    ltf = int(c["ltf_index"])
    rad = np.float32(float(c["rx_cfo_hz"]) * 2 * np.pi / 20e6)
    ev = np.zeros(1, abi.SIG_IN_DT)
    ev["raw_off"][0, 0] = ltf - 8
    ev["n_avail"] = len(x) - (ltf - 8)
    ev["cfo_rad"], ev["snr"], ev["rssi"], ev["n_ant"] = rad, 35.0, 1.0, 1
    pk, iq, chan = o.signal_events(ev, x)
    tags, llr, _ = o.demod(np.ascontiguousarray(pk), iq, chan, 0)
    fmt = int(c["fmt"])
    assert int(tags["status"]) == 0 and int(tags["format"]) == fmt and int(tags["mcs"]) == int(c["mcs"])
    bits = c["ht_bits"] if fmt == 1 else c["vht_a_bits"]
    if float(c["cfo_hz"]) != 0.0:
        assert not bool(c["ht_check"] if fmt == 1 else c["vht_a_check"])      # (the toolbox's phase restart, see above)
        return
    assert bool(c["ht_check"] if fmt == 1 else c["vht_a_check"])
    ok, f = (o.sig_ht if fmt == 1 else o.sig_vht_a)(bits)
    assert ok and f["mcs"] == int(c["mcs"]) and f["nSS"] == 1
    if fmt == 1:
        assert f["len"] == int(c["mpdu_len"]) == int(tags["len"])


def test_sig_checks_on_fuzzed_fields_against_the_toolbox():
    """procCheckSigLegacy / procCheckSigHt / procCheckSigVhtA (phy80211header.py:1063-1124) against the oracle's restatement of the
    C++ checks (cloud80211phy.cc:650-771) on 400 random words + every single-bit flip of a valid word per field.  The two code bases
    do not check the same things, by design: the toolbox also wants the L-SIG tail bits zero, the C++ also wants the HT-SIG / VHT-SIG-A
    reserved-as-zero and bandwidth bits right.  So: wherever the extra conditions hold, the verdicts are EQUAL; the stricter side
    never accepts what the other rejects for the shared conditions."""
    lw, lc = G["fuzz_l_bits"], G["fuzz_l_check"]
    n_ok = 0
    for w, py in zip(lw, lc):
        ok, f = o.sig_legacy(w)
        if w[18:24].sum() == 0:
            assert ok == bool(py)
        else:
            assert not py                                   # the toolbox rejects a non-zero tail; the C++ does not look at it
        n_ok += ok
    assert n_ok >= 100
    for bits, checks, fn, extra in ((G["fuzz_ht_bits"], G["fuzz_ht_check"], o.sig_ht, lambda b: b[5] + b[6] + b[7] + b[28] + b[29] + b[30] + b[32] + b[33] == 0),
                                    (G["fuzz_vht_bits"], G["fuzz_vht_check"], o.sig_vht_a, lambda b: b[0] + b[1] == 0)):
        n_ok = n_both = 0
        for w, py in zip(bits, checks):
            ok, _ = fn(w)
            if extra(w.astype(int)):
                assert ok == bool(py)
                n_both += 1
            else:
                assert not ok                               # the C++ rejects these whatever the toolbox says
            n_ok += ok
        assert n_ok >= 1 and n_both >= 10


def test_sig_parsers_and_rate_tables_against_the_toolbox():
    """procParserSigLegacy knows one rate (every branch tests `tmpRate == 11`, phy80211header.py:1088-1113): compared where it is
    right — 6 Mb/s words — on length and symbol count.  procParserSigHt + the `modulation` class against parse_ht for every fuzzed
    word whose MCS the toolbox supports at 20 MHz (0-15), and the rate tables (nSS, nBPSCS, nCBPS, nDBPS, nLTF) of every legacy / HT /
    VHT rate of one and two streams against the oracle's parsers fed a synthetic field."""
    for w, (mcs, ln, nsym) in zip(G["fuzz_l_bits"], G["fuzz_l_parse"]):
        ok, f = o.sig_legacy(w)
        rate = int(w[0]) | int(w[1]) << 1 | int(w[2]) << 2 | int(w[3]) << 3
        if ok and rate == 11:
            assert (f["mcs"], f["len"], f["nSym"]) == (int(mcs), int(ln), int(nsym))
    n = 0
    for w, (mcs, ln, nss, nb, ncbps, ndbps, init) in zip(G["fuzz_ht_bits"], G["fuzz_ht_parse"]):
        if not init or w[7] or w[31]:                       # 40 MHz / short GI: other tables
            continue
        _, f = o.sig_ht(w)
        assert (f["mcs"], f["len"], f["nSS"], f["nBPSCS"], f["nCBPS"], f["nDBPS"]) == (int(mcs), int(ln), int(nss), int(nb), int(ncbps), int(ndbps))
        n += 1
    assert n >= 20
    for fcode, mcs, nsts, nss, nb, ncbps, ndbps, nltf, nsd in G["mod_table"]:
        if fcode == 0:
            rate = {0: 11, 1: 15, 2: 10, 3: 14, 4: 9, 5: 13, 6: 8, 7: 12}[int(mcs)]
            w = np.array([(rate >> i) & 1 for i in range(4)] + [0] + [(100 >> i) & 1 for i in range(12)] + [0] * 7, np.uint8)
            w[17] = w[0:17].sum() % 2
            ok, f = o.sig_legacy(w)
            assert ok and (f["mcs"], f["nSS"], f["nBPSCS"], f["nCBPS"], f["nDBPS"]) == (int(mcs), 1, int(nb), int(ncbps), int(ndbps))
        elif fcode == 1:
            w = np.zeros(48, np.uint8)
            w[0:7] = [(int(mcs) >> i) & 1 for i in range(7)]
            w[8:24] = [(100 >> i) & 1 for i in range(16)]
            _, f = o.sig_ht(w)
            assert (f["nSS"], f["nBPSCS"], f["nCBPS"], f["nDBPS"], f["nLTF"]) == (int(nss), int(nb), int(ncbps), int(ndbps), int(nltf))
        else:
            if int(mcs) == 9 and int(nsts) in (1, 2):       # (VHT MCS 9 at 20 MHz exists for 3 streams only; the toolbox refuses it)
                continue
            w = np.zeros(48, np.uint8)
            w[2] = 1
            w[10:13] = [((int(nsts) - 1) >> i) & 1 for i in range(3)]
            w[28:32] = [(int(mcs) >> i) & 1 for i in range(4)]
            _, f = o.sig_vht_a(w)
            assert (f["nSS"], f["nBPSCS"], f["nCBPS"], f["nDBPS"], f["nLTF"]) == (int(nss), int(nb), int(ncbps), int(ndbps), int(nltf))
