#!/usr/bin/env python3
"""Generate the committed golden fixtures from the REFERENCE's own code.

Runs only in the build container (needs /root/reference); the GPU box and the test-suite only read
the files this script wrote.  Nothing of the reference's source text is stored: the outputs are
 (a) vectors produced by *importing and running* the reference's Python toolbox
     (tools/phy80211.py, phy80211header.py, mac80211.py) — TX IQ, PSDU bytes, coded bits, Viterbi
     decisions, CRC-8 values, scrambler output, interleaver permutations;
 (b) SHA-256 digests of the literal constant tables in lib/cloud80211phy.cc (digest only).

    MPLBACKEND=Agg python tests/golden/gen_golden.py
"""
import hashlib
import json
import os
import re
import sys
import zlib

import numpy as np

sys.dont_write_bytecode = True
REF = "/root/reference"
sys.path.insert(0, os.path.join(REF, "tools"))
os.environ.setdefault("MPLBACKEND", "Agg")
import mac80211          # noqa: E402
import phy80211          # noqa: E402
import phy80211header as p8h  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
MULT = 12.0   # tools/pktGenExample.py:187


# ----------------------------------------------------------------------------- (b) table digests
def table_digests():
    src = open(os.path.join(REF, "lib", "cloud80211phy.cc")).read()
    names_int = ["SV_PUNC_12", "SV_PUNC_23", "SV_PUNC_34", "SV_PUNC_56", "SV_STATE_NEXT", "SV_STATE_OUTPUT",
                 "mapDeintLegacyBpsk", "mapDeintLegacyQpsk", "mapDeintLegacy16Qam", "mapDeintLegacy64Qam",
                 "mapDeintNonlegacyBpsk", "mapDeintNonlegacyQpsk", "mapDeintNonlegacy16Qam",
                 "mapDeintNonlegacy64Qam", "mapDeintNonlegacy256Qam",
                 "mapDeintNonlegacyBpsk2", "mapDeintNonlegacyQpsk2", "mapDeintNonlegacy16Qam2",
                 "mapDeintNonlegacy64Qam2", "mapDeintNonlegacy256Qam2",
                 "mapDeintVhtSigB20", "mapIntelVhtSigB20", "FFT_NL_SIG_DEMAP", "FFT_L_SIG_DEMAP", "C8P_LEGACY_D_SC"]
    names_flt = ["PILOT_P", "PILOT_L", "PILOT_HT_1", "PILOT_HT_2_1", "PILOT_HT_2_2", "PILOT_VHT",
                 "LTF_L_26_F_FLOAT", "LTF_L_26_F_FLOAT2", "LTF_NL_28_F_FLOAT", "LTF_NL_28_F_FLOAT_VHT22",
                 "LTF_NL_28_F_FLOAT2"]
    out = {}
    for name, kind in [(n, "i") for n in names_int] + [(n, "f") for n in names_flt]:
        m = re.search(r"const\s+(?:int|float)\s+" + re.escape(name) + r"\s*(?:\[[^\]]*\])+\s*=\s*\{(.*?)\};", src, re.S)
        assert m, name
        body = re.sub(r"//[^\n]*", "", m.group(1))
        toks = re.findall(r"[-+]?\d*\.?\d+(?:[eE][-+]?\d+)?f?", body)
        vals = [float(t.rstrip("f")) for t in toks]
        arr = np.array(vals, np.float32 if kind == "f" else np.int32)
        out[name] = {"n": int(arr.size), "dtype": "f32" if kind == "f" else "i32",
                     "sha256": hashlib.sha256(arr.tobytes()).hexdigest()}
    return out


# ----------------------------------------------------------------------------- (a) small function vectors
def small_vectors(rng):
    v = {}
    # CRC-8 of the SIG fields: genBitBitCrc8 (phy80211header.py:87)
    bits = rng.integers(0, 2, size=(16, 34)).astype(np.uint8)
    v["crc8_in"] = bits
    v["crc8_out"] = np.array([p8h.genBitBitCrc8([int(b) for b in row]) for row in bits], np.uint8)
    # scrambler sequence (procScramble :800) for all 127 seeds on zeros = the sequence itself
    v["scr_seed"] = np.arange(1, 128, dtype=np.int32)
    v["scr_seq"] = np.array([p8h.procScramble([0] * 140, int(s)) for s in range(1, 128)], np.uint8)
    # BCC + puncture of random bits (procBcc :763)
    msg = rng.integers(0, 2, size=120).astype(np.uint8)
    v["bcc_in"] = msg
    for cr, nm in [(p8h.CR.CR12, "12"), (p8h.CR.CR23, "23"), (p8h.CR.CR34, "34"), (p8h.CR.CR56, "56")]:
        v["bcc_" + nm] = np.array(p8h.procBcc([int(b) for b in msg], cr), np.uint8)
    # interleaver permutations (procInterleaveLegacy :851, procInterleaveNonLegacy :868) applied to an index ramp
    for mcs in (0, 2, 4, 6):
        mod = p8h.modulation(p8h.F.L, mcs, p8h.BW.BW20, 1, False)
        mod.nSym = 1
        ramp = [list(range(mod.nCBPS))]
        v["intl_L_%d" % mod.nBPSCS] = np.array(p8h.procInterleaveLegacy(ramp, mod)[0], np.int32)
    for mcs in (0, 1, 3, 5, 8):
        for nss in (1, 2):
            mod = p8h.modulation(p8h.F.VHT, mcs, p8h.BW.BW20, nss, False)
            mod.nSym = 1
            ramp = [list(range(mod.nCBPSS)) for _ in range(nss)]
            r = p8h.procInterleaveNonLegacy(ramp, mod)
            for s in range(nss):
                v["intl_NL_%d_ss%d_of%d" % (mod.nBPSCS, s, nss)] = np.array(r[s], np.int32)
    # QAM map tables (C_QAM_MODU_TAB) so the demapper's sign conventions can be checked
    for idx, nm in [(0, "bpsk"), (2, "qpsk"), (3, "16qam"), (4, "64qam"), (5, "256qam")]:
        v["qam_" + nm] = np.array(p8h.C_QAM_MODU_TAB[idx], np.complex64)
    return v


def viterbi_vectors(rng):
    """procViterbiDecoder (phy80211header.py:980-1045) on LLRs quantised to 1/8 so that double (Python)
    and float32 (C++/HIP) path metrics are both exact — ties included."""
    v = {}
    crs = [(p8h.CR.CR12, 0), (p8h.CR.CR23, 1), (p8h.CR.CR34, 2), (p8h.CR.CR56, 3)]
    k = 0
    for cr, cri in crs:
        for mode in ("coded", "random", "coarse"):
            T = int(rng.integers(180, 420))
            msg = list(rng.integers(0, 2, size=T - 6)) + [0] * 6
            coded = np.array(p8h.procBcc([int(b) for b in msg], cr), np.float64)
            if mode == "coded":
                llr = (2 * coded - 1) * 2.0 + rng.normal(0, 1.6, size=coded.size)
                llr = np.clip(np.round(llr * 8) / 8, -7.875, 7.875)
            elif mode == "random":
                llr = np.round(rng.uniform(-4, 4, size=coded.size) * 8) / 8
            else:   # very coarse: integers in {-1,0,1} → ties everywhere
                llr = rng.integers(-1, 2, size=coded.size).astype(np.float64)
            bits = p8h.procViterbiDecoder([float(x) for x in llr], T, cr)
            v["vit%d_llr" % k] = llr.astype(np.float32)
            v["vit%d_T" % k] = np.int32(T)
            v["vit%d_cr" % k] = np.int32(cri)
            v["vit%d_bits" % k] = np.array(bits, np.uint8)
            k += 1
    v["n"] = np.int32(k)
    return v


# ----------------------------------------------------------------------------- (a) packets from the reference TX
def mpdu_bytes(rng, n):
    body = bytes(rng.integers(0, 256, size=n - 4, dtype=np.uint8))
    return body + zlib.crc32(body).to_bytes(4, "little")


def gen_packet(fmt, mcs, nss, mpdu, rng):
    ins = phy80211.phy80211(ifDebug=False)
    if fmt == "L":
        mod = p8h.modulation(p8h.F.L, mcs, p8h.BW.BW20, 1, False)
        ins.genFromMpdu(mpdu, mod)
        psdu = mpdu
    elif fmt == "HT":
        mod = p8h.modulation(p8h.F.HT, mcs, p8h.BW.BW20, nss, False)
        ins.genFromMpdu(mpdu, mod)
        psdu = mpdu
    else:
        mod = p8h.modulation(p8h.F.VHT, mcs, p8h.BW.BW20, nss, False)
        psdu = mac80211.genAmpduVHT([mpdu]) if len(mpdu) else b""
        ins.genFromAmpdu(psdu, mod, vhtPartialAid=0, vhtGroupId=0)
    assert mod.initRes
    mult = MULT * (np.sqrt(2) if nss == 2 else 1.0)     # pktGenExample.py:187,210
    sig = ins.genFinalSig(multiplier=mult, cfoHz=0.0, num=1, gap=False, gapLen=0)
    tx = np.array(sig, np.complex64)
    coded = np.array(ins.esCodedBits[0], np.uint8) if len(mpdu) else np.zeros(0, np.uint8)
    fcode = {"L": 0, "HT": 1, "VHT": 2}[fmt]
    if len(mpdu):
        pdu = bytes([fcode, len(mpdu) % 256, len(mpdu) // 256]) + mpdu + bytes([mcs])
    else:
        pdu = b""
    return {"tx": tx, "mpdu": np.frombuffer(mpdu, np.uint8), "pdu": np.frombuffer(pdu, np.uint8),
            "coded": coded, "fmt": np.int32(fcode), "mcs": np.int32(mcs), "nss": np.int32(nss),
            "nsym": np.int32(ins.m.nSym), "psdu_len": np.int32(len(psdu)),
            "lsig_len": np.int32(ins.m.legacyLen)}


def gen_mu(rng):
    """SURVEY Appendix B config-5 loop: NDP -> per-station feedback (reference Python) -> ZF Q
    (tools/cmu_v3/cmu_ap.py:182-189) -> 2-user MU packet (genAmpduMu)."""
    h = np.array([[1 + 0.2j, 0.5 - 0.4j], [0.3 + 0.6j, -0.9 + 0.1j]], np.complex64)
    ndp = gen_packet("VHT", 0, 2, b"", rng)
    tx = ndp["tx"]
    vfb = []
    for u in range(2):
        rx = h[u, 0] * tx[0] + h[u, 1] * tx[1]
        base = 184 + 224 + 240          # sync + consumed-by-signal + (SIG-A 160 + STF 80)
        chan = [complex(x) for x in rx[base + 8: base + 72]] + [complex(x) for x in rx[base + 88: base + 152]]
        ltf = []
        for k in range(2):
            ltf.append(p8h.procRemovePilots(p8h.procToneDescaling(p8h.procRmDcNonDataSc(
                p8h.procFftDemod(chan[64 * k:64 * k + 64]), p8h.F.VHT),
                p8h.C_SCALENTF_LTF_VHT[p8h.BW.BW20.value], 2)))
        vfb.append(p8h.procVhtChannelFeedback(ltf, p8h.BW.BW20, 2, 1))
    bfQ = []
    for k in range(len(vfb[0])):
        H11, H21 = vfb[0][k][0][0][0], vfb[0][k][0][0][1]
        H12, H22 = vfb[1][k][0][0][0], vfb[1][k][0][0][1]
        d1 = np.sqrt(abs(H21) ** 2 + abs(H11) ** 2)
        d2 = np.sqrt(abs(H22) ** 2 + abs(H12) ** 2)
        bfQ.append(np.array([[-1 * H22 / d2, -1 * H21 / d1], [H12 / d2, H11 / d1]]))
    one = np.ones_like(bfQ[0])
    bfQf = [one] * 4 + bfQ[0:28] + [one] + bfQ[28:56] + [one] * 3
    m0, m1 = mpdu_bytes(rng, 100), mpdu_bytes(rng, 60)
    ins = phy80211.phy80211(ifDebug=False)
    ins.genAmpduMu(nUser=2, bfQ=bfQf, groupId=2,
                   ampdu0=mac80211.genAmpduVHT([m0]), mod0=p8h.modulation(p8h.F.VHT, 3, p8h.BW.BW20, 1, False),
                   ampdu1=mac80211.genAmpduVHT([m1]), mod1=p8h.modulation(p8h.F.VHT, 3, p8h.BW.BW20, 1, False))
    sig = ins.genFinalSig(multiplier=MULT * np.sqrt(2), cfoHz=0.0, num=1, gap=False, gapLen=0)
    mu_tx = np.array(sig, np.complex64)
    return {"h": h, "ndp_tx": tx, "mu_tx": mu_tx,
            "mpdu0": np.frombuffer(m0, np.uint8), "mpdu1": np.frombuffer(m1, np.uint8),
            "pdu0": np.frombuffer(bytes([2, 100, 0]) + m0 + bytes([3]), np.uint8),
            "pdu1": np.frombuffer(bytes([2, 60, 0]) + m1 + bytes([3]), np.uint8),
            "lsig_len": np.int32(ins.m.legacyLen)}



def gen_mu_recorded():
    """BASELINE configs[4] with the reference's RECORDED NDP captures (tools/cmu_chan0.bin, cmu_chan1.bin: the two VHT-LTF windows each
    station received from the 2-antenna AP — the only data files the reference holds on this path; read as
    tools/cmu_v4/cmu_ap_offline.py:82-95 reads them): feedback per station with the toolbox (procFftDemod -> procRmDcNonDataSc ->
    procToneDescaling -> procRemovePilots -> procVhtChannelFeedback, as the offline script does, without its compress / quantise round
    trip), zero-forcing Q by the cmu_v3 formula (tools/cmu_v3/cmu_ap.py:182-189; the v4 matmul path fails on numpy 2), 2-user MU
    packet with genAmpduMu (MCS 3 / MCS 3, 100 / 60-byte MPDUs).  What each station then receives is computed HERE, through the
    channel the captures themselves describe: per transmit chain c and subcarrier k, G_c[k] = (capture's LTF combination) / (what the
    clean NDP puts there), interpolated over pilots / DC / guards, applied per OFDM symbol as a circular convolution (the model the
    receiver's own estimate assumes; the legacy training fields are periodic, so the same holds for them).  Stored: the steering
    matrices, both received streams, the MPDUs.  (With the impulse response cut to the guard interval instead, station 0 — whose chain-0
    path is 14 dB below its chain-1 path in the recording — sees 4 dB of residual inter-user interference and cannot decode MCS 3:
    the recording's estimation noise is then CSI error.)"""
    cap = [np.fromfile(os.path.join(REF, "tools", "cmu_chan%d.bin" % u), np.complex64) for u in range(2)]
    assert all(len(c) == 128 for c in cap)
    rng = np.random.default_rng(8021555)
    ndp = gen_packet("VHT", 0, 2, b"", rng)["tx"]
    base = 160 + 160 + 80 + 160 + 80                       # L-STF, L-LTF, L-SIG, VHT-SIG-A, VHT-STF
    win = lambda x, k: np.fft.fft(x[base + 80 * k + 8: base + 80 * k + 72])
    clean = [[win(ndp[c], 0), win(ndp[c], 1)] for c in range(2)]
    used = np.abs(clean[0][0]) > 1e-6
    pil = np.zeros(64, bool); pil[[7, 21, 43, 57]] = True
    data = used & ~pil
    sc = np.where(np.arange(64) < 32, np.arange(64), np.arange(64) - 64).astype(float)
    # stream separation of the 2 x 2 P matrix at the data tones: one chain appears in F1 + F2, the other in F1 - F2 (the clean NDP says which)
    plus_is_chain0 = np.abs((clean[0][0] + clean[0][1])[data]).mean() > np.abs((clean[0][0] - clean[0][1])[data]).mean()

    def per_chain(F1, F2):
        return [F1 + F2, F1 - F2] if plus_is_chain0 else [F1 - F2, F1 + F2]
    T = [per_chain(clean[c][0], clean[c][1])[c] for c in range(2)]
    Gk = np.zeros((2, 2, 64), complex)                      # [station][chain][bin]
    for u in range(2):
        S = per_chain(np.fft.fft(cap[u][:64]), np.fft.fft(cap[u][64:]))
        for c in range(2):
            g = S[c][data] / T[c][data]
            xs = sc[data]; idx = np.argsort(xs)
            Gk[u, c] = np.interp(sc, xs[idx], g.real[idx]) + 1j * np.interp(sc, xs[idx], g.imag[idx])
    vfb = []
    for u in range(2):
        ltfSym = [p8h.procRemovePilots(p8h.procToneDescaling(p8h.procRmDcNonDataSc(p8h.procFftDemod([complex(x) for x in cap[u][64 * k:64 * k + 64]]), p8h.F.VHT),
                                                             p8h.C_SCALENTF_LTF_VHT[p8h.BW.BW20.value], 2)) for k in range(2)]
        vfb.append(p8h.procVhtChannelFeedback(ltfSym, p8h.BW.BW20, 2, 1))
    bfQ = []
    for k in range(len(vfb[0])):
        H11, H21 = vfb[0][k][0][0][0], vfb[0][k][0][0][1]
        H12, H22 = vfb[1][k][0][0][0], vfb[1][k][0][0][1]
        d1 = np.sqrt(abs(H21) ** 2 + abs(H11) ** 2)
        d2 = np.sqrt(abs(H22) ** 2 + abs(H12) ** 2)
        bfQ.append(np.array([[-1 * H22 / d2, -1 * H21 / d1], [H12 / d2, H11 / d1]]))
    one = np.ones_like(bfQ[0])
    bfQf = [one] * 4 + bfQ[0:28] + [one] + bfQ[28:56] + [one] * 3
    m0, m1 = mpdu_bytes(rng, 100), mpdu_bytes(rng, 60)
    ins = phy80211.phy80211(ifDebug=False)
    ins.genAmpduMu(nUser=2, bfQ=bfQf, groupId=2,
                   ampdu0=mac80211.genAmpduVHT([m0]), mod0=p8h.modulation(p8h.F.VHT, 3, p8h.BW.BW20, 1, False),
                   ampdu1=mac80211.genAmpduVHT([m1]), mod1=p8h.modulation(p8h.F.VHT, 3, p8h.BW.BW20, 1, False))
    mu_tx = np.array(ins.genFinalSig(multiplier=MULT * np.sqrt(2), cfoHz=0.0, num=1, gap=False, gapLen=0), np.complex64)
    # undo the toolbox's half-amplitude symbol edges (as tests/c8p_cases.dewindow does), then the channel, symbol by symbol
    txc = np.array(mu_tx, np.complex128)
    for b in range(160, txc.shape[1], 80):
        if b == 160 or b >= 320:
            txc[:, b - 1] *= 2; txc[:, b] *= 2
    n = txc.shape[1]
    assert n % 80 == 0
    rx = np.zeros((2, n + 320 + 160), complex)
    for u in range(2):
        y = np.zeros(n, complex)
        for c in range(2):
            # L-STF (period 16) and L-LTF (32 + 64 + 64, period 64): one period block through the channel, then the same extension
            stf = np.fft.ifft(Gk[u, c] * np.fft.fft(txc[c][16:80]))
            y[0:160] += np.concatenate([stf[-16:], stf, stf, stf[:16]])
            lt = np.fft.ifft(Gk[u, c] * np.fft.fft(txc[c][192:256]))
            y[160:320] += np.concatenate([lt[-32:], lt, lt])
            for b in range(320, n, 80):
                body = np.fft.ifft(Gk[u, c] * np.fft.fft(txc[c][b + 16:b + 80]))
                y[b:b + 80] += np.concatenate([body[-16:], body])
        rx[u, :n] = y
    return {"rx0": rx[0].astype(np.complex64), "rx1": rx[1].astype(np.complex64), "mu_tx": mu_tx,
            "g": Gk.astype(np.complex64), "bfq": np.array(bfQf, np.complex64),
            "mpdu0": np.frombuffer(m0, np.uint8), "mpdu1": np.frombuffer(m1, np.uint8),
            "pdu0": np.frombuffer(bytes([2, 100, 0]) + m0 + bytes([3]), np.uint8),
            "pdu1": np.frombuffer(bytes([2, 60, 0]) + m1 + bytes([3]), np.uint8)}


def gen_mu_tx():
    """2-user MU-MIMO transmit vectors (genAmpduMu) with a full-rank random steering matrix per subcarrier, so that
    every term of the spatial mapping is pinned: both chains' IQ, both users' punctured coded bits, the Q matrices in
    FFT order (bin 0 = DC, as lib/modulation2_impl.cc:120 stores d_vhtMuBfQ), A-MPDUs, geometry."""
    out = {}
    cases = [(3, 3, 100, 60, 2), (0, 7, 40, 300, 5), (8, 1, 500, 32, 62), (5, 4, 64, 64, 1)]
    for ci, (mcs0, mcs1, n0, n1, gid) in enumerate(cases):
        rng = np.random.default_rng(8021500 + ci)
        q = (rng.normal(size=(56, 2, 2)) + 1j * rng.normal(size=(56, 2, 2))) / 2.0
        one = np.ones((2, 2), np.complex128)
        bfq = [one] * 4 + [q[i] for i in range(28)] + [one] + [q[28 + i] for i in range(28)] + [one] * 3   # -32 .. 31
        m0, m1 = mpdu_bytes(rng, n0), mpdu_bytes(rng, n1)
        a0, a1 = mac80211.genAmpduVHT([m0]), mac80211.genAmpduVHT([m1])
        ins = phy80211.phy80211(ifDebug=False)
        coded = []
        orig = ins._phy80211__genCodedBits

        def spy():
            orig()
            coded.append(np.array(ins.esCodedBits[0], np.uint8))
        ins._phy80211__genCodedBits = spy
        ins.genAmpduMu(nUser=2, bfQ=bfq, groupId=gid,
                       ampdu0=a0, mod0=p8h.modulation(p8h.F.VHT, mcs0, p8h.BW.BW20, 1, False),
                       ampdu1=a1, mod1=p8h.modulation(p8h.F.VHT, mcs1, p8h.BW.BW20, 1, False))
        sig = ins.genFinalSig(multiplier=MULT * np.sqrt(2), cfoHz=0.0, num=1, gap=False, gapLen=0)
        assert len(coded) == 2
        fftq = np.array([bfq[(b + 32) % 64] for b in range(64)], np.complex64).reshape(64, 4)     # bin b <- subcarrier b (b<32) / b-64
        out.update(flat("mu%d" % ci, {
            "tx": np.array(sig, np.complex64), "bfq": fftq, "ampdu0": np.frombuffer(a0, np.uint8),
            "ampdu1": np.frombuffer(a1, np.uint8), "mpdu0": np.frombuffer(m0, np.uint8), "mpdu1": np.frombuffer(m1, np.uint8),
            "coded0": coded[0], "coded1": coded[1], "mcs": np.array([mcs0, mcs1], np.int32), "gid": np.int32(gid),
            "nsym": np.int32(ins.m.nSym), "lsig_len": np.int32(ins.m.legacyLen)}))
    return out


# ----------------------------------------------------------------------------- (a) the toolbox's RECEIVE side
def gen_rx_side():
    """tools/phy80211.py:842-1016 (procSisoRx and the private steps it calls) on the toolbox's own transmissions at CFO 0 / +-50 kHz
    with 35 dB of AWGN: STF trigger index, coarse CFO, LTF sync index, coarse + fine CFO, LTF SNR, the legacy channel estimate, the L-SIG
    constellation / bits / SNR, HT-SIG and VHT-SIG-A bits, the VHT channel and SIG-B LLRs — every intermediate the method computes (it only
    prints them), taken by calling its steps in its own order.  And tools/phy80211header.py:1055-1175 on fuzzed bit fields: procCheckSig*,
    procParserSigLegacy / procParserSigHt, and the `modulation` tables for every HT / VHT rate."""
    out = {}
    rng = np.random.default_rng(802110)
    cases = [("L", 0, 60), ("L", 4, 200), ("L", 7, 333), ("HT", 0, 80), ("HT", 5, 300), ("VHT", 0, 70), ("VHT", 4, 150), ("VHT", 7, 400)]
    k = 0
    for fmt, mcs, ln in cases:
        for cfo in (0.0, 50000.0, -50000.0):
            mpdu = mpdu_bytes(rng, ln)
            ins = phy80211.phy80211(ifDebug=False)
            if fmt == "VHT":
                mod = p8h.modulation(p8h.F.VHT, mcs, p8h.BW.BW20, 1, False)
                ins.genFromAmpdu(mac80211.genAmpduVHT([mpdu]), mod, vhtPartialAid=0, vhtGroupId=0)
            else:
                mod = p8h.modulation(p8h.F.L if fmt == "L" else p8h.F.HT, mcs, p8h.BW.BW20, 1, False)
                ins.genFromMpdu(mpdu, mod)
            sig = ins.genFinalSig(multiplier=MULT, cfoHz=cfo, num=1, gap=False, gapLen=0)
            lead = 300 + 7 * k
            x = np.concatenate([np.zeros(lead), np.array(sig, np.complex128).reshape(-1), np.zeros(400)])
            sigma = np.sqrt(0.1875 ** 2 / 10 ** 3.5)
            x = x + (rng.normal(0, sigma / np.sqrt(2), len(x)) + 1j * rng.normal(0, sigma / np.sqrt(2), len(x)))
            x = x.astype(np.complex64)                 # what a receiver is handed (and what the fixture stores)
            rx = [complex(v) for v in x]
            R = phy80211.phy80211(ifDebug=False)
            R.rxSisoSig = rx
            trig = R._phy80211__procRxLegacyStfTrigger(rx)
            assert trig >= 0
            coarse = R._phy80211__procRxLegacyStfCoarseCfoEst(rx[trig:], 64)
            mid = R._phy80211__procRxLegacyLtfSync(rx[trig + 80:trig + 320], coarse)
            assert mid >= 0
            ltf = trig + 80 + mid + 10
            rad = R.rxCfo * 2 * np.pi / 20000000
            leg = [rx[ltf + i] * complex(np.cos(rad * i), np.sin(rad * i)) for i in range(0, 208)]
            R._phy80211__procRxLegacyChanEst(leg[0:128])
            qam = R._phy80211__procRxLegacySigDemod(leg[144:])
            inted = list(np.real(qam))
            coded = p8h.procDeinterleaveSigL(inted)
            bits = p8h.procViterbiDecoder(coded, 24, p8h.CR.CR12)
            d = {"x": x, "lead": np.int32(lead), "cfo_hz": np.float64(cfo), "trigger": np.int32(trig), "coarse_cfo_hz": np.float64(coarse),
                 "ltf_sync": np.int32(mid), "ltf_index": np.int32(ltf), "rx_cfo_hz": np.float64(R.rxCfo), "rx_snr_db": np.float64(R.rxSnrDb),
                 "chan_l": np.array(R.rxChanL, np.complex128), "lsig_qam": np.array(qam, np.complex128), "lsig_llr_coded": np.array(coded, np.float64),
                 "lsig_bits": np.array(bits, np.uint8), "lsig_snr": np.float64(p8h.procSnrSigLegacy(qam, bits)),
                 "lsig_check": np.int32(bool(p8h.procCheckSigLegacy(bits))), "fmt": np.int32({"L": 0, "HT": 1, "VHT": 2}[fmt]), "mcs": np.int32(mcs),
                 "mpdu_len": np.int32(ln)}
            if fmt != "L":
                _, _, nsym_l = p8h.procParserSigLegacy(bits)
                nsym_l = min(nsym_l, 8)                # (its rate table only knows 6 and 54 Mb/s: enough symbols for the SIG fields either way)
                ss = [rx[ltf + 208 + i] * complex(np.cos(rad * i), np.sin(rad * i)) for i in range(0, max(nsym_l, 6) * 80)]
                nl = R._phy80211__procRxNonLegacySigDemod(ss[16:80] + ss[96:160])
                ht_c = p8h.procDeinterleaveSigL(list(np.imag(nl[0:48]))) + p8h.procDeinterleaveSigL(list(np.imag(nl[48:96])))
                vht_c = p8h.procDeinterleaveSigL(list(np.real(nl[0:48]))) + p8h.procDeinterleaveSigL(list(np.imag(nl[48:96])))
                hb = p8h.procViterbiDecoder(ht_c, 48, p8h.CR.CR12)
                vb = p8h.procViterbiDecoder(vht_c, 48, p8h.CR.CR12)
                d.update({"nl_qam": np.array(nl, np.complex128), "ht_bits": np.array(hb, np.uint8), "vht_a_bits": np.array(vb, np.uint8),
                          "ht_check": np.int32(bool(p8h.procCheckSigHt(hb))), "vht_a_check": np.int32(bool(p8h.procCheckSigVhtA(vb)))})
            out.update(flat("rx%02d_%s%d" % (k, fmt, mcs), d))
            k += 1
    out["n_cases"] = np.int32(k)
    # the checks and parsers on fuzzed fields: 400 random words per field + every single-bit flip of a valid one
    fz = np.random.default_rng(802111)

    def words(n, valid):
        w = [fz.integers(0, 2, n).astype(np.uint8) for _ in range(400)]
        for b in range(n):
            v = valid.copy(); v[b] ^= 1
            w.append(v)
        w.append(valid.copy())
        return np.array(w, np.uint8)
    lv = np.array([1, 1, 0, 1, 0] + [(300 >> i) & 1 for i in range(12)] + [0] + [0] * 6, np.uint8)
    lv[17] = lv[0:17].sum() % 2
    lw = words(24, lv)
    for i in range(0, 400, 2):                         # half of the random ones: a valid rate field, tail and parity, random length
        lw[i][0:4] = [1, 1, 0, 1]; lw[i][4] = 0; lw[i][18:24] = 0; lw[i][17] = lw[i][0:17].sum() % 2
    out["fuzz_l_bits"] = lw
    out["fuzz_l_check"] = np.array([bool(p8h.procCheckSigLegacy([int(b) for b in w])) for w in lw], np.uint8)
    out["fuzz_l_parse"] = np.array([p8h.procParserSigLegacy([int(b) for b in w]) for w in lw], np.int32)
    hv = np.array([(5 >> i) & 1 for i in range(7)] + [0] + [(1234 >> i) & 1 for i in range(16)] + [1, 0, 1, 1, 0, 0, 0, 0, 0, 0], np.uint8)
    hv = np.concatenate([hv, np.array(p8h.genBitBitCrc8([int(b) for b in hv]), np.uint8), np.zeros(6, np.uint8)])
    hw = words(48, hv)
    for i in range(0, 400, 2):                         # half of the random ones carry a correct CRC-8
        hw[i][34:42] = p8h.genBitBitCrc8([int(b) for b in hw[i][0:34]])
    out["fuzz_ht_bits"] = hw
    out["fuzz_ht_check"] = np.array([bool(p8h.procCheckSigHt([int(b) for b in w])) for w in hw], np.uint8)
    hp = []
    for w in hw:
        m = p8h.procParserSigHt([int(b) for b in w])
        ln = sum(int(w[i + 8]) << i for i in range(16))
        hp.append([int(m.mcs) if m.initRes else -1, ln, int(m.nSS) if m.initRes else -1, int(m.nBPSCS) if m.initRes else -1,
                   int(m.nCBPS) if m.initRes else -1, int(m.nDBPS) if m.initRes else -1, int(m.initRes)])
    out["fuzz_ht_parse"] = np.array(hp, np.int32)
    vv = np.array([0, 0, 1, 0] + [0] * 6 + [0, 0, 0] + [0] * 9 + [0, 1, 0, 0, 0, 0] + [(7 >> i) & 1 for i in range(4)] + [0, 1], np.uint8)
    vv = np.concatenate([vv, np.array(p8h.genBitBitCrc8([int(b) for b in vv]), np.uint8), np.zeros(6, np.uint8)])
    vw = words(48, vv)
    for i in range(0, 400, 2):
        vw[i][34:42] = p8h.genBitBitCrc8([int(b) for b in vw[i][0:34]])
    out["fuzz_vht_bits"] = vw
    out["fuzz_vht_check"] = np.array([bool(p8h.procCheckSigVhtA([int(b) for b in w])) for w in vw], np.uint8)
    # the rate tables: modulation(format, mcs, BW20, nSTS) for every HT / VHT rate of one and two streams
    mt = []
    for f, fcode, rates in ((p8h.F.L, 0, range(8)), (p8h.F.HT, 1, range(16)), (p8h.F.VHT, 2, range(10))):
        for mcs in rates:
            for nsts in ((1,) if fcode == 0 else ((mcs // 8 + 1,) if fcode == 1 else (1, 2))):
                m = p8h.modulation(f, mcs, p8h.BW.BW20, nsts, False)
                if not m.initRes:
                    continue
                mt.append([fcode, mcs, nsts, int(m.nSS), int(m.nBPSCS), int(m.nCBPS), int(m.nDBPS), int(m.nLtf), int(m.nSD)])
    out["mod_table"] = np.array(mt, np.int32)
    return out


def flat(prefix, d):
    return {prefix + "/" + k: v for k, v in d.items()}


def main():
    if "--mu-recorded" in sys.argv:  # cfg 5 with the reference's recorded NDP captures (round 3); own seed, leaves the others untouched
        np.savez_compressed(os.path.join(OUT, "packets_mu_recorded.npz"), **gen_mu_recorded())
        print("packets_mu_recorded.npz written")
        return
    if "--rx-side" in sys.argv:     # round 5: the toolbox's receive side; own seeds, leaves the other files untouched
        np.savez_compressed(os.path.join(OUT, "rx_side.npz"), **gen_rx_side())
        print("rx_side.npz written")
        return
    if "--mu-tx" in sys.argv:       # added later than the other files; own seeds, leaves them untouched
        np.savez_compressed(os.path.join(OUT, "packets_mu_tx.npz"), **gen_mu_tx())
        print("packets_mu_tx.npz written")
        return
    rng = np.random.default_rng(80211)
    json.dump(table_digests(), open(os.path.join(OUT, "table_digests.json"), "w"), indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(OUT, "small_vectors.npz"), **small_vectors(rng))
    np.savez_compressed(os.path.join(OUT, "viterbi_vectors.npz"), **viterbi_vectors(rng))

    # BASELINE.json configs
    cfg = {}
    body = bytes([(5 * i + 1) & 255 for i in range(96)])
    cfg.update(flat("cfg1_L_mcs0_100B", gen_packet("L", 0, 1, body + zlib.crc32(body).to_bytes(4, "little"), rng)))
    for p in range(4):
        r = np.random.default_rng(8021100 + p)
        cfg.update(flat("cfg2_VHT_mcs7_1458B_%d" % p, gen_packet("VHT", 7, 1, mpdu_bytes(r, 1458), r)))
    cfg.update(flat("cfg3_HT_mcs15_2x2_1458B", gen_packet("HT", 15, 2, mpdu_bytes(rng, 1458), rng)))
    cfg.update(flat("cfg4_VHT_mcs8_1458B", gen_packet("VHT", 8, 1, mpdu_bytes(rng, 1458), rng)))
    np.savez_compressed(os.path.join(OUT, "packets_cfg.npz"), **cfg)

    # every supported MCS with a short MPDU
    sweep = {}
    for mcs in range(8):
        sweep.update(flat("L_%d" % mcs, gen_packet("L", mcs, 1, mpdu_bytes(rng, 40 + mcs), rng)))
    for mcs in range(8):
        sweep.update(flat("HT_%d" % mcs, gen_packet("HT", mcs, 1, mpdu_bytes(rng, 40 + mcs), rng)))
    for mcs in range(9):
        sweep.update(flat("VHT_%d" % mcs, gen_packet("VHT", mcs, 1, mpdu_bytes(rng, 40 + mcs), rng)))
    for mcs in range(8, 16):
        sweep.update(flat("HT2_%d" % mcs, gen_packet("HT", mcs, 2, mpdu_bytes(rng, 40 + mcs), rng)))
    for mcs in range(9):
        sweep.update(flat("VHT2_%d" % mcs, gen_packet("VHT", mcs, 2, mpdu_bytes(rng, 40 + mcs), rng)))
    sweep.update(flat("VHT_NDP", gen_packet("VHT", 0, 2, b"", rng)))
    np.savez_compressed(os.path.join(OUT, "packets_sweep.npz"), **sweep)

    np.savez_compressed(os.path.join(OUT, "packets_mu.npz"), **gen_mu(rng))
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
