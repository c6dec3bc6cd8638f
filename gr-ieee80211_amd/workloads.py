"""The BASELINE.json workloads built on the box by the engine's own transmitter, and the small channel helpers they
share with the tests: what bench.py measures and what the parity tests feed.  Product side only (the engine and numpy): c8p_tx ->
channel (torch, on the device) -> c8p_signal gives packets at the signal -> demod boundary, resident in HBM."""
import numpy as np

from . import abi

PAD = 320             # signal_impl.cc:194-202: zero items appended behind every packet
H22 = np.array([[1, 0.3j], [0.2, 0.9]], np.complex64)   # SURVEY §8d cfg 3 fixture channel


def dewindow(tx):
    """Undo the toolbox's half-amplitude symbol edges (procConcat2Symbol, phy80211header.py:894-899) so that a
    noiseless frame is a clean OFDM signal: every 80-sample symbol boundary carries two halved samples."""
    tx = np.array(tx, np.complex64, copy=True)
    n = tx.shape[-1]
    for b in range(160, n, 80):          # boundaries after L-STF(160), L-LTF(160) and every 80-sample field
        if b == 160 or b >= 320:
            tx[..., b - 1] *= 2
            tx[..., b] *= 2
    return tx


def awgn(shape, sigma, rng):
    return (rng.normal(0, sigma / np.sqrt(2), shape) + 1j * rng.normal(0, sigma / np.sqrt(2), shape)).astype(np.complex64)


def through_channel(tx, H=None, snr_db=None, rng=None, lead=0, sig_amp=0.1875):
    """tx: [nss][N] complex64 -> rx [n_rx][lead + N + PAD + 80]; H: [n_rx][nss] flat channel.
    Noise sigma follows tools/performance/perf_siso.py:92 (sigma = sqrt(amp^2 / 10^(snr/10)))."""
    tx = np.atleast_2d(tx)
    nss, n = tx.shape
    if H is None:
        H = np.eye(nss, dtype=np.complex64)
    H = np.asarray(H, np.complex64)
    rx = np.zeros((H.shape[0], lead + n + PAD + 160), np.complex64)
    rx[:, lead:lead + n] = (H @ tx).astype(np.complex64)
    if snr_db is not None:
        sigma = np.sqrt(sig_amp ** 2 / (10 ** (snr_db / 10.0)))
        rx = rx + awgn(rx.shape, sigma, rng)
    return rx.astype(np.complex64)



def ampdu_vht_multi(mpdus):
    """mac80211.genAmpduVHT(list) for any number of subframes: EOF is set only when there is a single one (:339-341)"""
    out = b""
    for mpdu in mpdus:
        n = len(mpdu)
        lb = [(n >> i) & 1 for i in range(14)]
        bits = [1 if len(mpdus) == 1 else 0, 0] + lb[12:14] + lb[0:12]
        c = [1] * 8
        for b in bits:
            c = [b ^ c[7], b ^ c[7] ^ c[0], b ^ c[7] ^ c[1]] + c[2:7]
        bits = bits + [1 - b for b in c[::-1]] + [(0x4E >> i) & 1 for i in range(8)]
        pkt = bytes(sum(bits[8 * i + j] << j for j in range(8)) for i in range(4)) + bytes(mpdu)
        out += pkt + b"\x00" * ((-len(pkt)) % 4)
    return out


def ampdu_vht(mpdu):
    """mac80211.genAmpduVHT([mpdu]) (tools/mac80211.py:333-360): delimiter {EOF, reserved, len[12:14], len[0:12], CRC-8,
    signature 0x4e} + MPDU, padded to a multiple of 4 bytes"""
    n = len(mpdu)
    lb = [(n >> i) & 1 for i in range(14)]
    bits = [1, 0] + lb[12:14] + lb[0:12]
    c = [1] * 8
    for b in bits:
        c = [b ^ c[7], b ^ c[7] ^ c[0], b ^ c[7] ^ c[1]] + c[2:7]
    bits = bits + [1 - b for b in c[::-1]] + [(0x4E >> i) & 1 for i in range(8)]
    d = bytes(sum(bits[8 * i + j] << j for j in range(8)) for i in range(4))
    pkt = d + bytes(mpdu)
    return pkt + b"\x00" * ((-len(pkt)) % 4)


H_MU = np.array([[1 + 0.2j, 0.5 - 0.4j], [0.3 + 0.6j, -0.9 + 0.1j]], np.complex64)     # SURVEY.md §8d cfg 5: rows = stations


# BASELINE.json configs as SURVEY.md §8d states them: (format, mcs [per stream], transmit chains, receive antennas, MPDU bytes,
# flat channel [n_ant][n_chain] or None, user position, seed base, what it is)
WORKLOADS = {
    "cfg1": (0, 0, 1, 1, 100, None, 0, 8021000, "configs[0]: legacy 802.11a MCS0 BPSK r=1/2 SISO, 100-byte MPDU"),
    "cfg2": (2, 7, 1, 1, 1458, None, 0, 8021100, "configs[1]: VHT MCS7 64-QAM r=5/6 SISO, 1458-byte MPDU"),
    "cfg3": (1, 7, 2, 2, 1458, H22, 0, 8021300, "configs[2]: HT MCS15 (64-QAM r=5/6 x 2 streams) 2x2, 1458-byte MPDU, demod2's zero-forcing path"),
    "cfg4": (2, 8, 1, 1, 1458, None, 0, 8021400, "configs[3] with the MCS8 substitution of SURVEY.md §7: VHT MCS8 256-QAM r=3/4 SISO, 1458-byte MPDU"),
    "cfg5u0": (3, 3, 2, 1, 100, H_MU[0:1], 0, 8021500, "configs[4]: VHT 2-user MU-MIMO (MCS3 / MCS3, 100 / 60-byte MPDUs, zero-forcing Q of the cmu_v3 formula), station 0"),
    "cfg5u1": (3, 3, 2, 1, 60, H_MU[1:2], 1, 8021500, "configs[4]: the same packets at station 1 (60-byte MPDUs)"),
    # not BASELINE configs: random formats / MCS / lengths per packet (differential and soak tests)
    "random1": (-1, 0, 1, 1, 0, None, 0, 8021700, "random legacy / HT / VHT, MCS and length (14-3000 bytes) per packet, one stream"),
    "random2": (-2, 0, 2, 2, 0, H22, 0, 8021800, "random HT MCS 8-15 / VHT 2 x MCS 0-8 and length (14-3000 bytes) per packet, two streams on two antennas"),
    "randommu0": (-3, 0, 2, 1, 0, H_MU[0:1], 0, 8021900, "VHT 2-user MU-MIMO, each user's MCS (0-8) and MPDU length (14-1500 bytes) random per packet, station 0"),
    "randommu1": (-3, 0, 2, 1, 0, H_MU[1:2], 1, 8021900, "the same packets at station 1"),
}


# a 3-tap frequency-selective channel (delays 0, 1, 2 samples, inside the 16-sample guard interval): -9 and -15 dB echoes with fixed
# phases, unit total power.  |H(f)| swings between 0.55 and 1.38 across the 56 tones (+2.8 / -5.2 dB).
TAPS3 = (np.array([1.0, 0.35 * np.exp(0.9j), 0.18 * np.exp(-2.1j)]) / np.sqrt(1.0 + 0.35 ** 2 + 0.18 ** 2)).astype(np.complex64)


def workload_on_device(kind, n, snr_db, seed, torch, dev, local=0, mcs=None, seed_base=None, window=False, taps=None):
    """A BASELINE.json config (WORKLOADS) as SURVEY.md §8d specifies it, generated on the box by the build's own transmitter:
    packet p carries its own MPDU (bytes from numpy.random.default_rng(seed_base + seed + p) + CRC-32; inside an A-MPDU for VHT),
    amplitude as tools/pktGenExample.py (multiplier 12, x sqrt(2) on two chains), clean OFDM symbols (C8P_TX_NO_WINDOW: the toolbox's
    half-amplitude symbol edges cost 19.6 dB of self-interference at the receiver's FFT window), a flat channel
    H e^{j theta_p} and AWGN at snr_db (sigma per tools/performance/perf_siso.py:92).  c8p_tx -> channel (torch, on the device) ->
    c8p_signal at the nominal sync item (L-LTF + 24, CFO 0) gives the packets at the signal -> demod boundary, all resident in HBM.
    window=True: the toolbox's half-amplitude symbol edges ARE applied (c8p_tx_pkt.flags = 0: the waveform of tools/phy80211.py, whose
    self-interference caps the effective SNR at 19.6 dB); taps: a complex FIR applied to every transmit chain's stream before H (a
    frequency-selective channel; TAPS3).
    Returns (descriptors [host], iq DevArray, chan DevArray, expected PDUs, keep-alive objects)."""
    import zlib
    from . import engine, shard
    fmt, mcs0, nchain, n_ant, mlen, H, mupos, sb, _ = WORKLOADS[kind]
    mcs = mcs0 if mcs is None else mcs
    seed_base = sb if seed_base is None else seed_base
    GAP = 120
    pk = np.zeros(n, abi.TX_PKT_DT)
    mu = np.zeros(n if fmt in (3, -3) else 0, abi.TX_MU_DT)
    arena, exp = [], []
    off = 0

    def put(b):
        nonlocal off
        arena.append(np.frombuffer(b, np.uint8))
        off += len(b)
        return off - len(b)
    fmt_cfg = fmt
    for p in range(n):
        rng = np.random.default_rng(seed_base + seed + p)
        if fmt_cfg in (-1, -2):                            # random1 / random2: this packet's own format, rate and length
            fmt = int(rng.integers(0, 3)) if fmt_cfg == -1 else int(rng.integers(1, 3))
            mcs = int(rng.integers(0, 8 if fmt < 2 else 9))
            mlen = int(rng.integers(14, 3000))
        mcs_u1 = mcs
        if fmt_cfg == -3:                                  # randommu0 / 1: both users' rate and length drawn per packet
            fmt = 3
            mcs, mcs_u1 = int(rng.integers(0, 9)), int(rng.integers(0, 9))
            # both users are padded to the longer one's symbol count, and the padded data field of EITHER must fit the decoder's trellis
            # limit (32 782 steps, decode_impl.cc:93): a common symbol budget first, then each user's length inside it
            ndbps = (26, 52, 78, 104, 156, 208, 234, 260, 312)
            nsym_t = int(rng.integers(2, min(110, 32782 // max(ndbps[mcs], ndbps[mcs_u1])) + 1))
            lens = tuple(int(rng.integers(14, max(15, min(1500, (nsym_t * ndbps[m_] - 22) // 8 - 8)))) for m_ in (mcs, mcs_u1))
        else:
            lens = (100, 60) if fmt == 3 else (mlen,)
        mpdus = []
        for ln in lens:
            body = bytes(rng.integers(0, 256, size=ln - 4, dtype=np.uint8))
            mpdus.append(body + zlib.crc32(body).to_bytes(4, "little"))
        ps = [ampdu_vht(m) if fmt >= 2 else m for m in mpdus]
        pk[p]["psdu_off"], pk[p]["len"], pk[p]["format"], pk[p]["mcs"] = put(ps[0]), len(ps[0]), fmt, mcs
        pk[p]["scr_seed"], pk[p]["gap"], pk[p]["nss"] = 93, GAP, nchain
        pk[p]["mult"] = 12.0 * (np.sqrt(2.0) if nchain == 2 else 1.0)
        pk[p]["flags"] = 0 if window else abi.TX_NO_WINDOW
        if fmt == 3:
            mu[p]["psdu_off"], mu[p]["len"], mu[p]["mcs"], mu[p]["group_id"], mu[p]["q_off"] = put(ps[1]), len(ps[1]), mcs_u1, 2, 0
        m = mpdus[mupos]
        exp.append(bytes([min(fmt, 2), len(m) % 256, len(m) // 256]) + m + bytes([mcs + 8 if (fmt == 1 and nchain == 2) else (mcs_u1 if (fmt == 3 and mupos == 1) else mcs)]))
    fmt = 3 if fmt_cfg == -3 else fmt_cfg
    bfq = None
    if fmt == 3:           # zero-forcing steering (tools/cmu_v3/cmu_ap.py:182-189): nulls the other user's stream at each station
        q = np.linalg.inv(np.asarray(H_MU, np.complex128))
        q = q / np.linalg.norm(q, axis=0, keepdims=True)
        bfq = np.tile(q.reshape(1, 4), (64, 1)).astype(np.complex64).reshape(256)
    e_tx = engine.Engine(device=local)
    e_tx.tx(pk, np.concatenate(arena), tail_gap=400, mu=mu if fmt == 3 else None, bfq=bfq)
    e_tx.sync()
    desc = e_tx.fetch(abi.BUF_TX_DESC, abi.TX_DESC_DT)
    p_iq, nbytes = e_tx.buffer(abi.BUF_TX_IQ)
    ntot = nbytes // 8 // nchain                               # items per transmit chain (chain 0 | chain 1)
    z = torch.view_as_complex(shard.device_bytes(p_iq, nbytes, dev).view(torch.float32).view(nchain, ntot, 2))
    g = torch.Generator(device=dev); g.manual_seed(seed_base + 100 + seed)
    theta = torch.rand(n, generator=g, device=dev, dtype=torch.float32) * (2 * np.pi)
    starts = torch.from_numpy(desc["iq_off"].astype(np.int64)).to(dev)
    idx = (torch.searchsorted(starts, torch.arange(ntot, device=dev), right=True) - 1).clamp(min=0)
    ph = torch.polar(torch.ones(n, device=dev), theta)[idx]
    Hm = torch.from_numpy(np.eye(1, dtype=np.complex64) if H is None else np.asarray(H, np.complex64)).to(dev)
    if taps is not None:                                       # echoes: z[t] <- sum_k taps[k] z[t - k] (the streams end in idle air)
        zt = torch.zeros_like(z)
        for k_, h_ in enumerate(np.asarray(taps, np.complex64)):
            zt[:, k_:] += complex(h_) * z[:, :ntot - k_]
        z = zt
    rx = (Hm @ z) * ph                                         # [n_ant][ntot]
    sigma = float(np.sqrt(0.1875 ** 2 / (10 ** (snr_db / 10.0))))
    noise = torch.randn(n_ant, ntot, 2, generator=g, device=dev, dtype=torch.float32) * (sigma / np.sqrt(2))
    noisy = (torch.view_as_real(rx) + noise).contiguous()      # antenna 0 | antenna 1
    x = engine.DevArray(noisy.data_ptr(), n_ant * ntot, noisy)
    ev = np.zeros(n, abi.SIG_IN_DT)
    for a in range(n_ant):
        ev["raw_off"][:, a] = a * ntot + desc["iq_off"] + 184  # L-LTF start + 24 (signal_impl's sync item)
    ev["n_avail"] = desc["n_samp"] - 184 + GAP                 # up to the next packet (the output slot is sized by this)
    ev["snr"], ev["rssi"], ev["n_ant"] = snr_db, 1.0, n_ant
    e_sig = engine.Engine(device=local)
    e_sig.signal(ev, x)
    e_sig.sync()
    spk = e_sig.fetch(abi.BUF_SIG_PKT, abi.PKT_IN_DT)
    ok = spk["n_items"] > 0
    spk = spk[ok].copy()
    spk["seq"] = np.arange(1, len(spk) + 1)
    spk["mupos"] = mupos
    exp = [e for e, k in zip(exp, ok) if k]
    p_i, b_i = e_sig.buffer(abi.BUF_SIG_IQ)
    p_c, b_c = e_sig.buffer(abi.BUF_SIG_CHAN)
    e_tx.close()
    return spk, engine.DevArray(p_i, b_i // 8, e_sig), engine.DevArray(p_c, b_c // 8, e_sig), exp, (e_sig, noisy)


def cfg2_workload_on_device(n, snr_db, seed, torch, dev, local=0, mcs=7, seed_base=8021100):
    """BASELINE configs[1] (and, with mcs = 8 and its own seed base, configs[3]): see workload_on_device"""
    return workload_on_device("cfg2", n, snr_db, seed, torch, dev, local, mcs=mcs, seed_base=seed_base)


def raw_stream_unit(local=0, snr_db=30.0, lead=600, seed=1, mpdu_len=1458, mcs=7):
    """one cfg-2 packet (VHT MCS7, 1458-byte MPDU in an A-MPDU, c8p_tx without the toolbox's symbol-edge windowing) preceded by `lead`
    items of idle air and followed by the 480 a receiver needs behind it, AWGN at snr_db: the unit the raw-stream legs of bench.py
    repeat.  Host array (complex64) + the PDU the chain must deliver for every copy."""
    import zlib
    from . import engine
    rng = np.random.default_rng(8021100 + seed)
    body = bytes(rng.integers(0, 256, size=mpdu_len - 4, dtype=np.uint8))
    mpdu = body + zlib.crc32(body).to_bytes(4, "little")
    psdu = ampdu_vht(mpdu)
    pk = np.zeros(1, abi.TX_PKT_DT)
    pk["psdu_off"], pk["len"], pk["format"], pk["mcs"], pk["scr_seed"], pk["gap"], pk["nss"] = 0, len(psdu), abi.F_VHT, mcs, 93, 0, 1
    pk["mult"], pk["flags"] = 12.0, abi.TX_NO_WINDOW
    e = engine.Engine(device=local)
    e.tx(pk, np.frombuffer(psdu, np.uint8))
    e.sync()
    tx = e.fetch(abi.BUF_TX_IQ, np.complex64)
    e.close()
    r = np.random.default_rng(seed)
    rx = through_channel(tx[None, :], None, snr_db, r, lead=lead)[0]
    return rx, bytes([2, len(mpdu) % 256, len(mpdu) // 256]) + mpdu + bytes([mcs])
