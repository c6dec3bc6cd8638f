"""Builds the mixed-format parity batches from the committed fixtures (used by GPU tests and smoke)."""
import os

import numpy as np

from c8p_testlib import GOLD, H22, Batch, load_npz, through_channel


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


def sweep_batch(snr_db=None, seed=1, dewin=False):
    """All 42 sweep packets + the BASELINE configs + NDP + both MU users in one batch.
    Returns (Batch, meta list)."""
    rng = np.random.default_rng(seed)
    b, meta = Batch(), []
    for fn in ("packets_sweep.npz", "packets_cfg.npz"):
        for name, d in load_npz(fn).items():
            nss = int(d["nss"])
            tx = dewindow(d["tx"]) if dewin else d["tx"]
            if name == "VHT_NDP":
                rx = through_channel(tx, np.array([[1 + 0.2j, 0.5 - 0.4j]], np.complex64), snr_db, rng)
                assert b.add_rx(rx, n_ant=1) >= 0
                meta.append((name + "/siso", d))
                rx = through_channel(tx, H22, snr_db, rng)
                assert b.add_rx(rx, n_ant=2) >= 0
                meta.append((name + "/2ant", d))
                continue
            rx = through_channel(tx, H22 if nss == 2 else None, snr_db, rng)
            if b.add_rx(rx, n_ant=nss) >= 0:
                meta.append((name, d))
    mu = np.load(os.path.join(GOLD, "packets_mu.npz"))
    for u in range(2):
        rx = through_channel(mu["mu_tx"], mu["h"][u:u + 1], snr_db, rng)
        if b.add_rx(rx, n_ant=1, mupos=u) >= 0:
            meta.append(("MU_user%d" % u, {"pdu": mu["pdu%d" % u]}))
    return b, meta


def multipath_batch(snr_db=None, seed=3):
    """every single-user fixture packet through a 3-path channel (echoes at 2 and 5 samples, inside the 16-sample guard
    interval); the 2-stream ones through a 2x2 channel with different echoes on every antenna pair"""
    from c8p_testlib import through_multipath
    rng = np.random.default_rng(seed)
    b, meta = Batch(), []
    t1 = [[[(0, 1.0), (2, 0.35 * np.exp(0.7j)), (5, 0.2 * np.exp(-2.1j))]]]
    t2 = [[[(0, 1.0), (3, 0.3 * np.exp(1.1j))], [(0, 0.3j), (1, 0.25)]],
          [[(0, 0.2), (4, 0.2 * np.exp(2.0j))], [(0, 0.9), (2, 0.3 * np.exp(-0.4j)), (6, 0.15)]]]
    for fn in ("packets_sweep.npz", "packets_cfg.npz"):
        for name, d in load_npz(fn).items():
            nss = int(d["nss"])
            if name == "VHT_NDP":
                continue
            rx = through_multipath(dewindow(d["tx"]), t2 if nss == 2 else t1, snr_db, rng)
            if b.add_rx(rx, n_ant=nss) >= 0:
                meta.append((name, d))
    return b, meta


def cfg2_batch(n, snr_db=30.0, seed=8021200, dewin=True):
    """n VHT MCS7 1458-byte packets (BASELINE.json configs[1]): the 4 reference-TX fixture frames, each sent
    through a per-packet random-phase flat channel plus AWGN."""
    g = load_npz("packets_cfg.npz")
    frames = [g["cfg2_VHT_mcs7_1458B_%d" % k] for k in range(4)]
    txs = [dewindow(f["tx"]) if dewin else f["tx"] for f in frames]
    b, exp = Batch(), []
    for p in range(n):
        rng = np.random.default_rng(seed + p)
        th = rng.uniform(0, 2 * np.pi)
        h = np.array([[np.exp(1j * th)]], np.complex64)
        rx = through_channel(txs[p % 4], h, snr_db, rng)
        if b.add_rx(rx, n_ant=1) >= 0:
            exp.append(frames[p % 4]["pdu"].tobytes())
    return b, exp


def replicated_batch(frame, n, n_ant, snr_db=30.0, seed=0, H=None, mupos=0):
    """n noisy copies of one reference-TX fixture frame (de-windowed), each through its own random-phase channel."""
    tx = dewindow(frame["tx"])
    b = Batch()
    for p in range(n):
        rng = np.random.default_rng(seed + p)
        th = rng.uniform(0, 2 * np.pi)
        Hp = (np.eye(tx.shape[0], dtype=np.complex64) if H is None else np.asarray(H, np.complex64)) * np.exp(1j * th)
        rx = through_channel(tx, Hp.astype(np.complex64), snr_db, rng)
        assert b.add_rx(rx, n_ant=n_ant, mupos=mupos) >= 0
    return b


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
}


def workload_on_device(kind, n, snr_db, seed, torch, dev, local=0, mcs=None, seed_base=None):
    """A BASELINE.json config (WORKLOADS) as SURVEY.md §8d specifies it, generated on the box by the build's own transmitter:
    packet p carries its own MPDU (bytes from numpy.random.default_rng(seed_base + seed + p) + CRC-32; inside an A-MPDU for VHT),
    amplitude as tools/pktGenExample.py (multiplier 12, x sqrt(2) on two chains), clean OFDM symbols (C8P_TX_NO_WINDOW: the toolbox's
    half-amplitude symbol edges cost 19.6 dB of self-interference at the receiver's FFT window), a flat channel
    H e^{j theta_p} and AWGN at snr_db (sigma per tools/performance/perf_siso.py:92).  c8p_tx -> channel (torch, on the device) ->
    c8p_signal at the nominal sync item (L-LTF + 24, CFO 0) gives the packets at the signal -> demod boundary, all resident in HBM.
    Returns (descriptors [host], iq DevArray, chan DevArray, expected PDUs, keep-alive objects)."""
    import zlib
    from c8p_testlib import abi
    from gr_ieee80211_amd import engine, shard
    fmt, mcs0, nchain, n_ant, mlen, H, mupos, sb, _ = WORKLOADS[kind]
    mcs = mcs0 if mcs is None else mcs
    seed_base = sb if seed_base is None else seed_base
    GAP = 120
    pk = np.zeros(n, abi.TX_PKT_DT)
    mu = np.zeros(n if fmt == 3 else 0, abi.TX_MU_DT)
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
        if fmt_cfg < 0:                                    # random1 / random2: this packet's own format, rate and length
            fmt = int(rng.integers(0, 3)) if fmt_cfg == -1 else int(rng.integers(1, 3))
            mcs = int(rng.integers(0, 8 if fmt < 2 else 9))
            mlen = int(rng.integers(14, 3000))
        lens = (100, 60) if fmt == 3 else (mlen,)
        mpdus = []
        for ln in lens:
            body = bytes(rng.integers(0, 256, size=ln - 4, dtype=np.uint8))
            mpdus.append(body + zlib.crc32(body).to_bytes(4, "little"))
        ps = [ampdu_vht(m) if fmt >= 2 else m for m in mpdus]
        pk[p]["psdu_off"], pk[p]["len"], pk[p]["format"], pk[p]["mcs"] = put(ps[0]), len(ps[0]), fmt, mcs
        pk[p]["scr_seed"], pk[p]["gap"], pk[p]["nss"] = 93, GAP, nchain
        pk[p]["mult"] = 12.0 * (np.sqrt(2.0) if nchain == 2 else 1.0)
        pk[p]["flags"] = abi.TX_NO_WINDOW
        if fmt == 3:
            mu[p]["psdu_off"], mu[p]["len"], mu[p]["mcs"], mu[p]["group_id"], mu[p]["q_off"] = put(ps[1]), len(ps[1]), mcs, 2, 0
        m = mpdus[mupos]
        exp.append(bytes([min(fmt, 2), len(m) % 256, len(m) // 256]) + m + bytes([mcs + 8 if (fmt == 1 and nchain == 2) else mcs]))
    fmt = fmt_cfg
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
