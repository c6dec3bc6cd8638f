#!/usr/bin/env python3
"""tools/soak_random.py — randomized differential soak: batches of packets of random format (legacy / HT / VHT), rate, length (14-3000
bytes) and stream count (one stream; two streams on two antennas; a 2-user MU-MIMO packet with each user's rate and length random,
received at station 0 or 1), each batch at a random SNR (6-34 dB), with or without the toolbox's windowed symbol edges, through a flat or a
random 2-5 tap channel, decoded by c8p_rx as it ships (the un-tapped default) and compared with the oracle PACKET BY PACKET on all host
cores: status, the whole tag set, failure counts, PDU bytes — failing frames included.

    python3 tools/soak_random.py [batches=24] [packets_per_batch=2048] [seed=1] [split]

split: every batch ALSO goes through the two entry points the block mirrors call in plain mode — c8p_demod (LLR items: |LLR - oracle|
<= 1e-5 on every item, the largest difference is printed) and c8p_decode on those items (PDUs, decoded and pre-descramble bits equal
to the oracle's, failing frames included).

TEST INFRASTRUCTURE (it drives the engine and checks it with oracle/ through tests/c8p_testlib, as tools/soak.py does): not part of the product.
Prints one line per batch and a summary; exits non-zero at the first difference.  (tests/test_gpu_parity.py's soak covers the BASELINE
configs at fixed operating points; this one covers what lies between them.)"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
torch.zeros(1, device="cuda")
from concurrent.futures import ThreadPoolExecutor  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_legs import usable_cores  # noqa: E402
from c8p_smoke import close_or_both_nan  # noqa: E402
from c8p_testlib import abi, oracle  # noqa: E402
from gr_ieee80211_amd import engine  # noqa: E402
from gr_ieee80211_amd.workloads import workload_on_device  # noqa: E402


def to_host(dev_arr):
    torch.cuda.synchronize()
    a = np.zeros(dev_arr.n, np.complex64)
    ctypes.CDLL("libamdhip64.so").hipMemcpy(ctypes.c_void_p(a.ctypes.data), ctypes.c_void_p(dev_arr.ptr), ctypes.c_size_t(a.nbytes), ctypes.c_int(2))
    return a


def random_taps(rng):
    """2-5 taps inside the guard interval: a unit first tap and echoes of -4 ... -20 dB with random phases, unit total power"""
    n = int(rng.integers(2, 6))
    g = np.zeros(n, np.complex128)
    g[0] = 1.0
    for k in range(1, n):
        g[k] = 10 ** (-rng.uniform(4, 20) / 20) * np.exp(2j * np.pi * rng.uniform()) * (rng.uniform() < 0.8)
    return (g / np.sqrt((np.abs(g) ** 2).sum())).astype(np.complex64)


def main():
    batches = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    split = len(sys.argv) > 4 and sys.argv[4] == "split"
    llr_max, llr_n, bits_n = 0.0, 0, 0
    rng = np.random.default_rng(seed)
    o = oracle()
    cores = usable_cores()
    e = engine.Engine(device=0)
    e2 = engine.Engine(device=0, keep_bits=True) if split else None
    dev = torch.device("cuda", 0)
    tot = n_ok = n_fail = n_rej = 0
    t0 = time.time()
    for k in range(batches):
        u = rng.uniform()
        kind = "random2" if u < 0.3 else ("randommu0" if u < 0.36 else ("randommu1" if u < 0.42 else "random1"))
        snr = float(rng.uniform(6.0, 34.0))
        window = bool(rng.uniform() < 0.25)
        taps = random_taps(rng) if rng.uniform() < 0.5 else None
        pk, A, C_, exp, keep = workload_on_device(kind, n, snr, 7919 * seed + 104729 * (k + 1), torch, dev, 0, window=window, taps=taps)
        e.rx(pk, A, C_)
        e.sync()
        tags, outs, area = e.tags(), e.outs(), e.fetch(abi.BUF_PDU, np.uint8)
        iq, chan = to_host(A), to_host(C_)
        step = -(-len(pk) // cores)
        with ThreadPoolExecutor(cores) as ex:
            parts = list(ex.map(lambda a: o.rx_batch(pk[a:a + step], iq, chan), range(0, len(pk), step)))
        p = b_ok = b_fail = b_rej = 0
        for t_o, o_o, pdu_o, _ in parts:
            stride = len(pdu_o) // max(len(t_o), 1)
            for j in range(len(t_o)):
                where = (k, kind, round(snr, 1), window, p)
                assert tags[p]["status"] == t_o[j]["status"], where
                for key in abi.TAG_KEYS + ("nsym", "nss", "seq"):
                    assert tags[p][key] == t_o[j][key], where + (key,)
                if t_o[j]["status"] == abi.ST_OK:
                    for key in ("sssnr0", "sssnr1"):
                        assert close_or_both_nan(tags[p][key], t_o[j][key], 1e-4), where + (key, float(tags[p][key]), float(t_o[j][key]))
                    for key in ("cfo", "snr", "rssi"):
                        assert np.float32(tags[p][key]).tobytes() == np.float32(t_o[j][key]).tobytes(), where + (key,)
                else:
                    b_rej += 1
                assert outs[p]["n_pdu"] == o_o[j]["n_pdu"] and outs[p]["n_crc_fail"] == o_o[j]["n_crc_fail"], where
                nb = int(o_o[j]["pdu_bytes"])
                assert int(outs[p]["pdu_bytes"]) == nb, where
                g0 = int(outs[p]["pdu_off"])
                assert np.array_equal(area[g0:g0 + nb], pdu_o[j * stride:j * stride + nb]), where
                b_ok += int(o_o[j]["n_pdu"]); b_fail += int(o_o[j]["n_crc_fail"])
                p += 1
        assert p == len(pk)
        tot += len(pk); n_ok += b_ok; n_fail += b_fail; n_rej += b_rej
        if split:
            e2.demod(pk, A, C_)
            e2.sync()
            tg = e2.tags().copy()
            arena = e2.fetch(abi.BUF_LLR, np.float32)
            mu_all = e2.fetch(abi.BUF_MUCHAN, np.complex64) if e2.buffer(abi.BUF_MUCHAN)[1] else None

            def one(q):
                t, llr, mu = o.demod(pk, iq, chan, q)
                out, pdus, sb, ub = o.decode(t, llr, mu)
                return t, llr, pdus, sb, ub, out
            with ThreadPoolExecutor(cores) as ex:
                ref = list(ex.map(one, range(len(pk))))
            e2.decode(tg, arena, mu_all)
            dres = e2.results(want_llr=False, want_bits=True)
            for q, (t, llr, pdus, sb, ub, out) in enumerate(ref):
                where = (k, kind, round(snr, 1), window, q, "split")
                assert tg[q]["status"] == t["status"], where
                if t["status"] == abi.ST_OK and len(llr):
                    got = arena[int(tg[q]["llr_off"]):int(tg[q]["llr_off"]) + len(llr)]
                    d = float(np.abs(got - llr).max())
                    assert d <= 1e-5, where + (d,)
                    llr_max = max(llr_max, d); llr_n += len(llr)
                assert dres[q]["pdus"] == pdus and dres[q]["out"]["n_crc_fail"] == out["n_crc_fail"], where
                if t["status"] == abi.ST_OK and t["trellis"] > 0 and not t["is_ndp"]:
                    assert np.array_equal(dres[q]["sbits"], sb) and np.array_equal(dres[q]["bits"], ub), where
                    bits_n += len(ub)
        print("batch %2d  %-9s  %4.1f dB  %s  %s: %d packets, %d PDUs, %d FCS failures, %d not decodable — all equal to the oracle"
              % (k, kind, snr, "windowed" if window else "clean   ", "flat  " if taps is None else "%d taps" % len(taps), len(pk), b_ok, b_fail, b_rej), flush=True)
        del keep
    e.close()
    print("soak_random ok: %d packets, %d PDUs, %d FCS failures, %d rejected by SIG checks, every packet equal to the oracle (%d host threads, %.0f s)"
          % (tot, n_ok, n_fail, n_rej, cores, time.time() - t0))
    if split:
        print("split entry points: c8p_demod's %d LLR items within %.3g of the oracle's (bound 1e-5); c8p_decode on them: %d decoded bits, "
              "the pre-descramble bits and every PDU equal to the oracle's" % (llr_n, llr_max, bits_n))


if __name__ == "__main__":
    main()
