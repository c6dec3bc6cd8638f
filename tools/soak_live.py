#!/usr/bin/env python3
"""tools/soak_live.py [streams] [packets per stream] [sessions] [seed offset] — the LIVE ingest of host/c8p_rawchain (push / round / drain; one and two
antennas per stream) against the oracle's detect -> signal (+ scan rule) -> demod + decode over each whole stream: random streams
(every fixture format, carrier offsets of +-150 kHz, 6-35 dB, gaps from 150 items), pushed in increments of random size per stream and
round; per stream the same kept packets (sync index, L-SIG fields), PDUs and FCS-failure counts, in order.  Sessions 2, 3 (mod 6) deliver
int16 I/Q pairs out of pinned memory (C8P_MEM_SC16), sessions 4, 5 int8 pairs (C8P_MEM_SC8, every radio scaled to its own peak); the
oracle then runs over their host-side conversion.  Run on the GPU box."""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
torch.cuda.init()                  # (before the engine library brings up its own view of the runtime: pinned sc16 buffers come from torch)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_legs import usable_cores  # noqa: E402
from c8p_testlib import oracle, oracle_rx  # noqa: E402
from gr_ieee80211_amd import blocks  # noqa: E402
from test_detect_frontend import scan_rule, sig_events  # noqa: E402
from test_multistream import random_stream, random_stream2, to_sc16  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 16
per = int(sys.argv[2]) if len(sys.argv) > 2 else 120
sessions = int(sys.argv[3]) if len(sys.argv) > 3 else 2
seed0 = int(sys.argv[4]) if len(sys.argv) > 4 else 0
o = oracle()
tot = {"sessions": 0, "streams": 0, "items": 0, "rounds": 0, "kept": 0, "pdus": 0, "fcs_fail": 0, "mismatch": 0}
t0 = time.time()


def chain(args):
    a0, a1 = args
    n = len(a0)
    arena = a0 if a1 is None else np.concatenate([a0, a1])
    ev = o.detect(a0)[0]
    s = sig_events(ev, n, n_ant=1 if a1 is None else 2)
    s["n_avail"] = np.minimum(s["n_avail"], 224 + 109280)
    if a1 is not None:
        s["raw_off"][:, 1] = n + s["raw_off"][:, 0]
    pk, iq, chan = o.signal_events(s, arena)
    kept, _ = scan_rule(s, pk)
    return pk[kept], [int(ev[i]["index"]) for i in kept], oracle_rx(pk[kept], iq, chan)


for sess in range(sessions):
    n_ant = 1 + sess % 2
    rng = np.random.default_rng(5000 + sess + 977 * seed0)
    xs = [(random_stream(7000 + 100 * sess + s + 100003 * seed0, per), None) if n_ant == 1 else random_stream2(7000 + 100 * sess + s + 100003 * seed0, per, snr=(6.0, 35.0)) for s in range(K)]
    bits = (0, 16, 8)[(sess // 2) % 3]
    sc16 = bits or False
    scale, qs = 0.0, None
    if bits == 8:
        xs = [tuple((a / np.float32(max(float(np.abs(x[0].view(np.float32)).max()), 1e-9))).astype(np.complex64) if a is not None else None for a in x) for x in xs]
    if sc16:
        scale = np.float32(1.02 * max(float(np.abs(a.view(np.float32)).max()) for x in xs for a in x if a is not None) / ((1 << (bits - 1)) - 1))
        conv = [[to_sc16(a, scale, bits) if a is not None else None for a in x] for x in xs]
        qs = [[torch.from_numpy(c[0].copy()).pin_memory() if c is not None else None for c in cs] for cs in conv]
        xs = [tuple(np.ascontiguousarray(c[1]) if c is not None else None for c in cs) for cs in conv]
        tot["sc16_sessions"] = tot.get("sc16_sessions", 0) + 1
    got = []
    h = blocks.RawChain(streams=K, stream_round_items=60000, stream_ant=n_ant, sc16_scale=float(scale), sc8_scale=float(scale))
    h.live(got.append)
    pos = [0] * K
    rounds = 0
    while any(p < len(x[0]) for p, x in zip(pos, xs)):
        for s, (a0, a1) in enumerate(xs):
            n = min(int(rng.choice([0, 0, 137, 1000, 2500, 9000, 20000, 59999])), len(a0) - pos[s])
            if n and sc16:
                p0 = qs[s][0].data_ptr() + (bits // 4) * pos[s]
                if n_ant == 1:
                    h.push(s, (p0, n), host="pinned", sc16=sc16)
                else:
                    h.push2(s, (p0, n), (qs[s][1].data_ptr() + (bits // 4) * pos[s], n), host="pinned", sc16=sc16)
                pos[s] += n
            elif n:
                if n_ant == 1:
                    h.push(s, a0[pos[s]:pos[s] + n])
                else:
                    h.push2(s, a0[pos[s]:pos[s] + n], a1[pos[s]:pos[s] + n])
                pos[s] += n
        h.round(wait=bool(rounds % 2))
        rounds += 1
    st = h.drain()
    h.close()
    per_s = [dict(idx=[], pk=[], outs=[], recs=[]) for _ in range(K)]
    for g in got:
        for i in range(len(g["pkts"])):
            P = per_s[int(g["pkt_stream"][i])]
            oo = g["outs"][i]
            P["idx"].append(int(g["pkt_index"][i])); P["pk"].append(g["pkts"][i]); P["outs"].append(oo)
            P["recs"].append(g["pdu_arena"][int(oo["pdu_off"]):int(oo["pdu_off"]) + int(oo["pdu_bytes"])].tobytes())
    with ThreadPoolExecutor(usable_cores()) as ex:
        refs = list(ex.map(chain, xs))
    for s, (pk, idx, ref) in enumerate(refs):
        P = per_s[s]
        ok = P["idx"] == idx
        if ok:
            for f in ("lsig_mcs", "lsig_len", "lsig_nsamp", "n_items", "n_ant"):
                ok = ok and [int(p[f]) for p in P["pk"]] == [int(v) for v in pk[f]]
        for r, oo, rec in zip(ref if ok else [], P["outs"], P["recs"]):
            want = b"".join(len(p).to_bytes(2, "little") + p for p in r["pdus"])
            if int(oo["n_pdu"]) != len(r["pdus"]) or int(oo["n_crc_fail"]) != int(r["out"]["n_crc_fail"]) or rec != want:
                ok = False
            tot["pdus"] += len(r["pdus"]); tot["fcs_fail"] += int(r["out"]["n_crc_fail"])
        tot["streams"] += 1; tot["items"] += len(xs[s][0]); tot["kept"] += len(idx); tot["mismatch"] += 0 if ok else 1
    tot["sessions"] += 1; tot["rounds"] += rounds
tot.setdefault("sc16_sessions", 0)
print("live-ingest soak: %(sessions)d sessions (one and two antennas alternating, %(sc16_sessions)d of them fed sc16 or sc8), %(streams)d streams, %(items)d items in %(rounds)d rounds, "
      "%(kept)d packets kept, %(pdus)d PDUs, %(fcs_fail)d FCS failures, %(mismatch)d streams that differ from the oracle's chain" % tot,
      "(%.0f s)" % (time.time() - t0))
sys.exit(1 if tot["mismatch"] else 0)
