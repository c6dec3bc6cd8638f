"""N>1 path on CPU: two gloo processes shard one arrival-ordered batch, each decodes its block (the oracle stands in
for the per-rank engine here — test only), rank 0 gathers the PDUs; the result must equal the single-process run."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from c8p_cases import sweep_batch
from c8p_testlib import ROOT, oracle_rx
from gr_ieee80211_amd import shard


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _arenas_from_oracle(res):
    """fixed-stride result arenas the way the engine lays them out (c8p_pkt_out records + a PDU arena), built from oracle
    results — the oracle stands in for the per-rank engine here (test only)"""
    from c8p_testlib import abi
    outs = np.zeros(len(res), abi.PKT_OUT_DT)
    chunks, off = [], 0
    for p, r in enumerate(res):
        area = b"".join(len(x).to_bytes(2, "little") + x for x in r["pdus"])
        outs[p]["n_pdu"], outs[p]["n_crc_fail"] = len(r["pdus"]), int(r["out"]["n_crc_fail"])
        outs[p]["pdu_off"], outs[p]["pdu_bytes"] = off, len(area)
        chunks.append(np.frombuffer(area, np.uint8))
        off += (len(area) + 15) & ~15
        chunks.append(np.zeros(off - sum(len(c) for c in chunks), np.uint8))
    pdu = np.concatenate(chunks) if chunks else np.zeros(0, np.uint8)
    return outs, pdu


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    for d in (ROOT, os.path.join(ROOT, "tests")):
        if d not in sys.path:
            sys.path.insert(0, d)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cpu = torch.device("cpu")
    n_total, pk_all, iq_all, chan_all = 0, None, None, None
    if rank == 0:                                           # the ingest rank holds the whole arrival-ordered stream
        b, _ = sweep_batch(snr_db=25.0, seed=9)
        pk_all, iq, chan = b.arrays()
        iq_all = torch.from_numpy(iq.view(np.float32).copy())
        chan_all = torch.from_numpy(chan.view(np.float32).copy())
        n_total = len(pk_all)
    nt = torch.tensor([n_total], dtype=torch.int64)
    dist.broadcast(nt, src=0)
    n_total = int(nt.item())
    pk, iq_l, ch_l = shard.scatter_stream(pk_all, iq_all, chan_all, n_total, cpu)
    lo, hi = shard.partition(n_total, world, rank)
    assert len(pk) == hi - lo
    res = oracle_rx(pk, iq_l.numpy().view(np.complex64), ch_l.numpy().view(np.complex64))
    outs, pdu = _arenas_from_oracle(res)
    counts, outs_all, pdu_all = shard.gather_results(torch.from_numpy(outs.view(np.uint8).copy()), torch.from_numpy(pdu.copy()), len(pk))
    got = shard.pdus_of(counts, outs_all, pdu_all, n_total)
    n_pdu, n_fail = shard.count_pdus(counts, outs_all)
    dist.barrier()
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)            # bench.py's elapsed-time reduction
    q.put((rank, got, float(t.item()), n_pdu, n_fail))
    dist.destroy_process_group()


def _worker_pipe(rank, world, port, q, sub):
    """the pipelined scatter (shard.StreamScatter): sub-batches posted up front, consumed in arrival order"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    for d in (ROOT, os.path.join(ROOT, "tests")):
        if d not in sys.path:
            sys.path.insert(0, d)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b, _ = sweep_batch(snr_db=25.0, seed=9)                # every rank builds the stream, ONLY to check what arrives against it
    pk_ref, iq_ref, chan_ref = b.arrays()
    n_total = len(pk_ref)
    pk_all = iq_all = chan_all = None
    if rank == 0:
        pk_all = pk_ref
        iq_all = torch.from_numpy(iq_ref.view(np.float32).copy())
        chan_all = torch.from_numpy(chan_ref.view(np.float32).copy())
    feed = shard.StreamScatter("cpu")
    got_all = []
    for step in range(2):                                   # two steps through the same object: buffers are reused
        pk = feed.begin(pk_all, iq_all, chan_all, n_total, sub)
        lo, hi = shard.partition(n_total, world, rank)
        assert len(pk) == hi - lo and feed.subs == shard.sub_batches(hi - lo, sub)
        if hi - lo > sub:
            assert len(feed.subs) >= 2 and feed.subs[-1][1] - feed.subs[-1][0] <= sub     # ragged last sub-batch
        res = []
        for k, (a, e) in enumerate(feed.subs):
            feed.arrived(k)
            # what has arrived is exactly this sub-batch of the stream, in arrival order, at the offsets its descriptors name
            iq_l, ch_l = feed.iq.numpy().view(np.complex64), feed.chan.numpy().view(np.complex64)
            for j in range(a, e):
                g, w = pk[j], pk_ref[lo + j]
                for f in ("n_items", "lsig_mcs", "lsig_len", "lsig_nsamp", "seq", "n_ant", "mupos"):
                    assert g[f] == w[f], (rank, j, f)
                for ant in range(int(w["n_ant"])):
                    x = iq_l[int(g["iq_off"][ant]):int(g["iq_off"][ant]) + int(g["n_items"])]
                    y = iq_ref[int(w["iq_off"][ant]):int(w["iq_off"][ant]) + int(w["n_items"])]
                    assert np.array_equal(x, y), (rank, j, ant)
                assert np.array_equal(ch_l[int(g["chan_off"]):int(g["chan_off"]) + 64], chan_ref[int(w["chan_off"]):int(w["chan_off"]) + 64])
            res += oracle_rx(pk[a:e], iq_l, ch_l)           # (the oracle stands in for c8p_rx on the sub-batch: test only)
        feed.finish()
        outs, pdu = _arenas_from_oracle(res)
        counts, outs_all, pdu_all = shard.gather_results(torch.from_numpy(outs.view(np.uint8).copy()), torch.from_numpy(pdu.copy()), len(pk))
        got_all.append(shard.pdus_of(counts, outs_all, pdu_all, n_total))
        dist.barrier()
    q.put((rank, got_all))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,sub", [(2, 7), (3, 5), (3, 64)])
def test_pipelined_scatter_delivers_sub_batches_in_arrival_order(world, sub):
    """shard.StreamScatter under gloo: exact-extent sub-batches (ragged: 53 packets in blocks of 26-27 or 17-18, sub-batches of 7 / 5;
    sub = 64: one sub-batch per rank), 1- and 2-antenna packets mixed; each rank checks every sub-batch against the stream the moment
    `arrived(k)` returns, decodes it, and the gathered PDU list equals the single-process run on both steps"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_pipe, args=(r, world, port, q, sub)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    b, _ = sweep_batch(snr_db=25.0, seed=9)
    pk, iq, chan = b.arrays()
    want = [(p, x) for p, r in enumerate(oracle_rx(pk, iq, chan)) for x in r["pdus"]]
    assert len(want) > 30
    for rank, got_all in outs:
        assert got_all == [want, want], rank


def test_partition_properties():
    for n in (0, 1, 7, 4096, 65536):
        for w in (1, 2, 3, 8):
            blocks = [shard.partition(n, w, r) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in blocks]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_stream_matches_single_process(world):
    """scatter_stream (contiguous blocks of a mixed 1- / 2-antenna stream from the ingest rank) -> per-rank RX -> gather_results
    (fixed-stride arenas, all-gathered): every rank ends up with the single-process PDU list in arrival order"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    b, _ = sweep_batch(snr_db=25.0, seed=9)
    pk, iq, chan = b.arrays()
    ref = oracle_rx(pk, iq, chan)
    want = [(p, x) for p, r in enumerate(ref) for x in r["pdus"]]
    assert len(want) > 30
    for rank, got, tmax, n_pdu, n_fail in outs:
        assert tmax == float(world)
        assert got == want
        assert n_pdu == len(want) and n_fail == sum(int(r["out"]["n_crc_fail"]) for r in ref)


# ------------------------------------------------------------------------------------------------ live streams over ranks
FAKE = os.path.join(ROOT, "tests", "fake_c8p", "libc8p_fake.so")


def _live_streams(K):
    from test_detect_frontend import CFOS, NAMES, stream
    rng = np.random.default_rng(5)
    xs = []
    for g in range(K):
        nm = NAMES[g % 3:g % 3 + 4 + g % 2]
        tail = 0.004 * (rng.normal(size=700) + 1j * rng.normal(size=700))
        xs.append(np.concatenate([stream(nm, CFOS[:len(nm)], snr=33.0, seed=300 + g)[0], tail]).astype(np.complex64))
    return xs


def _live_run(ls, xs, per_round):
    """every rank pushes ITS streams in increments; returns {global stream: [PDU bytes in order]}"""
    pos = {g: 0 for g in ls.mine()}
    while any(pos[g] < len(xs[g]) for g in pos):
        for g in pos:
            n = min(per_round + 37 * g, len(xs[g]) - pos[g])
            if n:
                ls.push(g, xs[g][pos[g]:pos[g] + n])
                pos[g] += n
        ls.round()
    return ls.drain()


def _worker_live(rank, world, port, q, K):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    for d in (ROOT, os.path.join(ROOT, "tests")):
        if d not in sys.path:
            sys.path.insert(0, d)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    xs = _live_streams(K)
    per = {}

    def on_round(g):
        from test_rawchain_cpu import pdus_of
        recs, k = pdus_of(g), 0
        for i in range(len(g["pkts"])):
            for _ in range(int(g["outs"]["n_pdu"][i])):
                per.setdefault(int(g["pkt_stream_global"][i]), []).append(recs[k]); k += 1
    ls = shard.LiveShard(K, 20000, on_round=on_round, engine_lib=FAKE)
    assert [ls.owner(g) for g in ls.mine()] == [rank] * len(ls.mine())
    with pytest.raises(ValueError):
        ls.push((ls.hi) % K if world > 1 else K, xs[0][:10])        # somebody else's stream
    _live_run(ls, xs, 6000)
    tot = ls.summary()
    ls.close()
    q.put((rank, per, tot))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,K", [(2, 5), (3, 4)])
def test_live_streams_sharded_over_ranks_equal_one_handle(world, K):
    """shard.LiveShard: K live streams over `world` ranks (contiguous blocks; every rank its own rawchain handle, here on the oracle-backed
    test engine), no collective in the data path: per GLOBAL stream the PDUs of a single-process handle that serves all K, and the
    all-reduced totals"""
    import subprocess
    from gr_ieee80211_amd import blocks
    from test_rawchain_cpu import pdus_of
    for d in ("oracle", os.path.join("tests", "fake_c8p"), os.path.join("gr-ieee80211_amd", "host")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, d)])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_live, args=(r, world, port, q, K)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    xs = _live_streams(K)
    ref = {}
    h = blocks.RawChain(engine_lib=FAKE, streams=K, stream_round_items=20000)

    def on_round(g):
        recs, k = pdus_of(g), 0
        for i in range(len(g["pkts"])):
            for _ in range(int(g["outs"]["n_pdu"][i])):
                ref.setdefault(int(g["pkt_stream"][i]), []).append(recs[k]); k += 1
    h.live(on_round)
    for g, x in enumerate(xs):
        h.push(g, x[:20000]); 
    h.round()
    for g, x in enumerate(xs):
        h.push(g, x[20000:])
    h.round()
    st = h.drain()
    h.close()
    merged = {}
    for rank, per, tot in outs:
        assert not set(per) & set(merged)
        merged.update(per)
        assert tot["pdus"] == st["pdus"] and tot["items"] == st["items"] == sum(len(x) for x in xs) and tot["events"] == st["events"]
    assert merged == ref and len(ref) == K and st["pdus"] >= 4 * K
