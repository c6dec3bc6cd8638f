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


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    for d in (ROOT, os.path.join(ROOT, "tests")):
        if d not in sys.path:
            sys.path.insert(0, d)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b, _ = sweep_batch(snr_db=25.0, seed=9)
    pk, iq, chan = b.arrays()
    lo, hi = shard.partition(len(pk), world, rank)
    mine = shard.slice_batch(pk, lo, hi)
    res = oracle_rx(mine, iq, chan)
    got = shard.gather_pdus([r["pdus"] for r in res], lo)
    dist.barrier()
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)            # bench.py's elapsed-time reduction
    if rank == 0:
        q.put((got, float(t.item())))
    dist.destroy_process_group()


def test_partition_properties():
    for n in (0, 1, 7, 4096, 65536):
        for w in (1, 2, 3, 8):
            blocks = [shard.partition(n, w, r) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in blocks]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_sharded_rx_matches_single_process():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got, tmax = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert tmax == 2.0
    b, _ = sweep_batch(snr_db=25.0, seed=9)
    pk, iq, chan = b.arrays()
    ref = oracle_rx(pk, iq, chan)
    want = [(p, x) for p, r in enumerate(ref) for x in r["pdus"]]
    assert got == want and len(want) > 30
