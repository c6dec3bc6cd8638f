#!/usr/bin/env python3
"""tools/probes/probe_contend.py — what a host-to-device bulk transfer in flight costs kernels that only touch HBM.

One stream keeps the link busy (c8p_copy_gather from pinned memory, by workgroup count; or c8p_copy_async = the DMA engine), another
runs a device-to-device copy (1 GiB read + 1 GiB write), a memset (1 GiB) and a small-kernel train (256 launches of 1 MB copies): each
timed alone and beside the transfer."""
import ctypes as C
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
import torch  # noqa: E402
from gr_ieee80211_amd import abi, engine  # noqa: E402

if len(sys.argv) == 1:
    for w in ("dma", "16", "64", "256", "1024"):
        env = dict(os.environ)
        if w != "dma":
            env["C8P_GATHER_WGS"] = w
        print(subprocess.run([sys.executable, __file__, w], env=env, capture_output=True, text=True).stdout, end="")
    sys.exit(0)

mode = sys.argv[1]
e = engine.Engine(device=0)
L = e.L
total = 256 << 20
pin = torch.empty(total, dtype=torch.uint8).pin_memory()
dst = L.c8p_dev_alloc(e.ctx, total)
n = 256
chunk = total // n
pcs = (abi.Piece * n)(*[abi.Piece(pin.data_ptr() + i * chunk, i * chunk // 8, chunk // 8, 1, 0) for i in range(n)])


def bulk(reps):
    for _ in range(reps):
        if mode == "dma":
            L.c8p_copy_async(e.ctx, C.c_void_p(dst), C.c_void_p(pin.data_ptr()), total)
        else:
            L.c8p_copy_gather(e.ctx, C.c_void_p(dst), pcs, n)


from gr_ieee80211_amd.workloads import cfg2_workload_on_device  # noqa: E402
dev = torch.device("cuda:0")
pk, iq_a, chan_a, _exp, _keep = cfg2_workload_on_device(4096, 30.0, 0, torch, dev)
rx_eng = engine.Engine(device=0)
a = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
b = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
small = [(a[i << 20:(i + 1) << 20], b[i << 20:(i + 1) << 20]) for i in range(256)]


def work(kind):
    if kind == "d2d":
        b.copy_(a)
    elif kind == "memset":
        b.zero_()
    elif kind == "c8p_rx":
        rx_eng.rx(pk, iq_a, chan_a)
        rx_eng.sync()
    else:
        for x, y in small:
            y.copy_(x)


def timed(kind, beside):
    if kind == "c8p_rx":                      # (its own streams: host clock around submit + sync)
        work(kind)
        L.c8p_copy_wait(e.ctx)
        if beside:
            bulk(12)
            time.sleep(0.01)
        t0 = time.perf_counter()
        for _ in range(8):
            work(kind)
        ms = 1e3 * (time.perf_counter() - t0) / 8
        L.c8p_copy_wait(e.ctx)
        return ms
    work(kind)
    torch.cuda.synchronize()
    L.c8p_copy_wait(e.ctx)
    if beside:
        bulk(12)                    # ~60 ms of link time queued
        time.sleep(0.01)            # the first transfer is under way
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(4):
        work(kind)
    ev1.record()
    ev1.synchronize()
    ms = ev0.elapsed_time(ev1) / 4
    L.c8p_copy_wait(e.ctx)
    return ms


t0 = time.perf_counter()
bulk(4)
L.c8p_copy_wait(e.ctx)
rate = 4 * total / (time.perf_counter() - t0) / 1e9
out = ["bulk = %-5s %5.1f GB/s" % (mode, rate)]
for kind in ("d2d", "memset", "train", "c8p_rx"):
    alone, with_bulk = timed(kind, False), timed(kind, True)
    out.append("%s %.3f -> %.3f ms (x%.2f)" % (kind, alone, with_bulk, with_bulk / alone))
print(" | ".join(out))
