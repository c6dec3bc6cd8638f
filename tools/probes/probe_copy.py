#!/usr/bin/env python3
"""tools/probes/probe_copy.py — c8p_copy_async from pinned host memory: cost of the call and rate, by copy size"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
import torch  # noqa: E402
from gr_ieee80211_amd import engine  # noqa: E402

e = engine.Engine(device=0)
L = e.L
total = 256 << 20
pin = torch.empty(total, dtype=torch.uint8).pin_memory()
dst = L.c8p_dev_alloc(e.ctx, total)
for chunk in (256 << 20, 32 << 20, 4 << 20, 1 << 20, 256 << 10):
    n = total // chunk
    for rep in range(3):
        t0 = time.perf_counter()
        for i in range(n):
            L.c8p_copy_async(e.ctx, C.c_void_p(dst + i * chunk), C.c_void_p(pin.data_ptr() + i * chunk), chunk)
        t1 = time.perf_counter()
        L.c8p_copy_wait(e.ctx)
        t2 = time.perf_counter()
    print("chunk %8d KB x %5d: enqueue %7.1f us per call, total %7.2f ms = %5.1f GB/s" % (chunk >> 10, n, 1e6 * (t1 - t0) / n, 1e3 * (t2 - t0), total / (t2 - t0) / 1e9))

from gr_ieee80211_amd import abi  # noqa: E402
for chunk in (256 << 20, 4 << 20, 1 << 20, 256 << 10, 64 << 10):
    n = total // chunk
    pcs = (abi.Piece * n)(*[abi.Piece(pin.data_ptr() + i * chunk, i * chunk // 8, chunk // 8, 1, 0) for i in range(n)])
    for rep in range(3):
        t0 = time.perf_counter()
        rc = L.c8p_copy_gather(e.ctx, C.c_void_p(dst), pcs, n)
        t1 = time.perf_counter()
        L.c8p_copy_wait(e.ctx)
        t2 = time.perf_counter()
    assert rc == 0, (rc, L.c8p_last_error(e.ctx))
    print("gather %8d KB x %5d pieces: call %7.1f us, total %7.2f ms = %5.1f GB/s" % (chunk >> 10, n, 1e6 * (t1 - t0), 1e3 * (t2 - t0), total / (t2 - t0) / 1e9))
dst2 = L.c8p_dev_alloc(e.ctx, 2 * total)                 # sc16: 4 B per item at the source, 8 B in the arena
for chunk in (256 << 20, 4 << 20, 1 << 20, 256 << 10):
    n = total // chunk
    pcs = (abi.Piece * n)(*[abi.Piece(pin.data_ptr() + i * chunk, i * chunk // 4, chunk // 4, 1, abi.FMT_SC16, 3.0e-5, 0) for i in range(n)])
    for rep in range(3):
        t0 = time.perf_counter()
        rc = L.c8p_copy_gather(e.ctx, C.c_void_p(dst2), pcs, n)
        t1 = time.perf_counter()
        L.c8p_copy_wait(e.ctx)
        t2 = time.perf_counter()
    assert rc == 0, (rc, L.c8p_last_error(e.ctx))
    print("gather sc16 %5d KB x %5d pieces: call %7.1f us, total %7.2f ms = %5.1f GB/s over the link" % (chunk >> 10, n, 1e6 * (t1 - t0), 1e3 * (t2 - t0), total / (t2 - t0) / 1e9))
for K, keep in ((64, 110000), (256, 110000), (1024, 110000)):           # the live carry: K spans of `keep` items between two arenas
    Cn = keep + 32768
    a0 = L.c8p_dev_alloc(e.ctx, 8 * K * Cn)
    a1 = L.c8p_dev_alloc(e.ctx, 8 * K * Cn)
    sp = (abi.Span * K)(*[abi.Span(s * Cn, s * Cn + 32768 - (s % 2), keep) for s in range(K)])
    for rep in range(3):
        t0 = time.perf_counter()
        L.c8p_copy_spans(e.ctx, C.c_void_p(a1), C.c_void_p(a0), sp, K)
        L.c8p_copy_wait(e.ctx)
        t2 = time.perf_counter()
    print("carry of %4d spans x %d items: %6.3f ms = %5.2f TB/s read + written" % (K, keep, 1e3 * (t2 - t0), 16.0 * K * keep / (t2 - t0) / 1e12))
    L.c8p_dev_free(e.ctx, C.c_void_p(a0)); L.c8p_dev_free(e.ctx, C.c_void_p(a1))
back = torch.empty(total, dtype=torch.uint8)
pin.copy_(torch.arange(total, dtype=torch.int64).to(torch.uint8))
n = 64
chunk = total // n
pcs = (abi.Piece * n)(*[abi.Piece(pin.data_ptr() + i * chunk + 8 * (i % 2), (i * chunk) // 8 + 3 * (i % 3), chunk // 8 - 7, 0, 0) for i in range(n)])
L.c8p_copy_gather(e.ctx, C.c_void_p(dst), pcs, n); L.c8p_copy_wait(e.ctx)
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemcpy(C.c_void_p(back.data_ptr()), C.c_void_p(dst), C.c_size_t(total), C.c_int(2))
ok = all(bool((back[((i * chunk) // 8 + 3 * (i % 3)) * 8:((i * chunk) // 8 + 3 * (i % 3)) * 8 + (chunk // 8 - 7) * 8] == pin[i * chunk + 8 * (i % 2):i * chunk + 8 * (i % 2) + (chunk // 8 - 7) * 8]).all()) for i in range(n))
print("gather with odd source / destination items and lengths: %s" % ("equal" if ok else "DIFFERENT"))
