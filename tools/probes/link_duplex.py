#!/usr/bin/env python3
"""tools/probes/link_duplex.py — what the link carries when the shader cores read pinned host memory (c8p_copy_gather, host -> device, the copy stream)
and write it (C8P_PIECE_TO_HOST pieces, the other copy stream) at the same time: 64 gathers of 1.4 MB down, 12 scatters of 8 MB up (the block chain's
plain mode per ~2300 packets), each direction alone and both together"""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
load_package()
import torch
from gr_ieee80211_amd import abi, engine

e = engine.Engine(device=0)
L = e.L
host_in = torch.empty(1 << 25, dtype=torch.float32).pin_memory(); host_in.normal_()       # 128 MB
host_out = torch.empty(1 << 25, dtype=torch.float32).pin_memory()
dev_in = torch.zeros(1 << 25, dtype=torch.float32, device="cuda:0")
dev_out = torch.randn(1 << 25, dtype=torch.float32, device="cuda:0")
torch.cuda.synchronize()
DOWN_F, N_DOWN = 350_000, 64          # floats per gather (1.4 MB), gathers
UP_F, N_UP = 2_000_000, 12            # floats per scatter (8 MB), scatters
TO_HOST = 1

UNORDERED = 2 if os.environ.get("UNORDERED") == "1" else 0
def down(i):
    off = (i * DOWN_F) % ((1 << 25) - DOWN_F)
    off -= off % 4
    pc = (abi.Piece * 1)(abi.Piece(host_in.data_ptr() + 4 * off, off, DOWN_F, 1, abi.FMT_F32, 1.0, UNORDERED))
    assert L.c8p_copy_gather(e.ctx, C.c_void_p(dev_in.data_ptr()), pc, 1) == 0

def up(i):
    off = (i * UP_F) % ((1 << 25) - UP_F)
    pc = (abi.Piece * 1)(abi.Piece(dev_out.data_ptr() + 4 * off, off, UP_F, 1, abi.FMT_F32, 1.0, TO_HOST))
    assert L.c8p_copy_gather(e.ctx, C.c_void_p(host_out.data_ptr()), pc, 1) == 0

def run(do_down, do_up):
    ts = []
    for rep in range(6):
        t0 = time.perf_counter()
        for i in range(max(N_DOWN if do_down else 0, N_UP if do_up else 0)):
            if do_down and i < N_DOWN: down(i)
            if do_up and i < N_UP: up(i)
        m = L.c8p_copy_mark(e.ctx)
        L.c8p_copy_wait_mark(e.ctx, m)
        ts.append(time.perf_counter() - t0)
    return min(ts[1:])

side = torch.cuda.Stream()
call_us = []
def up_dma(i):
    off = (i * UP_F) % ((1 << 25) - UP_F)
    t0 = time.perf_counter()
    with torch.cuda.stream(side):
        host_out[off:off + UP_F].copy_(dev_out[off:off + UP_F], non_blocking=True)
    call_us.append(1e6 * (time.perf_counter() - t0))

def run_dma(do_down):
    ts = []
    for rep in range(6):
        t0 = time.perf_counter()
        for i in range(max(N_DOWN if do_down else 0, N_UP)):
            if do_down and i < N_DOWN: down(i)
            if i < N_UP: up_dma(i)
        m = L.c8p_copy_mark(e.ctx)
        L.c8p_copy_wait_mark(e.ctx, m)
        side.synchronize()
        ts.append(time.perf_counter() - t0)
    return min(ts[1:])

gb_d, gb_u = 4e-9 * DOWN_F * N_DOWN, 4e-9 * UP_F * N_UP
td, tu, tb = run(True, False), run(False, True), run(True, True)
tud, tbd = run_dma(False), run_dma(True)
import numpy as np
print(json.dumps({"up_by_copy_engine_alone_GBps": round(gb_u / tud, 1), "down_by_shader_with_up_by_copy_engine_ms": round(1e3 * tbd, 3),
                  "sum_GBps": round((gb_d + gb_u) / tbd, 1), "memcpy_call_us_median_max": [round(float(np.median(call_us)), 1), round(max(call_us), 1)]}))
print(json.dumps({"unordered": UNORDERED != 0, "down_gather_us_apiece": round(1e6 * td / N_DOWN, 1), "copy_cus": os.environ.get("C8P_COPY_CUS"), "down_alone_GBps": round(gb_d / td, 1), "up_alone_GBps": round(gb_u / tu, 1), "both_ms": round(1e3 * tb, 3),
                  "both_down_plus_up_GBps": round((gb_d + gb_u) / tb, 1), "alone_sum_ms": round(1e3 * (td + tu), 3), "if_fully_overlapped_ms": round(1e3 * max(td, tu), 3)}))
e.close()
