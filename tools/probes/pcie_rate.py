#!/usr/bin/env python3
"""tools/probes/pcie_rate.py — the PCIe-inclusive rate of the same workload as bench.py (never the bench `value`): IQ + channel
estimates start in HOST memory (pageable numpy arrays, then pinned torch tensors), c8p_rx copies them in, and the PDU and
per-packet result arenas are fetched back to the host after every step.  Run on the GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes
import numpy as np, torch
torch.cuda.init(); dev = torch.device("cuda", 0)
from c8p_cases import cfg2_workload_on_device
from c8p_testlib import abi
from gr_ieee80211_amd import engine

pk, iq_dev, chan_dev, expected, _keep = cfg2_workload_on_device(4096, 30.0, 0, torch, dev, 0)
hip = ctypes.CDLL("libamdhip64.so")
iq = np.zeros(iq_dev.n, np.complex64); chan = np.zeros(chan_dev.n, np.complex64)
torch.cuda.synchronize()
hip.hipMemcpy(ctypes.c_void_p(iq.ctypes.data), ctypes.c_void_p(iq_dev.ptr), ctypes.c_size_t(iq.nbytes), 2)
hip.hipMemcpy(ctypes.c_void_p(chan.ctypes.data), ctypes.c_void_p(chan_dev.ptr), ctypes.c_size_t(chan.nbytes), 2)
iq_pin = torch.from_numpy(iq.view(np.float32)).pin_memory(); ch_pin = torch.from_numpy(chan.view(np.float32)).pin_memory()
cases = {"pageable host arrays": (iq, chan),
         "pinned host arrays": (iq_pin.numpy().view(np.complex64), ch_pin.numpy().view(np.complex64))}
for name, (a, c) in cases.items():
    eng = engine.Engine(device=0)
    for _ in range(3):
        eng.rx(pk, a, c); eng.sync()
    steps = 12
    t = time.perf_counter()
    for _ in range(steps):
        eng.rx(pk, a, c)
        outs = eng.outs()                                  # blocks until this batch is done; per-packet results to the host
        pdu = eng.fetch(abi.BUF_PDU, np.uint8)             # PDU arena to the host
    dt = time.perf_counter() - t
    ok = int((outs["n_pdu"] == 1).sum())
    print({"input": name, "mode": "fetch after every submit", "packets_per_s": round(4096 * steps / dt), "ms_per_step": round(1e3 * dt / steps, 3),
           "MB_in_per_step": round((a.nbytes + c.nbytes) / 1e6, 1), "MB_out_per_step": round((pdu.nbytes + outs.nbytes) / 1e6, 1), "pdus": ok})
    # pipelined with tickets: submit batch i+1, then download batch i (c8p_select)
    t = time.perf_counter()
    prev = None
    for i in range(steps + 1):
        if i < steps:
            eng.rx(pk, a, c)
            cur = eng.ticket()
        if prev is not None:
            eng.select(prev)
            outs = eng.outs()
            pdu = eng.fetch(abi.BUF_PDU, np.uint8)
        prev = cur
    dt = time.perf_counter() - t
    print({"input": name, "mode": "submit i+1, then fetch i (tickets)", "packets_per_s": round(4096 * steps / dt),
           "ms_per_step": round(1e3 * dt / steps, 3), "pdus": int((outs["n_pdu"] == 1).sum())})
    eng.close()
