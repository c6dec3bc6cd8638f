"""tools/probes/exp_stream.py — what a host-side extra per sub-batch or per step costs the resident 65536-packet cfg-4 loop (the yardstick of
bench.py's sharded_stream leg): extra ctypes calls, result copies, 0.4 ms of host work per step, a sync per step.  Run on the GPU box.
profiles/r3b/exp_stream.txt: nothing but the per-step sync shows (-5 %); what cost the stream leg 8 % was a host-blocking copy inside
shard.gather_results (the packet counts moved to the device behind the 192 MB all_gather)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
torch.zeros(1, device="cuda")
from c8p_cases import cfg2_workload_on_device
from c8p_testlib import abi
from gr_ieee80211_amd import engine, shard
dev = torch.device("cuda", 0)
N = 65536
pk_all, iq_d, chan_d, exp, keep = cfg2_workload_on_device(N, 30.0, 0, torch, dev, 0, mcs=8, seed_base=8021400)
eng = engine.Engine(device=0)
A = engine.DevArray(iq_d.ptr, iq_d.n); C_ = engine.DevArray(chan_d.ptr, chan_d.n)
subs = shard.sub_batches(len(pk_all), 4096)
outs_res = torch.zeros(len(pk_all) * 32, dtype=torch.uint8, device=dev)
pdu_res = torch.zeros(300 << 20, dtype=torch.uint8, device=dev)
def loop(mode, steps=8):
    eng.sync(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(steps):
        if "extents" in mode:
            shard.StreamScatter._extents(pk_all)
        if "sleep" in mode:
            time.sleep(0.0004)
        base = 0
        for a, b in subs:
            eng.rx(pk_all[a:b], A, C_)
            if "calls" in mode:
                _, ob = eng.buffer(abi.BUF_OUT); _, pb = eng.buffer(abi.BUF_PDU); eng.ticket()
            if "fetch" in mode:
                _, ob = eng.buffer(abi.BUF_OUT); _, pb = eng.buffer(abi.BUF_PDU)
                eng.fetch_async(abi.BUF_OUT, outs_res.data_ptr() + a * 32, 0, ob)
                eng.fetch_async(abi.BUF_PDU, pdu_res.data_ptr() + base, 0, pb)
                base += pb
        if "sync" in mode:
            eng.sync()
    eng.sync()
    dt = time.perf_counter() - t
    print("%-28s %.3f M packets/s  %.3f ms/step" % (mode, N * steps / dt / 1e6, 1e3 * dt / steps), flush=True)
for rep in range(2):
    for mode in ("plain", "calls", "fetch", "extents", "sleep", "extents+calls+fetch", "sync", "plain"):
        loop(mode)
