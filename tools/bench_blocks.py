#!/usr/bin/env python3
"""tools/bench_blocks.py — throughput of the drop-in BLOCK chain (demod -> decode mirrors of host/c8p_blocks.cc on the HIP
engine), fed the way the GNU Radio scheduler feeds blocks: tagged host item streams in bounded chunks, each block on its own
thread (thread-per-block).  Three figures per batch size:
  demod_block / decode_block   one block alone (its output is counted, not kept)
  chain                        demod -> decode through the float LLR stream, as rx.grc wires them: the stream crosses
                               PCIe twice (57 KB per packet down, 57 KB up)
  chain_fused                  batching_cfg::fused (what a demod -> decode flowgraph gets by default — fused_auto — or C8P_FUSED=1): demod runs c8p_rx, decode publishes behind the handle;
                               35 KB per packet up, 1.5 KB down
  rx_chain                     rx.grc's whole receive chain for K radios as ONE block (host/c8p_rx_chain.cc): raw 20 MS/s item streams in,
                               PDUs out; 8 B per item down, the PDUs up
Prints one JSON line per max_batch and a final {"best": ...} line.  Run on the GPU box."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from __graft_entry__ import load_package
load_package()
from gr_ieee80211_amd import blocks


PASSES = 5


def host_batch(n, local=0):
    """BASELINE configs[1] from the build's own generator (workloads.cfg2_workload_on_device), brought to the host: descriptors, the two
    arenas, the PDU every packet must yield"""
    import ctypes
    import torch
    from gr_ieee80211_amd.workloads import cfg2_workload_on_device
    dev = torch.device("cuda:%d" % local)
    pk, iq_dev, chan_dev, exp, keep = cfg2_workload_on_device(n, 30.0, 0, torch, dev, local)
    iq = np.zeros(iq_dev.n, np.complex64); chan = np.zeros(chan_dev.n, np.complex64)
    torch.cuda.synchronize(dev)
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy(ctypes.c_void_p(iq.ctypes.data), ctypes.c_void_p(iq_dev.ptr), ctypes.c_size_t(iq.nbytes), ctypes.c_int(2))
    hip.hipMemcpy(ctypes.c_void_p(chan.ctypes.data), ctypes.c_void_p(chan_dev.ptr), ctypes.c_size_t(chan.nbytes), ctypes.c_int(2))
    del keep
    return pk, iq, chan, exp


def rx_chain_rate(K=8, items_per_stream=1 << 22, per_round=1 << 19, local=0, bits=0):
    """the rx_chain mirror fed K packet-dense raw streams the way a scheduler feeds a block (bounded chunks, all ports advancing together)"""
    from gr_ieee80211_amd.workloads import raw_stream_unit
    unit, pdu = raw_stream_unit(local)
    reps = items_per_stream // len(unit)
    if bits:                                       # the ports carry int16 / int8 pairs, as a UHD source with cpu_format sc16 / sc8 emits them
        from gr_ieee80211_amd import abi
        top = (1 << (bits - 1)) - 1
        scale = np.float32(1.02 * float(np.abs(unit.view(np.float32)).max()) / top)
        unit = np.clip(np.rint(unit.view(np.float32) / scale), -top - 1, top).astype(np.int16 if bits == 16 else np.int8).reshape(-1, 2)
        b = blocks.Block(blocks.RX_CHAIN(K), max_batch=per_round, item_format=abi.FMT_SC16 if bits == 16 else abi.FMT_SC8, scale=float(scale))
    else:
        b = blocks.Block(blocks.RX_CHAIN(K), max_batch=per_round)
    ts, n_ok = [], 0
    for rep in range(3):
        for s in range(K):
            b.push_items(s, np.tile(np.roll(unit, 97 * s, axis=0), (reps, 1) if bits else reps))      # (rolled: the radios are not in step)
        t0 = time.perf_counter(); b.run(chunk=1 << 16, nout=1 << 16); ts.append(time.perf_counter() - t0)
    msgs = b.messages()
    n_ok = sum(1 for _, m in msgs if m == pdu)
    b.close()
    dt = float(np.median(ts[1:]))
    per_pass = len(msgs) / 3.0
    return {"streams": K, "bytes_per_item": (bits // 4) if bits else 8, "items_per_stream": reps * len(unit), "items_per_round": per_round, "packets_per_s": round(per_pass / dt),
            "x_realtime_each": round(reps * len(unit) / 20e6 / dt, 2), "pdus": len(msgs), "pdus_correct": n_ok,
            "pdus_in_the_streams": 3 * K * reps}


# What one general_work call is offered / may produce: (items of the demod input per call, LLR items of its output per call).
# "custom": the edge buffers come from the blocks' own buffer type (pinned host memory, c8p_block::buffer_alloc) and that type sizes
# them — 16 MB an edge (gnuradio/lib/c8p_pinned_buffer.h; a cfg-2 packet is 35 KB of samples and 57 KB of LLRs), of which a call sees
# half: 1 048 576 complex items in, 2 097 152 floats out.  "custom8": the sample edge at 8 MB, as the round's first profiles had it
# (the thread then waits for its gathers: fused 595-615 k against 686-697 k on one box).  "sched64k": the sizes rounds 1-4 measured
# with (512 KB / 1 MB a call), kept for continuity.
IO_SIZES = {"custom": (1 << 20, 1 << 21), "custom8": (1 << 19, 1 << 21), "custom32": (1 << 21, 1 << 22), "sched64k": (1 << 16, 1 << 18)}


def run(n=4096, batches=(256, 1024), quiet=False, batch=None, io="custom", pinned_io=None, pin="auto", defer_consume=None, passes=None):
    """every figure is the median of PASSES passes over the same block instances after a first untimed one (which grows the
    pinned staging and the device arenas, creates streams, pages the library in).  batch = (descriptors, iq, chan, expected PDUs) on
    the host; None: host_batch(n).  io: a key of IO_SIZES; pinned_io / defer_consume: batching_cfg's fields of those names (None: their defaults)"""
    pk, iq, chan, exp = batch if batch is not None else host_batch(n)
    pk, exp = pk[:n], exp[:n]
    PASSES = passes or globals()["PASSES"]
    chunk_in, nout_llr = IO_SIZES[io]
    # the chain's two scheduler threads on two cores that share an L3 (GNU Radio: the blocks' core affinity); pin=None: left to the OS
    if pin == "auto":
        from gr_ieee80211_amd import hostbind
        pin = hostbind.cpu_pair_sharing_l3()
    if pin:
        os.environ["C8P_CHAIN_PIN"] = "%d,%d" % tuple(pin)
    else:
        os.environ.pop("C8P_CHAIN_PIN", None)
    best = {}
    for max_batch in batches:
        dem = blocks.Block(blocks.DEMOD, max_batch=max_batch, fused=False, pinned_io=pinned_io, defer_consume=defer_consume)
        n_in = blocks.feed_demod(dem, pk, iq, chan)
        dem.run(chunk=chunk_in, nout=nout_llr)
        llr, tags = dem.output(), dem.tags()                 # the LLR stream + tags feed the decode-alone measurement
        dem.set_capture(False)
        ts = []
        for rep in range(1, 1 + PASSES):
            blocks.feed_demod(dem, pk, iq, chan, base=rep * n_in)
            t0 = time.perf_counter(); dem.run(chunk=chunk_in, nout=nout_llr); ts.append(time.perf_counter() - t0)
        t_dem = float(np.median(ts))
        dem.close()
        dec = blocks.Block(blocks.DECODE, max_batch=max_batch, fused=False, pinned_io=pinned_io, defer_consume=defer_consume)
        ts = []
        for rep in range(1 + PASSES):
            dec.push_items(0, llr)
            for off, key, val in tags:
                dec.push_tag(off + rep * len(llr), key, val)
            t0 = time.perf_counter(); dec.run(chunk=nout_llr, nout=1 << 16); ts.append(time.perf_counter() - t0)
        t_dec = float(np.median(ts[1:]))
        msgs = dec.messages()
        ok_dec = sum(1 for (ln, m), e in zip(msgs[-n:], exp) if m == e)
        dec.close()
        res = {"packets": n, "max_batch": max_batch, "io": io, "items_per_call": {"demod_in": chunk_in, "llr": nout_llr}, "demod_block_packets_per_s": round(n / t_dem),
               "decode_block_packets_per_s": round(n / t_dec), "pdus_ok_decode_alone": ok_dec}
        for fused in (False, True):
            dem = blocks.Block(blocks.DEMOD, max_batch=max_batch, fused=fused, pinned_io=pinned_io, defer_consume=defer_consume)
            dec = blocks.Block(blocks.DECODE, max_batch=max_batch, fused=fused, pinned_io=pinned_io, defer_consume=defer_consume)
            ts = []
            for rep in range(1 + PASSES):
                blocks.feed_demod(dem, pk, iq, chan, base=rep * n_in)
                t0 = time.perf_counter(); blocks.run_chain(dem, dec, chunk=chunk_in, nout=nout_llr); ts.append(time.perf_counter() - t0)
                if rep == 0:                                # (the first pass allocates the pinned staging: tens of ms)
                    dem.host_phase_reset(); dec.host_phase_reset()
            dt = float(np.median(ts[1:]))
            ok = sum(1 for (ln, m), e in zip(dec.messages()[-n:], exp) if m == e)
            key = "chain_fused" if fused else "chain"
            res[key + "_packets_per_s"] = round(n / dt); res["pdus_ok_" + key] = ok
            # host time inside the two blocks, microseconds per packet and phase (exclusive phases, the timed passes of this pair)
            tot = n * PASSES
            res[key + "_host_phase_us_per_packet"] = {"demod": {k: round(v / tot, 3) for k, v in dem.host_phase_us().items()},
                                                       "decode": {k: round(v / tot, 3) for k, v in dec.host_phase_us().items()},
                                                       "wall": round(1e6 * dt / n, 3)}
            dem.close(); dec.close()
        if "chain_default_mode" not in best:
            # what the pair does when nothing is said (fused=None, no C8P_FUSED): the demod front learns from the scheduler side that its only
            # reader is the decode mirror and runs fused (batching_cfg::fused_auto)
            dem = blocks.Block(blocks.DEMOD, max_batch=max_batch, pinned_io=pinned_io, defer_consume=defer_consume)
            dec = blocks.Block(blocks.DECODE, max_batch=max_batch, pinned_io=pinned_io, defer_consume=defer_consume)
            blocks.feed_demod(dem, pk, iq, chan)
            blocks.run_chain(dem, dec, chunk=chunk_in, nout=nout_llr)
            ok = sum(1 for (ln, m), e in zip(dec.messages()[-n:], exp) if m == e)
            best["chain_default_mode"] = {True: "fused", False: "plain", None: "undecided"}[dem.fused_state()] + (" (C8P_FUSED=%s)" % os.environ["C8P_FUSED"] if "C8P_FUSED" in os.environ else "")
            best["pdus_ok_chain_default"] = ok
            dem.close(); dec.close()
        if not quiet:
            print(json.dumps(res), flush=True)
        for k, v in res.items():
            if k.endswith("_packets_per_s"):
                if v > best.get(k, 0) and k.startswith("chain"):
                    best[k.replace("_packets_per_s", "_host_phase_us_per_packet")] = res[k.replace("_packets_per_s", "_host_phase_us_per_packet")]
                best[k] = max(best.get(k, 0), v)
        best["pdus_ok_min"] = min(best.get("pdus_ok_min", n), res["pdus_ok_chain"], res["pdus_ok_chain_fused"], ok_dec)
    best["packets"] = n
    best["chain_threads_on_cpus"] = list(pin) if pin else None
    os.environ.pop("C8P_CHAIN_PIN", None)
    best["io"] = io
    best["items_per_call"] = {"demod_in": chunk_in, "llr": nout_llr}
    return best


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    b = host_batch(n)
    print(json.dumps({"best": run(n, batch=b)}))
    print(json.dumps({"best_sched64k": run(n, batch=b, io="sched64k", quiet=True)}))
    print(json.dumps({"best_sched64k_pageable": run(n, batch=b, io="sched64k", quiet=True, pinned_io=False)}))
    print(json.dumps({"rx_chain": rx_chain_rate()}))
    print(json.dumps({"rx_chain_sc16": rx_chain_rate(bits=16)}))
