#!/usr/bin/env python3
"""tools/bench_legs.py — every measurement of bench.py that is NOT the headline: the other BASELINE configs, the headline's
sensitivity to its input, the sharded stream (configs[3]), the host-fed legs (PCIe-inclusive c8p_rx, the block chain), the front end
(detect, TX, raw stream -> PDUs, live ingest), the extra CPU legs and the VALU view of the dominant kernel.

bench.py measures the headline, then calls run_single_gpu() / run_multi_gpu() here with a wall-clock budget; what they find goes to
bench_detail.json (and stderr), never into the one short line on stdout except as one number per leg (`legs`).

ONE estimator for every packets/s figure here: rate_windows() — the MEDIAN of >= 3 timed windows, min and max beside it."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0
WINDOWS = 3                              # timed windows per leg; the median counts


def rate_windows(step, sync, units_per_step, window_s=0.25, windows=WINDOWS, probe_steps=8):
    """THE estimator of every leg: `probe_steps` untimed-for-the-figure steps size a window of ~window_s, then `windows` windows of k
    steps each between two syncs; the figure is the MEDIAN window, min / max beside it.  step(i) submits step i, sync() joins."""
    t = time.perf_counter()
    for i in range(probe_steps):
        step(i)
    sync()
    k = int(max(probe_steps, window_s / max((time.perf_counter() - t) / probe_steps, 1e-6)))
    dts = []
    for _w in range(max(1, windows)):
        t = time.perf_counter()
        for i in range(k):
            step(i)
        sync()
        dts.append(time.perf_counter() - t)
    dt = float(np.median(dts))
    return {"per_s": units_per_step * k / dt, "ms_per_step": 1e3 * dt / k, "steps": k, "seconds": dt,
            "windows_per_s": {"median": units_per_step * k / dt, "min": units_per_step * k / max(dts), "max": units_per_step * k / min(dts), "n": len(dts)}}


def multi_gpu_host_fed(torch, dev, local, rank, world, pk, iq_dev, chan_dev, streams_per_rank=64, rounds=20):
    """One rank's share of the host-fed legs at world > 1, started together on all ranks (barriers): live ingest of
    streams_per_rank x world radios through shard.LiveShard — a stream belongs to one rank, which keeps its detection state; no
    collective in the data path — and the PCIe-inclusive c8p_rx of this rank's batch.  Returns this rank's figures."""
    import torch.distributed as dist
    from gr_ieee80211_amd import shard
    from gr_ieee80211_amd.workloads import raw_stream_unit
    res = {"rank": rank, "gpu": local}
    unit, _pdu = raw_stream_unit(local)
    dist.barrier()
    ls = live_streams(torch, dev, local, unit, streams=streams_per_rank, rounds=rounds, shard_world=(shard, streams_per_rank * world))
    res["live_streams"] = {k: ls[k] for k in ("streams", "ms_per_round", "x_realtime_each", "streams_at_realtime", "pcie_GBps", "packets_per_s", "pdus_ok",
                                              "pdus_in_the_items_pushed", "round_closed_to_pdus_ms") if k in ls}
    res["live_streams"]["global_streams"] = ls.get("global_streams")
    dist.barrier()
    det = {}
    res["pcie_inclusive_packets_per_s"] = pcie_inclusive(pk, iq_dev, chan_dev, torch, dev, local, detail=det)
    res["pcie_host_to_device_GBps"] = det.get("host_to_device_GBps")
    dist.barrier()
    return res


def summarize_host_fed(every):
    """rank 0: the ranks' figures side by side + their sums (weak scaling: every rank serves its own radios / batches)"""
    ok = [e for e in every if isinstance(e, dict) and "error" not in e]
    out = {"per_rank": every, "ranks_ok": len(ok)}
    if ok:
        out["live_streams_at_realtime_total"] = sum(e["live_streams"]["streams_at_realtime"] for e in ok)
        out["live_pcie_GBps_total"] = sum(e["live_streams"]["pcie_GBps"] for e in ok)
        out["pcie_inclusive_packets_per_s_total"] = sum(e["pcie_inclusive_packets_per_s"] for e in ok)
        v = [e["live_streams"]["streams_at_realtime"] for e in ok]
        out["live_streams_at_realtime_min_max"] = [min(v), max(v)]
    out["note"] = ("all ranks run these at the same time between barriers: shard.LiveShard (each rank the streams partition() gives it, its own "
                   "c8p_rawchain handle, rings of pinned memory allocated after hostbind.bind_rank) and pcie_inclusive (one c8p_copy_gather + "
                   "c8p_rx + result fetch per step); totals are sums over ranks, no collective in either data path")
    return out


class LegGuard:
    """Watchdog around an optional leg that runs AFTER the headline was measured: if the leg does not finish in `timeout_s` (or
    `fire` is called), rank 0 prints the headline line as it stands, with the leg marked as failed, and the process exits with
    status 0 — on every rank, each by its own timer, because a rank stuck in a collective cannot be told anything."""

    def __init__(self, out, rank, timeout_s, name, emit):
        import threading
        self.out, self.rank, self.name, self.timeout_s, self.emit = out, rank, name, timeout_s, emit
        self._ev = threading.Event()
        self._th = threading.Thread(target=self._watch, daemon=True)
        self._th.start()

    def _watch(self):
        if not self._ev.wait(self.timeout_s):
            self.fire("no result after %.0f s" % self.timeout_s)

    def fire(self, why):
        if self._ev.is_set():
            return
        self._ev.set()
        if self.rank == 0 and self.out is not None:
            self.out[self.name] = {"error": why, "note": "the headline (`value`) was measured before this leg started and is unaffected"}
            self.emit(self.out)
        else:
            time.sleep(5.0)                              # let rank 0 print first
        sys.stderr.write("bench: leg %s abandoned on rank %d: %s\n" % (self.name, self.rank, why))
        sys.stderr.flush()
        os._exit(0)

    def done(self):
        self._ev.set()


def sharded_stream(torch, dev, local, rank, world, use_dist, n_total, steps, warmup, sub=None, snr_db=30.0):
    """BASELINE configs[3]: rank 0 (the ingest GPU) holds ONE arrival-ordered stream of n_total VHT MCS8
    packets.  Per step, as a PIPELINE (shard.StreamScatter): rank 0 posts the whole scatter up front — descriptors, then every
    rank's k-th sub-batch as one grouped send/recv, exact extents straight out of the stream's arenas — and works on a zero-copy
    view of its own block; every rank submits c8p_rx on sub-batch k the moment it has arrived, while k+1 is in flight (the
    engine's slots keep several sub-batches on the GPU); the fixed-stride c8p_pkt_out + PDU arenas are gathered once at the end,
    device-resident.  Two feeds are timed:
      rccl_scatter  the pipeline above (RCCL over xGMI)
      direct_h2d    every rank already holds its block in pinned host memory and uploads it itself (c8p_rx with host arenas)
    `timeline_ms` = when, after the step's start, each stage was reached (max over ranks): stages overlap, so they do not add up."""
    import torch.distributed as dist
    from gr_ieee80211_amd import abi, engine, shard
    from gr_ieee80211_amd.workloads import cfg2_workload_on_device
    own_group = False
    if not dist.is_initialized():                     # plain `python bench.py`: a world of one, still through RCCL
        import socket
        s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
        own_group = True
    if sub is None:                                   # alone, the headline's batch size; sharded, four sub-batches per rank at 64k / 8
        sub = int(os.environ.get("C8P_STREAM_SUB", "4096" if world == 1 else "2048"))
    t_gen = time.time()
    pk_all = iq_all = chan_all = expected = keep = None
    if rank == 0:
        pk_all, iq_d, chan_d, expected, keep = cfg2_workload_on_device(n_total, snr_db, 0, torch, dev, local, mcs=8, seed_base=8021400)
        iq_all = shard.device_bytes(iq_d.ptr, 8 * iq_d.n, dev).view(torch.float32)
        chan_all = shard.device_bytes(chan_d.ptr, 8 * chan_d.n, dev).view(torch.float32)
        n_total = len(pk_all)
    nt = torch.tensor([n_total], dtype=torch.int64, device=dev)
    dist.broadcast(nt, src=0)
    n_total = int(nt.item())
    t_gen = time.time() - t_gen
    lo, hi = shard.partition(n_total, world, rank)
    n_local = hi - lo
    # (C8P_STREAM_SLOTS: experiments; 3, 4, 5, 6 slots measured the same within the noise, profiles/r3b)
    slots_env = os.environ.get("C8P_SLOTS")
    os.environ["C8P_SLOTS"] = os.environ.get("C8P_STREAM_SLOTS", "3")
    eng = engine.Engine(device=local)
    if slots_env is None:
        del os.environ["C8P_SLOTS"]
    else:
        os.environ["C8P_SLOTS"] = slots_env
    feed = shard.StreamScatter(dev)

    def sync_all():
        eng.sync(); torch.cuda.synchronize(dev); dist.barrier(); torch.cuda.synchronize(dev)

    # result arenas of the rank, in sets: the gather of step i reads one while the sub-batches of step i+1 fill the next
    # (three sets since round 4: a set is rewritten only after the gather that read it has completed — an event, see Collector.take —
    # and with three sets that gather was issued two steps earlier, so the wait costs nothing; with two it cost 6 % of the rate)
    N_SETS = 3
    outs_res = [torch.zeros(max(n_local, 1) * abi.PKT_OUT_DT.itemsize, dtype=torch.uint8, device=dev) for _ in range(N_SETS)]
    pdu_res = [None] * N_SETS
    res = {"packets": n_total, "sub_batch": sub, "world": world, "generation_s": round(t_gen, 1), "feeds": {}}

    class Collector:
        """every sub-batch's result arenas are copied (device to device, behind its kernels on its slot's stream) into one of the rank's
        two contiguous arena sets before the slot is reused"""

        def __init__(self, n, which, t0):
            self.bases, self.base, self.n, self.w, self.t0, self.tickets = [], 0, n, which, t0, []
            self.waited = False

        def take(self, a, b):
            # this set was last read by the all_gather of two steps ago, which runs on RCCL's stream: wait for it before the engine's
            # streams write the set again (the host-side slot-reuse sync orders the engine's own work only).  Here, at the first copy
            # into the set, not when the step starts: the scatter and the first sub-batch are under way by now
            if not self.waited:
                self.waited = True
                wait_works(gather_works[self.w])
            _, ob = eng.buffer(abi.BUF_OUT)
            _, pb = eng.buffer(abi.BUF_PDU)
            pr = pdu_res[self.w]
            if pr is None or pr.numel() < self.base + pb:
                grown = torch.zeros(max(2 * (pr.numel() if pr is not None else 0), (self.n // max(b - a, 1) + 2) * pb), dtype=torch.uint8, device=dev)
                torch.cuda.synchronize(dev); eng.sync()
                if pr is not None:
                    grown[:pr.numel()] = pr
                pdu_res[self.w] = pr = grown
                torch.cuda.synchronize(dev)
            eng.fetch_async(abi.BUF_OUT, outs_res[self.w].data_ptr() + a * abi.PKT_OUT_DT.itemsize, 0, ob)
            eng.fetch_async(abi.BUF_PDU, pr.data_ptr() + self.base, 0, pb)
            self.bases.append((a, b, self.base))
            self.tickets.append(eng.ticket())
            self.base += pb

        def close(self):
            """waits for THIS step's sub-batches only (the next step's may already be running), then gathers"""
            for t in self.tickets[-8:]:               # (a ticket whose slot has been taken again belongs to a finished submit: only that)
                try:
                    eng.wait(t)
                except RuntimeError as e:
                    if "not resident" not in str(e):
                        raise
            wait_works(state.get("last_works") or [])   # the previous step's gather is complete (NOT a device-wide sync: the next
                                                      # step's sub-batches are running on the engine's streams)
            rec = outs_res[self.w][:self.n * abi.PKT_OUT_DT.itemsize].view(torch.int64).view(self.n, 4)
            key = tuple(self.bases)                   # pdu_off was relative to the sub-batch's own arena: one add of a cached vector
            if key not in off_cache:
                v = np.zeros(self.n, np.int64)
                for a, b, off in self.bases:
                    v[a:b] = off
                off_cache.clear()
                off_cache[key] = torch.from_numpy(v).to(dev)
            if self.n:
                rec[:, 2] += off_cache[key]
            t3 = time.perf_counter()
            # the collectives are issued asynchronously and their Work handles kept: whoever rewrites this set, or reads the gathered
            # arenas on the host, polls them.  (An event of our own after the gather — on torch's default stream or on a side stream —
            # cost 5-6 % of this leg's rate: 0.92 instead of 0.98 of resident; the handles' own events are there anyway.)
            works = []
            g = shard.gather_results(outs_res[self.w][:n_local * abi.PKT_OUT_DT.itemsize], pdu_res[self.w][:self.base], n_local, reuse_sizes=True, works=works)
            gather_works[self.w] = works
            state["last_works"] = works
            return g, t3 - self.t0

    checked = False
    host_keep = None
    off_cache = {}
    gather_works = [[], [], []]

    def wait_works(ws):
        for w_ in ws:
            while not w_.is_completed():
                time.sleep(0)
    state = {"pending": None, "i": 0, "g": None}

    def finalize():
        """closes the step whose sub-batches were submitted last: its results gathered; returns (seconds after that step's start at
        which its last sub-batch was done, ... its gather was issued) or None"""
        nonlocal checked
        col = state["pending"]
        state["pending"] = None
        if col is None:
            return None
        (counts, outs_all, pdu_all), t_rx = col.close()
        if not checked:                               # once: the gathered arenas hold every transmitted MPDU, in arrival order
            torch.cuda.synchronize(dev)
            if rank == 0:
                got = shard.pdus_of(counts, outs_all, pdu_all, n_total)
                okc = sum(1 for p, b in got if b == expected[p])
                res["pdus_gathered"], res["pdus_equal_to_transmitted"] = len(got), okc
                assert okc >= 0.97 * n_total and okc == len(got), "sharded stream: %d PDUs, %d correct of %d packets" % (len(got), okc, n_total)
            checked = True
        state["g"] = (counts, outs_all, pdu_all)
        return t_rx, time.perf_counter() - col.t0

    def step(feed_name):
        """one step of the stream, software-pipelined with the previous one: this step's scatter is posted and its first sub-batches
        are submitted BEFORE the previous step's results are gathered, so the engine never drains between steps"""
        nonlocal host_keep
        t0 = time.perf_counter()
        col = Collector(n_local, state["i"] % N_SETS, t0)
        state["i"] += 1
        prev = None
        if feed_name == "rccl_scatter":
            pk = feed.begin(pk_all, iq_all, chan_all, n_total, sub)
            t1 = t2 = time.perf_counter()
            A = engine.DevArray(feed.iq.data_ptr(), feed.iq.numel() // 2)
            C_ = engine.DevArray(feed.chan.data_ptr(), feed.chan.numel() // 2)
            for k, (a, b) in enumerate(feed.subs):
                feed.arrived(k)                       # host waits for sub-batch k only; k+1 .. keep arriving behind it
                t2 = time.perf_counter()
                eng.rx(pk[a:b], A, C_)
                col.take(a, b)
                if k == min(2, len(feed.subs) - 1):
                    prev = finalize()
            feed.finish()
            state["pk"] = pk
        else:
            pk, iq_h, ch_h = host_keep
            t1 = t2 = time.perf_counter()
            subs = shard.sub_batches(len(pk), sub)
            for k, (a, b) in enumerate(subs):         # a host arena is uploaded whole: hand over only this sub-batch's items
                (i0, i1), (c0, c1) = shard.block_extents(pk, a, b)
                d = pk[a:b].copy()
                d["iq_off"][:, 0] -= i0; d["chan_off"] -= c0
                eng.rx(d, iq_h[i0:i1], ch_h[c0:c1])
                col.take(a, b)
                if k == min(2, len(subs) - 1):
                    prev = finalize()
        if prev is None:
            prev = finalize()                         # (a rank without packets in this step)
        state["pending"] = col
        return [t1 - t0, t2 - t0], prev

    for feed_name in ("rccl_scatter", "direct_h2d"):
        if feed_name == "direct_h2d":                 # the block in pinned host memory, for the second feed
            eng.sync()
            host_keep = (state["pk"], feed.iq.cpu().pin_memory().numpy().view(np.complex64), feed.chan.cpu().pin_memory().numpy().view(np.complex64))
        # warm-up: at least `warmup` steps AND 0.15 s of load — the shader clock needs some tens of ms to come back up after the idle
        # time of the set-up (2 GB of pinned copies before the second feed); a timed region of ~60 ms would otherwise measure the ramp
        t_w, n_w = time.perf_counter(), 0
        while n_w < max(warmup, 1) or time.perf_counter() - t_w < 0.15:
            step(feed_name)
            n_w += 1
        finalize()
        # `steps` steps back to back between two barriers, as the headline is timed (a barrier before every step would let the GPU
        # idle and its clocks drop between steps of ~10 ms)
        vs = []
        for _w in range(WINDOWS):                     # the median window counts (every rank sees the same max-reduced clocks: the same choice)
            sync_all()
            t0 = time.perf_counter()
            tls = [step(feed_name) for _ in range(steps)]
            fin = [t[1] for t in tls[1:]] + [finalize()]
            sync_all()
            tl = np.mean(np.array([list(a[0]) + list(f) for a, f in zip(tls, fin)]), axis=0)
            v = torch.tensor([time.perf_counter() - t0] + list(tl), dtype=torch.float64, device=dev)
            dist.all_reduce(v, op=dist.ReduceOp.MAX)
            vs.append(v.tolist())
        v = sorted(vs, key=lambda x: x[0])[len(vs) // 2]
        win = [n_total * steps / x[0] for x in vs]
        counts, outs_all, pdu_all = state["g"]
        n_pdu, n_fail = shard.count_pdus(counts, outs_all)
        last_ok = None
        if rank == 0:                                 # the LAST timed step's gathered arenas, not only the first step's: every PDU again
            got = shard.pdus_of(counts, outs_all, pdu_all, n_total)
            last_ok = sum(1 for p, b in got if b == expected[p])
            assert last_ok == len(got) and last_ok >= 0.97 * n_total, "sharded stream, last step of %s: %d PDUs, %d correct" % (feed_name, len(got), last_ok)
        res["feeds"][feed_name] = {"packets_per_s": n_total * steps / v[0], "ms_per_step": 1e3 * v[0] / steps, "steps": steps, "warmup": max(warmup, 1),
                                   "windows_packets_per_s": {"median": float(np.median(win)), "min": min(win), "max": max(win), "n": len(win)},
                                   "timeline_ms": {k: round(1e3 * x, 3) for k, x in zip(("descriptors", "last_sub_batch_submitted", "rx_done", "gather_issued"), v[1:])},
                                   "timeline_note": "ms after the step's start, mean over the steps, max over ranks; a step's results are collected "
                                                    "while the next step's first sub-batches run, so rx_done / gather_issued lie beyond ms_per_step",
                                   "pdus_last_step": n_pdu, "fcs_failures_last_step": n_fail, "pdus_equal_to_transmitted_last_step": last_ok,
                                   "gathered_bytes_per_rank": int(outs_all.shape[1] + pdu_all.shape[1])}
    # the same stream resident on ONE GPU, no scatter and no gather: what the pipeline is measured against at world 1
    if rank == 0:
        A = engine.DevArray(iq_d.ptr, iq_d.n); C_ = engine.DevArray(chan_d.ptr, chan_d.n)
        subs_all = shard.sub_batches(n_total, sub)
        def one_pass():
            for a, b in subs_all:
                eng.rx(pk_all[a:b], A, C_)
        one_pass()
        eng.sync()
        dts = []
        for _w in range(WINDOWS):
            t = time.perf_counter()
            for _ in range(max(steps, 2)):
                one_pass()
            eng.sync()
            dts.append(time.perf_counter() - t)
        res["resident_one_gpu_packets_per_s"] = n_total * max(steps, 2) / float(np.median(dts))
        res["rccl_scatter_over_resident"] = res["feeds"]["rccl_scatter"]["packets_per_s"] / res["resident_one_gpu_packets_per_s"]
    res["note"] = ("unmeasured at N > 1 until the driver has an 8-GPU node; at N = 1 the ingest rank's block is a zero-copy view of the "
                   "stream (nothing to send) and the all_gather a self-copy through RCCL") if world == 1 else "measured at world %d" % world
    eng.close()
    del keep
    if own_group:
        dist.destroy_process_group()
    return res


def out_of_time(deadline):
    return deadline is not None and time.perf_counter() > deadline


CONFIG_LEGS = [("cfg1", "cfg1", 4096), ("cfg3", "cfg3", 4096), ("cfg4_one_gpu_share", "cfg4", 8192), ("cfg5_station0", "cfg5u0", 1024),
               ("cfg5_station1", "cfg5u1", 1024), ("cfg3_1024_packets", "cfg3", 1024)]


def config_legs(torch, dev, local, snr_db=30.0, window_s=0.12, deadline=None, only=None):
    """The other BASELINE.json configs beside the cfg-2 headline (all through c8p_rx, inputs resident in HBM, generated on the box by
    c8p_tx as SURVEY.md §8d states them — gr-ieee80211_amd/workloads.py WORKLOADS): packets/s, ms per step, algorithmic bytes per packet
    (8 n_ant n_items + 512 + 32 + PDU bytes) and the HBM fraction a step reaches.  Every published PDU is compared with
    the transmitted MPDU before anything is timed."""
    from gr_ieee80211_amd import engine
    from gr_ieee80211_amd.workloads import WORKLOADS, workload_on_device
    eng = engine.Engine(device=local)
    res = {}
    for name, kind, n in CONFIG_LEGS:
        if only is not None and name not in only:
            continue
        if out_of_time(deadline):
            res[name] = {"skipped": "the run's leg budget was spent (bench.py --legs full runs every leg)"}
            continue
        try:
            pk, A, C_, exp, keep = workload_on_device(kind, n, snr_db, 0, torch, dev, local)
            eng.rx(pk, A, C_)
            r = eng.results(want_llr=False)
            n_ok = 0
            for g, e in zip(r, exp):
                assert g["pdus"] in ([], [e]), "%s: a published PDU differs from the transmitted MPDU" % name
                n_ok += g["pdus"] == [e]
            assert n_ok >= 0.97 * len(pk), "%s: only %d of %d packets delivered" % (name, n_ok, len(pk))
            b_alg = float(np.mean(8.0 * pk["n_ant"] * pk["n_items"] + 512 + 32) + np.mean([len(e) for e in exp]))
            t = time.perf_counter()
            while time.perf_counter() - t < 0.1:          # clock pre-roll (the PDU check above left the GPU idle)
                eng.rx(pk, A, C_)
            eng.sync()
            w = rate_windows(lambda i: eng.rx(pk, A, C_), eng.sync, len(pk), window_s)
            pps = w["per_s"]
            res[name] = {"workload": WORKLOADS[kind][8] + ", %d packets/step, AWGN %.0f dB" % (len(pk), snr_db), "packets_per_s": pps,
                         "windows_packets_per_s": w["windows_per_s"],
                         "ms_per_step": w["ms_per_step"], "steps": w["steps"], "packets_per_step": len(pk), "fcs_ok_fraction": n_ok / len(pk),
                         "alg_bytes_per_packet": b_alg, "achieved_GBps": pps * b_alg / 1e9, "frac_step": pps * b_alg / 1e9 / HBM_PEAK_GBS}
            del keep, A, C_
        except Exception as e:                            # (a leg must not cost the headline)
            res[name] = {"error": "%s: %s" % (type(e).__name__, e)}
    eng.close()
    return res


def sensitivity_legs(torch, dev, local, n_pkt, headline, window_s=0.12, deadline=None):
    """The headline's sensitivity to its input: the same 4096-packet cfg-2 batch size, resident in HBM, the same estimator
    (rate_windows), with what the headline's input leaves out — the decoder's fast path
    is data-dependent (integer-compare blocks need provably positive path metrics, LLRs outside the lean range go to the clamp-exact
    kernel): 21 dB (most frames fail the FCS), the toolbox's symbol-edge windowing (tools/phy80211header.py:894-899), a 3-tap
    frequency-selective channel, a batch of mixed formats / rates / lengths (uneven trellis lengths), and the headline batch WITH
    delivery: every step's c8p_pkt_out records and PDU arena fetched to pinned host memory inside the timed region."""
    from gr_ieee80211_amd import abi, engine
    from gr_ieee80211_amd.workloads import TAPS3, workload_on_device
    cases = [("with_delivery", "cfg2", dict(snr=30.0, deliver=True), "the headline batch, c8p_pkt_out + PDU arena fetched to pinned host memory every step"),
             ("snr_21dB", "cfg2", dict(snr=21.0), "cfg 2 at 21 dB AWGN"),
             ("mixed_formats", "random1", dict(snr=30.0), "random legacy / HT / VHT, MCS and length (14-3000 B) per packet, 30 dB"),
             ("windowed_tx", "cfg2", dict(snr=30.0, window=True), "cfg 2, TX with the toolbox's half-amplitude symbol edges (flags = 0), 30 dB"),
             ("selective_3tap", "cfg2", dict(snr=30.0, taps=TAPS3), "cfg 2 through a 3-tap channel (0 / -9 / -15 dB at 0 / 1 / 2 samples), 30 dB")]
    res = {}
    eng = engine.Engine(device=local)
    for name, kind, o, what in cases:
        if out_of_time(deadline):
            res[name] = {"skipped": "the run's leg budget was spent (bench.py --legs full runs every leg)"}
            continue
        try:
            pk, A, C_, exp, keep = workload_on_device(kind, n_pkt, o["snr"], 0, torch, dev, local, window=o.get("window", False), taps=o.get("taps"))
            eng.rx(pk, A, C_)
            eng.sync()
            r = eng.results(want_llr=False)
            rng_flags, flt_blocks, _ = eng.decstat()
            tr = eng.tags()["trellis"].astype(np.int64)
            lean = rng_flags == 0
            n_ok = sum(1 for g, e in zip(r, exp) if g["pdus"] == [e])
            for g, e in zip(r, exp):                    # whatever is published is the transmitted MPDU, at any SNR
                assert g["pdus"] in ([], [e]), "%s: a published PDU differs from the transmitted MPDU" % name
            deliver = o.get("deliver", False)
            bufs = None
            if deliver:                                 # three pinned sets: step i's copies land while step i+1 is submitted
                _, ob = eng.buffer(abi.BUF_OUT)
                _, pb = eng.buffer(abi.BUF_PDU)
                bufs = [(torch.empty(ob, dtype=torch.uint8).pin_memory(), torch.empty(pb, dtype=torch.uint8).pin_memory()) for _ in range(3)]
            last = [0]

            def step(i):
                eng.rx(pk, A, C_)
                if deliver:
                    o_, p_ = bufs[i % 3]
                    eng.fetch_async(abi.BUF_OUT, o_.data_ptr(), 0, o_.numel())
                    eng.fetch_async(abi.BUF_PDU, p_.data_ptr(), 0, p_.numel())
                    last[0] = i % 3
            t = time.perf_counter()
            i = 0
            while time.perf_counter() - t < 0.1:
                step(i); i += 1
            eng.sync()
            w = rate_windows(step, eng.sync, len(pk), window_s)
            pps, k, dt = w["per_s"], w["steps"], w["seconds"]
            res[name] = {"workload": what + ", %d packets/step" % len(pk), "packets_per_s": pps, "ms_per_step": w["ms_per_step"], "steps": k,
                         "windows_packets_per_s": w["windows_per_s"],
                         "vs_headline": pps / headline if headline else None, "fcs_ok_fraction": n_ok / max(len(pk), 1),
                         "exact_kernel_packets": int(rng_flags.sum()),
                         "float_step_block_fraction": float(flt_blocks[lean].sum()) / max(1, int((tr[lean] // 30).sum())),
                         "trellis_steps_per_packet_mean": float(tr.mean()),
                         "trellis_steps_per_s": float(tr.sum()) * k / dt,
                         "items_per_packet_mean": float(pk["n_items"].mean()),
                         "air_items_per_s": float(pk["n_items"].astype(np.int64).sum()) * k / dt}
            if deliver:
                got = np.frombuffer(bufs[last[0]][0].numpy().tobytes(), abi.PKT_OUT_DT)
                res[name]["delivered_bytes_per_step"] = int(bufs[0][0].numel() + bufs[0][1].numel())
                assert int((got["n_pdu"] == 1).sum()) == n_ok, "with_delivery: the fetched c8p_pkt_out records disagree with the check pass"
            del keep, A, C_
        except Exception as e:                            # (a leg must not cost the headline)
            res[name] = {"error": "%s: %s" % (type(e).__name__, e)}
    eng.close()
    res["note"] = ("same batch size and estimator as every leg (inputs resident, median of %d windows of ~%.2f s between two syncs; windows_packets_per_s has min / max); "
                   "vs_headline = packets_per_s / value.  mixed_formats is NOT the same unit of work per packet: random rate and length put ~4 x the air "
                   "time of a cfg-2 packet into the average packet (items_per_packet_mean; a 3000-byte BPSK frame is 1000 symbols to demodulate for "
                   "24 000 trellis steps, a cfg-2 frame 46 symbols for 11 960) — compare air_items_per_s (cfg 2: 4400 items per packet); the engine "
                   "decodes a batch of mixed lengths longest first" % (WINDOWS, window_s))
    return res


def lean_step_mix(T, float_frac=0.0, cr_num=5):
    """VALU wave-instructions of the lean forward pass per packet of T trellis steps at code rate 5/6, by opcode, from the kernel's
    structure (csrc/c8p_viterbi_lean.h, tools/gen_lean_asm.py): every step one packed candidate fma, steps with both LLRs present
    (1 in 5) a packed mul + add more.  Integer-compare blocks (path metrics provably positive): exchange phases q=0,1,3 a borrow
    (v_subb_co_u32_dpp) and v_max_u32_dpp, q=2 one mov_dpp more, q=4,5 permlane swap + v_max_u32 + v_sub_co_u32; one carry-shift per
    step: 147 per 30 steps.  Float blocks (the fraction `float_frac`, measured in this run): sub_dpp + max_dpp + class test: 167."""
    assert cr_num == 5
    per30_int = {"v_pk_fma_f32": 30, "v_pk_mul_f32": 6, "v_pk_add_f32": 6, "v_subb_co_u32_dpp": 20, "v_max_u32_dpp": 20,
                 "v_mov_b32_dpp": 5, "v_permlane16_swap_b32": 5, "v_permlane32_swap_b32": 5, "v_max_u32": 10,
                 "v_sub_co_u32": 10, "v_addc_co_u32": 30, "v_xor_b32": 1}
    per30_flt = {"v_pk_fma_f32": 30, "v_pk_mul_f32": 6, "v_pk_add_f32": 6, "v_sub_f32_dpp": 20, "v_max_f32_dpp": 20,
                 "v_mov_b32_dpp": 5, "v_permlane16_swap_b32": 5, "v_permlane32_swap_b32": 5, "v_max_f32": 10,
                 "v_cmp_class_f32": 20, "v_cmp_gt_f32": 10, "v_addc_co_u32": 30}
    mix = {}
    for d, w in ((per30_int, 1.0 - float_frac), (per30_flt, float_frac)):
        for k, v in d.items():
            mix[k] = mix.get(k, 0.0) + w * v * T / 30.0
    return mix


def valu_roofline(kern_x, n_pkt, float_frac, profile_tag, T=11960):
    """The resource that binds the dominant kernel: VALU issue.  The kernel is k_rx_1ss_lds = symbol demodulation + the lean forward
    pass (LLRs through LDS).  ceiling = [sum over the forward pass's opcodes of (count x measured issue cost, profiles/*/opcode_costs.json from
    tools/probe_issue.hip) + the kernel's other VALU instructions (SQ_INSTS_VALU of the newest committed PMC pass minus the forward
    pass's count) at the 4.15-cycle slot every non-trivial opcode costs] / (1024 SIMDs x 2.4 GHz); frac = ceiling / exclusive
    launch duration of this run.  Pricing at the nominal 2.4 GHz makes the fraction a lower bound (the shader clock under this
    load is 2.2-2.35 GHz).  One batch alone gives every SIMD 4 decoder waves; the arbiter serves the oldest wave first, so they
    finish one after the other and the last runs alone, latency-bound (profiles/r2b/decoder_wave_cycles.txt: 27 SIMD cycles per
    wave-step at 4 waves, 24 at 8 waves, against 21.7 from the opcode costs) — the exclusive launch cannot reach the ceiling, the
    pipelined step (other batches fill the tail) is what `value` measures."""
    pdir = os.path.join(ROOT, "profiles", profile_tag)
    f_cost, f_pmc = os.path.join(pdir, "opcode_costs.json"), os.path.join(pdir, "pmc_hbm.json")
    if not os.path.exists(f_cost) or kern_x.get("viterbi_fwd", 0) <= 0:
        return None
    cost = json.load(open(f_cost))["lean_step_opcodes"]
    mix = lean_step_mix(T, float_frac)
    dec_instr = sum(mix.values()) * n_pkt
    cycles = sum(mix[k] * cost[k] for k in mix) * n_pkt            # SIMD-cycles summed over the chip
    other, other_src, busy = 0.0, None, None
    if os.path.exists(f_pmc):
        k = json.load(open(f_pmc))["kernels"].get("k_rx_1ss_lds")
        if k and k.get("sq_insts_valu"):
            other, other_src = max(0.0, k["sq_insts_valu"] * n_pkt / 4096.0 - dec_instr), os.path.relpath(f_pmc, ROOT)
            if k.get("valu_busy"):
                busy = {"valu_busy": k["valu_busy"], "sq_active_inst_valu_quad_cycles": k.get("sq_active_inst_valu"), "grbm_gui_active_cycles_8xcd": k.get("grbm_gui_active"),
                        "formula": "SQ_ACTIVE_INST_VALU x 4 / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs), averages per launch of the PMC passes (C8P_SLOTS=1)",
                        "source": os.path.relpath(f_pmc, ROOT) + " (= " + os.path.relpath(os.path.join(pdir, "pmc_summary.txt"), ROOT) + ")"}
    cycles += other * 4.15
    n_simd, clk = 256 * 4, 2.4e9
    ceiling_ms = cycles / n_simd / clk * 1e3
    dur_ms = kern_x["viterbi_fwd"]
    return {"bound": "valu issue", "kernel": "viterbi_fwd (k_rx_1ss_lds: symbol demodulation + lean forward pass, LLRs through LDS)", "unit": "ms per launch",
            "ceiling_ms": ceiling_ms, "achieved_ms": dur_ms, "frac": ceiling_ms / dur_ms,
            "forward_pass_instr_per_launch": dec_instr, "forward_pass_instr_per_step": sum(mix.values()) / T,
            "forward_pass_issue_cycles_per_step": sum(mix[k] * cost[k] for k in mix) / T, "float_step_block_fraction": float_frac,
            "other_instr_per_launch": other, "other_instr_source": other_src,
            "achieved_G_wave_instr_per_s": (dec_instr + other) / (dur_ms * 1e-3) / 1e9,
            "counters": busy,
            "source": os.path.relpath(f_cost, ROOT) + " (measured cycles per opcode, tools/probe_issue.hip on the box that took the profile) x lean_step_mix(); nominal 2.4 GHz"}


def residency_3slots(profile_tag):
    """the dominant kernel as the bench runs it (three slots): profiles/<tag>/residency_3slots.json from the kernel trace"""
    f = os.path.join(ROOT, "profiles", profile_tag, "residency_3slots.json")
    if not os.path.exists(f):
        return None
    r = json.load(open(f))
    r["file"] = os.path.relpath(f, ROOT)
    return r


def pmc_traffic(dom, profile_tag):
    """HBM bytes per launch of the dominant kernel, and of the three RX kernels of a c8p_rx step together, from the newest committed
    rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in separate runs, profiles/*/pmc_hbm.json).  FETCH_SIZE is doubled
    per MI355X_MICROARCH.md (gfx950 tallies a 128-byte request as 64 bytes; every kernel here reads with wide vector loads).
    Profiles from before the LLR stream moved into LDS hold k_rx_1ss instead of k_rx_1ss_lds: its scalar-cache reads of the LLRs were
    64-byte requests and counted in full, so that kernel = FETCH_SIZE as reported + half of the sample bytes.  The whole step also
    counts the second, nearly empty launch of k_rx_1ss (only_flagged) and k_viterbi_fwd.  See DESIGN.md §7."""
    names = {"viterbi_fwd": "k_rx_1ss_lds", "finish": "k_finish", "classify": "k_classify"}
    best = os.path.join(ROOT, "profiles", profile_tag, "pmc_hbm.json")
    if not os.path.exists(best):
        return None, None, None
    j = json.load(open(best))

    old_profile = "k_rx_1ss_lds" not in j["kernels"]
    if old_profile:
        names["viterbi_fwd"] = "k_rx_1ss"

    def hbm(name):
        k = j["kernels"].get(name)
        if not k:
            return None
        if "hbm_bytes" in k:
            return k["hbm_bytes"]
        if name == "k_rx_1ss" and old_profile:     # scalar-cache LLR reads (counted in full) + the samples (tallied at half)
            return (k["fetch_size_kb_raw"] + k["write_size_kb"]) * 1024.0 + 0.5 * 4096 * (8 * 4400 + 512)
        return (k["fetch_size_kb_raw"] * 2.0 + k["write_size_kb"]) * 1024.0

    per = {n: hbm(n) for n in names.values()}
    whole = sum(v for v in per.values() if v) if all(per.values()) else None
    if whole is not None and not old_profile:
        whole += sum(hbm(n) or 0.0 for n in ("k_rx_1ss", "k_viterbi_fwd"))
    return per[names[dom]], os.path.relpath(best, ROOT), whole


def pcie_inclusive(pk, iq_dev, chan_dev, torch, dev, local, steps=12, detail=None, with_dma_path=True):
    """PCIe-inclusive rate of the same batch: IQ + chan start in PINNED HOST memory every step and the per-packet results and the
    PDU arena end there.  The arenas go down with ONE c8p_copy_gather per step (the shader cores read the pinned memory over PCIe: no
    DMA engine, 54-57 GB/s against the ~42 the copy engines reach on this part) into one of two device arena sets, c8p_copy_fence
    orders the step's c8p_rx behind it on the device, and the step's c8p_pkt_out records + PDU arena are fetched into one of three
    pinned result sets behind its kernels (c8p_fetch_async); the host waits for step i - 1 after it has submitted step i.
    Round 4's leg (hipMemcpyAsync of the arenas inside c8p_rx, a synchronous fetch per step) is kept as `dma_path` in `detail`."""
    import ctypes
    from gr_ieee80211_amd import abi, engine
    hip = ctypes.CDLL("libamdhip64.so")
    iq_pin = torch.empty(2 * iq_dev.n, dtype=torch.float32).pin_memory()
    ch_pin = torch.empty(2 * chan_dev.n, dtype=torch.float32).pin_memory()
    torch.cuda.synchronize(dev)
    hip.hipMemcpy(ctypes.c_void_p(iq_pin.data_ptr()), ctypes.c_void_p(iq_dev.ptr), ctypes.c_size_t(8 * iq_dev.n), ctypes.c_int(2))
    hip.hipMemcpy(ctypes.c_void_p(ch_pin.data_ptr()), ctypes.c_void_p(chan_dev.ptr), ctypes.c_size_t(8 * chan_dev.n), ctypes.c_int(2))
    a, c = iq_pin.numpy().view(np.complex64), ch_pin.numpy().view(np.complex64)

    def dma_path():
        eng = engine.Engine(device=local)
        for _ in range(3):
            eng.rx(pk, a, c); eng.sync()
        t = time.perf_counter()
        prev = cur = None
        outs = None
        for i in range(steps + 1):
            if i < steps:
                eng.rx(pk, a, c)
                cur = eng.ticket()
            if prev is not None:
                eng.select(prev)
                outs = eng.outs()
                eng.fetch(abi.BUF_PDU, np.uint8)
            prev = cur
        dt = time.perf_counter() - t
        eng.close()
        assert int((outs["n_pdu"] == 1).sum()) >= 0.99 * len(pk)
        return len(pk) * steps / dt

    def gather_path():
        # the copy stream of this leg on 2 CUs of every XCD (C8P_COPY_CUS = 16, read when the ctx first copies; the engine's default for
        # complex64 pieces is 32, tuned on the live ingest): one 150 MB piece keeps few waves busy and every CU it does not hold stays with
        # the decoder — 16 / 32 / 64 / no mask: 1.49 / 1.44 / 1.41 / 1.20 M packets/s on one box (profiles/r5n/pcie_copy_cus.txt)
        cus_env = os.environ.get("C8P_COPY_CUS")
        if cus_env is None:
            os.environ["C8P_COPY_CUS"] = "16"
        try:
            return gather_path_()
        finally:
            if cus_env is None:
                del os.environ["C8P_COPY_CUS"]

    windows = {}

    def gather_path_():
        eng = engine.Engine(device=local)
        L = eng.L
        # one arena per set: [iq | chan] (chan starts on a 16-byte boundary: iq_dev.n is even or padded)
        n_iq = iq_dev.n + (iq_dev.n & 1)
        sets = [torch.zeros(2 * (n_iq + chan_dev.n), dtype=torch.float32, device=dev) for _ in range(2)]
        torch.cuda.synchronize(dev)
        pcs = (abi.Piece * 2)(abi.Piece(iq_pin.data_ptr(), 0, iq_dev.n, 1, abi.FMT_C32, 0.0, 0),
                              abi.Piece(ch_pin.data_ptr(), n_iq, chan_dev.n, 1, abi.FMT_C32, 0.0, 0))
        res_sets = None
        tickets = [None] * 3

        def submit(i):
            nonlocal res_sets
            s_ = sets[i % 2]
            eng._chk(L.c8p_copy_gather(eng.ctx, ctypes.c_void_p(s_.data_ptr()), pcs, 2))
            eng._chk(L.c8p_copy_fence(eng.ctx))
            eng.rx(pk, engine.DevArray(s_.data_ptr(), iq_dev.n), engine.DevArray(s_.data_ptr() + 8 * n_iq, chan_dev.n))
            if res_sets is None:
                _, ob = eng.buffer(abi.BUF_OUT)
                _, pb = eng.buffer(abi.BUF_PDU)
                res_sets = [(torch.empty(ob, dtype=torch.uint8).pin_memory(), torch.empty(pb, dtype=torch.uint8).pin_memory()) for _ in range(3)]
            o_, p_ = res_sets[i % 3]
            eng.fetch_async(abi.BUF_OUT, o_.data_ptr(), 0, o_.numel())
            eng.fetch_async(abi.BUF_PDU, p_.data_ptr(), 0, p_.numel())
            tickets[i % 3] = eng.ticket()
        for i in range(4):
            submit(i)
        eng.sync()
        dts = []
        for _ in range(WINDOWS):                           # windows of `steps` (~35 ms each): the median counts, as in every leg
            t = time.perf_counter()
            for i in range(steps + 1):
                if i < steps:
                    submit(i)                              # (set i % 2 was last read by step i - 2, waited for below one step ago)
                if i > 0:
                    eng.wait(tickets[(i - 1) % 3])         # step i - 1: kernels done, its results are in pinned host memory
            dts.append(time.perf_counter() - t)
        dt = float(np.median(dts))
        windows.update({"median": len(pk) * steps / dt, "min": len(pk) * steps / max(dts), "max": len(pk) * steps / min(dts), "n": len(dts)})
        outs = np.frombuffer(res_sets[(steps - 1) % 3][0].numpy().tobytes(), abi.PKT_OUT_DT)
        n_ok = int((outs["n_pdu"] == 1).sum())
        bytes_down = 8 * (iq_dev.n + chan_dev.n)
        bytes_up = int(res_sets[0][0].numel() + res_sets[0][1].numel())
        eng.close()
        assert n_ok >= 0.99 * len(pk), "pcie_inclusive: %d of %d packets delivered" % (n_ok, len(pk))
        return len(pk) * steps / dt, bytes_down, bytes_up
    rate, down, up = gather_path()
    if detail is not None:
        detail["gather_path_packets_per_s"] = rate
        detail["windows_packets_per_s"] = windows
        detail["copy_stream_cus"] = os.environ.get("C8P_COPY_CUS", "16 (this leg's choice; engine default 32)")
        detail["host_to_device_GBps"] = rate / len(pk) * down / 1e9
        detail["device_to_host_GBps"] = rate / len(pk) * up / 1e9
        detail["bytes_per_step"] = {"down": down, "up": up}
        if with_dma_path:
            try:
                detail["dma_path_packets_per_s"] = dma_path()
            except Exception as e:
                detail["dma_path_packets_per_s"] = "%s: %s" % (type(e).__name__, e)
    return rate


def frontend_detect(torch, dev, local, n_target=32 << 20, full=True):
    """SURVEY §8f-2 / f-1 stages, reported beside the headline metric: c8p_detect (presiso.grc + trigger + sync) over a raw
    20 MS/s stream resident in HBM — the dense kernel is k_detect_scan, an HBM-bound scan: 8 B read + 1 B (trigger byte)
    written per sample, the autocorrelation stream stays on the chip — the whole chain from that raw stream to PDUs
    (c8p_detect -> c8p_signal -> c8p_rx) as a pipeline, and the live ingest of many independent streams."""
    from gr_ieee80211_amd import engine
    from gr_ieee80211_amd.workloads import raw_stream_unit
    unit, _pdu = raw_stream_unit(local)                                                  # one cfg-2 packet + idle time, 30 dB
    reps = max(1, n_target // len(unit))
    x = torch.from_numpy(unit.view(np.float32)).to(dev).repeat(reps)
    n = x.numel() // 2
    eng = engine.Engine(device=local, profile=True)
    arr = engine.DevArray(x.data_ptr(), n, x)
    ev = eng.detect(arr)
    eng.profile_reset()
    K, dts = 3, []
    for _w in range(WINDOWS):                     # (c8p_detect returns the events: every call is synchronous)
        t0 = time.perf_counter()
        for _ in range(K):
            ev = eng.detect(arr)
        dts.append((time.perf_counter() - t0) / K)
    dt = float(np.median(dts))
    kt = {k: (ms / cnt if cnt else 0.0) for k, (ms, cnt) in eng.kernel_times().items()}
    eng.close()
    pre_s = kt["presiso"] * 1e-3
    tx = tx_rate(engine, local)
    chain = raw_stream_chain(torch, dev, local, arr, n, reps, unit, full=full)
    return {"tx_packets_per_s": tx, "samples": n, "packets_in_stream": reps, "sync_events": int(len(ev)), "samples_per_s": n / dt,
            "x_realtime_20MSps": n / dt / 20e6, "ms_per_call": 1e3 * dt,
            "windows_samples_per_s": {"median": n / dt, "min": n / max(dts), "max": n / min(dts), "n": len(dts)},
            "kernel_ms": {"scan": round(kt["presiso"], 4), "candidates": round(kt["trigger"], 4), "syncevt": round(kt["syncevt"], 4)},
            "scan_roofline": {"bound": "hbm", "kernel": "k_detect_scan", "alg_bytes_per_sample": 9, "achieved": 9.0 * n / pre_s / 1e9 if pre_s > 0 else 0.0,
                              "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": (9.0 * n / pre_s / 1e9) / HBM_PEAK_GBS if pre_s > 0 else 0.0},
            "raw_stream_to_pdus": chain, "_unit": unit}


def chain_report(st, tl, packets_per_segment):
    """what a run of the raw chain says about itself: the steady-state rate (everything but the pipeline's fill), the cost of the first
    segment beside it, every stage's host clocks, the per-stage durations from the segment marks, and the counters that would explain
    a slow run (arena growths inside the run, look-back give-ups)"""
    def ms(v):
        return round(1e3 * float(v), 4)
    stage = {}
    if tl:
        det = [m["det_end"] - m["det_begin"] for m in tl]
        sig = [m["sig_end"] - m["sig_begin"] for m in tl]
        rxd = [m["rx_done"] - m["rx_submit"] for m in tl]
        lat = [m["rx_done"] - m["det_begin"] for m in tl]
        for k, v in (("detect", det), ("signal", sig), ("rx_submit_to_results", rxd), ("segment_latency", lat)):
            stage[k] = {"median_ms": ms(np.median(v)), "max_ms": ms(np.max(v)), "first_ms": ms(v[0])}
    ss = max(st["steady_s"], 1e-9)
    return {"segments": st["segments"], "events": st["events"], "packets": st["packets"], "pdus_ok": st["pdus"], "packets_per_segment": packets_per_segment,
            "ms_per_segment": 1e3 * ss / max(st["steady_segments"], 1), "packets_per_s": st["steady_pdus"] / ss, "samples_per_s": st["steady_items"] / ss,
            "x_realtime_20MSps": st["steady_items"] / ss / 20e6, "steady_segments": st["steady_segments"],
            "first_segment_ms": ms(st["first_segment_s"]), "whole_run_ms_per_segment": 1e3 * st["seconds"] / max(st["segments"], 1),
            "whole_run_packets_per_s": st["pdus"] / max(st["seconds"], 1e-9),
            "host_thread_s": {k: {f: round(v, 5) for f, v in st[k].items()} for k in ("detect", "signal", "rx")},
            "stage_ms": stage, "arena_growths_in_run": st["arena_growths"], "arena_GB": round(st["arena_bytes"] / 1e9, 3),
            "lookback_giveups": st["det_giveups"], "ticket_scans": st["det_ticket_scans"]}


def raw_stream_chain(torch, dev, local, arr, n, reps, unit, segments=24, full=True):
    """the chain from the RAW stream (rx.grc from the SDR source on) as a pipeline over `segments` passes over one resident
    33.5 M-item segment: host/c8p_rawchain.cc — segment i+2 in c8p_detect while i+1 is in c8p_signal (consume: the block's scan
    rule inside the engine, the kept packets back to back) and i in c8p_rx; each stage has its own engine context (own HIP
    streams) and host thread.  ONE handle for everything: its arenas are reserved at create, the untimed warm-up runs through the
    same contexts as the timed run, and the run reports where its time went (chain_report).  Every PDU is counted from the
    per-packet results fetched back."""
    from gr_ieee80211_amd import blocks
    unit_len = len(unit)
    out = {"samples_per_segment": n,
           "note": "c8p_detect -> c8p_signal (consume) -> c8p_rx through one host/c8p_rawchain handle (three host threads, four engine contexts "
                   "that outlive the runs, arenas reserved at create), everything resident in HBM, no state carried between segments; "
                   "packets_per_s / ms_per_segment = the steady state (the last steady_segments segments), first_segment_ms and the "
                   "whole-run figures beside it; compare `value` (the same packets handed over at the demod boundary)"}
    h = None
    try:
        # (this stream raises two sync flags per packet: the second falls inside the packet and the scan rule skips it)
        h = blocks.RawChain(device=local, reserve_seg_items=n, reserve_seg_events=2 * reps + reps // 4 + 64, reserve_seg_packets=reps + 64,
                            reserve_pkt_items=unit_len, reserve_host=True)
        segs = [(arr.ptr, n)] * segments
        warm = h.run(segs[:4])                                                # untimed, the same handle: streams, clocks
        runs = []
        for _r in range(WINDOWS):                                             # runs of 24 segments (~35 ms each): the median run counts, as in every leg
            st = h.run(segs)
            runs.append((st, h.timeline()))
        rate = [r_[0]["steady_pdus"] / max(r_[0]["steady_s"], 1e-9) for r_ in runs]
        st, tl = runs[int(np.argsort(rate)[len(rate) // 2])]
        out.update(chain_report(st, tl, reps))
        out["windows_packets_per_s"] = {"median": float(np.median(rate)), "min": min(rate), "max": max(rate), "n": len(rate)}
        out["warmup"] = {"segments": warm["segments"], "arena_growths": warm["arena_growths"], "first_segment_ms": round(1e3 * warm["first_segment_s"], 3)}
    except Exception as e:
        out["error"] = "%s: %s" % (type(e).__name__, e)
        if h is not None:
            h.close()
        return out
    # what a host-fed caller sees: the same segment in pinned host memory, crossing PCIe once per pass (inside c8p_detect)
    pin = None
    if not full:                                    # the short run: the resident chain above + the 64-radio live ingest
        h.close()
        try:
            out["live_streams"] = live_streams(torch, dev, local, unit, rounds=24)
        except Exception as e:
            out["live_streams"] = {"error": "%s: %s" % (type(e).__name__, e)}
        return out
    try:
        pin = torch.empty(2 * n, dtype=torch.float32).pin_memory()
        pin.copy_(arr.keepalive)
        torch.cuda.synchronize(dev)
        # 40 segments: with 12 the steady window is 8 completions long and they come in bunches — 55-70 GB/s "over PCIe" from run to run
        # (tools/probes/rawchain_pcie.py); 36 completions give 47-49 GB/s every time, and the whole run with its fill 45-47
        hs = [(pin.data_ptr(), n)] * 40
        h.run(hs[:3], host=True)
        t_run = time.perf_counter()
        sp = h.run(hs, host=True)
        t_run = time.perf_counter() - t_run
        pc = chain_report(sp, h.timeline(), reps)
        pc["GB_per_s_over_pcie"] = 8.0 * pc["samples_per_s"] / 1e9
        pc["whole_run"] = {"segments": len(hs), "ms": round(1e3 * t_run, 2), "GB_per_s_over_pcie": round(8.0 * n * len(hs) / t_run / 1e9, 2),
                           "packets_per_s": round(sp["pdus"] / t_run)}
        out["pcie_inclusive"] = pc
    except Exception as e:
        out["pcie_inclusive"] = {"error": "%s: %s" % (type(e).__name__, e)}
    h.close()
    try:
        out["live_streams"] = live_streams(torch, dev, local, unit)
        keys = ("streams", "items_per_stream_per_round", "rounds", "ms_per_round", "x_realtime_each", "streams_at_realtime", "pcie_GBps",
                "bytes_per_item_over_pcie", "packets_per_s", "pdus_ok", "pdus_in_the_items_pushed", "round_closed_to_pdus_ms",
                "arena_growths_in_session", "lookback_giveups")
        # the same link, four times as many radios, a quarter of the increment: what the per-round costs leave of the rate
        more = live_streams(torch, dev, local, unit, streams=256, round_items=1 << 17)
        out["live_streams"]["streams_256"] = {k: more[k] for k in keys}
        # the radios' own wire formats: int16 I/Q pairs (rx.grc's UHD source, otw sc16) converted on the device instead of on the host;
        # UHD's narrower one, int8 pairs: 2 B per item — the link no longer binds alone
        for name, K, bits in (("sc16", 64, 16), ("sc16_128", 128, 16), ("sc8_128", 128, 8)):
            more = live_streams(torch, dev, local, unit, streams=K, bits=bits)
            out["live_streams"][name] = {k: more[k] for k in keys}
        # at the radios' own pace: 256 streams x 6.5 ms rounds of complex64 = 85-90 % of what the link carries; how long after a round
        # is closed its PDUs are out when the pipeline is not kept full on purpose
        more = live_streams(torch, dev, local, unit, streams=256, round_items=1 << 17, rounds=80, pace=1.0)
        out["live_streams"]["paced_256"] = {k: more[k] for k in keys + ("paced_at", "behind_ms")}
    except Exception as e:
        out["live_streams"] = {"error": "%s: %s" % (type(e).__name__, e)}
    return out


def live_streams(torch, dev, local, unit, streams=64, round_items=1 << 19, rounds=40, bits=0, pace=0.0, shard_world=None):
    """north_star's unit of parallelism: many independent 20 MHz streams on one GPU.  `streams` radios deliver round_items items each
    per round (26 ms of air at 20 MS/s) out of pinned host memory through the live ingest of host/c8p_rawchain: per round ONE
    c8p_detect_multi over all streams, one c8p_signal (scan rule per stream), one c8p_rx over the packets of all streams; detection
    state, commit point and scan position per stream live in the handle.  PCIe is in the loop.  x_realtime_each = air time of a round
    / wall time of a round: how many times faster than real time EVERY stream is served; streams x that = 20 MS/s streams one GPU
    sustains at this packet density.
    Every stream has its OWN ring of pinned memory (as a driver's receive ring: two rounds long, the packet-dense capture `unit`
    repeated, read from a different phase per stream, wrapping): no two streams share a source byte, a ring comes round again only
    after all the others have been read (>= 67 MB later), so every byte crosses the link every round.  (Until late in round 4 the
    streams read overlapping stretches of ONE capture: with 2-byte items two streams' windows lay 1 MB apart in the gather kernel's
    walk and the L2 served part of the reads — 146 GB/s "over PCIe".)
    bits = 16 / 8: the rings hold int16 / int8 I/Q pairs as the radio sends them (4 / 2 B per item over PCIe, C8P_MEM_SC16 / _SC8):
    the gather kernel converts.
    pace > 0: the rounds are closed at `pace` times the radios' own pace (1.0 = a round every round_items / 20 MS/s seconds), not as
    fast as they go: what round_closed_to_pdus_ms is when the machine is not saturated, and whether it keeps up (behind_ms)."""
    from gr_ieee80211_amd import blocks
    unit_len = len(unit)
    item_bytes = bits // 4 if bits else 8
    scale = 0.0
    if bits:
        top = (1 << (bits - 1)) - 1
        scale = np.float32(1.02 * float(np.abs(unit.view(np.float32)).max()) / top)
        unit_q = np.clip(np.rint(unit.view(np.float32) / scale), -top - 1, top).astype(np.int16 if bits == 16 else np.int8)
    else:
        unit_q = unit.view(np.float32)
    m = -(-2 * round_items // unit_len)                 # units per ring: two rounds, rounded up
    W = m * unit_len                                    # items per ring
    ring = torch.empty((streams, 2 * W), dtype={0: torch.float32, 16: torch.int16, 8: torch.int8}[bits]).pin_memory()
    ring.numpy()[:] = np.tile(unit_q, m)[None, :]
    per_round_pk = streams * (round_items // unit_len + 2)
    kw = dict(reserve_seg_events=2 * per_round_pk + per_round_pk // 4, reserve_seg_packets=per_round_pk, reserve_pkt_items=unit_len,
              sc16_scale=float(scale), sc8_scale=float(scale))
    lshard = None
    if shard_world is not None:     # (shard module, streams of the whole job): this rank's `streams` of them through shard.LiveShard
        lshard = shard_world[0].LiveShard(shard_world[1], round_items, device=local, **kw)
        assert lshard.hi - lshard.lo == streams, "LiveShard gave this rank %d streams, the leg feeds %d" % (lshard.hi - lshard.lo, streams)
        h = lshard.h
    else:
        h = blocks.RawChain(device=local, streams=streams, stream_round_items=round_items, **kw)
        h.live(None)                # (the per-packet results are fetched by the rx stage and counted; no Python in the pipeline's threads)
    base = np.uint64(ring.data_ptr()) + np.arange(streams, dtype=np.uint64) * np.uint64(W * item_bytes)
    pos0 = (np.arange(streams, dtype=np.uint64) * np.uint64(7919 * 131)) % np.uint64(W)      # every radio at its own point of the capture
    sid = np.arange(streams, dtype=np.uint32)

    def one_round(r):
        p = (pos0 + np.uint64(r * round_items)) % np.uint64(W)
        n1 = np.minimum(np.uint64(round_items), np.uint64(W) - p)             # up to the end of the ring ...
        n2 = np.uint64(round_items) - n1                                      # ... and the rest from its start
        wrap = n2 > 0
        h.pushv(np.concatenate([sid, sid[wrap]]), np.concatenate([base + p * np.uint64(item_bytes), base[wrap]]), np.concatenate([n1, n2[wrap]]),
                host="pinned", sc16=bits or False)                            # one call for all streams' increments
        h.round(wait=False)             # the rings stay where they are: no need to wait for the copies, the next round's follow at once
    for r in range(3):                                                           # untimed: the same handle
        one_round(r)
    t0 = time.perf_counter()
    behind = 0.0
    for r in range(3, 3 + rounds):
        if pace > 0:
            due = t0 + (r - 3) * (round_items / 20e6) / pace
            while time.perf_counter() < due:
                pass
            behind = max(behind, time.perf_counter() - due)
        one_round(r)
    t_push = time.perf_counter() - t0
    st = h.drain()
    dt = time.perf_counter() - t0
    lat = sorted(1e3 * (m["rx_done"] - m["det_begin"]) for m in h.timeline()[3:]) or [0.0]      # round closed -> its PDUs fetched
    h.close()
    del ring
    air = round_items / 20e6
    per_round = dt / rounds
    want_pdus = streams * ((rounds + 3) * round_items // unit_len)
    return {"streams": streams, "global_streams": ([lshard.lo, lshard.hi, lshard.n_streams] if lshard is not None else None),
            "items_per_stream_per_round": round_items, "rounds": rounds, "ms_per_round": 1e3 * per_round,
            "x_realtime_each": air / per_round, "streams_at_realtime": streams * air / per_round,
            "pcie_GBps": float(item_bytes) * streams * round_items / per_round / 1e9, "bytes_per_item_over_pcie": item_bytes,
            "packets_per_s": st["steady_pdus"] / max(st["steady_s"], 1e-9),
            "pdus_ok": st["pdus"], "pdus_in_the_items_pushed": [want_pdus - 2 * streams, want_pdus + streams],
            "push_and_round_calls_s": round(t_push, 4), "paced_at": pace or None, "behind_ms": round(1e3 * behind, 2) if pace > 0 else None,
            "round_closed_to_pdus_ms": {"median": round(lat[len(lat) // 2], 2), "p90": round(lat[(9 * len(lat)) // 10], 2), "max": round(lat[-1], 2)},
            "host_thread_s": {k: {f: round(v, 5) for f, v in st[k].items()} for k in ("detect", "signal", "rx")},
            "arena_growths_in_session": st["arena_growths"], "lookback_giveups": st["det_giveups"],
            "note": "host-fed: every stream its own ring of pinned memory (%d B per item over PCIe), moved by the engine's gather kernel — the shader "
                    "cores read the host memory, no DMA engine; c8p_rawchain_round_async: the pipeline's first stage waits for its round's items; "
                    "warm-up rounds are inside the session, so arena_growths_in_session counts them too; pdus_in_the_items_pushed: every stream's "
                    "items hold that many whole packets, give or take the one at either end" % item_bytes}


def tx_rate(engine, local, n=4096):
    """SURVEY §8f-4 stage: c8p_tx packets/s for the cfg-2 packet shape (4096 x VHT MCS7, 1464-byte A-MPDU), PSDUs resident
    on the host (0.15 MB upload per call inside the timing)."""
    from gr_ieee80211_amd import abi
    rng = np.random.default_rng(3)
    pk = np.zeros(n, abi.TX_PKT_DT)
    pk["psdu_off"] = 1464 * np.arange(n); pk["len"] = 1464; pk["format"] = abi.F_VHT; pk["mcs"] = 7
    pk["scr_seed"] = 93; pk["mult"] = 12.0; pk["gap"] = 100; pk["nss"] = 1
    arena = rng.integers(0, 256, size=1464 * n, dtype=np.uint8)
    eng = engine.Engine(device=local)
    for _ in range(3):
        eng.tx(pk, arena)
    eng.sync()
    w = rate_windows(lambda i: eng.tx(pk, arena), eng.sync, n, window_s=0.05, probe_steps=3)
    eng.close()
    return w["per_s"]


def cpu_baseline_raw_stream(unit, n_packets=256):
    """the chain from the raw stream on one host core: the oracle's detect -> signal (+ the block's scan rule) -> demod + decode over
    `n_packets` copies of the unit the raw_stream_to_pdus leg repeats (same packet, same noise)"""
    from gr_ieee80211_amd import abi
    from oracle import oracle_py as o
    x = np.tile(unit, n_packets).astype(np.complex64)
    t0 = time.perf_counter()
    ev = o.detect(x)[0]
    t1 = time.perf_counter()
    s = np.zeros(len(ev), abi.SIG_IN_DT)
    s["raw_off"][:, 0] = ev["index"]
    s["n_avail"] = np.minimum(len(x) - ev["index"], 8192)          # (sizes the oracle's output slots: what these packets need, not the 109 504 of the longest one)
    s["cfo_rad"], s["snr"], s["rssi"], s["n_ant"] = ev["rad"], ev["snr"], ev["rssi"], 1
    pk, iq, chan = o.signal_events(s, x)
    keep, pos = [], 0
    for i in range(len(pk)):                             # signal_impl's scan rule (signal_impl.cc:76-111,154-160)
        idx = int(ev[i]["index"])
        if idx < pos:
            continue
        if pk[i]["n_items"] > 0:
            keep.append(i)
            pos = idx + 224 + int(pk[i]["lsig_nsamp"])
        else:
            pos = idx + 80
    t2 = time.perf_counter()
    _, outs, _, npdu = o.rx_batch(pk[keep], iq, chan)
    t3 = time.perf_counter()
    dt = t3 - t0
    return {"packets_per_s": npdu / dt, "samples_per_s": len(x) / dt, "cores": 1, "kind": "port",
            "sample": "%d items (%d packets), %d sync events, %d PDUs, %.1f s: detect %.2f s, signal %.2f s, demod + decode %.2f s"
                      % (len(x), n_packets, len(ev), npdu, dt, t1 - t0, t2 - t1, t3 - t2)}


def usable_cores():
    """host cores this process may actually use: affinity mask, capped by the container's CPU quota (cgroup v2 cpu.max /
    v1 cfs quota) — the GPU box shows 256 logical CPUs but grants a 16-CPU quota"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return n


def cpu_baseline_all_cores(pk, iq, chan, per_core_rate, budget_s=6.0):
    """one independent packet stream per host core (the oracle is re-entrant C, ctypes drops the GIL)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle_py as o
    cores = usable_cores()
    per = int(max(4, min(len(pk) // cores, 16)))                 # packets per pass and thread: short passes, checked against a deadline
    slices = [pk[i * per:(i + 1) * per] for i in range(cores) if len(pk[i * per:(i + 1) * per])]
    t = time.perf_counter()
    deadline = t + budget_s

    def work(s):
        done = pdus = 0
        while time.perf_counter() < deadline:
            pdus += o.rx_batch(s, iq, chan)[3]
            done += len(s)
        return done, pdus

    with ThreadPoolExecutor(len(slices)) as ex:
        res = list(ex.map(work, slices))
    dt = time.perf_counter() - t
    n = sum(r[0] for r in res)
    return {"value": n / dt, "unit": "packets/s", "cores": len(slices), "kind": "port",
            "sample": "%d packets (passes of %d of the same batch per thread until %.0f s), oracle demod+decode, %.1f s, %d PDUs; "
                      "built with '%s'" % (n, per, budget_s, dt, sum(r[1] for r in res), o.build_flags())}


# ------------------------------------------------------------------------------------------------ what bench.py calls
def arenas_to_host(torch, dev, iq_dev, chan_dev):
    """the two arenas at the signal -> demod boundary as host arrays (the CPU legs and the block mirrors read host memory)"""
    import ctypes
    iq = np.zeros(iq_dev.n, np.complex64); chan = np.zeros(chan_dev.n, np.complex64)
    torch.cuda.synchronize(dev)
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy(ctypes.c_void_p(iq.ctypes.data), ctypes.c_void_p(iq_dev.ptr), ctypes.c_size_t(iq.nbytes), ctypes.c_int(2))
    hip.hipMemcpy(ctypes.c_void_p(chan.ctypes.data), ctypes.c_void_p(chan_dev.ptr), ctypes.c_size_t(chan.nbytes), ctypes.c_int(2))
    return iq, chan


def block_chain_legs(out, pk, iq, chan, expected, local, full):
    """the gr::ieee80211::demod -> decode mirrors (host/c8p_blocks.cc) fed tagged host item streams on two threads.  The figure in
    `legs` is ONE configuration — the blocks' own pinned edge buffers, max_batch 512, threads on two cores that share an L3, median of
    the passes; --legs full adds the other configurations beside it (block_chain_variants), none of which replaces it."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_blocks", os.path.join(ROOT, "tools", "bench_blocks.py"))
    bb = importlib.util.module_from_spec(spec); spec.loader.exec_module(bb)
    b4 = (pk, iq, chan, expected)
    r = bb.run(4096, batches=(512,), quiet=True, batch=b4, passes=5 if full else 3)
    out["block_chain_packets_per_s"] = r["chain_packets_per_s"]
    out["block_chain_fused_packets_per_s"] = r["chain_fused_packets_per_s"]
    out["block_demod_packets_per_s"] = r["demod_block_packets_per_s"]
    out["block_decode_packets_per_s"] = r["decode_block_packets_per_s"]
    out["block_chain"] = {"max_batch": 512, "items_per_call": r["items_per_call"], "chain_threads_on_cpus": r.get("chain_threads_on_cpus"),
                          "mode_when_nothing_is_said": r.get("chain_default_mode"), "pdus_ok_in_that_mode": r.get("pdus_ok_chain_default"),
                          "pdus_ok_min": r["pdus_ok_min"], "passes": 5 if full else 3}
    out["host_phase_us"] = {"chain": r.get("chain_host_phase_us_per_packet"), "chain_fused": r.get("chain_fused_host_phase_us_per_packet")}
    if not full:
        return
    v = {}
    for name, kw in (("max_batch_256", dict(batches=(256,))), ("max_batch_1024", dict(batches=(1024,))),
                     ("threads_floating", dict(batches=(512,), pin=None)), ("consume_at_once", dict(batches=(512,), defer_consume=False)),
                     ("sched64k_pinned", dict(batches=(256,), io="sched64k")), ("sched64k_pageable", dict(batches=(256,), io="sched64k", pinned_io=False))):
        try:
            q = bb.run(4096, quiet=True, batch=b4, **kw)
            v[name] = {"items_per_call": q["items_per_call"], "chain": q["chain_packets_per_s"], "chain_fused": q["chain_fused_packets_per_s"]}
        except Exception as e:
            v[name] = {"error": "%s: %s" % (type(e).__name__, e)}
    out["block_chain_variants"] = v
    try:
        out["block_rx_chain"] = bb.rx_chain_rate(local=local)
        out["block_rx_chain"]["sc16_ports"] = bb.rx_chain_rate(local=local, bits=16)
    except Exception as e:
        out["block_rx_chain"] = {"error": "%s: %s" % (type(e).__name__, e)}


def run_single_gpu(ctx, out, args, full, deadline, emit):
    """rank 0 at N = 1, after the headline: the legs in order of what the short line reports first.  A leg that would start after
    `deadline` (time.perf_counter() seconds; None = no limit, --legs full) is recorded as skipped; every leg's wall time is in
    out["leg_wall_s"].  A leg that fails is recorded with its error and costs nothing else."""
    torch, dev, local = ctx["torch"], ctx["dev"], ctx["local"]
    pk, iq_dev, chan_dev, expected = ctx["pk"], ctx["iq_dev"], ctx["chan_dev"], ctx["expected"]
    walls = out.setdefault("leg_wall_s", {})
    host = {}

    def host_arenas():
        if "iq" not in host:
            host["iq"], host["chan"] = arenas_to_host(torch, dev, iq_dev, chan_dev)
        return host["iq"], host["chan"]

    def leg(name, fn, enabled=True):
        if not enabled:
            return
        if out_of_time(deadline):
            out.setdefault("legs_skipped", []).append(name)
            return
        t = time.perf_counter()
        try:
            fn()
        except Exception as e:
            import traceback
            out[name + "_error"] = "%s: %s | %s" % (type(e).__name__, e, traceback.format_exc(limit=3).replace("\n", " / "))
        walls[name] = round(time.perf_counter() - t, 2)

    other = args.workload != "cfg2"

    def _configs():
        out["configs"] = config_legs(torch, dev, local, args.snr, deadline=deadline,
                                     only=None if full else ("cfg1", "cfg3", "cfg4_one_gpu_share", "cfg5_station0", "cfg5_station1"))

    def _pcie():
        out["pcie_inclusive_detail"] = {}
        out["pcie_inclusive_packets_per_s"] = pcie_inclusive(pk, iq_dev, chan_dev, torch, dev, local, detail=out["pcie_inclusive_detail"], with_dma_path=full)

    def _blocks():
        iq, chan = host_arenas()
        block_chain_legs(out, pk, iq, chan, expected, local, full)

    def _stream():
        guard = LegGuard(out, 0, float(os.environ.get("C8P_STREAM_LEG_TIMEOUT_S", "180")), "sharded_stream", emit)
        try:
            out["sharded_stream"] = sharded_stream(torch, dev, local, 0, 1, ctx["use_dist"], args.stream_packets, args.stream_steps, 1)
        finally:
            guard.done()

    def _sens():
        out["sensitivity"] = sensitivity_legs(torch, dev, local, len(pk), out["value"], deadline=deadline)

    def _frontend():
        out["frontend_detect"] = frontend_detect(torch, dev, local, full=full)
        ctx["fe_unit"] = out["frontend_detect"].pop("_unit", None)

    leg("configs", _configs, not args.no_configs)
    leg("pcie_inclusive", _pcie, not args.no_host_legs)
    leg("block_chain", _blocks, not args.no_host_legs)
    leg("sensitivity", _sens, not args.no_sensitivity and not other)
    leg("sharded_stream", _stream, not args.no_stream)
    leg("frontend_detect", _frontend, not args.no_frontend)
    return host_arenas if host else None


def run_multi_gpu(ctx, out, args, emit):
    """every rank, world > 1 (or a one-rank torch.distributed.run with C8P_TEST_MULTI_LEG=1): BASELINE configs[3] as ONE stream scattered
    from rank 0, and the host-fed legs on all ranks at once.  Each runs under a watchdog on every rank, so that neither an exception on
    one rank nor a collective that never returns can cost the headline, which is complete when this is called."""
    import signal
    import torch.distributed as dist
    torch, dev, local, rank, world = ctx["torch"], ctx["dev"], ctx["local"], ctx["rank"], ctx["world"]
    # These legs have never met more than one GPU.  An exception is caught and a hang is timed out (LegGuard); if a PEER dies instead
    # (a signal inside a collective), the launcher ends the survivors with SIGTERM: rank 0 then prints the headline line as it stands
    # — measured and complete before any of this started — and leaves.
    if rank == 0 and out is not None:
        def on_term(signum, frame):
            out["multi_gpu_legs_error"] = "a peer rank died during the multi-GPU legs (SIGTERM from the launcher); the headline was measured before them"
            emit(out)
            os._exit(0)
        try:
            signal.signal(signal.SIGTERM, on_term)
        except (ValueError, OSError):
            pass
    if not args.no_stream and world > 1:
        guard = LegGuard(out, rank, float(os.environ.get("C8P_STREAM_LEG_TIMEOUT_S", "120")), "sharded_stream", emit)
        try:
            stream_res = sharded_stream(torch, dev, local, rank, world, True, args.stream_packets, args.stream_steps, 1)
        except Exception as e:                           # (the peers are then stuck in a collective: their guards end them)
            stream_res = {"error": "%s: %s" % (type(e).__name__, e), "world": world}
            guard.fire("rank %d: %s" % (rank, stream_res["error"]))
        dist.barrier()                                   # still guarded: a rank that failed above never arrives
        guard.done()
        if rank == 0:
            out["sharded_stream"] = stream_res
    if not args.no_host_legs:
        guard = LegGuard(out, rank, float(os.environ.get("C8P_HOST_LEG_TIMEOUT_S", "150")), "multi_gpu_host_fed", emit)
        try:
            mine = multi_gpu_host_fed(torch, dev, local, rank, world, ctx["pk"], ctx["iq_dev"], ctx["chan_dev"])
        except Exception as e:
            mine = {"rank": rank, "error": "%s: %s" % (type(e).__name__, e)}
        every = [None] * world
        dist.all_gather_object(every, mine)
        guard.done()
        if rank == 0:
            out["multi_gpu_host_fed"] = summarize_host_fed(every)
    if rank == 0 and out is not None:
        try:
            signal.signal(signal.SIGTERM, signal.SIG_DFL)
        except (ValueError, OSError):
            pass
