#!/usr/bin/env python3
"""bench.py — RX packets/s of the demod+decode hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: starts the N ranks itself, under torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One step = one pass of c8p_rx() over one batch of BASELINE.json configs[1]: 4096 packets of 20 MHz VHT MCS7,
1458-byte MPDU (46 OFDM symbols, 4400 complex64 boundary items, 11960 trellis steps each).  The batch (IQ + legacy
channel estimates) is resident in HBM before the timed region; packet descriptors (48 B each) are host structs, as
they are in the GNU Radio block, and their upload is inside the timed region.  Packets are independent, so N GPUs
each take their own batch (weak scaling, no data-path collective); the only collectives are the barrier and the
MAX-reduction of the elapsed time.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_ALG = 8 * 4400 + 512 + 32 + 1462       # algorithmic bytes per packet, SURVEY.md §8(d) = 37 206
HBM_PEAK_GBS = 8000.0                    # MI355X_MICROARCH.md: HBM3E 8 TB/s
PROFILE_EVERY = 4                        # c8p_cfg.profile: sample the per-kernel HIP events on every 4th submit
# The committed rocprofv3 profile the line's counter-derived figures come from (roofline.traffic, roofline_valu.counters, the opcode
# costs): ONE directory, named here and in the line — taken with tools/profile_gpu.sh + tools/summarize_prof.py on the code as it is.
PROFILE_TAG = os.environ.get("C8P_PROFILE_TAG", "r5t")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--repeats", type=int, default=5, help="the K-step timed region is run this many times; value = the median repeat")
    ap.add_argument("--min-timed-s", type=float, default=1.2, help="repeat the K-step region until the timed wall adds up to this (0: --repeats only)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-configs", action="store_true", help="skip the legs of the other BASELINE configs (`configs` object of the line)")
    ap.add_argument("--no-sensitivity", action="store_true", help="skip the headline's sensitivity legs (21 dB, windowed TX, 3-tap channel, mixed formats, with delivery)")
    ap.add_argument("--batch", type=int, default=4096, help="packets per GPU per step")
    ap.add_argument("--snr", type=float, default=30.0)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--dry-launch", action="store_true", help="no GPU: the launcher, rendezvous (gloo), barriers, max-over-ranks timing and the "
                    "assembly of the line with a sleep standing in for the step; the line says dry_launch and carries no value")
    ap.add_argument("--no-frontend", action="store_true", help="skip the detection front-end leg (extra object in the JSON line)")
    ap.add_argument("--workload", choices=("cfg2", "cfg4", "cfg1", "cfg3", "cfg4r", "cfg5u0", "cfg5u1"), default="cfg2",
                    help="cfg2: BASELINE configs[1], one 4096-packet batch per GPU and step (weak scaling; the headline). "
                         "cfg1 / cfg3 / cfg4r / cfg5u0 / cfg5u1: the same measurement on another BASELINE config resident in HBM (profiling: "
                         "tools/profile_gpu.sh takes the PMC passes of the two-stream kernel with cfg3); the line then names that config. "
                         "cfg4: BASELINE configs[3], ONE 65536-packet VHT MCS8 stream held by rank 0, scattered over the ranks, "
                         "decoded, results gathered device-resident (strong scaling); without the flag it is reported as the extra "
                         "object `sharded_stream` of the cfg2 line")
    ap.add_argument("--stream-packets", type=int, default=65536)
    ap.add_argument("--stream-steps", type=int, default=12)
    ap.add_argument("--no-stream", action="store_true", help="skip the sharded-stream leg")
    ap.add_argument("--no-host-legs", action="store_true", help="skip the PCIe-inclusive and block-chain legs (extra keys in the JSON line)")
    ap.add_argument("--no-check", action="store_true", help="kernel-variant experiments only: skip the PDU gate (line is then INVALID)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "RANK" not in os.environ:
        return launch_ranks(args)                        # N ranks under torch.distributed.run; this process only waits for them
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if env_world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE): refusing to report a line whose n_gpus is not "
                         "what was asked for" % (args.gpus, env_world))
    if args.dry_launch:
        return dry_launch(args)

    # stdout carries ONE JSON line and nothing else: libraries that chat on file descriptor 1 (RCCL prints a version banner when
    # a communicator is created) are pointed at stderr; the line itself goes to a private duplicate of the original stdout
    global _JSON_OUT
    sys.stdout.flush()
    _JSON_OUT = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)   # any torch.distributed.run launch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    if torch.cuda.device_count() < world:
        raise SystemExit("bench.py: --gpus %d but only %d GPU(s) are visible" % (world, torch.cuda.device_count()))
    torch.cuda.set_device(local)                         # before the process group: RCCL binds the communicator to the current device
    dev = torch.device("cuda", local)
    from __graft_entry__ import load_package
    load_package()
    from gr_ieee80211_amd import hostbind
    # this rank's host side — the engines' stage threads, the fronts' timers, RCCL's proxy threads, every pinned allocation — on the NUMA
    # node its GPU hangs off (reported in the line: host_binding)
    try:
        binding = hostbind.bind_rank(local, torch)
    except Exception as e:                               # (a platform whose sysfs this has not met: the rank runs unbound and says so)
        binding = {"gpu": int(local), "affinity": "kept (bind_rank failed: %s: %s)" % (type(e).__name__, e), "mempolicy": "kept"}
    binding["rank"] = rank
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl")
        if dist.get_world_size() != args.gpus:
            raise SystemExit("bench.py: the RCCL communicator has %d ranks, --gpus says %d" % (dist.get_world_size(), args.gpus))

    from gr_ieee80211_amd import engine
    from gr_ieee80211_amd.workloads import WORKLOADS, workload_on_device
    engine.load_library()
    binding["hip_runtimes_mapped"] = hostbind.assert_single_hip_runtime()      # torch's and the engine's must be ONE (raises otherwise)

    if args.workload == "cfg4":
        st = sharded_stream(torch, dev, local, rank, world, use_dist, args.stream_packets, max(args.steps, 1) if args.steps != 200 else 5,
                            args.warmup if args.warmup != 5 else 1)
        if use_dist:
            import torch.distributed as dist
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            best = st["feeds"]["rccl_scatter"]
            emit(({
                "metric": "RX packets/s (20 MHz VHT MCS8 1458 B, one %d-packet stream sharded over the GPUs)" % st["packets"],
                "value": best["packets_per_s"], "unit": "packets/s", "n_gpus": world, "steps": best["steps"], "warmup": best["warmup"],
                "ms_per_step": best["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
                "data": "synthetic (c8p_tx, distinct payloads, seeds 8021400+p)",
                "config": {"workload": "BASELINE configs[3] with the MCS8 substitution of SURVEY.md §7: VHT MCS8 256-QAM r=3/4 SISO, "
                                       "1458-byte MPDU, one %d-packet stream on rank 0, contiguous blocks per rank, AWGN 30 dB" % st["packets"],
                           "parallelism": "scatter (RCCL) -> c8p_rx per rank in sub-batches of %d -> all_gather of the fixed-stride "
                                          "result arenas (device-resident)" % st["sub_batch"]},
                "sharded_stream": st}))
        return

    # ---- synthetic workload, resident in HBM before anything is timed
    t_gen = time.time()
    # 4096 distinct MPDUs, generated on the box by the build's own transmitter (c8p_tx), per-packet
    # flat-channel phase + AWGN on the device, c8p_signal to the signal -> demod boundary
    kind = {"cfg4r": "cfg4"}.get(args.workload, args.workload)
    pk, iq_dev, chan_dev, expected, _keep = workload_on_device(kind, args.batch, args.snr, 100000 * rank, torch, dev, local)
    iq, chan = None, None
    data_note = ("synthetic (c8p_tx on the device, distinct payloads; clean OFDM symbols: the reference toolbox's half-amplitude symbol edges "
                 "are not applied — C8P_TX_NO_WINDOW, DESIGN.md §6 — AWGN + random-phase flat channel)")
    n_pkt = len(pk)
    t_gen = time.time() - t_gen
    other = args.workload != "cfg2"        # not the headline config: algorithmic bytes from the batch itself (SURVEY §8d formula)
    global B_ALG
    if other:
        B_ALG = float(np.mean(8.0 * pk["n_ant"] * pk["n_items"] + 512 + 32) + np.mean([len(e) for e in expected]))

    # engine-owned streams: consecutive batches are software-pipelined across C8P_SLOTS slots (upload / classify /
    # symbol demod of batch i+1 overlap the decoder of batch i); eng.sync() joins all of them
    # HIP events bracket each kernel on every 4th submit of the timed region (bracketing every launch costs ~5 % of the step)
    eng = engine.Engine(device=local, profile=0 if os.environ.get("C8P_BENCH_NOPROFILE") else PROFILE_EVERY)

    def barrier():
        eng.sync()
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # correctness gate (outside the timed region, before the warm-up so that the GPU does not idle between warm-up and timing):
    # every published PDU is the transmitted MPDU
    eng.rx(pk, iq_dev, chan_dev)
    eng.sync()
    res = eng.results(want_llr=False)
    rng_flags, flt_blocks, _ = eng.decstat()       # decoder diagnostics: packets sent to the clamp-exact kernel, float-compare blocks
    trellis = int(np.median(eng.tags()["trellis"]))
    dec_path = {"exact_kernel_packets": int(rng_flags.sum()), "float_step_block_fraction":
                float(flt_blocks[rng_flags == 0].sum()) / max(1, int((rng_flags == 0).sum()) * (trellis // 30))}
    n_ok = 0
    for r, e in zip(res, expected):
        if r["pdus"] and not args.no_check:
            assert r["pdus"] == [e], "PDU mismatch"
            n_ok += 1
    # a regression that makes every frame fail the FCS must not print a headline number
    gate_ok = args.no_check or (args.snr < 25.0) or n_ok >= 0.99 * n_pkt
    assert gate_ok, "correctness gate: only %d of %d packets delivered their MPDU at %.0f dB" % (n_ok, n_pkt, args.snr)
    # clock pre-roll (untimed, independent of --warmup): the PDU check above leaves the GPU idle for ~50 ms and its shader clock
    # needs some tens of ms of load to come back up; a short --steps run would otherwise measure the ramp
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < 0.15:
        eng.rx(pk, iq_dev, chan_dev)
    for _ in range(args.warmup):
        eng.rx(pk, iq_dev, chan_dev)
    eng.sync()
    eng.profile_reset()

    # the timed region: EXACTLY --steps steps between two barriers (+ device syncs), max over ranks; repeated --repeats times so
    # that the headline is a median with its spread instead of one sample of a few tens of ms
    rep_elapsed, rank_elapsed = [], []               # max over the ranks / this rank's own clock, per repeat
    n_rep = max(1, args.repeats)
    r_i = 0
    while r_i < n_rep:
        r_i += 1
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            eng.rx(pk, iq_dev, chan_dev)
        eng.sync()
        barrier()
        el = time.perf_counter() - t0
        rank_elapsed.append(el)
        if use_dist:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        rep_elapsed.append(el)
        if r_i == 1 and args.min_timed_s > 0:
            # a K-step region of a few tens of ms says little to an outside observer (the driver's SMI sampler never saw one): repeat
            # the region until the timed wall adds up to --min-timed-s of back-to-back load (every rank derives the same count from
            # the reduced first sample)
            n_rep = min(2000, max(n_rep, int(np.ceil(args.min_timed_s / max(el, 1e-6)))))
    elapsed = float(np.median(rep_elapsed))
    # every rank's own rate (its median repeat by its own clock): the spread says whether one GPU holds the others back
    per_rank = [n_pkt * args.steps / float(np.median(rank_elapsed))]
    if use_dist:
        t = torch.zeros(world, dtype=torch.float64, device=dev)
        t[rank] = per_rank[0]
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        per_rank = [float(v) for v in t.tolist()]
    kt = eng.kernel_times()
    eng.close()                              # its arenas (one set per slot, sized for the worst case per packet) are not needed any more

    # exclusive per-kernel durations: the same batch on an engine with ONE slot (one stream: every launch has the GPU to itself),
    # every launch bracketed, 40 steps back to back after 8 untimed ones — the GPU stays busy, so the clocks are the ones of the timed
    # region (with a sync after every step the part idles in between and the same kernel takes ~10 % longer); this is what
    # `C8P_SLOTS=1 rocprofv3 --kernel-trace --stats` averages (profiles/*/kernel_stats_exclusive.csv)
    slots_env = os.environ.get("C8P_SLOTS")
    os.environ["C8P_SLOTS"] = "1"
    eng_x = engine.Engine(device=local, profile=1)
    if slots_env is None:
        del os.environ["C8P_SLOTS"]
    else:
        os.environ["C8P_SLOTS"] = slots_env
    for i in range(48):
        eng_x.rx(pk, iq_dev, chan_dev)
        if i == 7:
            eng_x.sync()
            eng_x.profile_reset()
    eng_x.sync()
    kt_excl = eng_x.kernel_times()
    eng_x.close()

    out = None
    fe_unit = None
    if rank == 0:
        total_pkts = n_pkt * world * args.steps
        value = total_pkts / elapsed
        kern = {k: (ms / cnt if cnt else 0.0) for k, (ms, cnt) in kt.items()}
        kern_x = {k: (ms / cnt if cnt else 0.0) for k, (ms, cnt) in kt_excl.items()}
        # the dominant kernel is the one with the most work (longest when it has the GPU to itself); inside the timed region
        # the launches of 4 batches share the GPU, so elapsed durations there mostly reflect who waited for whom
        dom = max(("classify", "demod_sym", "viterbi_fwd", "finish"), key=lambda k: kern_x[k])
        # priced on the dominant kernel's EXCLUSIVE launch duration (one batch at a time: a per-launch cost that matches
        # profiles/*/pmc_summary.txt avg_ns and is <= ms_per_step); the event durations inside the pipelined region are
        # longer than a whole step because four batches share the GPU there, so they are reported but not priced
        dom_s = kern_x[dom] * 1e-3
        achieved = (B_ALG * n_pkt / dom_s) / 1e9 if dom_s > 0 else 0.0
        step_s = elapsed / args.steps
        achieved_step = (B_ALG * n_pkt / step_s) / 1e9
        traffic, traffic_src, traffic_step = pmc_traffic(dom)
        vals = sorted(n_pkt * world * args.steps / e for e in rep_elapsed)
        out = {
            "metric": "RX packets/s (20 MHz VHT MCS7 1458 B)" if not other else "RX packets/s (NOT the headline config: %s)" % WORKLOADS[kind][8],
            "value": value, "unit": "packets/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "per_rank_packets_per_s": {"values": per_rank, "min": min(per_rank), "max": max(per_rank),
                                       "note": "each rank's own packets/s by its own clock (median repeat); `value` = all ranks' packets / the slowest rank's time"},
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": data_note if not args.no_check else "INVALID (--no-check experiment)",
            "repeats": {"n": len(rep_elapsed), "value_median": value, "value_min": vals[0], "value_max": vals[-1],
                        "timed_region_ms": [round(1e3 * e, 3) for e in rep_elapsed],
                        "note": "each repeat times exactly --steps steps between barriers; value / ms_per_step are the median repeat"},
            "config": {"workload": ("BASELINE configs[1]: VHT MCS7 64-QAM r=5/6 SISO, 1458-byte MPDU" if not other else "BASELINE " + WORKLOADS[kind][8]) +
                                   ", %d packets/GPU/step, AWGN %.0f dB, random-phase flat channel, TX without the toolbox's symbol-edge windowing" % (n_pkt, args.snr),
                       "packets_per_gpu_per_step": n_pkt, "fcs_ok_fraction": n_ok / max(n_pkt, 1), "decoder_path": dec_path,
                       "parallelism": "packet-sharded x%d, no data-path collective" % world},
            # achieved / frac = the algorithmic bytes of a launch / the dominant kernel's average launch duration (HIP events, one batch at a
            # time: what `C8P_SLOTS=1 rocprofv3 --stats` averages) — the prescribed figure.  The same bytes over ms_per_step (frac_step) are
            # slightly HIGHER: the launches of consecutive batches overlap, so a step takes less than one exclusive launch
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "profile": "profiles/" + PROFILE_TAG, "residency_3slots": residency_3slots(),
                         "traffic_whole_step": traffic_step,
                         "frac_step": achieved_step / HBM_PEAK_GBS, "achieved_step": achieved_step,
                         "frac_exclusive": achieved / HBM_PEAK_GBS, "achieved_exclusive": achieved,
                         "alg_bytes_per_packet": B_ALG, "packets_per_launch": n_pkt,
                         "binding_resource": "VALU issue slots, not HBM (see roofline_valu); the HBM figure is the prescribed one",
                         "exclusive_avg_launch_ms": {k: round(v, 4) for k, v in kern_x.items()},
                         "pipelined_avg_launch_ms": {k: round(v, 4) for k, v in kern.items()},
                         "note": "achieved (= achieved_exclusive) = alg_bytes_per_packet x packets_per_launch / exclusive_avg_launch_ms[kernel]; "
                                 "achieved_step = the same bytes / ms_per_step; exclusive_avg_launch_ms: HIP events on the "
                                 "engine's stream, every launch, 40 steps back to back on ONE slot: one batch at a time, the average "
                                 "`C8P_SLOTS=1 rocprofv3 --stats` reports); "
                                 "pipelined_avg_launch_ms = events on every %d-th launch inside the timed region, where the launches of "
                                 "%s batches overlap (not a per-launch cost)" % (PROFILE_EVERY, os.environ.get("C8P_SLOTS", "3"))},
        }
        rv = None if other else valu_roofline(kern_x, n_pkt, dec_path["float_step_block_fraction"])
        if rv:
            out["roofline_valu"] = rv
        if not args.no_frontend and world == 1:
            try:                                     # (a leg must not cost the headline)
                out["frontend_detect"] = frontend_detect(torch, dev, local)
                fe_unit = out["frontend_detect"].pop("_unit", None)
            except Exception as e:
                out["frontend_detect"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if not args.no_configs and world == 1:
            out["configs"] = config_legs(torch, dev, local, args.snr)
        if not args.no_sensitivity and world == 1 and not other:
            out["sensitivity"] = sensitivity_legs(torch, dev, local, n_pkt, value)
    # BASELINE configs[3] beside the headline: ONE stream scattered from rank 0, decoded, gathered (every rank takes part).  The
    # headline above is complete at this point and this leg's collectives have never run on more than one GPU: it runs under a
    # watchdog on every rank, so that neither an exception on one rank nor a collective that never returns can cost the line.
    if not args.no_stream:
        guard = LegGuard(out, rank, float(os.environ.get("C8P_STREAM_LEG_TIMEOUT_S", "180")), "sharded_stream")
        try:
            stream_res = sharded_stream(torch, dev, local, rank, world, use_dist, args.stream_packets, args.stream_steps, 1)
        except Exception as e:                           # (the peers, if any, are then stuck in a collective: their guards end them)
            stream_res = {"error": "%s: %s" % (type(e).__name__, e), "world": world}
            if world > 1:
                guard.fire("rank %d: %s" % (rank, stream_res["error"]))
        if use_dist:
            dist.barrier()                               # still guarded: a rank that failed above never arrives
        guard.done()
        if rank == 0:
            out["sharded_stream"] = stream_res
    # world > 1: the host-fed legs, EVERY rank at the same time (each GPU has its own link; whether the host feeds them all at once is what
    # is asked): live ingest through shard.LiveShard (64 streams per rank) and the PCIe-inclusive c8p_rx, per-rank figures in the line
    bindings = [binding]
    if use_dist:
        bindings = [None] * world
        dist.all_gather_object(bindings, binding)
    # (C8P_TEST_MULTI_LEG=1 under a one-rank torch.distributed.run: the leg's code path — LiveShard, the gathers of the per-rank
    # records — on the one GPU a development box has; tools/probes and the GPU test that exercises it)
    if (world > 1 or (use_dist and os.environ.get("C8P_TEST_MULTI_LEG") == "1")) and not args.no_host_legs:
        guard = LegGuard(out, rank, float(os.environ.get("C8P_HOST_LEG_TIMEOUT_S", "240")), "multi_gpu_host_fed")
        try:
            mine = multi_gpu_host_fed(torch, dev, local, rank, world, pk, iq_dev, chan_dev)
        except Exception as e:
            mine = {"rank": rank, "error": "%s: %s" % (type(e).__name__, e)}
        every = [None] * world
        dist.all_gather_object(every, mine)
        guard.done()
        if rank == 0:
            out["multi_gpu_host_fed"] = summarize_host_fed(every)
    if rank == 0:
        out["host_binding"] = bindings
        if not args.no_host_legs and world == 1:
            try:                                 # (a host leg must not cost the headline: the line is printed whatever happens here)
                # what a host-fed caller sees — never `value`: inputs start in pinned host memory and the results come back
                out["pcie_inclusive_detail"] = {}
                out["pcie_inclusive_packets_per_s"] = pcie_inclusive(pk, iq_dev, chan_dev, torch, dev, local, detail=out["pcie_inclusive_detail"])
                import importlib.util
                spec = importlib.util.spec_from_file_location("bench_blocks", os.path.join(ROOT, "tools", "bench_blocks.py"))
                bb = importlib.util.module_from_spec(spec); spec.loader.exec_module(bb)
                if iq is None:      # the block mirrors are fed host item streams: bring the boundary arenas back once
                    iq, chan = arenas_to_host(torch, dev, iq_dev, chan_dev)
                r = bb.run(4096, batches=(256, 512, 1024), quiet=True, batch=(pk, iq, chan, expected))           # the blocks' own (pinned) buffers
                r64 = bb.run(4096, batches=(256,), quiet=True, batch=(pk, iq, chan, expected), io="sched64k")      # rounds 1-4's call sizes
                rpg = bb.run(4096, batches=(256,), quiet=True, batch=(pk, iq, chan, expected), io="sched64k", pinned_io=False)   # ... and pageable buffers
                rfl = bb.run(4096, batches=(256,), quiet=True, batch=(pk, iq, chan, expected), pin=None)           # custom buffers, threads left to the OS
                rnd = bb.run(4096, batches=(512,), quiet=True, batch=(pk, iq, chan, expected), defer_consume=False)  # ... every call waits for its own gathers
                try:
                    out["block_rx_chain"] = bb.rx_chain_rate(local=local)
                    out["block_rx_chain"]["sc16_ports"] = bb.rx_chain_rate(local=local, bits=16)
                    out["block_rx_chain"]["note"] = ("gr::ieee80211::rx_chain (host/c8p_rx_chain.cc): rx.grc's whole receive chain for 8 radios as ONE "
                                                     "block of the mirror, fed raw item streams in scheduler-sized chunks; every port's items are "
                                                     "copied once into pinned memory, one round per 524 288 items per stream")
                except Exception as e:
                    out["block_rx_chain"] = {"error": "%s: %s" % (type(e).__name__, e)}
                out["block_chain_packets_per_s"] = r["chain_packets_per_s"]
                out["block_chain_fused_packets_per_s"] = r["chain_fused_packets_per_s"]
                out["block_chain_variants"] = {
                    "custom_buffers": {"items_per_call": r["items_per_call"], "chain": r["chain_packets_per_s"], "chain_fused": r["chain_fused_packets_per_s"],
                                       "chain_threads_on_cpus": r.get("chain_threads_on_cpus"),
                                   "mode_when_nothing_is_said": r.get("chain_default_mode"), "pdus_ok_in_that_mode": r.get("pdus_ok_chain_default")},
                    "custom_buffers_threads_floating": {"items_per_call": rfl["items_per_call"], "chain": rfl["chain_packets_per_s"], "chain_fused": rfl["chain_fused_packets_per_s"],
                                                        "note": "the two scheduler threads wherever the OS puts them (no core affinity set on the blocks)"},
                    "custom_buffers_consume_at_once": {"items_per_call": rnd["items_per_call"], "chain": rnd["chain_packets_per_s"], "chain_fused": rnd["chain_fused_packets_per_s"],
                                                       "note": "batching_cfg::defer_consume off (C8P_DEFER_CONSUME=0): a call consumes what it looked at and waits for its gathers "
                                                               "before it returns; the default — items let go of two calls late — needs a buffer that announces when it moves "
                                                               "items (c8p::buffer_will_move; gnuradio/lib/c8p_pinned_buffer.h does)"},
                    "sched64k_pinned": {"items_per_call": r64["items_per_call"], "chain": r64["chain_packets_per_s"], "chain_fused": r64["chain_fused_packets_per_s"]},
                    "sched64k_pageable": {"items_per_call": rpg["items_per_call"], "chain": rpg["chain_packets_per_s"], "chain_fused": rpg["chain_fused_packets_per_s"],
                                          "note": "what rounds 1-4 measured: ordinary scheduler buffers, every packet staged through pinned copies on the host"},
                    "note": "block_chain_* (and legs.block_chain / block_fused) = custom_buffers: the edge buffers are allocated through the mirrors' buffer type "
                            "(GNU Radio 3.10 custom buffers: pinned host memory, 16 MB an edge, a call sees half), the packets are gathered out of them by the "
                            "device and demod's LLR items are written into them by the device; the two scheduler threads sit on two cores that share an L3 "
                            "(hostbind.cpu_pair_sharing_l3: what a flowgraph sets with the blocks' core affinity)"}
                out["block_demod_packets_per_s"] = r["demod_block_packets_per_s"]
                out["block_decode_packets_per_s"] = r["decode_block_packets_per_s"]
                out["host_phase_us"] = {"chain": r.get("chain_host_phase_us_per_packet"), "chain_fused": r.get("chain_fused_host_phase_us_per_packet"),
                                        "note": "host microseconds per packet inside the demod / decode block mirrors, by phase (exclusive): "
                                                "cut_and_stage = tag reads + the copy of the items into pinned staging, submit = the c8p_* call, "
                                                "wait = blocked on the batch, fetch = synchronous result copies, replay = tags / PDUs / debug lines, "
                                                "drain = the LLR items into the output buffer; wall = chain time per packet (two threads)"}
                out["host_legs_note"] = ("pcie_inclusive: the arenas start in pinned host memory every step (one c8p_copy_gather + c8p_copy_fence, then "
                                         "c8p_rx on the device copy), c8p_pkt_out + PDU arena fetched back to pinned memory behind the kernels, step i+1 "
                                         "submitted before step i is waited for; block_chain: the gr::ieee80211::demod -> decode "
                                         "mirrors (host/c8p_blocks.cc) fed tagged host item streams on two threads, LLR stream through "
                                         "host memory; block_chain_fused: the same with batching_cfg::fused (the default of a demod whose only reader is decode: fused_auto); medians of 5 passes "
                                         "over 4096 packets, max_batch 256 / 512 / 1024 (the best), %d of 4096 PDUs correct in the worst leg" % r["pdus_ok_min"])
            except Exception as e:
                import traceback
                out["host_legs_error"] = "%s: %s | %s" % (type(e).__name__, e, traceback.format_exc(limit=3).replace("\n", " / "))
        if not args.no_cpu and world == 1:      # the CPU leg runs on rank 0 at N = 1 only
            if iq is None:      # the oracle runs on the host: bring the boundary arenas back once
                iq, chan = arenas_to_host(torch, dev, iq_dev, chan_dev)
            out["cpu_baseline"] = cpu_baseline(pk, iq, chan)
            out["speedup_vs_cpu_1thread"] = value / out["cpu_baseline"]["value"]
            try:                                     # (the extra CPU legs must not cost the line either)
                out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(pk, iq, chan, out["cpu_baseline"]["value"])
                out["speedup_vs_cpu_all_cores"] = value / out["cpu_baseline_all_cores"]["value"]
                if isinstance(out.get("frontend_detect"), dict) and fe_unit is not None:
                    out["frontend_detect"]["raw_stream_cpu_baseline"] = cpu_baseline_raw_stream(fe_unit)
            except Exception as e:
                out["cpu_extra_legs_error"] = "%s: %s" % (type(e).__name__, e)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        out["legs"] = legs_summary(out)                   # last key of the line: what survives when only the tail of the line is kept
        emit(out)


def arenas_to_host(torch, dev, iq_dev, chan_dev):
    """the two arenas at the signal -> demod boundary as host arrays (the CPU legs and the block mirrors read host memory)"""
    import ctypes
    iq = np.zeros(iq_dev.n, np.complex64); chan = np.zeros(chan_dev.n, np.complex64)
    torch.cuda.synchronize(dev)
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy(ctypes.c_void_p(iq.ctypes.data), ctypes.c_void_p(iq_dev.ptr), ctypes.c_size_t(iq.nbytes), ctypes.c_int(2))
    hip.hipMemcpy(ctypes.c_void_p(chan.ctypes.data), ctypes.c_void_p(chan_dev.ptr), ctypes.c_size_t(chan.nbytes), ctypes.c_int(2))
    return iq, chan


def legs_summary(out):
    """every leg's one number (packets/s unless the name says otherwise), compact, as the LAST key of the line: the driver's record
    keeps the contract keys and the tail of stdout, and tests/test_bench_contract.py compares these with the committed line"""
    def g(*path):
        v = out
        for k in path:
            if not isinstance(v, dict) or k not in v:
                return None
            v = v[k]
        return round(float(v), 1) if isinstance(v, (int, float)) else None
    legs = {"headline": g("value"), "cfg1": g("configs", "cfg1", "packets_per_s"), "cfg3": g("configs", "cfg3", "packets_per_s"),
            "cfg4_share": g("configs", "cfg4_one_gpu_share", "packets_per_s"), "cfg5_sta0": g("configs", "cfg5_station0", "packets_per_s"),
            "sharded_rccl": g("sharded_stream", "feeds", "rccl_scatter", "packets_per_s"), "pcie": g("pcie_inclusive_packets_per_s"),
            "block_chain": g("block_chain_packets_per_s"), "block_fused": g("block_chain_fused_packets_per_s"),
            "block_rx_chain": g("block_rx_chain", "packets_per_s"), "block_rx_chain_sc16": g("block_rx_chain", "sc16_ports", "packets_per_s"),
            "detect_samples_per_s": g("frontend_detect", "samples_per_s"), "tx": g("frontend_detect", "tx_packets_per_s"),
            "raw_chain": g("frontend_detect", "raw_stream_to_pdus", "packets_per_s"),
            "raw_chain_first_segment_ms": g("frontend_detect", "raw_stream_to_pdus", "first_segment_ms"),
            "raw_chain_pcie": g("frontend_detect", "raw_stream_to_pdus", "pcie_inclusive", "packets_per_s"),
            "live_streams_at_realtime": g("frontend_detect", "raw_stream_to_pdus", "live_streams", "streams_at_realtime"),
            "live_streams_sc16_at_realtime": max(g("frontend_detect", "raw_stream_to_pdus", "live_streams", "sc16", "streams_at_realtime") or 0.0,
                                                 g("frontend_detect", "raw_stream_to_pdus", "live_streams", "sc16_128", "streams_at_realtime") or 0.0) or None,
            "live_streams_sc8_at_realtime": g("frontend_detect", "raw_stream_to_pdus", "live_streams", "sc8_128", "streams_at_realtime"),
            "cpu_1thread": g("cpu_baseline", "value"),
            "sens_21dB": g("sensitivity", "snr_21dB", "packets_per_s"), "sens_windowed": g("sensitivity", "windowed_tx", "packets_per_s"),
            "sens_3tap": g("sensitivity", "selective_3tap", "packets_per_s"), "sens_mixed": g("sensitivity", "mixed_formats", "packets_per_s"),
            "sens_mixed_steps_per_s": g("sensitivity", "mixed_formats", "trellis_steps_per_s"),
            "sens_mixed_air_items_per_s": g("sensitivity", "mixed_formats", "air_items_per_s"), "headline_air_items_per_s": g("sensitivity", "with_delivery", "air_items_per_s"),
            "with_delivery": g("sensitivity", "with_delivery", "packets_per_s"),
            "host_wall_us_fused": g("host_phase_us", "chain_fused", "wall"), "host_wall_us_chain": g("host_phase_us", "chain", "wall")}
    return {k: v for k, v in legs.items() if v is not None}


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


def launch_ranks(args):
    """`python bench.py --gpus N` started by hand (no RANK in the environment): start the N ranks, one per GPU, exactly as the driver
    does — python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <same arguments>
    — and pass their exit status on.  Rank 0 prints the line."""
    import socket
    import subprocess
    if not args.dry_launch:
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            raise SystemExit("bench.py: --gpus %d but only %d GPU(s) are visible (no line is printed for a world that cannot exist)" % (args.gpus, have))
    s_ = socket.socket()
    s_.bind(("127.0.0.1", 0))
    port = s_.getsockname()[1]
    s_.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # the host driver only supports dmabuf IPC (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    rc = subprocess.call(cmd, env=env)
    if rc != 0:
        raise SystemExit("bench.py: the %d-rank launch failed with status %d" % (args.gpus, rc))


def dry_launch(args):
    """--dry-launch: everything around the measurement without a GPU — the ranks the launcher started rendezvous over gloo, the timed
    region is bracketed by the same barriers, the elapsed time is the max over the ranks, every rank's own rate is gathered, rank 0
    prints ONE line with n_gpus = the world that actually ran.  A sleep stands in for the step: the line carries no value."""
    global _JSON_OUT
    sys.stdout.flush()
    _JSON_OUT = os.fdopen(os.dup(1), "w")                 # (gloo chats on file descriptor 1 as RCCL does: the line goes to a private duplicate)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    use_dist = "RANK" in os.environ and "MASTER_PORT" in os.environ
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")
        if dist.get_world_size() != args.gpus:
            raise SystemExit("bench.py: the communicator has %d ranks, --gpus says %d" % (dist.get_world_size(), args.gpus))
    # what every rank does before it touches its GPU: find the device's NUMA node, bind to it (no GPU here: nothing to find, and the
    # record says so), check that one HIP runtime is mapped; and the partition of the live streams over the ranks (shard.LiveShard's)
    from __graft_entry__ import load_package
    load_package()
    from gr_ieee80211_amd import hostbind, shard
    binding = hostbind.bind_rank(int(os.environ.get("LOCAL_RANK", "0")), torch)
    binding["rank"] = rank
    binding["hip_runtimes_mapped"] = hostbind.assert_single_hip_runtime()
    lo, hi = shard.partition(64 * world, world, rank)
    my_leg = {"rank": rank, "live_streams": {"global_streams": [lo, hi, 64 * world], "streams": hi - lo}, "pcie_inclusive_packets_per_s": None}
    n_pkt = args.batch
    for _ in range(args.warmup):
        time.sleep(1e-4)
    if use_dist:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(2e-3 * (1 + rank))                    # (rank r is r times slower: the max-reduction and the per-rank spread have something to show)
    mine = time.perf_counter() - t0
    if use_dist:
        dist.barrier()
    el = time.perf_counter() - t0
    per_rank = [n_pkt * args.steps / mine]
    if use_dist:
        t = torch.tensor([el], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
        v = torch.zeros(world, dtype=torch.float64)
        v[rank] = per_rank[0]
        dist.all_reduce(v, op=dist.ReduceOp.SUM)
        per_rank = [float(x) for x in v.tolist()]
    bindings, legs_ = [binding], [my_leg]
    if use_dist:
        bindings, legs_ = [None] * world, [None] * world
        dist.all_gather_object(bindings, binding)
        dist.all_gather_object(legs_, my_leg)
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        emit({"metric": "RX packets/s (20 MHz VHT MCS7 1458 B)", "value": None, "host_binding": bindings,
              "multi_gpu_host_fed": {"dry": True, "per_rank": legs_, "ranks_ok": len(legs_),
                                     "note": "dry launch: the partition of the live streams over the ranks and the gathering of the per-rank records; "
                                             "no engine, no figures — UNMEASURED at N > 1"}, "unit": "packets/s", "n_gpus": world, "steps": args.steps,
              "warmup": args.warmup, "ms_per_step": 1e3 * el / max(args.steps, 1), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
              "dtype": "f32", "data": "none (dry launch)", "dry_launch": True,
              "per_rank_packets_per_s": {"values": per_rank, "min": min(per_rank), "max": max(per_rank), "note": "sleeps, not packets"},
              "config": {"workload": "dry launch: launcher, rendezvous, barriers and reductions only; NOT a measurement",
                         "parallelism": "packet-sharded x%d, no data-path collective" % world}})


_JSON_OUT = None


class LegGuard:
    """Watchdog around an optional leg that runs AFTER the headline was measured: if the leg does not finish in `timeout_s` (or
    `fire` is called), rank 0 prints the headline line as it stands, with the leg marked as failed, and the process exits with
    status 0 — on every rank, each by its own timer, because a rank stuck in a collective cannot be told anything."""

    def __init__(self, out, rank, timeout_s, name):
        import threading
        self.out, self.rank, self.name, self.timeout_s = out, rank, name, timeout_s
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
            emit(self.out)
        else:
            time.sleep(5.0)                              # let rank 0 print first
        sys.stderr.write("bench.py: leg %s abandoned on rank %d: %s\n" % (self.name, self.rank, why))
        sys.stderr.flush()
        os._exit(0)

    def done(self):
        self._ev.set()


def emit(obj):
    f = _JSON_OUT or sys.stdout
    f.write(json.dumps(obj) + "\n")
    f.flush()


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
        sync_all()
        t0 = time.perf_counter()
        tls = [step(feed_name) for _ in range(steps)]
        fin = [t[1] for t in tls[1:]] + [finalize()]
        sync_all()
        tl = np.mean(np.array([list(a[0]) + list(f) for a, f in zip(tls, fin)]), axis=0)
        v = torch.tensor([time.perf_counter() - t0] + list(tl), dtype=torch.float64, device=dev)
        dist.all_reduce(v, op=dist.ReduceOp.MAX)
        v = v.tolist()
        counts, outs_all, pdu_all = state["g"]
        n_pdu, n_fail = shard.count_pdus(counts, outs_all)
        last_ok = None
        if rank == 0:                                 # the LAST timed step's gathered arenas, not only the first step's: every PDU again
            got = shard.pdus_of(counts, outs_all, pdu_all, n_total)
            last_ok = sum(1 for p, b in got if b == expected[p])
            assert last_ok == len(got) and last_ok >= 0.97 * n_total, "sharded stream, last step of %s: %d PDUs, %d correct" % (feed_name, len(got), last_ok)
        res["feeds"][feed_name] = {"packets_per_s": n_total * steps / v[0], "ms_per_step": 1e3 * v[0] / steps, "steps": steps, "warmup": max(warmup, 1),
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
        t = time.perf_counter()
        for _ in range(max(steps, 2)):
            one_pass()
        eng.sync()
        res["resident_one_gpu_packets_per_s"] = n_total * max(steps, 2) / (time.perf_counter() - t)
        res["rccl_scatter_over_resident"] = res["feeds"]["rccl_scatter"]["packets_per_s"] / res["resident_one_gpu_packets_per_s"]
    res["note"] = ("unmeasured at N > 1 until the driver has an 8-GPU node; at N = 1 the ingest rank's block is a zero-copy view of the "
                   "stream (nothing to send) and the all_gather a self-copy through RCCL") if world == 1 else "measured at world %d" % world
    eng.close()
    del keep
    if own_group:
        dist.destroy_process_group()
    return res


def config_legs(torch, dev, local, snr_db=30.0, timed_s=0.35):
    """The other BASELINE.json configs beside the cfg-2 headline (all through c8p_rx, inputs resident in HBM, generated on the box by
    c8p_tx as SURVEY.md §8d states them — gr-ieee80211_amd/workloads.py WORKLOADS): packets/s, ms per step, algorithmic bytes per packet
    (8 n_ant n_items + 512 + 32 + PDU bytes) and the HBM fraction a step reaches.  Every published PDU is compared with
    the transmitted MPDU before anything is timed."""
    from gr_ieee80211_amd import engine
    from gr_ieee80211_amd.workloads import WORKLOADS, workload_on_device
    legs = [("cfg1", "cfg1", 4096), ("cfg3", "cfg3", 4096), ("cfg3_1024_packets", "cfg3", 1024), ("cfg4_one_gpu_share", "cfg4", 8192),
            ("cfg5_station0", "cfg5u0", 1024), ("cfg5_station1", "cfg5u1", 1024)]
    eng = engine.Engine(device=local)
    res = {}
    for name, kind, n in legs:
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
            while time.perf_counter() - t < 0.1:
                eng.rx(pk, A, C_)
            eng.sync()
            t = time.perf_counter()
            for _ in range(8):
                eng.rx(pk, A, C_)
            eng.sync()
            k = int(max(8, timed_s / max((time.perf_counter() - t) / 8, 1e-6)))
            t = time.perf_counter()
            for _ in range(k):
                eng.rx(pk, A, C_)
            eng.sync()
            dt = time.perf_counter() - t
            pps = len(pk) * k / dt
            res[name] = {"workload": WORKLOADS[kind][8] + ", %d packets/step, AWGN %.0f dB" % (len(pk), snr_db), "packets_per_s": pps,
                         "ms_per_step": 1e3 * dt / k, "steps": k, "packets_per_step": len(pk), "fcs_ok_fraction": n_ok / len(pk),
                         "alg_bytes_per_packet": b_alg, "achieved_GBps": pps * b_alg / 1e9, "frac_step": pps * b_alg / 1e9 / HBM_PEAK_GBS}
            del keep, A, C_
        except Exception as e:                            # (a leg must not cost the headline)
            res[name] = {"error": "%s: %s" % (type(e).__name__, e)}
    eng.close()
    return res


def sensitivity_legs(torch, dev, local, n_pkt, headline, timed_s=0.35):
    """The headline's sensitivity to its input: the same 4096-packet cfg-2 batch size, resident in HBM, the
    same timing code (warm-up, then K steps of c8p_rx between two syncs), with what the headline's input leaves out — the decoder's fast path
    is data-dependent (integer-compare blocks need provably positive path metrics, LLRs outside the lean range go to the clamp-exact
    kernel): 21 dB (most frames fail the FCS), the toolbox's symbol-edge windowing (tools/phy80211header.py:894-899), a 3-tap
    frequency-selective channel, a batch of mixed formats / rates / lengths (uneven trellis lengths), and the headline batch WITH
    delivery: every step's c8p_pkt_out records and PDU arena fetched to pinned host memory inside the timed region."""
    from gr_ieee80211_amd import abi, engine
    from gr_ieee80211_amd.workloads import TAPS3, workload_on_device
    cases = [("snr_21dB", "cfg2", dict(snr=21.0), "cfg 2 at 21 dB AWGN"),
             ("windowed_tx", "cfg2", dict(snr=30.0, window=True), "cfg 2, TX with the toolbox's half-amplitude symbol edges (flags = 0), 30 dB"),
             ("selective_3tap", "cfg2", dict(snr=30.0, taps=TAPS3), "cfg 2 through a 3-tap channel (0 / -9 / -15 dB at 0 / 1 / 2 samples), 30 dB"),
             ("mixed_formats", "random1", dict(snr=30.0), "random legacy / HT / VHT, MCS and length (14-3000 B) per packet, 30 dB"),
             ("with_delivery", "cfg2", dict(snr=30.0, deliver=True), "the headline batch, c8p_pkt_out + PDU arena fetched to pinned host memory every step")]
    res = {}
    eng = engine.Engine(device=local)
    for name, kind, o, what in cases:
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
            if deliver:                                 # two pinned sets: step i's copies land while step i+1 is submitted
                _, ob = eng.buffer(abi.BUF_OUT)
                _, pb = eng.buffer(abi.BUF_PDU)
                bufs = [(torch.empty(ob, dtype=torch.uint8).pin_memory(), torch.empty(pb, dtype=torch.uint8).pin_memory()) for _ in range(3)]

            def step(i):
                eng.rx(pk, A, C_)
                if deliver:
                    o_, p_ = bufs[i % 3]
                    eng.fetch_async(abi.BUF_OUT, o_.data_ptr(), 0, o_.numel())
                    eng.fetch_async(abi.BUF_PDU, p_.data_ptr(), 0, p_.numel())
            t = time.perf_counter()
            i = 0
            while time.perf_counter() - t < 0.1:
                step(i); i += 1
            eng.sync()
            t = time.perf_counter()
            for i in range(8):
                step(i)
            eng.sync()
            k = int(max(8, timed_s / max((time.perf_counter() - t) / 8, 1e-6)))
            dts = []
            for _w in range(2):                         # two timed windows, the better one counts: these hosts are shared, and one window in
                t = time.perf_counter()                 # ten or twenty meets a neighbour's burst (a leg once read 0.88 of the headline, the
                for i in range(k):                      # next run 1.04: profiles/r5v)
                    step(i)
                eng.sync()
                dts.append(time.perf_counter() - t)
            dt = min(dts)
            pps = len(pk) * k / dt
            res[name] = {"workload": what + ", %d packets/step" % len(pk), "packets_per_s": pps, "ms_per_step": 1e3 * dt / k, "steps": k,
                         "windows_ms_per_step": [round(1e3 * d_ / k, 4) for d_ in dts],
                         "vs_headline": pps / headline if headline else None, "fcs_ok_fraction": n_ok / max(len(pk), 1),
                         "exact_kernel_packets": int(rng_flags.sum()),
                         "float_step_block_fraction": float(flt_blocks[lean].sum()) / max(1, int((tr[lean] // 30).sum())),
                         "trellis_steps_per_packet_mean": float(tr.mean()),
                         "trellis_steps_per_s": float(tr.sum()) * k / dt,
                         "items_per_packet_mean": float(pk["n_items"].mean()),
                         "air_items_per_s": float(pk["n_items"].astype(np.int64).sum()) * k / dt}
            if deliver:
                got = np.frombuffer(bufs[(k - 1) % 3][0].numpy().tobytes(), abi.PKT_OUT_DT)
                res[name]["delivered_bytes_per_step"] = int(bufs[0][0].numel() + bufs[0][1].numel())
                assert int((got["n_pdu"] == 1).sum()) == n_ok, "with_delivery: the fetched c8p_pkt_out records disagree with the check pass"
            del keep, A, C_
        except Exception as e:                            # (a leg must not cost the headline)
            res[name] = {"error": "%s: %s" % (type(e).__name__, e)}
    eng.close()
    res["note"] = ("same batch size and timing code as the headline (inputs resident, K steps of c8p_rx between two syncs, two windows of ~%.2f s each, the better one counts: windows_ms_per_step has both); "
                   "vs_headline = packets_per_s / value.  mixed_formats is NOT the same unit of work per packet: random rate and length put ~4 x the air "
                   "time of a cfg-2 packet into the average packet (items_per_packet_mean; a 3000-byte BPSK frame is 1000 symbols to demodulate for "
                   "24 000 trellis steps, a cfg-2 frame 46 symbols for 11 960) — compare air_items_per_s (cfg 2: 4400 items per packet); the engine "
                   "decodes a batch of mixed lengths longest first" % timed_s)
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


def valu_roofline(kern_x, n_pkt, float_frac, T=11960):
    """The resource that binds the dominant kernel: VALU issue.  The kernel is k_rx_1ss_lds = symbol demodulation + the lean forward
    pass (LLRs through LDS).  ceiling = [sum over the forward pass's opcodes of (count x measured issue cost, profiles/*/opcode_costs.json from
    tools/probe_issue.hip) + the kernel's other VALU instructions (SQ_INSTS_VALU of the newest committed PMC pass minus the forward
    pass's count) at the 4.15-cycle slot every non-trivial opcode costs] / (1024 SIMDs x 2.4 GHz); frac = ceiling / exclusive
    launch duration of this run.  Pricing at the nominal 2.4 GHz makes the fraction a lower bound (the shader clock under this
    load is 2.2-2.35 GHz).  One batch alone gives every SIMD 4 decoder waves; the arbiter serves the oldest wave first, so they
    finish one after the other and the last runs alone, latency-bound (profiles/r2b/decoder_wave_cycles.txt: 27 SIMD cycles per
    wave-step at 4 waves, 24 at 8 waves, against 21.7 from the opcode costs) — the exclusive launch cannot reach the ceiling, the
    pipelined step (other batches fill the tail) is what `value` measures."""
    pdir = os.path.join(ROOT, "profiles", PROFILE_TAG)
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


def residency_3slots():
    """the dominant kernel as the bench runs it (three slots): profiles/<tag>/residency_3slots.json from the kernel trace"""
    f = os.path.join(ROOT, "profiles", PROFILE_TAG, "residency_3slots.json")
    if not os.path.exists(f):
        return None
    r = json.load(open(f))
    r["file"] = os.path.relpath(f, ROOT)
    return r


def pmc_traffic(dom):
    """HBM bytes per launch of the dominant kernel, and of the three RX kernels of a c8p_rx step together, from the newest committed
    rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in separate runs, profiles/*/pmc_hbm.json).  FETCH_SIZE is doubled
    per MI355X_MICROARCH.md (gfx950 tallies a 128-byte request as 64 bytes; every kernel here reads with wide vector loads).
    Profiles from before the LLR stream moved into LDS hold k_rx_1ss instead of k_rx_1ss_lds: its scalar-cache reads of the LLRs were
    64-byte requests and counted in full, so that kernel = FETCH_SIZE as reported + half of the sample bytes.  The whole step also
    counts the second, nearly empty launch of k_rx_1ss (only_flagged) and k_viterbi_fwd.  See DESIGN.md §7."""
    names = {"viterbi_fwd": "k_rx_1ss_lds", "finish": "k_finish", "classify": "k_classify"}
    best = os.path.join(ROOT, "profiles", PROFILE_TAG, "pmc_hbm.json")
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


def pcie_inclusive(pk, iq_dev, chan_dev, torch, dev, local, steps=12, detail=None):
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
        dt = None
        for _ in range(3):                                 # three windows of `steps`, the best one: a window is ~10 ms, and one host
            t = time.perf_counter()                        # hiccup inside it read 0.50 M on one box of round 5 (profiles/r5p, first retake)
            for i in range(steps + 1):
                if i < steps:
                    submit(i)                              # (set i % 2 was last read by step i - 2, waited for below one step ago)
                if i > 0:
                    eng.wait(tickets[(i - 1) % 3])         # step i - 1: kernels done, its results are in pinned host memory
            d_ = time.perf_counter() - t
            dt = d_ if dt is None else min(dt, d_)
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
        detail["copy_stream_cus"] = os.environ.get("C8P_COPY_CUS", "16 (this leg's choice; engine default 32)")
        detail["host_to_device_GBps"] = rate / len(pk) * down / 1e9
        detail["device_to_host_GBps"] = rate / len(pk) * up / 1e9
        detail["bytes_per_step"] = {"down": down, "up": up}
        try:
            detail["dma_path_packets_per_s"] = dma_path()
        except Exception as e:
            detail["dma_path_packets_per_s"] = "%s: %s" % (type(e).__name__, e)
    return rate


def frontend_detect(torch, dev, local, n_target=32 << 20):
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
    t0 = time.perf_counter()
    K = 5
    for _ in range(K):
        ev = eng.detect(arr)
    dt = (time.perf_counter() - t0) / K
    kt = {k: (ms / cnt if cnt else 0.0) for k, (ms, cnt) in eng.kernel_times().items()}
    eng.close()
    pre_s = kt["presiso"] * 1e-3
    tx = tx_rate(engine, local)
    chain = raw_stream_chain(torch, dev, local, arr, n, reps, unit)
    return {"tx_packets_per_s": tx, "samples": n, "packets_in_stream": reps, "sync_events": int(len(ev)), "samples_per_s": n / dt,
            "x_realtime_20MSps": n / dt / 20e6, "ms_per_call": 1e3 * dt,
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


def raw_stream_chain(torch, dev, local, arr, n, reps, unit, segments=24):
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
        for _r in range(2):                                                   # two runs of 24 segments (~35 ms each), the better one counts: on these
            st = h.run(segs)                                                  # shared hosts one run in ten meets a burst of a neighbour's (a run once
            runs.append((st, h.timeline()))                                   # read 2.4 M packets/s between two of 4.3-4.6 M: profiles/r5v)
        rate = [r_[0]["steady_pdus"] / max(r_[0]["steady_s"], 1e-9) for r_ in runs]
        st, tl = runs[int(np.argmax(rate))]
        out.update(chain_report(st, tl, reps))
        out["runs_packets_per_s"] = [round(v) for v in rate]
        out["warmup"] = {"segments": warm["segments"], "arena_growths": warm["arena_growths"], "first_segment_ms": round(1e3 * warm["first_segment_s"], 3)}
    except Exception as e:
        out["error"] = "%s: %s" % (type(e).__name__, e)
        if h is not None:
            h.close()
        return out
    # what a host-fed caller sees: the same segment in pinned host memory, crossing PCIe once per pass (inside c8p_detect)
    pin = None
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
    t0 = time.perf_counter()
    K = 10
    for _ in range(K):
        eng.tx(pk, arena)
    eng.sync()
    dt = (time.perf_counter() - t0) / K
    eng.close()
    return n / dt


def cpu_baseline(pk, iq, chan, budget_s=12.0):
    """The oracle (scalar CPU restatement, "port") timed on one host core over a bounded sample of the same batch."""
    from oracle import oracle_py as o          # the checker, timed here as the CPU baseline ("port"): never part of the measured path
    t = time.perf_counter()
    o.rx_batch(pk[:8], iq, chan)
    per = (time.perf_counter() - t) / 8
    n = int(max(16, min(len(pk), budget_s / max(per, 1e-6))))
    t = time.perf_counter()
    _, outs, _, npdu = o.rx_batch(pk[:n], iq, chan)
    dt = time.perf_counter() - t
    return {"value": n / dt, "unit": "packets/s", "cores": 1, "kind": "port",
            "sample": "first %d packets of the same batch, oracle demod+decode, 1 thread, %.1f s, %d PDUs; built with '%s'; the "
                      "oracle's 64-point FFT is an O(N^2) DFT accumulated in double (51 x 4096 complex MACs per packet, ~4 %% of "
                      "its time next to the 11960-step Viterbi)" % (n, dt, npdu, o.build_flags())}


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


if __name__ == "__main__":
    main()
