#!/usr/bin/env python3
"""bench.py — RX packets/s of the demod+decode hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: starts the N ranks itself, under torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One step = one pass of c8p_rx() over one batch of BASELINE.json configs[1]: 4096 packets of 20 MHz VHT MCS7, 1458-byte MPDU
(46 OFDM symbols, 4400 complex64 boundary items, 11960 trellis steps each).  The batch (IQ + legacy channel estimates) is
resident in HBM before the timed region; packet descriptors (48 B each) are host structs, as they are in the GNU Radio block,
and their upload is inside the timed region.  Packets are independent, so N GPUs each take their own batch (weak scaling, no
data-path collective); the only collectives are the barrier and the MAX-reduction of the elapsed time.

stdout: ONE JSON line of at most 4 KB on rank 0 (short_line(): the contract keys, `roofline`, `cpu_baseline`, one number per
leg).  Everything else — every leg's own record, notes, per-rank bindings — goes to bench_detail.json (--detail) and, as one
line, to stderr.  The legs themselves live in tools/bench_legs.py; without --legs full they run under a wall-clock budget.
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_ALG = 8 * 4400 + 512 + 32 + 1462       # algorithmic bytes per packet, SURVEY.md §8(d) = 37 206
HBM_PEAK_GBS = 8000.0                    # MI355X_MICROARCH.md: HBM3E 8 TB/s
PROFILE_EVERY = 4                        # c8p_cfg.profile: sample the per-kernel HIP events on every 4th submit
# The committed rocprofv3 profile the line's counter-derived figures come from (roofline.traffic*): ONE directory, named here and in
# the line — taken with tools/profile_gpu.sh + tools/summarize_prof.py on the code as it is.
PROFILE_TAG = os.environ.get("C8P_PROFILE_TAG", "r6a")
sys.path.insert(1, os.path.join(ROOT, "tools"))
import bench_line                        # noqa: E402  (tools/bench_line.py: short_line / finish / launch_ranks / dry_launch)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=5, help="the K-step timed region is run this many times; value = the median repeat")
    ap.add_argument("--min-timed-s", type=float, default=1.2, help="repeat the K-step region until the timed wall adds up to this (0: --repeats only)")
    ap.add_argument("--batch", type=int, default=4096, help="packets per GPU per step")
    ap.add_argument("--snr", type=float, default=30.0)
    ap.add_argument("--workload", choices=("cfg2", "cfg4", "cfg1", "cfg3", "cfg4r", "cfg5u0", "cfg5u1"), default="cfg2",
                    help="cfg2: BASELINE configs[1] (the headline).  cfg1 / cfg3 / cfg4r / cfg5u0 / cfg5u1: the same measurement on another "
                         "BASELINE config resident in HBM (profiling); the line names that config.  cfg4: BASELINE configs[3], ONE 65536-packet "
                         "VHT MCS8 stream held by rank 0, scattered over the ranks, decoded, gathered (strong scaling)")
    ap.add_argument("--legs", choices=("budget", "full", "none"), default="budget",
                    help="budget: the legs beside the headline run in order until --legs-budget-s is spent (the rest are listed as skipped); "
                         "full: every leg and variant, no limit (what profiles/*/bench_detail.json is taken with); none: headline + cpu_baseline")
    ap.add_argument("--legs-budget-s", type=float, default=16.0)
    ap.add_argument("--detail", default=None, help="where the full record goes (default: bench_detail.json beside bench.py, and gpurun_out/)")
    ap.add_argument("--cpu-budget-s", type=float, default=10.0, help="seconds of one host core spent on the cpu_baseline sample")
    ap.add_argument("--stream-packets", type=int, default=65536)
    ap.add_argument("--stream-steps", type=int, default=6)
    ap.add_argument("--dry-launch", action="store_true", help="no GPU: the launcher, rendezvous (gloo), barriers, max-over-ranks timing and the "
                    "assembly of the line with a sleep standing in for the step; the line says dry_launch and carries no value")
    ap.add_argument("--no-check", action="store_true", help="kernel-variant experiments only: skip the PDU gate (line is then INVALID)")
    for leg in ("cpu", "configs", "sensitivity", "frontend", "stream", "host-legs"):
        ap.add_argument("--no-" + leg, action="store_true", help="skip that leg")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.legs == "none":
        args.no_configs = args.no_sensitivity = args.no_frontend = args.no_stream = args.no_host_legs = True
    return args


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        return bench_line.launch_ranks(args, __file__)                        # N ranks under torch.distributed.run; this process only waits for them
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if env_world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE): refusing to report a line whose n_gpus is not "
                         "what was asked for" % (args.gpus, env_world))
    if args.dry_launch:
        return bench_line.dry_launch(args)
    t_start = time.perf_counter()
    bench_line.private_stdout()

    import torch
    rank, world, local = (int(os.environ.get(k, "0")) for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"))
    world = max(world, 1)
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)   # any torch.distributed.run launch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    if torch.cuda.device_count() < world:
        raise SystemExit("bench.py: --gpus %d but only %d GPU(s) are visible" % (world, torch.cuda.device_count()))
    torch.cuda.set_device(local)                         # before the process group: RCCL binds the communicator to the current device
    dev = torch.device("cuda", local)
    from __graft_entry__ import load_package
    load_package()
    from gr_ieee80211_amd import engine, hostbind
    from gr_ieee80211_amd.workloads import WORKLOADS, workload_on_device
    # this rank's host side (stage threads, RCCL's proxy threads, every pinned allocation) on the NUMA node its GPU hangs off
    try:
        binding = hostbind.bind_rank(local, torch)
    except Exception as e:                               # (a platform whose sysfs this has not met: the rank runs unbound and says so)
        binding = {"gpu": int(local), "affinity": "kept (bind_rank failed: %s: %s)" % (type(e).__name__, e), "mempolicy": "kept"}
    binding["rank"] = rank
    dist = None
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl")
        if dist.get_world_size() != args.gpus:
            raise SystemExit("bench.py: the RCCL communicator has %d ranks, --gpus says %d" % (dist.get_world_size(), args.gpus))
    engine.load_library()
    binding["hip_runtimes_mapped"] = hostbind.assert_single_hip_runtime()      # torch's and the engine's must be ONE (raises otherwise)
    legs = bench_line.load_legs()

    if args.workload == "cfg4":                          # BASELINE configs[3] as the line itself (strong scaling)
        st = legs.sharded_stream(torch, dev, local, rank, world, use_dist, args.stream_packets, max(args.steps, 1) if args.steps != 200 else 5,
                                 args.warmup if args.warmup != 5 else 1)
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            best = st["feeds"]["rccl_scatter"]
            bench_line.finish({"metric": "RX packets/s (20 MHz VHT MCS8 1458 B, one %d-packet stream sharded over the GPUs)" % st["packets"],
                    "value": best["packets_per_s"], "unit": "packets/s", "n_gpus": world, "steps": best["steps"], "warmup": best["warmup"],
                    "ms_per_step": best["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
                    "data": "synthetic (c8p_tx, distinct payloads, seeds 8021400+p)",
                    "config": {"workload": "BASELINE configs[3] (MCS8 substitution, SURVEY.md §7): VHT MCS8 256-QAM r=3/4 SISO, 1458-byte MPDU, "
                                           "one %d-packet stream on rank 0, contiguous blocks per rank, AWGN 30 dB" % st["packets"],
                               "parallelism": "RCCL scatter -> c8p_rx per rank in sub-batches of %d -> all_gather of the result arenas" % st["sub_batch"]},
                    "sharded_stream": st}, args)
        return

    # ---- synthetic workload, resident in HBM before anything is timed (distinct MPDUs from the build's own transmitter
    # c8p_tx, per-packet flat-channel phase + AWGN on the device, c8p_signal to the signal -> demod boundary)
    kind = {"cfg4r": "cfg4"}.get(args.workload, args.workload)
    pk, iq_dev, chan_dev, expected, _keep = workload_on_device(kind, args.batch, args.snr, 100000 * rank, torch, dev, local)
    n_pkt = len(pk)
    other = args.workload != "cfg2"        # not the headline config: algorithmic bytes from the batch itself (SURVEY §8d formula)
    b_alg = float(np.mean(8.0 * pk["n_ant"] * pk["n_items"] + 512 + 32) + np.mean([len(e) for e in expected])) if other else float(B_ALG)

    # engine-owned streams: consecutive batches are software-pipelined across C8P_SLOTS slots (upload / classify / symbol demod of
    # batch i+1 overlap the decoder of batch i); eng.sync() joins all of them.  HIP events bracket each kernel on every 4th submit
    eng = engine.Engine(device=local, profile=0 if os.environ.get("C8P_BENCH_NOPROFILE") else PROFILE_EVERY)

    def barrier():
        eng.sync()
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # correctness gate (outside the timed region): every published PDU is the transmitted MPDU
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
    assert args.no_check or args.snr < 25.0 or n_ok >= 0.99 * n_pkt, \
        "correctness gate: only %d of %d packets delivered their MPDU at %.0f dB" % (n_ok, n_pkt, args.snr)
    # clock pre-roll (untimed, independent of --warmup): the PDU check leaves the GPU idle for ~50 ms and its shader clock needs some
    # tens of ms of load to come back up
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < 0.15:
        eng.rx(pk, iq_dev, chan_dev)
    for _ in range(args.warmup):
        eng.rx(pk, iq_dev, chan_dev)
    eng.sync()
    eng.profile_reset()

    # the timed region: EXACTLY --steps steps between two barriers (+ device syncs), max over ranks; repeated so that the headline is
    # the MEDIAN of the repeats with min / max beside it (the estimator of every leg: median of >= 3 windows)
    rep_elapsed, rank_elapsed = [], []               # max over the ranks / this rank's own clock, per repeat
    n_rep, r_i = max(3, args.repeats), 0
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
            # a K-step region of a few tens of ms says little to an outside observer (the driver's SMI sampler): repeat it until the
            # timed wall adds up to --min-timed-s (every rank derives the same count from the reduced first sample)
            n_rep = min(2000, max(n_rep, int(np.ceil(args.min_timed_s / max(el, 1e-6)))))
    elapsed = float(np.median(rep_elapsed))
    per_rank = [n_pkt * args.steps / float(np.median(rank_elapsed))]
    if use_dist:
        t = torch.zeros(world, dtype=torch.float64, device=dev)
        t[rank] = per_rank[0]
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        per_rank = [float(v) for v in t.tolist()]
    kt = eng.kernel_times()
    eng.close()                              # its arenas (one set per slot, sized for the worst case per packet) are not needed any more

    # the dominant kernel's average launch duration, measured live: the same batch on an engine with ONE slot (one stream: every launch
    # has the GPU to itself), HIP events on the engine's stream around every launch, 40 steps back to back after 8 untimed ones — what
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
    ctx = {"torch": torch, "dev": dev, "local": local, "rank": rank, "world": world, "use_dist": use_dist, "pk": pk, "iq_dev": iq_dev,
           "chan_dev": chan_dev, "expected": expected}
    if rank == 0:
        value = n_pkt * world * args.steps / elapsed
        kern = {k: (ms / cnt if cnt else 0.0) for k, (ms, cnt) in kt.items()}
        kern_x = {k: (ms / cnt if cnt else 0.0) for k, (ms, cnt) in kt_excl.items()}
        dom = max(("classify", "demod_sym", "viterbi_fwd", "finish"), key=lambda k: kern_x[k])     # the longest launch when alone
        dom_s, step_s = kern_x[dom] * 1e-3, elapsed / args.steps
        achieved = (b_alg * n_pkt / dom_s) / 1e9 if dom_s > 0 else 0.0
        achieved_step = (b_alg * n_pkt / step_s) / 1e9
        traffic, traffic_src, traffic_step = legs.pmc_traffic(dom, PROFILE_TAG)
        vals = sorted(n_pkt * world * args.steps / e for e in rep_elapsed)
        out = {
            "metric": "RX packets/s (20 MHz VHT MCS7 1458 B)" if not other else "RX packets/s (NOT the headline config: %s)" % WORKLOADS[kind][8],
            "value": value, "unit": "packets/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * step_s, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": ("synthetic: c8p_tx on the device, distinct payloads, AWGN + random-phase flat channel, clean OFDM symbol edges (C8P_TX_NO_WINDOW, DESIGN.md §6)"
                     if not args.no_check else "INVALID (--no-check experiment)"),
            "config": {"workload": ("BASELINE configs[1]: VHT MCS7 64-QAM r=5/6 SISO, 1458-byte MPDU" if not other else "BASELINE " + WORKLOADS[kind][8]) +
                                   ", %d packets/GPU/step, AWGN %.0f dB" % (n_pkt, args.snr),
                       "packets_per_gpu_per_step": n_pkt, "fcs_ok_fraction": n_ok / max(n_pkt, 1), "decoder_path": dec_path,
                       "parallelism": "packet-sharded x%d, no data-path collective" % world},
            "per_rank_packets_per_s": {"values": per_rank, "min": min(per_rank), "max": max(per_rank)},
            "repeats": {"n": len(rep_elapsed), "value_median": value, "value_min": vals[0], "value_max": vals[-1],
                        "timed_region_ms": [round(1e3 * e, 3) for e in rep_elapsed]},
            # achieved / frac = the algorithmic bytes of a launch / the dominant kernel's average launch duration (the prescribed figure).
            # The same bytes over ms_per_step (frac_step) are slightly HIGHER: launches of consecutive batches overlap
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "launch_ms": kern_x[dom], "achieved_step": achieved_step, "frac_step": achieved_step / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_whole_step": traffic_step, "traffic_source": traffic_src, "profile": "profiles/" + PROFILE_TAG,
                         "alg_bytes_per_packet": b_alg, "packets_per_launch": n_pkt,
                         "binding_resource": "VALU issue slots, not HBM (see roofline_valu in the detail); the HBM figure is the prescribed one",
                         "exclusive_avg_launch_ms": {k: round(v, 4) for k, v in kern_x.items()},
                         "pipelined_avg_launch_ms": {k: round(v, 4) for k, v in kern.items()},
                         "residency_3slots": legs.residency_3slots(PROFILE_TAG)},
            "startup_s": round(time.perf_counter() - t_start, 1),
        }
        rv = None if other else legs.valu_roofline(kern_x, n_pkt, dec_path["float_step_block_fraction"], PROFILE_TAG)
        if rv:
            out["roofline_valu"] = rv

    def emit_short(o):                                   # (what a leg's watchdog prints if the leg hangs: the headline as it stands)
        bench_line.finish(o, args)

    full = args.legs == "full"
    if world == 1 and rank == 0 and not (use_dist and os.environ.get("C8P_TEST_MULTI_LEG") == "1"):
        deadline = None if full else time.perf_counter() + args.legs_budget_s
        legs.run_single_gpu(ctx, out, args, full, deadline, emit_short)
    bindings = [binding]
    if use_dist:
        bindings = [None] * world
        dist.all_gather_object(bindings, binding)
        if world > 1 or os.environ.get("C8P_TEST_MULTI_LEG") == "1":
            legs.run_multi_gpu(ctx, out, args, emit_short)
    if rank == 0:
        out["host_binding"] = bindings
        if not args.no_cpu and world == 1:      # the CPU leg runs on rank 0 at N = 1 only
            iq, chan = legs.arenas_to_host(torch, dev, iq_dev, chan_dev)
            t_cpu = time.perf_counter()
            out["cpu_baseline"] = cpu_baseline(pk, iq, chan, args.cpu_budget_s)
            out["speedup_vs_cpu_1thread"] = out["value"] / out["cpu_baseline"]["value"]
            try:                                     # (the extra CPU legs must not cost the line)
                out["cpu_baseline_all_cores"] = legs.cpu_baseline_all_cores(pk, iq, chan, out["cpu_baseline"]["value"], budget_s=6.0 if full else 2.0)
                out["speedup_vs_cpu_all_cores"] = out["value"] / out["cpu_baseline_all_cores"]["value"]
                if full and ctx.get("fe_unit") is not None:
                    out["frontend_detect"]["raw_stream_cpu_baseline"] = legs.cpu_baseline_raw_stream(ctx["fe_unit"])
            except Exception as e:
                out["cpu_extra_legs_error"] = "%s: %s" % (type(e).__name__, e)
            out.setdefault("leg_wall_s", {})["cpu"] = round(time.perf_counter() - t_cpu, 2)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        out["wall_s"] = round(time.perf_counter() - t_start, 1)
        bench_line.finish(out, args)


def cpu_baseline(pk, iq, chan, budget_s=10.0):
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
            "sample": "first %d packets of the same batch, oracle demod+decode, 1 thread, %.1f s, %d PDUs" % (n, dt, npdu),
            "build": o.build_flags(), "note": "the oracle's 64-point FFT is an O(N^2) DFT accumulated in double (51 x 4096 complex MACs per "
                                              "packet, ~4 % of its time next to the 11960-step Viterbi)"}


if __name__ == "__main__":
    main()
