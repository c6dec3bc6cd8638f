"""tools/bench_line.py — bench.py's stdout contract: ONE JSON line of at most MAX_LINE bytes (short_line) out of the full record, which
goes to bench_detail.json and stderr (finish); and the launcher side of bench.py: `--gpus N` starting N ranks (launch_ranks) and the
GPU-less rehearsal of everything around the measurement (dry_launch)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAX_LINE = 4096                          # bytes of the stdout line (tests/test_bench_contract.py holds bench.py to it)
_JSON_OUT = None


def load_legs():
    import importlib.util
    if "bench_legs" in sys.modules:
        return sys.modules["bench_legs"]
    spec = importlib.util.spec_from_file_location("bench_legs", os.path.join(ROOT, "tools", "bench_legs.py"))
    m = importlib.util.module_from_spec(spec)
    sys.modules["bench_legs"] = m
    spec.loader.exec_module(m)
    return m


def private_stdout():
    """stdout carries ONE JSON line and nothing else: libraries that chat on file descriptor 1 (RCCL and gloo print banners when a
    communicator is created) are pointed at stderr; the line itself goes to a private duplicate of the original stdout"""
    global _JSON_OUT
    sys.stdout.flush()
    _JSON_OUT = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)



# ------------------------------------------------------------------------------------------------ the line
LEG_PATHS = [   # (name in `legs`, path in the full record) — one number per leg, packets/s unless the name says otherwise; in order of importance
    ("headline", ("value",)), ("cfg1", ("configs", "cfg1", "packets_per_s")), ("cfg3", ("configs", "cfg3", "packets_per_s")),
    ("cfg4_share", ("configs", "cfg4_one_gpu_share", "packets_per_s")), ("cfg5_sta0", ("configs", "cfg5_station0", "packets_per_s")),
    ("cfg5_sta1", ("configs", "cfg5_station1", "packets_per_s")), ("pcie", ("pcie_inclusive_packets_per_s",)),
    ("block_chain", ("block_chain_packets_per_s",)), ("block_fused", ("block_chain_fused_packets_per_s",)),
    ("with_delivery", ("sensitivity", "with_delivery", "packets_per_s")), ("sens_21dB", ("sensitivity", "snr_21dB", "packets_per_s")),
    ("sens_mixed", ("sensitivity", "mixed_formats", "packets_per_s")), ("sens_mixed_air_items_per_s", ("sensitivity", "mixed_formats", "air_items_per_s")),
    ("headline_air_items_per_s", ("sensitivity", "with_delivery", "air_items_per_s")),
    ("sens_windowed", ("sensitivity", "windowed_tx", "packets_per_s")), ("sens_3tap", ("sensitivity", "selective_3tap", "packets_per_s")),
    ("sharded_rccl", ("sharded_stream", "feeds", "rccl_scatter", "packets_per_s")),
    ("detect_samples_per_s", ("frontend_detect", "samples_per_s")), ("tx", ("frontend_detect", "tx_packets_per_s")),
    ("raw_chain", ("frontend_detect", "raw_stream_to_pdus", "packets_per_s")),
    ("raw_chain_pcie", ("frontend_detect", "raw_stream_to_pdus", "pcie_inclusive", "packets_per_s")),
    ("live_streams_at_realtime", ("frontend_detect", "raw_stream_to_pdus", "live_streams", "streams_at_realtime")),
    ("live_streams_sc16_at_realtime", ("frontend_detect", "raw_stream_to_pdus", "live_streams", "sc16_128", "streams_at_realtime")),
    ("live_streams_sc8_at_realtime", ("frontend_detect", "raw_stream_to_pdus", "live_streams", "sc8_128", "streams_at_realtime")),
    ("block_rx_chain", ("block_rx_chain", "packets_per_s")), ("block_rx_chain_sc16", ("block_rx_chain", "sc16_ports", "packets_per_s")),
    ("cpu_1thread", ("cpu_baseline", "value")), ("cpu_all_cores", ("cpu_baseline_all_cores", "value")),
    ("multi_live_streams_total", ("multi_gpu_host_fed", "live_streams_at_realtime_total")),
    ("multi_pcie_total", ("multi_gpu_host_fed", "pcie_inclusive_packets_per_s_total"))]


def dig(obj, path):
    for k in path:
        if not isinstance(obj, dict) or k not in obj:
            return None
        obj = obj[k]
    return obj


def legs_summary(out):
    legs = {}
    for name, path in LEG_PATHS:
        v = dig(out, path)
        if isinstance(v, (int, float)) and not isinstance(v, bool):
            legs[name] = round(float(v), 1)
    return legs


def short_line(out, detail_path=None):
    """the <= MAX_LINE byte stdout line out of the full record: the contract keys, the prescribed roofline / cpu_baseline objects with
    short values, and `legs`"""
    def num(v, nd=4):                                    # floats rounded, strings and lists bounded: nothing a leg adds can lengthen the line
        if isinstance(v, float):
            return round(v, nd)
        if isinstance(v, str) and len(v) > 200:
            return v[:197] + "..."
        if isinstance(v, list) and len(v) > 8:
            return v[:8] + ["+%d more" % (len(v) - 8)]
        return v
    line = {k: num(out.get(k)) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                         "vs_baseline", "dtype", "data")}
    line["value"], line["ms_per_step"] = num(line["value"], 1), num(line["ms_per_step"], 5)
    cfg = out.get("config") or {}
    line["config"] = {k: num(cfg[k]) for k in ("workload", "packets_per_gpu_per_step", "fcs_ok_fraction", "parallelism") if k in cfg}
    if out.get("dry_launch"):
        line["dry_launch"] = True
    pr, rp = out.get("per_rank_packets_per_s"), out.get("repeats")
    if pr:
        line["per_rank_packets_per_s"] = {"min": num(pr["min"], 1), "max": num(pr["max"], 1)}
    if rp:
        line["repeats"] = {"n": rp["n"], "estimator": "median", "value_min": num(rp["value_min"], 1), "value_max": num(rp["value_max"], 1)}
    r = out.get("roofline")
    if r:
        line["roofline"] = {k: num(r.get(k), 5) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "launch_ms", "frac_step", "traffic",
                                                          "traffic_whole_step", "traffic_source", "profile", "alg_bytes_per_packet", "packets_per_launch")}
    c = out.get("cpu_baseline")
    if c:
        line["cpu_baseline"] = {k: num(c.get(k), 2) for k in ("value", "unit", "cores", "kind", "sample")}
        line["speedup_vs_cpu_1thread"] = num(out.get("speedup_vs_cpu_1thread"), 1)
    if out.get("n_gpus", 1) > 1:
        line["multi_gpu"] = "measured at world %d" % out["n_gpus"] if not out.get("dry_launch") else "dry launch: UNMEASURED at N > 1"
    elif "sharded_stream" in out:
        line["multi_gpu"] = "unmeasured at N > 1 (this run: world 1; sharded_rccl = configs[3] through RCCL at world 1)"
    for k in ("legs_skipped",):
        if out.get(k):
            line[k] = num(out[k])
    errs = sorted(k for k in out if k.endswith("_error")) + sorted(k for k, v in out.items() if isinstance(v, dict) and "error" in v)
    if errs:
        line["legs_failed"] = num(errs)
    if "wall_s" in out:
        line["wall_s"] = out["wall_s"]
    line["detail"] = os.path.relpath(detail_path, ROOT) if detail_path else None
    line["legs"] = legs_summary(out)                     # last key: what survives when only the tail of stdout is kept
    names = [n for n, _ in LEG_PATHS]
    while len(json.dumps(line)) > MAX_LINE and line["legs"]:      # (cannot happen with the keys above; the bound is kept whatever a leg adds)
        line["legs"].pop(max(line["legs"], key=names.index))
    return line


def finish(out, args):
    """rank 0: the full record to --detail (+ gpurun_out/ when bench.py runs from the repository on a GPU box) and to stderr; the short
    line to stdout"""
    out["legs"] = legs_summary(out)
    paths = [args.detail] if args.detail else [os.path.join(ROOT, "bench_detail.json"), os.path.join(ROOT, "gpurun_out", "bench_detail.json")]
    written = None
    for p in paths:
        try:
            os.makedirs(os.path.dirname(os.path.abspath(p)), exist_ok=True)
            with open(p, "w") as f:
                json.dump(out, f)
            written = written or p
        except OSError as e:
            sys.stderr.write("bench.py: could not write %s: %s\n" % (p, e))
    sys.stderr.write("bench.py detail: " + json.dumps(out) + "\n")
    sys.stderr.flush()
    f = _JSON_OUT or sys.stdout
    f.write(json.dumps(short_line(out, written)) + "\n")
    f.flush()


# ------------------------------------------------------------------------------------------------ the launcher
def launch_ranks(args, bench_py):
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
           "--master-port", str(port), os.path.abspath(bench_py)] + sys.argv[1:]
    rc = subprocess.call(cmd, env=env)
    if rc != 0:
        raise SystemExit("bench.py: the %d-rank launch failed with status %d" % (args.gpus, rc))


def dry_launch(args):
    """--dry-launch: everything around the measurement without a GPU — the ranks the launcher started rendezvous over gloo, the timed
    region is bracketed by the same barriers, the elapsed time is the max over the ranks, every rank's own rate and host binding are
    gathered, rank 0 prints ONE short line with n_gpus = the world that actually ran.  A sleep stands in for the step: no value."""
    private_stdout()
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    use_dist = "RANK" in os.environ and "MASTER_PORT" in os.environ
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")
        if dist.get_world_size() != args.gpus:
            raise SystemExit("bench.py: the communicator has %d ranks, --gpus says %d" % (dist.get_world_size(), args.gpus))
    # what every rank does before it touches its GPU: find the device's NUMA node, bind to it (no GPU here: nothing to find, and the
    # record says so), check that one HIP runtime is mapped; and the partition of the live streams over the ranks (shard.LiveShard's)
    sys.path.insert(0, ROOT) if ROOT not in sys.path else None
    from __graft_entry__ import load_package
    load_package()
    from gr_ieee80211_amd import hostbind, shard
    binding = hostbind.bind_rank(int(os.environ.get("LOCAL_RANK", "0")), torch)
    binding["rank"] = rank
    binding["hip_runtimes_mapped"] = hostbind.assert_single_hip_runtime()
    lo, hi = shard.partition(64 * world, world, rank)
    my_leg = {"rank": rank, "live_streams": {"global_streams": [lo, hi, 64 * world], "streams": hi - lo, "streams_at_realtime": 0.0, "pcie_GBps": 0.0},
              "pcie_inclusive_packets_per_s": 0.0}
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
    bindings, legs_ = [binding], [my_leg]
    if use_dist:
        t = torch.tensor([el], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
        v = torch.zeros(world, dtype=torch.float64)
        v[rank] = per_rank[0]
        dist.all_reduce(v, op=dist.ReduceOp.SUM)
        per_rank = [float(x) for x in v.tolist()]
        bindings, legs_ = [None] * world, [None] * world
        dist.all_gather_object(bindings, binding)
        dist.all_gather_object(legs_, my_leg)
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        mg = load_legs().summarize_host_fed(legs_)
        mg.update({"dry": True, "note": "dry launch: the partition of the live streams over the ranks and the gathering of the per-rank records; "
                                        "no engine, no figures — UNMEASURED at N > 1"})
        finish({"metric": "RX packets/s (20 MHz VHT MCS7 1458 B)", "value": None, "unit": "packets/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": 1e3 * el / max(args.steps, 1), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "f32", "data": "none (dry launch)", "dry_launch": True, "host_binding": bindings, "multi_gpu_host_fed": mg,
                "per_rank_packets_per_s": {"values": per_rank, "min": min(per_rank), "max": max(per_rank), "note": "sleeps, not packets"},
                "config": {"workload": "dry launch: launcher, rendezvous, barriers and reductions only; NOT a measurement",
                           "packets_per_gpu_per_step": n_pkt, "parallelism": "packet-sharded x%d, no data-path collective" % world}}, args)


