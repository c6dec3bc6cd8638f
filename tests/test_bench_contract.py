"""The bench line's contract.  (1) The newest committed line (profiles/*/bench.json — written by `python bench.py` on an MI355X): the
keys the driver and the judge read are there, the roofline object is consistent with the headline, the CPU baseline is bounded.
(2) The DRIVER's own records (BENCH_rNN.json at the repository root, written after each round on a box the builder never touched):
every leg that can be read out of what the driver kept — the parsed contract keys and the tails of stdout — within a stated factor of
the line committed in the same round, failing with the leg's name otherwise; a regression that is known and answered is listed in
ACKNOWLEDGED with the answer, anything else fails.  (3) `bench.py --gpus N` starts N ranks by itself (--dry-launch: no GPU)."""
import glob
import json
import os
import re
import subprocess
import sys

import pytest

from c8p_testlib import ROOT

FACTOR = 0.7          # a leg of the driver's run may be at most this much worse than the committed line's


def newest_line():
    fs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "bench.json")))
    assert fs, "no committed bench line"
    return json.load(open(fs[-1])), fs[-1]


def test_bench_line_carries_the_contract_keys():
    j, f = newest_line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in j, (f, k)
    assert j["unit"] == "packets/s" and j["higher_is_better"] is True and j["scaling"] == "weak" and j["vs_baseline"] is None and j["dtype"] == "f32"
    assert "workload" in j["config"] and "model" not in j["config"]
    assert "symbol-edge windowing" in j["config"]["workload"] or "C8P_TX_NO_WINDOW" in j["data"]      # the de-windowed TX is stated
    r = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    # achieved / frac: the algorithmic bytes of a launch over the dominant kernel's average launch duration (lines up to profiles/r5t: over
    # ms_per_step, the exclusive figure beside it as achieved_exclusive); the other of the two is in the object as well
    bytes_launch = r["alg_bytes_per_packet"] * r["packets_per_launch"]
    want_step = bytes_launch / (j["ms_per_step"] * 1e-3) / 1e9
    assert abs(r["achieved_step"] - want_step) / want_step < 1e-6
    assert abs(r["achieved"] - r["achieved_exclusive"]) < 1e-9 or abs(r["achieved"] - r["achieved_step"]) < 1e-9
    k_ms = r["exclusive_avg_launch_ms"][r["kernel"]]
    assert abs(r["achieved_exclusive"] - bytes_launch / (k_ms * 1e-3) / 1e9) / r["achieved_exclusive"] < 2e-3      # (k_ms is rounded to 0.1 us in the line)
    assert abs(j["value"] - j["n_gpus"] * r["packets_per_launch"] / (j["ms_per_step"] * 1e-3)) / j["value"] < 1e-6
    c = j["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["cores"] == 1 and 100 < c["value"] < 5000


def test_bench_line_reports_every_other_baseline_config_and_the_pipelined_stream():
    j, f = newest_line()
    cfg = j.get("configs", {})
    for k in ("cfg1", "cfg3", "cfg3_1024_packets", "cfg4_one_gpu_share", "cfg5_station0", "cfg5_station1"):
        assert k in cfg and "error" not in cfg[k], (f, k, cfg.get(k))
        for q in ("packets_per_s", "ms_per_step", "alg_bytes_per_packet", "frac_step", "fcs_ok_fraction", "workload"):
            assert q in cfg[k], (k, q)
        assert cfg[k]["fcs_ok_fraction"] >= 0.97
    assert cfg["cfg3"]["packets_per_s"] >= 0.9 * j["value"]          # the two-stream path is on chip: within 10 % of the headline
    s = j["sharded_stream"]
    assert s["world"] == 1 and s["rccl_scatter_over_resident"] >= 0.95 and "timeline_ms" in s["feeds"]["rccl_scatter"]
    assert "host_phase_us" in j and j["block_chain_packets_per_s"] >= 200 * j["cpu_baseline"]["value"]


# ------------------------------------------------------------------------------------------------ the driver's records
def flatten(obj, pre=()):
    out = {}
    if isinstance(obj, dict):
        for k, v in obj.items():
            out.update(flatten(v, pre + (k,)))
    elif isinstance(obj, (int, float)) and not isinstance(obj, bool):
        out[pre] = float(obj)
    return out


def scan_fragment(text):
    """{key path (as far as the fragment shows it): number} out of a PIECE of a JSON line — the driver keeps the tails of stdout, cut
    anywhere.  Scanning starts at the first key that follows a complete value; `"key": {` pushes, `}` pops, numbers are recorded,
    strings and arrays are skipped."""
    m = re.search(r'[\]}0-9el"], "', text)
    if not m:
        return {}
    i, n = m.end() - 1, len(text)
    stack, out = [], {}

    def string_at(i):           # text[i] == '"': returns (value, index after the closing quote) or (None, n)
        j = i + 1
        while j < n:
            if text[j] == "\\":
                j += 2
                continue
            if text[j] == '"':
                return text[i + 1:j], j + 1
            j += 1
        return None, n
    while i < n:
        c = text[i]
        if c == '"':
            key, j = string_at(i)
            if key is None:
                break
            k = j
            while k < n and text[k] == " ":
                k += 1
            if k < n and text[k] == ":":
                k += 1
                while k < n and text[k] == " ":
                    k += 1
                if k >= n:
                    break
                if text[k] == "{":
                    stack.append(key)
                    i = k + 1
                    continue
                num = re.match(r"-?[0-9][0-9.eE+-]*", text[k:])
                if num:
                    end = k + num.end()
                    if end < n:                      # (a number cut off by the end of the fragment is not a number)
                        out[tuple(stack) + (key,)] = float(num.group(0))
                    i = end
                    continue
                if text[k] == "[":
                    depth, k = 1, k + 1
                    while k < n and depth:
                        if text[k] == '"':
                            _, k = string_at(k)
                            continue
                        depth += text[k] == "["
                        depth -= text[k] == "]"
                        k += 1
                    i = k
                    continue
                i = k                                # a string value (or true / false / null): the loop skips it
                continue
            i = j
            continue
        if c == "}" and stack:
            stack.pop()
        i += 1
    return out


# legs by the END of their key path; True: bigger is better
LEGS = [(("legs", "headline"), True), (("legs", "cfg1"), True), (("legs", "cfg3"), True), (("legs", "cfg4_share"), True), (("legs", "cfg5_sta0"), True),
        (("legs", "sharded_rccl"), True), (("legs", "pcie"), True), (("legs", "block_chain"), True), (("legs", "block_fused"), True),
        (("legs", "block_rx_chain"), True),
        (("legs", "detect_samples_per_s"), True), (("legs", "tx"), True), (("legs", "raw_chain"), True), (("legs", "raw_chain_pcie"), True),
        (("legs", "live_streams_at_realtime"), True), (("legs", "live_streams_sc16_at_realtime"), True), (("legs", "live_streams_sc8_at_realtime"), True),
        (("value",), True), (("raw_stream_to_pdus", "packets_per_s"), True), (("raw_stream_to_pdus", "ms_per_segment"), False),
        (("raw_stream_to_pdus", "pcie_inclusive", "ms_per_segment"), False), (("raw_stream_to_pdus", "pcie_inclusive", "packets_per_s"), True),
        (("frontend_detect", "samples_per_s"), True), (("pcie_inclusive_packets_per_s",), True), (("block_chain_packets_per_s",), True),
        (("block_chain_fused_packets_per_s",), True), (("cfg3", "packets_per_s"), True), (("cfg1", "packets_per_s"), True),
        (("rccl_scatter", "packets_per_s"), True), (("cpu_baseline", "value"), True)]

# a driver record's regressions that are known, with what was done about them (anything not listed here fails the test)
_R3_RAWCHAIN = ("round 3: c8p_rawchain_run made four fresh engine contexts per call and grew every arena (hipMalloc / hipHostMalloc behind stream "
                "syncs) inside its timed window; on the driver's box that cost 10x (VERDICT r3 weak 4).  Round 4: one c8p_rawchain handle that "
                "outlives the runs, arenas reserved at create, the warm-up runs through the same contexts, and the leg reports first-segment "
                "time, per-stage clocks, arena growths and look-back give-ups (bench.py chain_report)")
ACKNOWLEDGED = {
    "BENCH_r03.json": {k: _R3_RAWCHAIN for k in ("raw_stream_to_pdus.packets_per_s", "raw_stream_to_pdus.ms_per_segment",
                                                 "raw_stream_to_pdus.pcie_inclusive.ms_per_segment", "raw_stream_to_pdus.pcie_inclusive.packets_per_s")},
}


def find(flat, suffix):
    hits = [v for k, v in flat.items() if k[-len(suffix):] == suffix]
    return hits[0] if hits else None


def regressions(driver_flat, committed_flat):
    bad = []
    for suffix, bigger in LEGS:
        d, c = find(driver_flat, suffix), find(committed_flat, suffix)
        if d is None or c is None or c <= 0 or d <= 0:
            continue
        ratio = d / c if bigger else c / d
        if ratio < FACTOR:
            bad.append((".".join(suffix), d, c, round(ratio, 3)))
    return bad


def driver_records():
    return sorted(glob.glob(os.path.join(ROOT, "BENCH_r[0-9][0-9].json")))


def committed_for_round(n):
    """the newest line committed in round n (profiles/r<n><letter>/bench.json), else the newest older one"""
    for r in range(n, 0, -1):
        fs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r%d[a-z]*" % r, "bench.json")))
        if fs:
            return fs[-1]
    return None


def test_fragment_scanner_reads_a_line_cut_anywhere():
    j, _ = newest_line()
    line = json.dumps(j)
    whole = flatten(j)
    for cut in (0, 17, len(line) // 3, len(line) // 2, len(line) - 2500, len(line) - 900):
        got = scan_fragment(line[max(cut, 0):])
        assert len(got) >= 5, cut
        for path, v in got.items():                      # everything the scanner reports is a number of the line, under the same key path's end
            assert any(k[-len(path):] == path and abs(w - v) <= 1e-9 * max(1.0, abs(w)) for k, w in whole.items()), (cut, path, v)


def test_the_drivers_own_lines_are_within_a_factor_of_the_committed_ones():
    recs = driver_records()
    if not recs:
        return
    checked = 0
    for f in recs:
        name = os.path.basename(f)
        d = json.load(open(f))
        n = int(re.search(r"r(\d+)", name).group(1))
        cf = committed_for_round(n)
        if cf is None:
            continue
        flat = {("parsed",) + k: v for k, v in flatten(d.get("parsed") or {}).items()}
        flat = {k[1:]: v for k, v in flat.items()}        # parsed holds the line's own top-level keys
        for piece in ((d.get("run") or {}).get("stdout_tail") or "", (d.get("tail") or "").split("---- stderr ----")[0]):
            for k, v in scan_fragment(piece).items():
                flat.setdefault(k, v)
        bad = regressions(flat, flatten(json.load(open(cf))))
        known = ACKNOWLEDGED.get(name, {})
        new = [b for b in bad if b[0] not in known]
        assert not new, "%s against %s: %s" % (name, os.path.relpath(cf, ROOT), new)
        stale = [k for k in known if k not in [b[0] for b in bad]]
        assert not stale, "%s: acknowledged regressions that the check no longer sees: %s" % (name, stale)
        checked += 1
    assert checked == len([f for f in recs if committed_for_round(int(re.search(r"r(\d+)", os.path.basename(f)).group(1)))])


# ------------------------------------------------------------------------------------------------ the launcher
def run_bench(*argv):
    env = dict(os.environ)
    env.pop("RANK", None); env.pop("WORLD_SIZE", None); env.pop("MASTER_PORT", None); env.pop("LOCAL_RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), capture_output=True, text=True, timeout=300, env=env)
    return p


def test_bench_gpus_n_starts_n_ranks_by_itself():
    """`python bench.py --gpus 2` with no launcher around it: two ranks under torch.distributed.run, rendezvous on 127.0.0.1, ONE line on
    stdout with n_gpus = 2 and both ranks' own rates (--dry-launch: gloo, a sleep for the step, no value)"""
    p = run_bench("--gpus", "2", "--dry-launch", "--steps", "5", "--warmup", "1")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["dry_launch"] is True and j["value"] is None and j["steps"] == 5 and j["warmup"] == 1
    pr = j["per_rank_packets_per_s"]
    assert len(pr["values"]) == 2 and pr["min"] == min(pr["values"]) > 0 and pr["values"][0] > pr["values"][1]       # (rank 1 sleeps twice as long)
    assert j["ms_per_step"] >= 1e3 * 4096 / pr["min"] * 0.99                                                          # the slowest rank sets the time
    # round 5: every rank reports where its host side runs (hostbind.bind_rank: the GPU's NUMA node, or that there is none to find) and
    # that ONE HIP runtime is mapped; the live leg's streams are partitioned over the ranks as shard.LiveShard does; the line still says
    # that nothing was measured
    hb = j["host_binding"]
    assert [b["rank"] for b in hb] == [0, 1] and all("numa_node" in b and "affinity" in b and len(b["hip_runtimes_mapped"]) <= 1 for b in hb)
    mg = j["multi_gpu_host_fed"]
    assert mg["dry"] is True and "UNMEASURED" in mg["note"]
    assert [r["live_streams"]["global_streams"] for r in mg["per_rank"]] == [[0, 64, 128], [64, 128, 128]]


def test_bench_refuses_a_world_that_is_not_what_was_asked_for():
    """a launcher that started 2 ranks for --gpus 3 (or none for --gpus 1): no line"""
    env = dict(os.environ, RANK="0", WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--dry-launch"], capture_output=True, text=True, timeout=120, env=env)
    assert p.returncode != 0 and "refusing" in p.stderr and not p.stdout.strip()
    p = run_bench("--gpus", "2")                         # no GPUs here: the launcher says so instead of measuring one GPU
    assert p.returncode != 0 and "GPU(s) are visible" in p.stderr and not p.stdout.strip()


@pytest.mark.gpu
def test_multi_gpu_host_fed_leg_runs_under_a_one_rank_launch():
    """bench.py's world > 1 leg (live ingest through shard.LiveShard + the PCIe-inclusive c8p_rx, per-rank records gathered) on the one
    GPU a development box has: a one-rank torch.distributed.run with C8P_TEST_MULTI_LEG=1.  The line carries the rank's figures, the
    NUMA binding record, and every pushed PDU was delivered."""
    import socket
    s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
    env = dict(os.environ, C8P_TEST_MULTI_LEG="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "10", "--warmup", "2", "--repeats", "1", "--min-timed-s", "0", "--no-cpu", "--no-frontend",
           "--no-configs", "--no-sensitivity", "--no-stream"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    j = json.loads([ln for ln in p.stdout.splitlines() if ln.strip().startswith("{")][-1])
    mg = j["multi_gpu_host_fed"]
    assert mg["ranks_ok"] == 1 and "error" not in mg["per_rank"][0], mg
    r = mg["per_rank"][0]
    assert r["live_streams"]["global_streams"] == [0, 64, 64] and r["live_streams"]["streams_at_realtime"] > 50
    lo, hi = r["live_streams"]["pdus_in_the_items_pushed"]
    assert lo <= r["live_streams"]["pdus_ok"] <= hi
    assert r["pcie_inclusive_packets_per_s"] > 3e5
    assert j["host_binding"][0]["rank"] == 0 and "numa_node" in j["host_binding"][0]
