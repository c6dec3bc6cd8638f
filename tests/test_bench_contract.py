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
    """(the short stdout line, the full record beside it, the line's path) of the newest committed run: profiles/r*/bench.json +
    bench_detail.json (round 6 on: bench.py prints <= 4 KB and writes the rest to the detail file)"""
    fs = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r*", "bench.json")) if os.path.exists(os.path.join(os.path.dirname(f), "bench_detail.json")))
    assert fs, "no committed bench line with its bench_detail.json"
    return json.load(open(fs[-1])), json.load(open(os.path.join(os.path.dirname(fs[-1]), "bench_detail.json"))), fs[-1]


CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")
ROOFLINE = ("bound", "kernel", "achieved", "peak", "unit", "frac", "launch_ms", "frac_step", "traffic", "traffic_whole_step", "traffic_source", "profile",
            "alg_bytes_per_packet", "packets_per_launch")
LINE_KEYS = set(CONTRACT) | {"roofline", "cpu_baseline", "speedup_vs_cpu_1thread", "per_rank_packets_per_s", "repeats", "multi_gpu", "legs_skipped",
                             "legs_failed", "wall_s", "detail", "legs", "dry_launch"}


def check_short_line(j, text=None):
    if text is not None:
        assert len(text.encode()) <= 4096, len(text)
    assert len(json.dumps(j)) <= 4096
    for k in CONTRACT:
        assert k in j, k
    assert set(j) <= LINE_KEYS, set(j) - LINE_KEYS                 # nothing else: no prose notes, no per-leg records
    assert "workload" in j["config"] and "model" not in j["config"]
    assert list(j)[-1] == "legs" and all(isinstance(v, (int, float)) for v in j["legs"].values())


def test_bench_line_carries_the_contract_keys():
    j, d, f = newest_line()
    check_short_line(j, open(f).read().strip())
    assert j["unit"] == "packets/s" and j["higher_is_better"] is True and j["scaling"] == "weak" and j["vs_baseline"] is None and j["dtype"] == "f32"
    assert "C8P_TX_NO_WINDOW" in j["data"]      # the de-windowed TX is stated
    r = j["roofline"]
    assert set(r) == set(ROOFLINE)
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5
    # achieved / frac: the algorithmic bytes of a launch over the dominant kernel's average launch duration, measured live (HIP events);
    # frac_step: the same bytes over ms_per_step
    bytes_launch = r["alg_bytes_per_packet"] * r["packets_per_launch"]
    assert r["alg_bytes_per_packet"] == 37206 and r["packets_per_launch"] == 4096
    assert abs(r["achieved"] - bytes_launch / (r["launch_ms"] * 1e-3) / 1e9) / r["achieved"] < 1e-3
    assert abs(r["frac_step"] - bytes_launch / (j["ms_per_step"] * 1e-3) / 1e9 / r["peak"]) / r["frac_step"] < 1e-3
    assert abs(j["value"] - j["n_gpus"] * r["packets_per_launch"] / (j["ms_per_step"] * 1e-3)) / j["value"] < 1e-4
    assert r["traffic"] and r["traffic"] > bytes_launch and r["profile"].startswith("profiles/r") and os.path.isdir(os.path.join(ROOT, r["profile"]))
    c = j["cpu_baseline"]
    assert set(c) == {"value", "unit", "cores", "kind", "sample"}
    assert c["kind"] == "port" and c["cores"] == 1 and 100 < c["value"] < 5000 and c["unit"] == "packets/s"
    assert j["repeats"]["n"] >= 3 and j["repeats"]["value_min"] <= j["value"] <= j["repeats"]["value_max"]      # the median of >= 3 windows
    assert j["legs"]["headline"] == round(j["value"], 1)
    # the full record: the same headline, the same roofline figures, and the line names it
    assert d["value"] == pytest.approx(j["value"], rel=1e-6) and d["roofline"]["achieved"] == pytest.approx(r["achieved"], rel=1e-4)
    assert j["detail"].endswith("bench_detail.json")


def test_every_leg_uses_one_estimator():
    """every packets/s figure of the record is the median of >= 3 timed windows with min / max beside it (the headline
    was a median, the sensitivity legs the better of two windows, the PCIe leg the best of three)"""
    j, d, f = newest_line()
    seen = 0
    for grp in ("configs", "sensitivity"):
        for name, leg in d.get(grp, {}).items():
            if isinstance(leg, dict) and "packets_per_s" in leg:
                w = leg["windows_packets_per_s"]
                assert w["n"] >= 3 and w["min"] <= leg["packets_per_s"] <= w["max"] and leg["packets_per_s"] == pytest.approx(w["median"]), (grp, name)
                seen += 1
    w = d["pcie_inclusive_detail"]["windows_packets_per_s"]
    assert w["n"] >= 3 and w["min"] <= d["pcie_inclusive_packets_per_s"] <= w["max"]
    assert seen >= 8
    # a leg that shares the headline's work cannot beat it by more than the spread of the windows
    for name in ("snr_21dB", "windowed_tx", "selective_3tap", "with_delivery"):
        if name in d.get("sensitivity", {}) and "vs_headline" in d["sensitivity"][name]:
            assert d["sensitivity"][name]["vs_headline"] < 1.08, name


def test_bench_record_reports_every_other_baseline_config_and_the_pipelined_stream():
    j, d, f = newest_line()
    cfg = d.get("configs", {})
    for k in ("cfg1", "cfg3", "cfg4_one_gpu_share", "cfg5_station0", "cfg5_station1"):
        assert k in cfg and "error" not in cfg[k] and "skipped" not in cfg[k], (f, k, cfg.get(k))
        for q in ("packets_per_s", "ms_per_step", "alg_bytes_per_packet", "frac_step", "fcs_ok_fraction", "workload"):
            assert q in cfg[k], (k, q)
        assert cfg[k]["fcs_ok_fraction"] >= 0.97
    assert cfg["cfg3"]["packets_per_s"] >= 0.9 * d["value"]          # the two-stream path is on chip: within 10 % of the headline
    s = d["sharded_stream"]
    assert s["world"] == 1 and s["rccl_scatter_over_resident"] >= 0.95 and "timeline_ms" in s["feeds"]["rccl_scatter"]
    assert "host_phase_us" in d and d["block_chain_packets_per_s"] >= 200 * d["cpu_baseline"]["value"]
    for k in ("cfg1", "cfg3", "cfg4_share", "cfg5_sta0", "pcie", "block_chain", "block_fused", "sharded_rccl", "cpu_1thread"):
        assert k in j["legs"], k


def test_short_line_stays_short_whatever_the_record_holds():
    """tools/bench_line.short_line on a record with every leg present, eight ranks, long strings and failed legs: <= 4096 bytes, `legs` last"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_line
    j, d, f = newest_line()
    big = json.loads(json.dumps(d))
    big["n_gpus"] = 8
    big["per_rank_packets_per_s"] = {"values": [6.5e6 + i for i in range(8)], "min": 6.5e6, "max": 6.5e6 + 7}
    big["host_binding"] = [{"rank": i, "note": "x" * 500} for i in range(8)]
    big["multi_gpu_host_fed"] = {"per_rank": [{"rank": i, "blob": "y" * 2000} for i in range(8)], "live_streams_at_realtime_total": 2400.0,
                                 "pcie_inclusive_packets_per_s_total": 1.1e7}
    for i in range(40):
        big["made_up_leg_%d_error" % i] = "z" * 300
    big["legs_skipped"] = ["sensitivity", "sharded_stream", "frontend_detect"]
    for name, path in bench_line.LEG_PATHS:                            # every leg of the table present
        o = big
        for k in path[:-1]:
            o = o.setdefault(k, {}) if isinstance(o.get(k), dict) or k not in o else o[k]
        if isinstance(o, dict):
            o.setdefault(path[-1], 123456789.123)
    line = bench_line.short_line(big, os.path.join(ROOT, "bench_detail.json"))
    text = json.dumps(line)
    assert len(text) <= bench_line.MAX_LINE
    assert list(line)[-1] == "legs" and line["legs"]["headline"] == round(d["value"], 1) and "roofline" in line and "cpu_baseline" in line
    assert len(line["legs"]) >= 20


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


# legs by key path; True: bigger is better.  A path of ONE key is a top-level key of the line and is looked up only in complete documents
# (the driver's `parsed`, a committed line) — never in a fragment, where `cpu_baseline.value` cut after `cpu_baseline` would pass for it
# (that is how BENCH_r05.json, whose line the driver could not parse, read as a 0.00005x regression of the headline).  Longer paths match
# by their END, in fragments too: the lines of rounds 1-5 nest them at different depths.
LEGS = [(("legs", "headline"), True), (("legs", "cfg1"), True), (("legs", "cfg3"), True), (("legs", "cfg4_share"), True), (("legs", "cfg5_sta0"), True),
        (("legs", "sharded_rccl"), True), (("legs", "pcie"), True), (("legs", "block_chain"), True), (("legs", "block_fused"), True),
        (("legs", "block_rx_chain"), True),
        (("legs", "detect_samples_per_s"), True), (("legs", "tx"), True), (("legs", "raw_chain"), True), (("legs", "raw_chain_pcie"), True),
        (("legs", "live_streams_at_realtime"), True), (("legs", "live_streams_sc16_at_realtime"), True), (("legs", "live_streams_sc8_at_realtime"), True),
        (("legs", "with_delivery"), True), (("legs", "sens_21dB"), True), (("legs", "sens_mixed"), True), (("legs", "cpu_1thread"), True),
        (("value",), True), (("raw_stream_to_pdus", "packets_per_s"), True), (("raw_stream_to_pdus", "ms_per_segment"), False),
        (("raw_stream_to_pdus", "pcie_inclusive", "ms_per_segment"), False), (("raw_stream_to_pdus", "pcie_inclusive", "packets_per_s"), True),
        (("frontend_detect", "samples_per_s"), True), (("pcie_inclusive_packets_per_s",), True), (("block_chain_packets_per_s",), True),
        (("block_chain_fused_packets_per_s",), True), (("cfg3", "packets_per_s"), True), (("cfg1", "packets_per_s"), True),
        (("rccl_scatter", "packets_per_s"), True), (("cpu_baseline", "value"), True), (("roofline", "achieved"), True)]

# a driver record's regressions that are known, with what was done about them (anything not listed here fails the test)
_R3_RAWCHAIN = ("round 3: c8p_rawchain_run made four fresh engine contexts per call and grew every arena (hipMalloc / hipHostMalloc behind stream "
                "syncs) inside its timed window; on the driver's box that cost 10x (VERDICT r3 weak 4).  Round 4: one c8p_rawchain handle that "
                "outlives the runs, arenas reserved at create, the warm-up runs through the same contexts, and the leg reports first-segment "
                "time, per-stage clocks, arena growths and look-back give-ups (bench.py chain_report)")
ACKNOWLEDGED = {
    "BENCH_r03.json": {k: _R3_RAWCHAIN for k in ("raw_stream_to_pdus.packets_per_s", "raw_stream_to_pdus.ms_per_segment",
                                                 "raw_stream_to_pdus.pcie_inclusive.ms_per_segment", "raw_stream_to_pdus.pcie_inclusive.packets_per_s")},
}


def find(whole, frag, path):
    """`path` in a complete document's flattened keys (`whole`) or, failing that and only for paths of two keys or more, at the END of a
    key path scanned out of a fragment (`frag`)"""
    if len(path) == 1:
        return whole.get(path)
    hits = [v for k, v in whole.items() if k[-len(path):] == path] or [v for k, v in frag.items() if k[-len(path):] == path]
    return hits[0] if hits else None


# legs whose figure is made by host threads and the PCIe link of a SHARED box (two scheduler threads on two cores, pinned-memory rings):
# one driver run in three on this pool meets a neighbour's burst that takes a whole 2 s leg down to a third (gpurun_out/r6z2: 735 k,
# 215 k, 730 k packets/s fused, the headline of the same three runs 6.56 / 6.57 / 6.57 M).  They are compared with a factor that still
# catches a broken path (an order of magnitude) and does not turn the host's noise into a red suite.
HOST_BOUND = ("block_chain", "block_fused", "block_rx_chain", "pcie", "raw_chain_pcie", "live_streams", "pcie_inclusive", "ms_per_segment")
HOST_FACTOR = 0.25


# ... and the legs with a host thread in their loop (c8p_detect returns the events: every call is synchronous; the raw chain's three stage
# threads; the RCCL scatter's host-side posting) move 20-30 % from box to box: half is still a broken path, not a slow neighbour
HOST_IN_LOOP = ("detect", "samples_per_s", "raw_chain", "raw_stream_to_pdus", "tx", "sharded", "rccl")
MID_FACTOR = 0.5


def factor_of(path):
    if any(h in k for k in path for h in HOST_BOUND):
        return HOST_FACTOR
    return MID_FACTOR if any(h in k for k in path for h in HOST_IN_LOOP) else FACTOR


def regressions(driver_whole, driver_frag, committed_flat):
    bad = []
    for path, bigger in LEGS:
        d, c = find(driver_whole, driver_frag, path), find(committed_flat, {}, path)
        if d is None or c is None or c <= 0 or d <= 0:
            continue
        ratio = d / c if bigger else c / d
        if ratio < factor_of(path):
            bad.append((".".join(path), d, c, round(ratio, 3)))
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


def old_line():
    """a line of the long format (rounds 1-5): what the driver's records of those rounds hold pieces of"""
    return json.load(open(os.path.join(ROOT, "profiles", "r5v", "bench.json")))


def test_fragment_scanner_reads_a_line_cut_anywhere():
    j = old_line()
    line = json.dumps(j)
    whole = flatten(j)
    for cut in (0, 17, len(line) // 3, len(line) // 2, len(line) - 2500, len(line) - 900):
        got = scan_fragment(line[max(cut, 0):])
        assert len(got) >= 5, cut
        for path, v in got.items():                      # everything the scanner reports is a number of the line, under the same key path's end
            assert any(k[-len(path):] == path and abs(w - v) <= 1e-9 * max(1.0, abs(w)) for k, w in whole.items()), (cut, path, v)


def test_a_fragment_cannot_pass_for_the_headline():
    """the tail of a line that starts after the top-level `value` still shows `cpu_baseline.value`: a key path ENDING in `value`, not the
    line's `value` (BENCH_r05.json, parsed null: the old suffix match read 378 packets/s as the headline)"""
    j = old_line()
    line = json.dumps(j)
    frag = scan_fragment(line[line.index('"host_legs_note"') - 40:])
    assert ("cpu_baseline", "value") in frag and ("value",) not in frag
    assert find({}, frag, ("value",)) is None                             # the comparison does not take it for the headline
    assert find({}, frag, ("cpu_baseline", "value")) == pytest.approx(j["cpu_baseline"]["value"])
    assert find({}, frag, ("legs", "headline")) == pytest.approx(j["legs"]["headline"])
    assert not regressions({}, frag, flatten(j))


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
        whole = flatten(d.get("parsed") or {})            # parsed holds the line's own top-level keys
        frag = {}
        for piece in ((d.get("run") or {}).get("stdout_tail") or "", (d.get("tail") or "").split("---- stderr ----")[0]):
            for k, v in scan_fragment(piece).items():
                frag.setdefault(k, v)
        bad = regressions(whole, frag, flatten(json.load(open(cf))))
        known = ACKNOWLEDGED.get(name, {})
        new = [b for b in bad if b[0] not in known]
        assert not new, "%s against %s: %s" % (name, os.path.relpath(cf, ROOT), new)
        stale = [k for k in known if k not in [b[0] for b in bad]]
        assert not stale, "%s: acknowledged regressions that the check no longer sees: %s" % (name, stale)
        ran_ok = d.get("rc") == 0 and not d.get("lease_fault") and (d.get("run") or {}).get("status", "ok") == "ok"
        if n >= 6 and ran_ok:                             # round 6 on: a run that ended well must have been PARSED (a line the driver cannot
            p_ = d.get("parsed")                          # parse is unmeasured); a box that died under the driver is not this test's business
            assert p_ and "roofline" in p_ and "cpu_baseline" in p_ and p_.get("value"), "%s: the driver did not parse the bench line" % name
        checked += 1
    assert checked == len([f for f in recs if committed_for_round(int(re.search(r"r(\d+)", os.path.basename(f)).group(1)))])


# ------------------------------------------------------------------------------------------------ the launcher
def run_bench(*argv):
    env = dict(os.environ)
    env.pop("RANK", None); env.pop("WORLD_SIZE", None); env.pop("MASTER_PORT", None); env.pop("LOCAL_RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), capture_output=True, text=True, timeout=300, env=env)
    return p


def dry(world, tmp_path):
    detail = str(tmp_path / ("detail%d.json" % world))
    p = run_bench("--gpus", str(world), "--dry-launch", "--steps", "5", "--warmup", "1", "--detail", detail)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines                       # ONE line on stdout, whatever gloo prints
    j = json.loads(lines[0])
    check_short_line(j, lines[0])
    return j, json.load(open(detail)), p


def test_bench_gpus_n_starts_n_ranks_by_itself(tmp_path):
    """`python bench.py --gpus 2` with no launcher around it: two ranks under torch.distributed.run, rendezvous on 127.0.0.1, ONE short line
    on stdout with n_gpus = 2 and the ranks' own rates (--dry-launch: gloo, a sleep for the step, no value); the per-rank records in the
    detail file"""
    j, d, p = dry(2, tmp_path)
    assert j["n_gpus"] == 2 and j["dry_launch"] is True and j["value"] is None and j["steps"] == 5 and j["warmup"] == 1
    assert "UNMEASURED" in j["multi_gpu"]
    pr = d["per_rank_packets_per_s"]
    assert len(pr["values"]) == 2 and pr["min"] == min(pr["values"]) > 0 and pr["values"][0] > pr["values"][1]       # (rank 1 sleeps twice as long)
    assert j["per_rank_packets_per_s"] == {"min": round(pr["min"], 1), "max": round(pr["max"], 1)}
    assert j["ms_per_step"] >= 1e3 * 4096 / pr["min"] * 0.99                                                          # the slowest rank sets the time
    # every rank reports where its host side runs (hostbind.bind_rank: the GPU's NUMA node, or that there is none to find) and that ONE
    # HIP runtime is mapped; the live leg's streams are partitioned over the ranks as shard.LiveShard does
    hb = d["host_binding"]
    assert [b["rank"] for b in hb] == [0, 1] and all("numa_node" in b and "affinity" in b and len(b["hip_runtimes_mapped"]) <= 1 for b in hb)
    mg = d["multi_gpu_host_fed"]
    assert mg["dry"] is True and "UNMEASURED" in mg["note"]
    assert [r["live_streams"]["global_streams"] for r in mg["per_rank"]] == [[0, 64, 128], [64, 128, 128]]
    assert "bench.py detail: {" in p.stderr             # the full record is on stderr as well


def test_bench_dry_launch_at_world_8(tmp_path):
    """the world the driver's 8-GPU node will run: eight ranks, the line still <= 4 KB with min / max of the ranks only, the eight
    per-rank records (binding, stream partition) gathered into the detail file"""
    j, d, p = dry(8, tmp_path)
    assert j["n_gpus"] == 8 and set(j["per_rank_packets_per_s"]) == {"min", "max"} and "host_binding" not in j
    assert len(d["per_rank_packets_per_s"]["values"]) == 8 and [b["rank"] for b in d["host_binding"]] == list(range(8))
    mg = d["multi_gpu_host_fed"]
    assert mg["ranks_ok"] == 8 and [r["live_streams"]["global_streams"] for r in mg["per_rank"]] == [[64 * i, 64 * i + 64, 512] for i in range(8)]
    assert j["ms_per_step"] >= 2.0 * 8 * 0.99           # rank 7 sleeps 16 ms a step: the max over the ranks


def test_bench_refuses_a_world_that_is_not_what_was_asked_for():
    """a launcher that started 2 ranks for --gpus 3 (or none for --gpus 1): no line"""
    env = dict(os.environ, RANK="0", WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--dry-launch"], capture_output=True, text=True, timeout=120, env=env)
    assert p.returncode != 0 and "refusing" in p.stderr and not p.stdout.strip()
    p = run_bench("--gpus", "2")                         # no GPUs here: the launcher says so instead of measuring one GPU
    assert p.returncode != 0 and "GPU(s) are visible" in p.stderr and not p.stdout.strip()


@pytest.mark.gpu
def test_multi_gpu_host_fed_leg_runs_under_a_one_rank_launch(tmp_path):
    """bench.py's world > 1 leg (live ingest through shard.LiveShard + the PCIe-inclusive c8p_rx, per-rank records gathered) on the one
    GPU a development box has: a one-rank torch.distributed.run with C8P_TEST_MULTI_LEG=1.  The line carries the rank's figures, the
    NUMA binding record, and every pushed PDU was delivered."""
    import socket
    s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
    env = dict(os.environ, C8P_TEST_MULTI_LEG="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    detail = str(tmp_path / "detail.json")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "10", "--warmup", "2", "--repeats", "1", "--min-timed-s", "0", "--no-cpu", "--no-frontend",
           "--no-configs", "--no-sensitivity", "--no-stream", "--detail", detail]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [ln for ln in p.stdout.splitlines() if ln.strip().startswith("{")][-1]
    check_short_line(json.loads(line), line)
    j = json.load(open(detail))
    mg = j["multi_gpu_host_fed"]
    assert mg["ranks_ok"] == 1 and "error" not in mg["per_rank"][0], mg
    r = mg["per_rank"][0]
    assert r["live_streams"]["global_streams"] == [0, 64, 64] and r["live_streams"]["streams_at_realtime"] > 50
    lo, hi = r["live_streams"]["pdus_in_the_items_pushed"]
    assert lo <= r["live_streams"]["pdus_ok"] <= hi
    assert r["pcie_inclusive_packets_per_s"] > 3e5
    assert j["host_binding"][0]["rank"] == 0 and "numa_node" in j["host_binding"][0]


def test_the_one_estimator_is_a_median_of_windows():
    """tools/bench_legs.rate_windows: probe steps size a window, then >= 3 windows between two syncs; the figure is the MEDIAN window, min
    and max beside it — one slow window (a neighbour's burst on a shared host) moves neither the figure nor is it hidden"""
    import time
    from c8p_testlib import bench_legs
    bl = bench_legs()
    calls = {"n": 0, "sync": 0}
    slow_window = {"at": None}

    def step(i):
        calls["n"] += 1
        time.sleep(2e-4)

    def sync():
        calls["sync"] += 1
        if calls["sync"] == 3:                           # the second timed window (sync 1 ends the probe) is disturbed
            time.sleep(0.05)
    w = bl.rate_windows(step, sync, units_per_step=100, window_s=0.02, windows=3, probe_steps=4)
    ws = w["windows_per_s"]
    assert ws["n"] == 3 and ws["min"] < 0.6 * ws["median"] and ws["median"] <= ws["max"] and w["per_s"] == ws["median"]
    assert calls["sync"] == 4 and calls["n"] == 4 + 3 * w["steps"]
    assert 100 / 6e-4 < w["per_s"] < 100 / 2e-4          # ~0.2-0.3 ms a step: the disturbed window did not become the figure


def test_a_leg_that_hangs_costs_the_leg_not_the_line(tmp_path):
    """tools/bench_legs.LegGuard around a leg that never returns (a collective whose peer is gone): after the timeout rank 0 prints the
    headline line as it stands — short, parseable, the leg named as failed — and the process leaves with status 0"""
    detail = str(tmp_path / "d.json")
    code = r'''
import argparse, json, os, sys, time
sys.path.insert(0, os.path.join(%r, "tools"))
import bench_legs, bench_line
args = argparse.Namespace(detail=%r)
out = {"metric": "RX packets/s (20 MHz VHT MCS7 1458 B)", "value": 6.5e6, "unit": "packets/s", "n_gpus": 8, "steps": 20, "warmup": 5, "ms_per_step": 5.04,
       "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": {"workload": "w", "parallelism": "p"},
       "roofline": {"bound": "hbm", "kernel": "viterbi_fwd", "achieved": 238.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.03, "traffic": 5.4e8}}
g = bench_legs.LegGuard(out, 0, 0.3, "sharded_stream", lambda o: bench_line.finish(o, args))
time.sleep(30)                      # the leg: never returns
print("NOT REACHED")
''' % (ROOT, detail)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert p.returncode == 0, p.stderr[-1500:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and "NOT REACHED" not in p.stdout
    j = json.loads(lines[0])
    check_short_line(j, lines[0])
    assert j["value"] == 6.5e6 and j["n_gpus"] == 8 and j["legs_failed"] == ["sharded_stream"] and j["roofline"]["frac"] == 0.03
    d = json.load(open(detail))
    assert "no result after" in d["sharded_stream"]["error"] and "abandoned" in p.stderr


def test_leg_budget_skips_what_does_not_fit_and_isolates_a_failing_leg(monkeypatch):
    """tools/bench_legs.run_single_gpu with the legs stubbed: they run in the order of the line's `legs`, a leg that would start after the
    deadline is listed in legs_skipped (the driver's command keeps its wall clock), a leg that throws is recorded and the next one runs"""
    import argparse
    import time
    from c8p_testlib import bench_legs
    bl = bench_legs()
    ran = []
    monkeypatch.setattr(bl, "config_legs", lambda *a, **k: ran.append("configs") or {"cfg1": {"packets_per_s": 1.0}})

    def pcie(*a, **k):
        ran.append("pcie")
        raise RuntimeError("link down")
    monkeypatch.setattr(bl, "pcie_inclusive", pcie)
    monkeypatch.setattr(bl, "arenas_to_host", lambda *a: (None, None))
    monkeypatch.setattr(bl, "block_chain_legs", lambda out, *a: (ran.append("blocks"), time.sleep(0.25), out.update(block_chain_packets_per_s=3.0)))
    monkeypatch.setattr(bl, "sensitivity_legs", lambda *a, **k: ran.append("sens") or {})
    monkeypatch.setattr(bl, "sharded_stream", lambda *a, **k: ran.append("stream") or {})
    monkeypatch.setattr(bl, "frontend_detect", lambda *a, **k: ran.append("frontend") or {})
    args = argparse.Namespace(workload="cfg2", snr=30.0, no_configs=False, no_host_legs=False, no_sensitivity=False, no_stream=False, no_frontend=False,
                              stream_packets=1, stream_steps=1)
    ctx = {"torch": None, "dev": None, "local": 0, "pk": [0] * 4, "iq_dev": None, "chan_dev": None, "expected": [], "use_dist": False}
    out = {"value": 1.0}
    bl.run_single_gpu(ctx, out, args, False, time.perf_counter() + 0.2, lambda o: None)
    assert ran == ["configs", "pcie", "blocks"]                    # the 0.25 s leg ran past the deadline: what follows is skipped, in order
    assert out["legs_skipped"] == ["sensitivity", "sharded_stream", "frontend_detect"]
    assert "link down" in out["pcie_inclusive_error"] and out["block_chain_packets_per_s"] == 3.0 and out["configs"]["cfg1"]["packets_per_s"] == 1.0
    assert set(out["leg_wall_s"]) == {"configs", "pcie_inclusive", "block_chain"} and out["leg_wall_s"]["block_chain"] >= 0.25
    ran.clear()
    out = {"value": 1.0}
    bl.run_single_gpu(ctx, out, args, True, None, lambda o: None)   # --legs full: no deadline, every leg
    assert ran == ["configs", "pcie", "blocks", "sens", "stream", "frontend"] and "legs_skipped" not in out
