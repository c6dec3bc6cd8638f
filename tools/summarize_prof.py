#!/usr/bin/env python3
"""tools/summarize_prof.py <tag> — condense gpurun_out/prof_<tag>/ (tools/profile_gpu.sh) into profiles/<tag>/:
bench.json, kernel_stats.csv (rocprofv3 --kernel-trace --stats), pmc_hbm.json (FETCH_SIZE / WRITE_SIZE per launch,
separate passes; FETCH bytes corrected x2 for gfx950 per MI355X_MICROARCH.md) and pmc_summary.txt (SQ counters)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

tag = sys.argv[1]
src = os.path.join("gpurun_out", "prof_" + tag)
dst = os.path.join("profiles", tag)
os.makedirs(dst, exist_ok=True)


def short(name):
    return name.split("(")[0]


b = open(os.path.join(src, "bench.json")).read().strip().splitlines()
open(os.path.join(dst, "bench.json"), "w").write(b[-1] + "\n" if b else "")
for extra in ("bench_detail.json", "bench_driver_cmd.json", "bench_driver_cmd_detail.json"):      # the full record (--legs full) and the driver's command
    if os.path.exists(os.path.join(src, extra)):
        open(os.path.join(dst, extra), "w").write(open(os.path.join(src, extra)).read().strip() + "\n")
for f in glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True):
    open(os.path.join(dst, "kernel_stats.csv"), "w").write(open(f).read())
for f in glob.glob(os.path.join(src, "stats1", "**", "*kernel_stats.csv"), recursive=True):      # C8P_SLOTS=1: averages = exclusive durations
    open(os.path.join(dst, "kernel_stats_exclusive.csv"), "w").write(open(f).read())

counters = defaultdict(lambda: defaultdict(list))      # kernel -> counter -> values
for f in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        counters[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
durs = defaultdict(list)
for f in glob.glob(os.path.join(src, "pmc_FETCH_SIZE", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        durs[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))

avg = lambda v: sum(v) / len(v) if v else 0.0
hbm = {"source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, C8P_SLOTS=1 (no overlap), 4096 packets per launch",
       "kernels": {}}
lines = ["rocprofv3 PMC passes (each counter set in its own run, --kernel-trace only), bench.py --steps 3 --warmup 1 --no-cpu, C8P_SLOTS=1",
         "FETCH_SIZE / WRITE_SIZE in KB as reported; FETCH bytes doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests as 64 B)", "",
         "kernel, launches, FETCH_SIZE_KB(avg raw), FETCH_bytes_corrected, WRITE_SIZE_KB(avg), WRITE_bytes, HBM_bytes_total, avg_ns(pmc run)"]
for k in sorted(counters):
    c = counters[k]
    if "FETCH_SIZE" not in c:
        continue
    fk, wk = avg(c["FETCH_SIZE"]), avg(c.get("WRITE_SIZE", []))
    hbm["kernels"][k] = {"fetch_size_kb_raw": fk, "write_size_kb": wk}
    for extra_c in ("SQ_INSTS_VALU", "SQ_WAVES", "GRBM_GUI_ACTIVE", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES"):
        if extra_c in c:
            hbm["kernels"][k][extra_c.lower()] = avg(c[extra_c])
    if "SQ_ACTIVE_INST_VALU" in c and "GRBM_GUI_ACTIVE" in c and avg(c["GRBM_GUI_ACTIVE"]) > 0:
        # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the chip's 1024 SIMDs; GRBM_GUI_ACTIVE cycles summed over its 8 XCDs:
        # VALU-busy cycles per SIMD / cycles the launch lasted
        hbm["kernels"][k]["valu_busy"] = (avg(c["SQ_ACTIVE_INST_VALU"]) * 4.0 / 1024.0) / (avg(c["GRBM_GUI_ACTIVE"]) / 8.0)
    lines.append("%s, %d, %.1f, %d, %.1f, %d, %d, %d" % (k, len(c["FETCH_SIZE"]), fk, fk * 2048, wk, wk * 1024, fk * 2048 + wk * 1024, avg(durs.get(k, []))))
lines += ["", "SQ / GRBM counters, average per launch (SQ_*_CYCLES, SQ_ACTIVE_*, SQ_WAIT_* are in quad-cycles summed over waves / SIMDs)"]
for k in sorted(counters):
    c = counters[k]
    names = [n for n in sorted(c) if n not in ("FETCH_SIZE", "WRITE_SIZE")]
    if not names:
        continue
    lines.append("%s: " % k + ", ".join("%s=%d" % (n, avg(c[n])) for n in names))
    if "SQ_INSTS_VALU" in c and "SQ_WAVES" in c and avg(c["SQ_WAVES"]) > 0:
        extra = "   VALU instr / wave = %.0f" % (avg(c["SQ_INSTS_VALU"]) / avg(c["SQ_WAVES"]))
        if "SQ_ACTIVE_INST_VALU" in c and "SQ_WAVE_CYCLES" in c and avg(c["SQ_WAVE_CYCLES"]) > 0:
            extra += "; VALU-busy share of a wave's lifetime = %.3f" % (avg(c["SQ_ACTIVE_INST_VALU"]) / avg(c["SQ_WAVE_CYCLES"]))
        lines.append(extra)
# the two-stream kernel on BASELINE configs[2] (cfg3_* passes of tools/profile_gpu.sh)
c3 = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(src, "cfg3_pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        c3[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
if c3:
    lines += ["", "BASELINE configs[2] (HT MCS15 2x2, 4096 packets per launch, bench.py --workload cfg3, C8P_SLOTS=1): average per launch",
              "(42 966 algorithmic bytes per packet = 176 MB per launch; the LLR stream of this config would be 14 352 x 4 B x 4096 = 235 MB each way)"]
    hbm["cfg3_kernels"] = {}
    for k in sorted(c3):
        c = c3[k]
        if not k.startswith("k_"):
            continue
        fk, wk = avg(c.get("FETCH_SIZE", [])), avg(c.get("WRITE_SIZE", []))
        hbm["cfg3_kernels"][k] = {"fetch_size_kb_raw": fk, "write_size_kb": wk, **{n.lower(): avg(v) for n, v in c.items() if n not in ("FETCH_SIZE", "WRITE_SIZE")}}
        lines.append("%s: FETCH x2 = %.1f MB, WRITE = %.1f MB, " % (k, fk * 2048 / 1e6, wk * 1024 / 1e6) +
                     ", ".join("%s=%d" % (n, avg(c[n])) for n in sorted(c) if n not in ("FETCH_SIZE", "WRITE_SIZE")))
for f in glob.glob(os.path.join(src, "cfg3_stats1", "**", "*kernel_stats.csv"), recursive=True):
    open(os.path.join(dst, "kernel_stats_exclusive_cfg3.csv"), "w").write(open(f).read())
# the front end (raw_* passes): the chain from the raw stream, one call after the other
cr = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(src, "raw_pmc_*", "**", "*counter_collection.csv"), recursive=True):
    per = defaultdict(float)                                   # (kernel, dispatch, counter): a dispatch's rows add up
    for r in csv.DictReader(open(f)):
        per[(short(r["Kernel_Name"]), r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (k, _, n), v in per.items():
        cr[k][n].append(v)
rd = defaultdict(list)
for f in glob.glob(os.path.join(src, "raw_stats", "**", "*kernel_stats.csv"), recursive=True):
    open(os.path.join(dst, "kernel_stats_raw_chain.csv"), "w").write(open(f).read())
    for r in csv.DictReader(open(f)):
        rd[short(r["Name"])] = float(r["AverageNs"])
if cr:
    lines += ["", "the chain from the RAW stream (tools/exp_rawchain.py: one 33 549 040-item stream of 6034 cfg-2 packets, c8p_detect -> c8p_signal (consume) ->",
              "c8p_rx one call after the other, 7 passes): average per launch; avg_us from the --stats pass (kernel_stats_raw_chain.csv)",
              "k_detect_scan's algorithmic bytes: 9 B per item (8 read + 1 trigger byte written) = 301.9 MB per launch"]
    hbm["raw_chain_kernels"] = {}
    for k in sorted(cr):
        c = cr[k]
        if not k.startswith("k_"):
            continue
        fk, wk = avg(c.get("FETCH_SIZE", [])), avg(c.get("WRITE_SIZE", []))
        hbm["raw_chain_kernels"][k] = {"fetch_size_kb_raw": fk, "write_size_kb": wk, "avg_us": rd.get(k, 0.0) / 1e3,
                                       **{n.lower(): avg(v) for n, v in c.items() if n not in ("FETCH_SIZE", "WRITE_SIZE")}}
        lines.append("%s: %.1f us, FETCH x2 = %.1f MB, WRITE = %.1f MB, " % (k, rd.get(k, 0.0) / 1e3, fk * 2048 / 1e6, wk * 1024 / 1e6) +
                     ", ".join("%s=%d" % (n, avg(c[n])) for n in sorted(c) if n not in ("FETCH_SIZE", "WRITE_SIZE")))
# opcode issue costs measured in the same session (tools/probe_issue.hip -> tools/opcode_costs.py)
pi = os.path.join(src, "probe_issue.txt")
if os.path.exists(pi):
    import subprocess
    open(os.path.join(dst, "probe_issue.txt"), "w").write(open(pi).read())
    subprocess.check_call([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "opcode_costs.py"), os.path.join(dst, "probe_issue.txt"),
                           os.path.join(dst, "opcode_costs.json")], stdout=subprocess.DEVNULL)
# the dominant kernel AS THE BENCH RUNS IT: three slots, launches of consecutive batches co-resident.  From the kernel trace of the
# default run: the k_rx_1ss_lds dispatches of the timed steps (the long ones: a second, nearly empty launch per step handles flagged
# packets), the middle half of them — per step: the period between consecutive starts, the time during which at least one such
# launch is resident (union of the intervals / steps), the mean duration of a launch and how many are resident on average
for f in glob.glob(os.path.join(src, "trace3", "**", "*kernel_trace.csv"), recursive=True):
    iv = []
    for r in csv.DictReader(open(f)):
        if short(r["Kernel_Name"]).startswith("k_rx_1ss_lds"):
            iv.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    iv.sort()
    if len(iv) >= 16:
        med = sorted(e - b for b, e in iv)[len(iv) // 2]
        iv = [(b, e) for b, e in iv if e - b > 0.5 * med]                      # the full-size launches
        iv = iv[len(iv) // 4: len(iv) - len(iv) // 4]                           # steady state: the middle half
        union, cur_b, cur_e = 0, iv[0][0], iv[0][1]
        for b, e in iv[1:]:
            if b > cur_e:
                union += cur_e - cur_b; cur_b, cur_e = b, e
            else:
                cur_e = max(cur_e, e)
        union += cur_e - cur_b
        n = len(iv)
        span = iv[-1][1] - iv[0][0]
        res = {"source": "rocprofv3 --kernel-trace of bench.py --steps 60 (C8P_SLOTS default 3), k_rx_1ss_lds dispatches, middle half",
               "launches": n, "step_period_us": (iv[-1][0] - iv[0][0]) / (n - 1) / 1e3, "launch_duration_us_mean": sum(e - b for b, e in iv) / n / 1e3,
               "resident_us_per_step": union / n / 1e3, "resident_fraction_of_span": union / span,
               "launches_resident_on_average": sum(e - b for b, e in iv) / span}
        json.dump(res, open(os.path.join(dst, "residency_3slots.json"), "w"), indent=1)
        lines += ["", "k_rx_1ss_lds as the bench runs it (3 slots, kernel trace): " + json.dumps(res)]
json.dump(hbm, open(os.path.join(dst, "pmc_hbm.json"), "w"), indent=1)
open(os.path.join(dst, "pmc_summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
