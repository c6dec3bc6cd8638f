#!/usr/bin/env python3
"""tools/probes/rawchain_repro_table.py gpurun_out/repro_*.jsonl > profiles/<round>/rawchain_repro.txt — the table of tools/probes/rawchain_repro.sh runs
(one fresh MI355X lease per file, five variants per lease)."""
import json
import sys

rows = []
for f in sys.argv[1:]:
    for ln in open(f):
        j = json.loads(ln)
        j["lease"] = f.split("repro_")[-1].split(".")[0]
        rows.append(j)
print("raw stream -> PDUs (bench.py frontend_detect.raw_stream_to_pdus), one c8p_rawchain handle, arenas reserved at create, warm-up through the")
print("same handle; 24 passes over one resident 33.5 M-item segment (6034 cfg-2 packets); steady state = the last 20 segments")
print()
print("%-8s %-14s %9s | %11s %9s %9s %8s %8s | %10s %8s | %9s %9s" % ("lease", "variant", "headline", "chain pk/s", "ms/seg", "first ms", "growths", "giveups",
                                                                        "pcie pk/s", "GB/s", "live x RT", "streams"))
for j in rows:
    c, p, l = j["chain"], j["pcie"], j["live"]
    print("%-8s %-14s %8.2fM | %10.2fM %9.3f %9.2f %8s %8s | %9.2fM %8.1f | %9.2f %9.0f" % (
        j["lease"], j["variant"], j["headline"] / 1e6, (c["packets_per_s"] or 0) / 1e6, c["ms_per_segment"] or 0, c["first_segment_ms"] or 0,
        c["arena_growths_in_run"], c["lookback_giveups"], (p.get("packets_per_s") or 0) / 1e6, p.get("GB_per_s_over_pcie") or 0,
        l.get("x_realtime_each") or 0, l.get("streams_at_realtime") or 0))
v = [j["chain"]["packets_per_s"] for j in rows if j["chain"].get("packets_per_s")]
print()
print("chain packets/s over all %d runs: min %.2f M, median %.2f M, max %.2f M" % (len(v), min(v) / 1e6, sorted(v)[len(v) // 2] / 1e6, max(v) / 1e6))
print("variants: default = as shipped; tickets = C8P_DET_TICKETS=1 (every k_detect_scan tile takes its number from a counter); hwq2 / hwq8 =")
print("GPU_MAX_HW_QUEUES=2 / 8 (the HIP runtime's hardware-queue count; the chain runs 3 + 3 + 3 + 3 engine streams on one GPU).")
print("per-stage detail of every run: gpurun_out/repro_<lease>.jsonl (stage_ms, host_thread_s), copied beside this file.")
