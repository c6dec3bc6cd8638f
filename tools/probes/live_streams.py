#!/usr/bin/env python3
"""tools/probes/live_streams.py [streams ...] — bench.py's live_streams leg alone, for several stream counts (and round lengths): how many
independent 20 MS/s streams one MI355X serves, host-fed over PCIe, through host/c8p_rawchain's live ingest."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench_legs as bench  # noqa: E402  (the legs moved out of bench.py in round 6)
from __graft_entry__ import load_package  # noqa: E402

load_package()
import torch  # noqa: E402
from gr_ieee80211_amd.workloads import raw_stream_unit  # noqa: E402

dev = torch.device("cuda:0")
unit, _ = raw_stream_unit(0, lead=int(os.environ.get("C8P_LIVE_LEAD", "600")))      # C8P_LIVE_LEAD: idle items in front of every packet (600: packet-dense)
for arg in (sys.argv[1:] or ["64"]):                 # K[:items per round][:sc16 | sc8][:pace]
    f = arg.split(":")
    K, items = int(f[0]), int(f[1]) if len(f) > 1 and f[1] else 524288
    bits = {"sc16": 16, "sc8": 8}.get(f[2] if len(f) > 2 else "", 0)
    pace = float(f[3]) if len(f) > 3 and f[3] else 0.0
    r = bench.live_streams(torch, dev, 0, unit, streams=K, round_items=items, rounds=80 if pace else 40, bits=bits, pace=pace)
    print(json.dumps({k: r[k] for k in ("streams", "items_per_stream_per_round", "rounds", "ms_per_round", "x_realtime_each", "streams_at_realtime",
                                        "pcie_GBps", "packets_per_s", "pdus_ok", "pdus_in_the_items_pushed", "round_closed_to_pdus_ms", "paced_at", "behind_ms", "push_and_round_calls_s", "host_thread_s")}))
