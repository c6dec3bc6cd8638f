#!/usr/bin/env python3
"""tools/probes/pcie_gather.py — bench.py's pcie_inclusive leg alone (the gather path), for A/B runs under C8P_COPY_CUS / C8P_GATHER_WGS"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
load_package()
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench_legs as bench  # noqa: E402  (the legs moved out of bench.py in round 6)
from gr_ieee80211_amd.workloads import workload_on_device
dev = torch.device("cuda:0")
pk, iq_dev, chan_dev, exp, keep = workload_on_device("cfg2", 4096, 30.0, 0, torch, dev, 0)
det = {}
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 24
r = bench.pcie_inclusive(pk, iq_dev, chan_dev, torch, dev, 0, steps=steps, detail=det)
print(json.dumps({"copy_cus": os.environ.get("C8P_COPY_CUS"), "gather_wgs": os.environ.get("C8P_GATHER_WGS"), "packets_per_s": round(r), "down_GBps": round(det["host_to_device_GBps"], 2),
                  "dma_path": round(det["dma_path_packets_per_s"]) if isinstance(det.get("dma_path_packets_per_s"), float) else None}))
