#!/usr/bin/env python3
"""tools/probes/exp_scan.py — k_detect_scan alone: its duration (HIP events) over a 33.5 M-item stream that is (a) the bench stream of cfg-2
packets (about a third of its tiles hold a plateau) and (b) noise only (every tile takes the quiet path).  Run on the GPU box; under
rocprofv3 --pmc the two launches can be told apart by their order (a first)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from __graft_entry__ import load_package
load_package()
from gr_ieee80211_amd import engine
from gr_ieee80211_amd.workloads import raw_stream_unit

dev = torch.device("cuda:0")
unit, _ = raw_stream_unit(0)
reps = (32 << 20) // len(unit)
xa = torch.from_numpy(unit.view(np.float32)).to(dev).repeat(reps)
n = xa.numel() // 2
g = torch.Generator(device=dev); g.manual_seed(5)
xb = (0.006 * torch.randn(2 * n, generator=g, device=dev, dtype=torch.float32)).contiguous()
out = {}
for name, x in (("packets", xa), ("noise", xb)):
    eng = engine.Engine(device=0, profile=True)
    arr = engine.DevArray(x.data_ptr(), n, x)
    ev = eng.detect(arr)
    eng.profile_reset()
    for _ in range(int(os.environ.get("EXP_SCAN_REPS", "5"))):
        ev = eng.detect(arr)
    kt = {k: (ms / cnt if cnt else 0.0) for k, (ms, cnt) in eng.kernel_times().items()}
    out[name] = {"scan_us": round(1e3 * kt["presiso"], 1), "GBps_alg": round(9.0 * n / (kt["presiso"] * 1e-3) / 1e9), "events": int(len(ev))}
    eng.close()
print(json.dumps(out))
