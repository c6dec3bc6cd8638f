import os, sys
sys.path.insert(0, "/root/repo")
os.environ["C8P_DET_TLOG"] = "1"
from __graft_entry__ import load_package
load_package()
import numpy as np, torch
from gr_ieee80211_amd import engine
from gr_ieee80211_amd.workloads import raw_stream_unit
dev = torch.device("cuda:0")
unit, _ = raw_stream_unit(0)
reps = (32 << 20) // len(unit)
x = torch.from_numpy(unit.view(np.float32)).to(dev).repeat(reps)
n = x.numel() // 2
eng = engine.Engine(device=0)
arr = engine.DevArray(x.data_ptr(), n, x)
for i in range(4):
    sys.stderr.write("--- call %d\n" % i)
    eng.detect(arr)
