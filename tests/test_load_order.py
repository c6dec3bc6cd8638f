"""The Python wrapper and PyTorch in one process, in either import order: PyTorch-ROCm wheels bring their own HIP runtime and the engine
library is linked against the system's; whichever is loaded first serves both (gr_ieee80211_amd.engine.prefer_torch_hip_runtime)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

ENGINE_FIRST = """
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
from __graft_entry__ import load_package
load_package()
assert "torch" not in sys.modules
from gr_ieee80211_amd import engine
e = engine.Engine(device=0)            # the engine brings up HIP before torch exists in the process
assert e.counter(5) >= 1               # C8P_CNT_SLOTS
import torch
assert torch.cuda.is_available(), "torch lost the GPU to the engine's runtime"
t = torch.zeros(1024).pin_memory().to("cuda:0")
torch.cuda.synchronize()
e.close()
print("ok")
"""


@pytest.mark.gpu
def test_engine_first_then_torch():
    r = subprocess.run([sys.executable, "-c", ENGINE_FIRST % (ROOT, os.path.join(ROOT, "tests"))], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]
