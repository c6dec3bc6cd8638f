#!/usr/bin/env python3
"""tools/probes/run_detect.py — c8p_detect over a 32M-sample synthetic stream, a few calls (profiling driver)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench_legs as bench  # noqa: E402  (the legs moved out of bench.py in round 6)
r = bench.frontend_detect(torch, torch.device("cuda", 0), 0)
print(r)
