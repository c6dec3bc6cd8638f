#!/usr/bin/env python3
"""tools/probes/run_detect.py — c8p_detect over a 32M-sample synthetic stream, a few calls (profiling driver)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
r = bench.frontend_detect(torch, torch.device("cuda", 0), 0)
print(r)
