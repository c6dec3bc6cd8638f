import os, sys, time, json
sys.path.insert(0, "."); sys.path.insert(0, "tests")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench_legs as bench  # noqa: E402  (the legs moved out of bench.py in round 6)
import torch
torch.zeros(1, device="cuda")
r = bench.config_legs(torch, torch.device("cuda", 0), 0, 30.0)
print(os.environ.get("C8P_LIB", "default")[-20:], {k: round(v["packets_per_s"]) for k, v in r.items() if k.startswith("cfg3")})
