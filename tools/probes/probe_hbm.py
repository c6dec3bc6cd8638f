#!/usr/bin/env python3
"""tools/probes/probe_hbm.py — what a trivial streaming kernel reaches on this box, as a yardstick for k_presiso (8 B in, 4 B out per
sample): torch.abs on complex64 (same traffic shape), a float32 copy, and a float32 fill."""
import time
import torch
torch.cuda.init()
n = 33_549_040
x = torch.randn(n, 2, device="cuda").view(torch.float32)
z = torch.view_as_complex(x.view(n, 2))
out = torch.empty(n, device="cuda")
big = torch.empty(n * 3, device="cuda")
dst = torch.empty_like(big)

def t(f, it=30):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it

a = t(lambda: torch.abs(z, out=out)); print("abs(complex64) -> f32: %.4f ms, %.2f TB/s" % (a * 1e3, n * 12 / a / 1e12))
c = t(lambda: dst.copy_(big)); print("f32 copy (%d MB): %.4f ms, %.2f TB/s (read + write)" % (big.numel() * 4 // 2**20, c * 1e3, big.numel() * 8 / c / 1e12))
f = t(lambda: dst.fill_(1.0)); print("f32 fill: %.4f ms, %.2f TB/s" % (f * 1e3, big.numel() * 4 / f / 1e12))
