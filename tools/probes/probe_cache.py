import ctypes as C, os, sys, time
sys.path.insert(0, os.getcwd())
from __graft_entry__ import load_package
load_package()
import torch
from gr_ieee80211_amd import abi, engine
e = engine.Engine(device=0)
L = e.L
for mb in (16, 64, 128, 512):
    total = mb << 20
    pin = torch.empty(total, dtype=torch.uint8).pin_memory()
    pin.random_(0, 255)
    dst = L.c8p_dev_alloc(e.ctx, 4 * total)
    n = 64
    chunk = total // n
    for fmt, ib, name in ((abi.FMT_C32, 8, "c32"), (abi.FMT_SC16, 4, "sc16"), (abi.FMT_SC8, 2, "sc8")):
        pcs = (abi.Piece * n)(*[abi.Piece(pin.data_ptr() + i * chunk, i * chunk // ib, chunk // ib, 1, fmt, 1e-3, 0) for i in range(n)])
        ts = []
        for rep in range(6):
            t0 = time.perf_counter()
            L.c8p_copy_gather(e.ctx, C.c_void_p(dst), pcs, n)
            L.c8p_copy_wait(e.ctx)
            ts.append(time.perf_counter() - t0)
        print("%4d MB pinned, %-4s: first %6.2f ms, later %6.2f ms = %6.1f GB/s of source bytes" % (mb, name, 1e3 * ts[0], 1e3 * min(ts[1:]), total / min(ts[1:]) / 1e9))
    L.c8p_dev_free(e.ctx, C.c_void_p(dst))
