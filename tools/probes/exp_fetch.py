import os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from c8p_cases import cfg2_batch
from c8p_testlib import abi
from gr_ieee80211_amd import engine
b, exp = cfg2_batch(256, snr_db=30.0, seed=1)
pk, iq, chan = b.arrays()
eng = engine.Engine(device=0)
L = eng.L
# pinned staging for inputs
def pinned(arr):
    p = L.c8p_host_alloc(arr.nbytes)
    C.memmove(p, arr.ctypes.data, arr.nbytes)
    return np.ctypeslib.as_array((C.c_uint8 * arr.nbytes).from_address(p)).view(arr.dtype)
iq_p, ch_p = pinned(iq), pinned(chan)
eng.demod(pk, iq_p, ch_p); eng.sync()
_, nb = eng.buffer(abi.BUF_LLR)
dst = L.c8p_host_alloc(nb)
pag = np.zeros(nb, np.uint8)
print("LLR arena bytes", nb)
for rep in range(4):
    for name, d in (("pinned", dst), ("pageable", pag.ctypes.data)):
        t0 = time.perf_counter(); eng.demod(pk, iq_p, ch_p); t1 = time.perf_counter()
        eng.fetch_async(abi.BUF_LLR, d, 0, nb); t2 = time.perf_counter()
        eng.sync(); t3 = time.perf_counter()
        print("%-9s submit %.3f ms  fetch_async call %.3f ms  sync %.3f ms" % (name, 1e3*(t1-t0), 1e3*(t2-t1), 1e3*(t3-t2)))
