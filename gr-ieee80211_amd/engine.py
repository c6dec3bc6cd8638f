"""ctypes binding of libc8p_rx.so (the HIP engine).  Plumbing only: no arithmetic happens here.

There is deliberately no fallback: if the shared library is missing or a symbol of include/c8p_rx.h is
not exported, loading fails loudly; if no HIP device is usable, Engine() raises.
"""
import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# C8P_LIB: load another build of the same library (A/B runs of kernel variants); still no fallback
LIB_PATH = os.environ.get("C8P_LIB") or os.path.join(_HERE, "libc8p_rx.so")

# every entry point include/c8p_rx.h declares
SYMBOLS = ["c8p_create", "c8p_destroy", "c8p_last_error", "c8p_demod", "c8p_decode", "c8p_rx", "c8p_sync",
           "c8p_buffer", "c8p_fetch", "c8p_kernel_time", "c8p_profile_reset", "c8p_table", "c8p_abi_sizes", "c8p_signal",
           "c8p_detect", "c8p_tx", "c8p_tx_plan_of", "c8p_tx_plan_mu", "c8p_ticket", "c8p_select", "c8p_fetch_async", "c8p_wait", "c8p_done",
           "c8p_host_alloc", "c8p_host_free", "c8p_selftest", "c8p_detect_multi", "c8p_reserve", "c8p_counter", "c8p_option",
           "c8p_dev_alloc", "c8p_dev_free", "c8p_copy_async", "c8p_copy_spans", "c8p_copy_wait", "c8p_copy_mark", "c8p_copy_wait_mark", "c8p_copy_mark_done", "c8p_copy_gather", "c8p_copy_fence", "c8p_fetch_gather"]

_LIB = None


def prefer_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels carry their own libamdhip64.so; libc8p_rx.so is linked against the system's.
    Whichever is loaded first serves both (same soname) — and torch on top of the system's newer runtime reports "No HIP GPUs are
    available", while the engine runs on either.  So if torch is installed but not imported yet, its runtime is loaded first (the
    library only, not the package): a later `import torch` in the same process then finds the runtime it was built for."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    p = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(p):
        try:
            C.CDLL(p, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load_library():
    global _LIB
    if _LIB is not None:
        return _LIB
    prefer_torch_hip_runtime()
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libc8p_rx.so is not built (run `python __graft_entry__.py` / make -C gr-ieee80211_amd/csrc); "
                           "there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    for s in SYMBOLS:
        if not hasattr(L, s):
            raise RuntimeError("libc8p_rx.so does not export %s" % s)
    L.c8p_create.restype = C.c_void_p
    L.c8p_create.argtypes = [C.POINTER(abi.Cfg)]
    L.c8p_destroy.argtypes = [C.c_void_p]
    L.c8p_last_error.restype = C.c_char_p
    L.c8p_last_error.argtypes = [C.c_void_p]
    for f in ("c8p_demod", "c8p_rx"):
        getattr(L, f).argtypes = [C.c_void_p, C.POINTER(abi.BatchIn)]
    L.c8p_decode.argtypes = [C.c_void_p, C.POINTER(abi.DecodeIn)]
    L.c8p_signal.argtypes = [C.c_void_p, C.POINTER(abi.SignalIn)]
    L.c8p_detect.argtypes = [C.c_void_p, C.POINTER(abi.DetectIn)]
    L.c8p_detect_multi.argtypes = [C.c_void_p, C.POINTER(abi.DetectMultiIn)]
    L.c8p_reserve.argtypes = [C.c_void_p, C.POINTER(abi.ReserveIn)]
    L.c8p_counter.restype = C.c_int64
    L.c8p_counter.argtypes = [C.c_void_p, C.c_int]
    L.c8p_option.argtypes = [C.c_void_p, C.c_int, C.c_int64]
    L.c8p_dev_alloc.restype = C.c_void_p
    L.c8p_dev_alloc.argtypes = [C.c_void_p, C.c_uint64]
    L.c8p_dev_free.argtypes = [C.c_void_p, C.c_void_p]
    L.c8p_copy_async.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
    L.c8p_copy_spans.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
    L.c8p_copy_wait.argtypes = [C.c_void_p]
    L.c8p_copy_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
    L.c8p_copy_mark.restype = C.c_uint64
    L.c8p_copy_mark.argtypes = [C.c_void_p]
    L.c8p_copy_wait_mark.argtypes = [C.c_void_p, C.c_uint64]
    L.c8p_copy_mark_done.argtypes = [C.c_void_p, C.c_uint64]
    L.c8p_copy_fence.argtypes = [C.c_void_p]
    L.c8p_fetch_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
    L.c8p_tx.argtypes = [C.c_void_p, C.POINTER(abi.TxIn)]
    L.c8p_tx_plan_of.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(abi.TxPlan)]
    L.c8p_tx_plan_mu.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(abi.TxPlan)]
    L.c8p_sync.argtypes = [C.c_void_p]
    L.c8p_ticket.argtypes = [C.c_void_p]
    L.c8p_select.argtypes = [C.c_void_p, C.c_int]
    L.c8p_buffer.restype = C.c_void_p
    L.c8p_buffer.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]
    L.c8p_fetch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.c_uint64]
    L.c8p_fetch_async.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.c_uint64]
    L.c8p_wait.argtypes = [C.c_void_p, C.c_int]
    L.c8p_done.argtypes = [C.c_void_p, C.c_int]
    L.c8p_host_alloc.restype = C.c_void_p
    L.c8p_host_alloc.argtypes = [C.c_uint64]
    L.c8p_host_free.argtypes = [C.c_void_p]
    L.c8p_selftest.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64]
    L.c8p_kernel_time.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    L.c8p_profile_reset.argtypes = [C.c_void_p]
    L.c8p_table.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int]
    L.c8p_abi_sizes.argtypes = [C.c_void_p, C.c_int]
    sizes = np.zeros(23, np.int32)
    L.c8p_abi_sizes(sizes.ctypes.data_as(C.c_void_p), 23)
    want = [abi.ABI_VERSION, C.sizeof(abi.PktIn), C.sizeof(abi.PktTags), C.sizeof(abi.PktOut), C.sizeof(abi.Cfg),
            C.sizeof(abi.BatchIn), C.sizeof(abi.DecodeIn), C.sizeof(abi.SigIn), C.sizeof(abi.SignalIn),
            C.sizeof(abi.SyncEvt), C.sizeof(abi.DetectIn), C.sizeof(abi.DetState), C.sizeof(abi.TxPkt), C.sizeof(abi.TxDesc),
            C.sizeof(abi.TxIn), C.sizeof(abi.TxPlan), C.sizeof(abi.TxMu), C.sizeof(abi.DetStream), C.sizeof(abi.DetectMultiIn),
            C.sizeof(abi.ReserveIn), C.sizeof(abi.Span), C.sizeof(abi.Piece), C.sizeof(abi.FetchPiece)]
    if list(sizes) != want:
        raise RuntimeError("ABI mismatch between abi.py and libc8p_rx.so: %s vs %s" % (list(sizes), want))
    _LIB = L
    return L


def table(which, sub=0):
    buf = np.zeros(512, np.int32)
    n = load_library().c8p_table(which, sub, buf.ctypes.data_as(C.c_void_p), 512)
    if n < 0:
        raise ValueError("c8p_table(%d,%d) -> %d" % (which, sub, n))
    return buf[:n].copy()


class DevArray:
    """A caller-owned device buffer (e.g. torch tensor storage): raw pointer + item count.

    The engine works on its OWN streams (unless Engine(stream=...) hands it the caller's), so the buffer's contents must be
    complete before an engine call reads them.  If `keepalive` is a CUDA torch tensor, its producer stream is synchronised
    here, when the DevArray is made — i.e. make the DevArray after the operations that fill the tensor have been issued."""

    def __init__(self, ptr, n, keepalive=None):
        self.ptr, self.n, self.keepalive = int(ptr), int(n), keepalive
        if getattr(keepalive, "is_cuda", False):
            import torch
            torch.cuda.current_stream(keepalive.device).synchronize()


class Engine:
    def __init__(self, device=0, profile=False, keep_bits=False, stream=None, keep_llr=False):
        """keep_llr (c8p_rx): False / 0 = the LLR stream stays on chip, True / 1 = it goes through C8P_BUF_LLR (k_rx_1ss), 2 = on chip with a
        copy of every symbol's LLRs written to C8P_BUF_LLR on the way (the parity tests' view of the default kernels)"""
        self.L = load_library()
        cfg = abi.Cfg(abi.ABI_VERSION, device, int(profile), int(keep_bits), stream, int(keep_llr))
        self.ctx = self.L.c8p_create(C.byref(cfg))
        if not self.ctx:
            raise RuntimeError("c8p_create failed: no usable HIP device %d (there is no CPU fallback)" % device)
        self._keep = None

    def close(self):
        if self.ctx:
            self.L.c8p_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError("c8p error %d: %s" % (rc, self.L.c8p_last_error(self.ctx).decode()))

    @staticmethod
    def _arena(x, dtype):
        if isinstance(x, DevArray):
            return x.ptr, x.n, abi.MEM_DEVICE, x
        a = np.ascontiguousarray(x, dtype)
        return a.ctypes.data, a.size, abi.MEM_HOST, a

    def _batch(self, pk, iq, chan):
        pk = np.ascontiguousarray(pk, abi.PKT_IN_DT)
        iq_p, n_iq, m1, k1 = self._arena(iq, np.complex64)
        ch_p, n_ch, m2, k2 = self._arena(chan, np.complex64)
        if m1 != m2:
            raise ValueError("iq and chan must live in the same memory kind")
        b = abi.BatchIn(len(pk), m1, pk.ctypes.data, iq_p, n_iq, ch_p, n_ch)
        self._keep = (pk, k1, k2, b)
        self.n = len(pk)
        return b

    def demod(self, pk, iq, chan):
        self._chk(self.L.c8p_demod(self.ctx, C.byref(self._batch(pk, iq, chan))))

    def rx(self, pk, iq, chan):
        self._chk(self.L.c8p_rx(self.ctx, C.byref(self._batch(pk, iq, chan))))

    def decode(self, tags, llr, mu=None):
        tags = np.ascontiguousarray(tags, abi.PKT_TAGS_DT)
        l_p, n_l, m, k1 = self._arena(llr, np.float32)
        mu_p, k2 = None, None
        if mu is not None:
            mu_p, _, m2, k2 = self._arena(mu, np.complex64)
            if m2 != m:
                raise ValueError("llr and mu2x1chan must live in the same memory kind")
        d = abi.DecodeIn(len(tags), m, tags.ctypes.data, l_p, n_l, mu_p)
        self._keep = (tags, k1, k2, d)
        self.n = len(tags)
        self._chk(self.L.c8p_decode(self.ctx, C.byref(d)))

    def signal(self, evts, raw, consume=False, n_streams=0, scan_from=None):
        """signal / signal2 front-end: sync events + raw samples -> boundary descriptors, de-rotated items, chan.
        consume=True: evts are one stream's flags in order and the engine applies the block's scan rule (flags inside a
        copied packet are never looked at, SIG_SKIPPED); the kept packets lie back to back in BUF_SIG_IQ; returns the
        descriptors (host copy of BUF_SIG_PKT; n_items = 0 for flags that yield no packet)."""
        evts = np.ascontiguousarray(evts, abi.SIG_IN_DT)
        r_p, n_r, m, k1 = self._arena(raw, np.complex64)
        pk = np.zeros(len(evts), abi.PKT_IN_DT) if consume else None
        K = int(n_streams)
        if K > 1:       # K independent streams: evts grouped by evts["stream"]; scan_from = K positions (None: zeros)
            sf = np.ascontiguousarray(scan_from if scan_from is not None else np.zeros(K), np.uint64)
            self.scan_next = np.zeros(K, np.uint64)
            d = abi.SignalIn(len(evts), m, evts.ctypes.data, r_p, n_r, 1, K, pk.ctypes.data if len(evts) else None, 0,
                             self.scan_next.ctypes.data, sf.ctypes.data)
        else:
            sf = None
            self.scan_next = np.zeros(1, np.uint64)
            d = abi.SignalIn(len(evts), m, evts.ctypes.data, r_p, n_r, int(bool(consume)), 0, pk.ctypes.data if consume and len(evts) else None,
                             int(scan_from or 0), self.scan_next.ctypes.data if consume else None, None)
        self._keep = (evts, k1, d, pk, sf)
        self.n = len(evts)
        self._chk(self.L.c8p_signal(self.ctx, C.byref(d)))
        return pk

    def detect(self, raw, keep=False, n_hist=0, state=None, stream=False):
        """presiso + trigger + sync over one raw stream segment (antenna 0); returns the sync events (host copy).
        stream=True: also returns (state at the new commit point, commit index into raw) for the next call, which gets
        raw[commit - DET_HIST:] + new items, n_hist = DET_HIST (or commit, if smaller) and that state."""
        r_p, n_r, m, k1 = self._arena(raw, np.complex64)
        st_out = abi.DetState()
        commit = C.c_uint64(0)
        d = abi.DetectIn(m, int(keep), r_p, n_r, int(n_hist), C.addressof(state) if state is not None else None,
                         C.addressof(st_out), C.addressof(commit), None, 0)
        self._keep = (k1, d, state, st_out, commit)
        self._chk(self.L.c8p_detect(self.ctx, C.byref(d)))
        ev = self.fetch(abi.BUF_DET_EVT, abi.SYNC_EVT_DT)
        return (ev, st_out, commit.value) if stream else ev

    def detect_multi(self, raw, streams):
        """c8p_detect_multi: K independent streams whose items lie in one arena.  streams = list of dicts {off, n, n_hist, state
        (abi.DetState or None)}; returns (events per stream [list of SYNC_EVT_DT arrays, index relative to off], the list of dicts
        updated with commit / trig_off / state)"""
        r_p, n_r, m, k1 = self._arena(raw, np.complex64)
        K = len(streams)
        arr = (abi.DetStream * max(K, 1))()
        for s, d in enumerate(streams):
            arr[s].off, arr[s].n, arr[s].n_hist = int(d["off"]), int(d["n"]), int(d.get("n_hist", 0))
            st = d.get("state")
            arr[s].has_state = int(st is not None)
            if st is not None:
                C.memmove(C.addressof(arr[s].state), C.addressof(st), C.sizeof(abi.DetState))
        din = abi.DetectMultiIn(m, K, r_p, n_r, C.addressof(arr), None, 0)
        self._keep = (k1, din, arr)
        self._chk(self.L.c8p_detect_multi(self.ctx, C.byref(din)))
        ev = self.fetch(abi.BUF_DET_EVT, abi.SYNC_EVT_DT)
        out, evs, o = [], [], 0
        for s in range(K):
            st = abi.DetState()
            C.memmove(C.addressof(st), C.addressof(arr[s].state), C.sizeof(abi.DetState))
            out.append(dict(streams[s], commit=int(arr[s].commit), trig_off=int(arr[s].trig_off), state=st, n_evt=int(arr[s].n_evt)))
            evs.append(ev[o:o + int(arr[s].n_evt)].copy())
            o += int(arr[s].n_evt)
        assert o == len(ev)
        return evs, out

    def reserve(self, **kw):
        """c8p_reserve(fields of c8p_reserve_in by name)"""
        r = abi.ReserveIn()
        for k, v in kw.items():
            setattr(r, k, int(v))
        self._chk(self.L.c8p_reserve(self.ctx, C.byref(r)))

    def counter(self, which):
        return int(self.L.c8p_counter(self.ctx, int(which)))

    def tx(self, pkts, psdu, tail_gap=0, mu=None, bfq=None):
        """transmit chain: pkts = TX_PKT_DT records, psdu = byte arena (host array or DevArray); MU-MIMO packets
        (format F_VHT_MU) take their second user from mu (TX_MU_DT records, in order) and their steering matrices from
        bfq (complex64 arena of the same memory kind as psdu, 256 items per matrix)"""
        pkts = np.ascontiguousarray(pkts, abi.TX_PKT_DT)
        p_p, n_p, m, k1 = self._arena(psdu, np.uint8)
        mu_p, n_mu, q_p, n_q, k2 = None, 0, None, 0, None
        if mu is not None and len(mu):
            mu = np.ascontiguousarray(mu, abi.TX_MU_DT)
            mu_p, n_mu = mu.ctypes.data, len(mu)
            q_p, n_q, m2, k2 = self._arena(bfq, np.complex64)
            if m2 != m:
                raise ValueError("psdu and bfq must be of the same memory kind")
        d = abi.TxIn(len(pkts), m, pkts.ctypes.data, p_p, n_p, int(tail_gap), n_mu, mu_p, q_p, n_q)
        self._keep = (pkts, k1, d, mu, k2)
        self._chk(self.L.c8p_tx(self.ctx, C.byref(d)))

    def sync(self):
        self._chk(self.L.c8p_sync(self.ctx))

    def ticket(self):
        """id of the latest submit; select(ticket) points buffer() / fetch() at it for as long as its slot is not reused"""
        return self.L.c8p_ticket(self.ctx)

    def select(self, ticket):
        self._chk(self.L.c8p_select(self.ctx, int(ticket)))

    def selftest(self, which, operands):
        """device arithmetic primitives over operand tuples (0: |z| of (n,2); 1: a/b of (n,4) -> (n,2); 2: sqrtf(x^2+y^2)/e of (n,3))"""
        a = np.ascontiguousarray(operands, np.float32)
        n = a.shape[0]
        out = np.zeros((n, 2) if which == 1 else n, np.float32)
        self._chk(self.L.c8p_selftest(self.ctx, which, a.ctypes.data, out.ctypes.data, n))
        return out

    def done(self, ticket):
        """c8p_done: True once the submit behind `ticket` has finished; never blocks"""
        r = self.L.c8p_done(self.ctx, int(ticket))
        if r < 0:
            self._chk(r)
        return r == 1

    def wait(self, ticket):
        """wait for ONE submit (uploads, kernels, the fetch_async copies behind it)"""
        self._chk(self.L.c8p_wait(self.ctx, int(ticket)))

    def fetch_async(self, which, dst_ptr, offset_bytes, nbytes):
        """enqueue a copy of a result arena of the selected submit behind its kernels; dst_ptr = pinned host or device memory"""
        self._chk(self.L.c8p_fetch_async(self.ctx, which, C.c_void_p(int(dst_ptr)), int(offset_bytes), int(nbytes)))

    def buffer(self, which):
        nbytes = C.c_uint64(0)
        p = self.L.c8p_buffer(self.ctx, which, C.byref(nbytes))
        return p, nbytes.value

    def fetch(self, which, dtype, offset_items=0, n_items=None):
        dt = np.dtype(dtype)
        _, nbytes = self.buffer(which)
        if n_items is None:
            n_items = nbytes // dt.itemsize - offset_items
        out = np.zeros(n_items, dt)
        self._chk(self.L.c8p_fetch(self.ctx, which, out.ctypes.data, offset_items * dt.itemsize, n_items * dt.itemsize))
        return out

    def tags(self):
        return self.fetch(abi.BUF_TAGS, abi.PKT_TAGS_DT)

    def decstat(self):
        """(range-check flags, float-step block counts, negative-part bounds) of the last decode: C8P_BUF_DECSTAT, 3 x uint32 per packet"""
        a = self.fetch(abi.BUF_DECSTAT, np.uint32)
        n = len(a) // 3
        return a[:n], a[n:2 * n], a[2 * n:].view(np.float32)

    def outs(self):
        return self.fetch(abi.BUF_OUT, abi.PKT_OUT_DT)

    def kernel_times(self):
        res = {}
        for k in range(abi.K_COUNT):
            ms, cnt = C.c_double(0), C.c_int64(0)
            self.L.c8p_kernel_time(self.ctx, k, C.byref(ms), C.byref(cnt))
            res[abi.K_NAMES[k]] = (ms.value, cnt.value)
        return res

    def profile_reset(self):
        self.L.c8p_profile_reset(self.ctx)

    # convenience: everything of a finished rx() as python objects (test / smoke use)
    def results(self, want_llr=True, want_bits=False):
        self.sync()
        tags, outs = self.tags(), self.outs()
        pdu = self.fetch(abi.BUF_PDU, np.uint8)
        llr = self.fetch(abi.BUF_LLR, np.float32) if want_llr and self.buffer(abi.BUF_LLR)[1] else None
        mu = self.fetch(abi.BUF_MUCHAN, np.complex64) if self.buffer(abi.BUF_MUCHAN)[1] else None
        bits = self.fetch(abi.BUF_BITS, np.uint8) if want_bits else None
        sbits = self.fetch(abi.BUF_SBITS, np.uint8) if want_bits else None
        res = []
        for p in range(len(tags)):
            t, o = tags[p], outs[p]
            r = dict(tags=t, out=o)
            area = pdu[int(o["pdu_off"]):int(o["pdu_off"]) + int(o["pdu_bytes"])].tobytes()
            r["pdus"] = abi.split_pdus(area, int(o["n_pdu"]))
            if llr is not None:
                tot = int(t["total"]) if t["status"] == abi.ST_OK else 0
                r["llr"] = llr[int(t["llr_off"]):int(t["llr_off"]) + tot]
            if mu is not None:
                r["mu"] = mu[128 * p:128 * p + 128]
            if want_bits:
                T = int(t["trellis"]) if t["status"] == abi.ST_OK else 0
                r["bits"] = bits[int(o["bits_off"]):int(o["bits_off"]) + T]
                r["sbits"] = sbits[int(o["bits_off"]):int(o["bits_off"]) + T]
            res.append(r)
        return res
