"""ctypes driver of libc8p_blocks.so — the C++ mirror of gr::ieee80211::demod / demod2 / decode / signal / signal2.

Feeds tagged item streams through general_work() in scheduler-sized chunks and collects the output stream,
its tags, and the published PDUs.  Plumbing for tests and examples; a GNU Radio install uses the C++ classes
directly.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libc8p_blocks.so")
# (C8P_LIB: another build of the same library, as in engine.py — A/B runs; still no fallback)
ENGINE_LIB = os.environ.get("C8P_LIB") or os.path.join(_HERE, "libc8p_rx.so")
_LIB = None

DEMOD, DEMOD2, DECODE, SIGNAL, SIGNAL2, DETECT = 0, 1, 2, 3, 4, 5


def RX_CHAIN(n_streams):
    """kind of the rx_chain mirror (rx.grc's whole receive chain for n_streams radios as one block): n_streams complex input ports,
    max_batch = items per stream and round, messages() / message_streams() = the PDUs and whose they are"""
    return 100 + int(n_streams)


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libc8p_blocks.so is not built (make -C gr-ieee80211_amd/host)")
        from .engine import prefer_torch_hip_runtime
        prefer_torch_hip_runtime()         # (libc8p_blocks.so dlopens the engine library, which brings in a HIP runtime)
        L = C.CDLL(LIB_PATH)
        L.c8ph_create.restype = C.c_void_p
        L.c8ph_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_int]
        L.c8ph_create_ex.restype = C.c_void_p
        L.c8ph_create_ex.argtypes = [C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int]
        L.c8ph_create_ex2.restype = C.c_void_p
        L.c8ph_create_ex2.argtypes = [C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int]
        L.c8ph_set_poison.argtypes = [C.c_void_p, C.c_int]
        L.c8ph_set_relocate.argtypes = [C.c_void_p, C.c_int]
        L.c8ph_set_move_out.argtypes = [C.c_void_p, C.c_int]
        L.c8ph_moves_done.argtypes = [C.c_void_p]; L.c8ph_moves_done.restype = C.c_long
        L.c8ph_try_move.argtypes = [C.c_int]
        L.c8ph_input_pinned.argtypes = [C.c_void_p, C.c_int]
        L.c8ph_output_pinned.argtypes = [C.c_void_p]
        L.c8ph_fused_state.argtypes = [C.c_void_p]
        L.c8ph_tags_emitted.argtypes = [C.c_void_p]
        L.c8ph_tags_emitted.restype = C.c_long
        L.c8ph_run_chain.restype = C.c_long
        L.c8ph_run_chain.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.c8ph_destroy.argtypes = [C.c_void_p]
        L.c8p_rawchain_create.restype = C.c_void_p
        L.c8p_rawchain_create.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        L.c8p_rawchain_destroy.argtypes = [C.c_void_p]
        L.c8p_rawchain_stream.restype = C.c_int
        L.c8p_rawchain_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_char_p, C.c_int]
        L.c8p_rawchain_run.restype = C.c_int
        L.c8p_rawchain_run.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_int]
        L.c8p_rawchain_run_once.restype = C.c_int
        L.c8p_rawchain_run_once.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_int]
        L.c8p_rawchain_stream_once.restype = C.c_int
        L.c8p_rawchain_stream_once.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_char_p, C.c_int]
        L.c8p_rawchain_timeline.restype = C.c_int64
        L.c8p_rawchain_timeline.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.c8p_rawchain_live.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.c8p_rawchain_push.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_int32]
        L.c8p_rawchain_pushv.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
        L.c8p_rawchain_push2.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32]
        L.c8p_rawchain_round.argtypes = [C.c_void_p]
        L.c8p_rawchain_round_async.argtypes = [C.c_void_p]
        L.c8p_rawchain_ingested.argtypes = [C.c_void_p]
        L.c8p_rawchain_drain.argtypes = [C.c_void_p, C.c_void_p]
        L.c8p_rawchain_error.restype = C.c_char_p
        L.c8p_rawchain_error.argtypes = [C.c_void_p]
        L.c8ph_push_items.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_long]
        L.c8ph_push_tag_long.argtypes = [C.c_void_p, C.c_uint64, C.c_char_p, C.c_long]
        L.c8ph_push_tag_float.argtypes = [C.c_void_p, C.c_uint64, C.c_char_p, C.c_float]
        L.c8ph_push_tag_c32.argtypes = [C.c_void_p, C.c_uint64, C.c_char_p, C.c_void_p, C.c_int]
        L.c8ph_run.restype = C.c_long
        L.c8ph_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.c8ph_run_live.restype = C.c_long
        L.c8ph_run_live.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.c8ph_host_phase_reset.argtypes = [C.c_void_p]
        L.c8ph_host_phase_us.restype = C.c_char_p
        L.c8ph_host_phase_us.argtypes = [C.c_void_p]
        L.c8ph_set_jitter.argtypes = [C.c_void_p, C.c_ulonglong]
        L.c8ph_set_capture.argtypes = [C.c_void_p, C.c_int]
        L.c8ph_error.restype = C.c_char_p
        L.c8ph_error.argtypes = [C.c_void_p]
        for f in ("c8ph_out_count", "c8ph_tag_count", "c8ph_msg_count", "c8ph_dbg_count"):
            getattr(L, f).restype = C.c_long
            getattr(L, f).argtypes = [C.c_void_p]
        L.c8ph_out_copy.argtypes = [C.c_void_p, C.c_void_p]
        L.c8ph_out1_copy.argtypes = [C.c_void_p, C.c_void_p]
        L.c8ph_tag_get.argtypes = [C.c_void_p, C.c_long, C.POINTER(C.c_uint64), C.c_char_p, C.c_int, C.POINTER(C.c_long),
                                   C.POINTER(C.c_float), C.POINTER(C.c_int)]
        L.c8ph_tag_vec.argtypes = [C.c_void_p, C.c_long, C.c_void_p]
        L.c8ph_msg_len.restype = C.c_long
        L.c8ph_msg_len.argtypes = [C.c_void_p, C.c_long]
        L.c8ph_create_rx_chain.restype = C.c_void_p
        L.c8ph_create_rx_chain.argtypes = [C.c_int, C.c_int, C.c_float, C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_int]
        L.c8ph_msg_stream.restype = C.c_long
        L.c8ph_msg_stream.argtypes = [C.c_void_p, C.c_long]
        L.c8ph_msg_copy.argtypes = [C.c_void_p, C.c_long, C.c_void_p]
        L.c8ph_dbg_get.argtypes = [C.c_void_p, C.c_long, C.c_char_p, C.c_int]
        _LIB = L
    return _LIB


class Block:
    def __init__(self, kind, mupos=0, ifdebug=False, engine_lib=None, max_batch=256, flush_when_idle=True, fused=None, item_format=0, scale=0.0,
                 pinned_io=None, defer_consume=None):
        """fused: batching_cfg::fused (None = the defaults: the environment's C8P_FUSED, and when that is unset the flowgraph decides —
        a demod whose LLR port is read by the decode mirror only, as in run_chain, runs fused); item_format / scale: rx_chain only — the
        ports carry complex64 (0), int16 pairs (abi.FMT_SC16) or int8 pairs (abi.FMT_SC8) and the engine converts with `scale`;
        pinned_io / defer_consume: batching_cfg's fields of those names (None = their defaults, C8P_PINNED_IO / C8P_DEFER_CONSUME): the
        harness allocates a port's buffer through the block's buffer type — pinned host memory — when the block names one"""
        self.L = lib()
        err = C.create_string_buffer(512)
        if kind >= 100 and item_format:
            self.h = self.L.c8ph_create_rx_chain(kind - 100, int(item_format), float(scale), (engine_lib or ENGINE_LIB).encode(), max_batch,
                                                 int(flush_when_idle), err, 512)
        else:
            self.h = self.L.c8ph_create_ex2(kind, mupos, int(ifdebug), (engine_lib or ENGINE_LIB).encode(), max_batch,
                                            int(flush_when_idle), -1 if fused is None else int(fused), -1 if pinned_io is None else int(pinned_io),
                                            -1 if defer_consume is None else int(defer_consume), err, 512)
        if not self.h:
            raise RuntimeError(err.value.decode())
        self.kind = kind

    def close(self):
        if self.h:
            self.L.c8ph_destroy(self.h)
            self.h = None

    def push_items(self, port, arr):
        a = np.ascontiguousarray(arr)
        self.L.c8ph_push_items(self.h, port, a.ctypes.data, len(a))

    def push_tag(self, offset, key, value):
        k = key.encode()
        if isinstance(value, (int, np.integer)):
            self.L.c8ph_push_tag_long(self.h, offset, k, int(value))
        elif isinstance(value, (float, np.floating)):
            self.L.c8ph_push_tag_float(self.h, offset, k, float(value))
        else:
            v = np.ascontiguousarray(value, np.complex64)
            self.L.c8ph_push_tag_c32(self.h, offset, k, v.ctypes.data, len(v))

    def set_capture(self, on):
        """off: run() counts the produced items instead of keeping them (throughput measurements)"""
        self.L.c8ph_set_capture(self.h, int(on))

    def set_poison(self, on):
        """from now on consumed input items are overwritten at once (a test: nothing may still be reading them)"""
        self.L.c8ph_set_poison(self.h, int(on))

    def set_relocate(self, every):
        """every n-th general_work call of run() finds the unread items of its pinned input buffers moved to other memory (announced with
        c8p::buffer_will_move, the old memory overwritten): what a single-mapped GNU Radio buffer does at its end.  0: never"""
        self.L.c8ph_set_relocate(self.h, int(every))

    def set_move_out(self, every):
        """every n-th general_work call of run() finds its output position moved somewhere else in the output buffer WITHOUT the items
        placed ahead being drained first (output_ahead), the move bracketed by c8p::buffer_move_scope and everything else in the buffer
        overwritten: what GNU Radio's single-mapped buffers do from their blocked callbacks (gnuradio/lib/c8p_pinned_buffer.h).  0: never"""
        self.L.c8ph_set_move_out(self.h, int(every))

    def moves_done(self):
        return int(self.L.c8ph_moves_done(self.h))

    def input_pinned(self, port=0):
        return bool(self.L.c8ph_input_pinned(self.h, port))

    def output_pinned(self):
        return bool(self.L.c8ph_output_pinned(self.h))

    def fused_state(self):
        """demod / demod2: True / False = the front runs fused / plain; None = not decided yet (batching_cfg::fused_auto decides at the
        first general_work call, from who reads the LLR port) or not a demod block"""
        v = int(self.L.c8ph_fused_state(self.h))
        return None if v < 0 else bool(v)

    def tags_emitted(self):
        """stream tags the block has written so far (every add_item_tag call of its life)"""
        return int(self.L.c8ph_tags_emitted(self.h))

    def set_jitter(self, seed):
        """random item counts per general_work call from now on (0 = off): what a busy scheduler hands a block"""
        self.L.c8ph_set_jitter(self.h, int(seed))

    def run(self, chunk=4096, nout=8192, flush_at_end=True):
        n = self.L.c8ph_run(self.h, chunk, nout, int(flush_at_end))
        if n < 0:
            raise RuntimeError(self.L.c8ph_error(self.h).decode())
        return n

    def run_live(self, chunk=4096, nout=8192, linger_ms=50):
        """a LIVE scheduler: general_work is only called while there is input (or the block's forecast asks for none), nothing is
        flushed; after the input is used up the loop idles for linger_ms, woken by the block's latency timer"""
        n = self.L.c8ph_run_live(self.h, chunk, nout, int(linger_ms))
        if n < 0:
            raise RuntimeError(self.L.c8ph_error(self.h).decode())
        return n

    def host_phase_reset(self):
        self.L.c8ph_host_phase_reset(self.h)

    def host_phase_us(self):
        """{phase: microseconds} of host time inside this block since it was made (exclusive phases)"""
        txt = self.L.c8ph_host_phase_us(self.h).decode()
        return {k: int(v) for k, v in (kv.split("=") for kv in txt.split(";") if kv)}

    def output(self, port=0):
        """the block's output stream: float LLRs (demod), complex items (signal / signal2, port 0 or 1)"""
        n = self.L.c8ph_out_count(self.h)
        out = np.zeros(n, np.float32)
        if n:
            (self.L.c8ph_out1_copy if port == 1 else self.L.c8ph_out_copy)(self.h, out.ctypes.data)
        if self.kind == DETECT:
            return out.astype(np.uint8)
        return out.view(np.complex64) if self.kind in (SIGNAL, SIGNAL2) else out

    def tags(self):
        res = []
        for i in range(self.L.c8ph_tag_count(self.h)):
            off, l, f, vn = C.c_uint64(0), C.c_long(0), C.c_float(0), C.c_int(0)
            key = C.create_string_buffer(64)
            kind = self.L.c8ph_tag_get(self.h, i, C.byref(off), key, 64, C.byref(l), C.byref(f), C.byref(vn))
            if kind == 0:
                val = l.value
            elif kind == 1:
                val = f.value
            else:
                v = np.zeros(vn.value, np.complex64)
                self.L.c8ph_tag_vec(self.h, i, v.ctypes.data)
                val = v
            res.append((off.value, key.value.decode(), val))
        return res

    def messages(self):
        res = []
        for i in range(self.L.c8ph_msg_count(self.h)):
            n = self.L.c8ph_msg_len(self.h, i)
            b = np.zeros(n, np.uint8)
            self.L.c8ph_msg_copy(self.h, i, b.ctypes.data)
            res.append((n, b.tobytes()))
        return res

    def message_streams(self):
        """rx_chain: the input port (radio) of every message, in messages() order (-1 for the other blocks)"""
        return [int(self.L.c8ph_msg_stream(self.h, i)) for i in range(self.L.c8ph_msg_count(self.h))]

    def debug_lines(self):
        res = []
        for i in range(self.L.c8ph_dbg_count(self.h)):
            buf = C.create_string_buffer(1024)
            self.L.c8ph_dbg_get(self.h, i, buf, 1024)
            res.append(buf.value.decode())
        return res


class _RawSeg(C.Structure):
    _fields_ = [("raw", C.c_void_p), ("n", C.c_uint64), ("ant1_off", C.c_uint64), ("mem", C.c_int32), ("rsv", C.c_int32)]


class _RawStage(C.Structure):
    _fields_ = [("busy", C.c_double), ("wait_in", C.c_double), ("wait_out", C.c_double)]


class _RawStats(C.Structure):
    _fields_ = [("seconds", C.c_double), ("segments", C.c_int64), ("items", C.c_int64), ("events", C.c_int64),
                ("packets", C.c_int64), ("pdus", C.c_int64), ("first_segment_s", C.c_double), ("steady_s", C.c_double),
                ("steady_segments", C.c_int64), ("steady_items", C.c_int64), ("steady_packets", C.c_int64), ("steady_pdus", C.c_int64),
                ("detect", _RawStage), ("signal", _RawStage), ("rx", _RawStage), ("arena_growths", C.c_int64), ("arena_bytes", C.c_int64),
                ("det_giveups", C.c_int64), ("det_ticket_scans", C.c_int64)]


class _RawMark(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("det_begin", "det_end", "sig_begin", "sig_end", "rx_submit", "rx_done")]


class _RawOut(C.Structure):
    _fields_ = [("segment", C.c_int64), ("n_events", C.c_int32), ("n_packets", C.c_int32), ("pkts", C.c_void_p), ("outs", C.c_void_p),
                ("rx", C.c_void_p), ("pkt_stream", C.c_void_p), ("pkt_index", C.c_void_p)]


class _RawCfg(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("lib", C.c_char_p), ("reserve_seg_items", C.c_uint64),
                ("reserve_seg_events", C.c_uint32), ("reserve_seg_packets", C.c_uint32), ("reserve_pkt_items", C.c_uint32),
                ("reserve_host", C.c_int32), ("reserve_ant", C.c_int32), ("streams", C.c_uint32), ("stream_ant", C.c_uint32),
                ("stream_round_items", C.c_uint64), ("sc16_scale", C.c_float), ("sc8_scale", C.c_float)]


def _stats_dict(st):
    d = {}
    for k, t in _RawStats._fields_:
        v = getattr(st, k)
        d[k] = {f: getattr(v, f) for f, _ in _RawStage._fields_} if t is _RawStage else v
    return d


_CB_T = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(_RawOut))


def _segment_callback(on_segment, engine_lib):
    """the C callback that hands on_segment(dict) numpy copies of one segment's descriptors, results and PDU arena"""
    from . import abi
    eng = C.CDLL(engine_lib)
    eng.c8p_buffer.restype = C.c_void_p
    eng.c8p_buffer.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]
    eng.c8p_fetch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.c_uint64]

    def cb(_, out):
        o = out.contents
        n = o.n_packets
        pk = np.frombuffer(C.string_at(o.pkts, n * C.sizeof(abi.PktIn)), abi.PKT_IN_DT).copy() if n else np.zeros(0, abi.PKT_IN_DT)
        outs = np.frombuffer(C.string_at(o.outs, n * C.sizeof(abi.PktOut)), abi.PKT_OUT_DT).copy() if n else np.zeros(0, abi.PKT_OUT_DT)
        strm = np.frombuffer(C.string_at(o.pkt_stream, 4 * n), np.uint32).copy() if n else np.zeros(0, np.uint32)
        idx = np.frombuffer(C.string_at(o.pkt_index, 8 * n), np.uint64).copy() if n else np.zeros(0, np.uint64)
        nb = C.c_uint64(0)
        eng.c8p_buffer(o.rx, abi.BUF_PDU, C.byref(nb))
        pdu = np.zeros(nb.value, np.uint8)
        if nb.value:
            eng.c8p_fetch(o.rx, abi.BUF_PDU, pdu.ctypes.data, 0, nb.value)
        on_segment({"segment": o.segment, "n_events": o.n_events, "pkts": pk, "outs": outs, "pdu_arena": pdu, "pkt_stream": strm, "pkt_index": idx})

    return _CB_T(cb)


class RawChain:
    """host/c8p_rawchain.h: a handle that owns the pipeline's four engine contexts (streams, arenas) for as long as it lives.
    reserve_*: arena shapes sized at create; streams / stream_round_items: live ingest of K independent streams (push / round / drain)."""

    def __init__(self, device=0, engine_lib=None, reserve_seg_items=0, reserve_seg_events=0, reserve_seg_packets=0, reserve_pkt_items=0,
                 reserve_host=False, reserve_ant=1, streams=0, stream_round_items=0, stream_ant=1, sc16_scale=0.0, sc8_scale=0.0):
        self.L = lib()
        self.engine_lib = engine_lib or ENGINE_LIB
        cfg = _RawCfg(C.sizeof(_RawCfg), device, self.engine_lib.encode(), int(reserve_seg_items), int(reserve_seg_events),
                      int(reserve_seg_packets), int(reserve_pkt_items), int(bool(reserve_host)), int(reserve_ant), int(streams), int(stream_ant),
                      int(stream_round_items), float(sc16_scale), float(sc8_scale))
        err = C.create_string_buffer(512)
        self.h = self.L.c8p_rawchain_create(C.byref(cfg), err, len(err))
        if not self.h:
            raise RuntimeError("c8p_rawchain_create: " + err.value.decode())
        self._cb = None

    def close(self):
        if self.h:
            self.L.c8p_rawchain_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, err=None):
        if rc != 0:
            msg = err.value.decode() if err is not None and err.value else (self.L.c8p_rawchain_error(self.h) or b"").decode()
            raise RuntimeError("c8p_rawchain: %d %s" % (rc, msg))

    def run(self, segments, on_segment=None, stream_items=0, host=False):
        """segments = [(pointer, items[, item offset of antenna 1's stream inside the same allocation: rx2.grc's chain]), ...];
        on_segment(dict) is called per segment in order.  host=True: the pointers are host memory (pinned for the DMA rate).
        stream_items > 0: segments[0] is ONE capture worked through in pieces of that many items, state carried across the cuts.
        Returns the run's c8p_rawchain_stats as a dict."""
        segs = (_RawSeg * len(segments))(*[_RawSeg(int(t[0]), int(t[1]), int(t[2]) if len(t) > 2 else 0, 0 if host else 1, 0) for t in segments])
        st = _RawStats()
        err = C.create_string_buffer(512)
        cbp = _segment_callback(on_segment, self.engine_lib) if on_segment else None
        if stream_items:
            (p0, n0), a1 = segments[0][:2], (segments[0][2] if len(segments[0]) > 2 else 0)
            rc = self.L.c8p_rawchain_stream(self.h, int(p0), int(n0), int(a1), int(stream_items), cbp, None, C.byref(st), err, len(err))
        else:
            rc = self.L.c8p_rawchain_run(self.h, segs, len(segments), cbp, None, C.byref(st), err, len(err))
        self._chk(rc, err)
        return _stats_dict(st)

    def timeline(self):
        """the last run's marks: one dict per segment, seconds since the start of the run"""
        n = self.L.c8p_rawchain_timeline(self.h, None, 0)
        arr = (_RawMark * max(int(n), 1))()
        self.L.c8p_rawchain_timeline(self.h, arr, n)
        return [{k: getattr(arr[i], k) for k, _ in _RawMark._fields_} for i in range(int(n))]

    # live ingest
    def live(self, on_round=None):
        self._cb = _segment_callback(on_round, self.engine_lib) if on_round else None
        self._chk(self.L.c8p_rawchain_live(self.h, self._cb, None))

    @staticmethod
    def _mem(host):
        """host: True = any host memory (C8P_MEM_HOST), False = device memory, "pinned" = host memory the device can read (the caller vouches)"""
        return 2 if host == "pinned" else (0 if host else 1)

    def _items(self, a, sc16):
        """(pointer, items) of a push argument: a (pointer, n) tuple, a complex64 array, or — sc16 — an int16 array of I/Q pairs"""
        if isinstance(a, tuple):
            return int(a[0]), int(a[1])
        a = np.ascontiguousarray(a, np.int8 if sc16 == 8 else (np.int16 if sc16 else np.complex64))
        self._keep = getattr(self, "_keep", [])
        self._keep.append(a)
        return a.ctypes.data, a.size // 2 if sc16 else len(a)

    @staticmethod
    def _fmt(sc16):
        """sc16: False = complex64 items, True / 16 = int16 pairs (C8P_MEM_SC16), 8 = int8 pairs (C8P_MEM_SC8)"""
        return 0x200 if sc16 == 8 else (0x100 if sc16 else 0)

    def push(self, stream, items, host=True, sc16=False):
        """items: a numpy complex64 array (host; kept alive until the next round()) or (pointer, n); host: see _mem;
        sc16: the items are int16 I/Q pairs (C8P_MEM_SC16: pinned or device memory only), n counts pairs"""
        ptr, n = self._items(items, sc16)
        self._chk(self.L.c8p_rawchain_push(self.h, int(stream), ptr, n, self._mem(host) | self._fmt(sc16)))

    def pushv(self, streams, ptrs, counts, host=True, sc16=False):
        """c8p_rawchain_pushv: many pushes in one call; streams / ptrs / counts = integer sequences (ptrs = addresses the caller keeps valid)"""
        st = np.ascontiguousarray(streams, np.uint32)
        pt = np.ascontiguousarray(ptrs, np.uint64)
        ct = np.ascontiguousarray(counts, np.uint64)
        self._chk(self.L.c8p_rawchain_pushv(self.h, len(st), st.ctypes.data, pt.ctypes.data, ct.ctypes.data, self._mem(host) | self._fmt(sc16)))

    def push2(self, stream, ant0, ant1, host=True, sc16=False):
        """two antennas per stream (stream_ant=2): numpy arrays of equal length, or (pointer, n) tuples"""
        (p0, n0), (p1, n1) = self._items(ant0, sc16), self._items(ant1, sc16)
        assert n0 == n1
        self._chk(self.L.c8p_rawchain_push2(self.h, int(stream), p0, p1, n0, self._mem(host) | self._fmt(sc16)))

    def round(self, wait=True):
        """wait=False: c8p_rawchain_round_async — the pushed arrays must stay alive until their items are on the device.  They are kept
        here: this round's until the round AFTER the next has been closed (a round call returns only once an arena is free again, and
        of the three arenas the one it frees belongs to the round two back, which c8p_signal — and so that round's gather — is through
        with), or until ingested() / drain()."""
        if wait:
            self._chk(self.L.c8p_rawchain_round(self.h))          # returns when everything pushed so far is on the device
            self._keep, self._held = [], []
        else:
            self._chk(self.L.c8p_rawchain_round_async(self.h))
            held = getattr(self, "_held", [])
            self._held = held[-1:] + [getattr(self, "_keep", [])]   # batches of the last two async rounds
            self._keep = []

    def ingested(self):
        self._chk(self.L.c8p_rawchain_ingested(self.h))
        self._held = []

    def drain(self):
        st = _RawStats()
        try:
            self._chk(self.L.c8p_rawchain_drain(self.h, C.byref(st)))
        finally:
            self._held = []
        return _stats_dict(st)


def rawchain_run(segments, device=0, engine_lib=None, on_segment=None, stream_items=0, host=False):
    """one-shot RawChain(...).run(...): a handle is made for this call alone (its contexts and arenas with it)"""
    rc = RawChain(device=device, engine_lib=engine_lib)
    try:
        return rc.run(segments, on_segment=on_segment, stream_items=stream_items, host=host)
    finally:
        rc.close()


def run_chain(a, b, chunk=1 << 16, nout=1 << 16):
    """a -> b with each block on its own thread, as the GNU Radio scheduler runs a flowgraph (a's items are not captured)"""
    n = a.L.c8ph_run_chain(a.h, b.h, chunk, nout)
    if n < 0:
        raise RuntimeError((a.L.c8ph_error(a.h) or b"").decode() + " | " + (b.L.c8ph_error(b.h) or b"").decode())
    return n


def feed_signal(blk, sync_events, raws):
    """sync byte stream (port 0, flags tagged rad / snr / rssi as sync_impl.cc:124-141 does) + raw antenna streams"""
    n = len(raws[0])
    sync = np.zeros(n, np.uint8)
    for e in sync_events:
        i = int(e["index"])
        sync[i] = 1
        blk.push_tag(i, "rad", float(e["rad"])); blk.push_tag(i, "snr", float(e["snr"])); blk.push_tag(i, "rssi", float(e["rssi"]))
    blk.push_items(0, sync)
    for a, x in enumerate(raws):
        blk.push_items(1 + a, np.ascontiguousarray(x, np.complex64))


def feed_demod(blk, pk, iq, chan, base=0):
    """Lay a batch out as the tagged stream signal/signal2 would produce: packets back to back.  base = items already pushed
    into the block (a second pass over the same block).  Returns the number of items pushed per port."""
    pos = base
    for p in pk:
        n = int(p["n_items"])
        for a in range(int(p["n_ant"])):
            blk.push_items(a, iq[int(p["iq_off"][a]):int(p["iq_off"][a]) + n])
        blk.push_tag(pos, "cfo", float(p["cfo"])); blk.push_tag(pos, "snr", float(p["snr"])); blk.push_tag(pos, "rssi", float(p["rssi"]))
        blk.push_tag(pos, "seq", int(p["seq"])); blk.push_tag(pos, "mcs", int(p["lsig_mcs"])); blk.push_tag(pos, "len", int(p["lsig_len"]))
        blk.push_tag(pos, "nsamp", int(p["lsig_nsamp"])); blk.push_tag(pos, "chan", chan[int(p["chan_off"]):int(p["chan_off"]) + 64])
        pos += n
    return pos - base
