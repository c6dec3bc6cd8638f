"""The engine's copy streams (include/c8p_rx.h, c8p_copy_*): what orders what.  ADVICE r5 found two gaps that depend on timing — a
c8p_copy_async that was not ordered behind C8P_PIECE_UNORDERED gathers on the copy stream's twin, and a mark that forgot the twin's
earlier events — so these tests make the timing: the gather that must be waited for is LONG (hundreds of megabytes out of pinned host
memory, ~5 ms over the link) and what depends on it follows at once."""
import ctypes as C

import numpy as np
import pytest

from c8p_testlib import abi
from gr_ieee80211_amd import engine

pytestmark = pytest.mark.gpu
UNORDERED = 2                       # C8P_PIECE_UNORDERED


def pinned(L, n_items, fill):
    p = L.c8p_host_alloc(8 * n_items)
    assert p
    a = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(2 * n_items,))
    a[:] = fill
    return p, a


def test_copy_async_runs_behind_unordered_gathers_on_either_copy_stream():
    eng = engine.Engine(device=0)
    L = eng.L
    n_small, n_big = 1 << 10, 24 << 20                              # 8 KB and 192 MB of complex64 items
    hs, a_s = pinned(L, n_small, 1.0)
    hb, a_b = pinned(L, n_big, 7.0)
    a_b[-2:] = (123.0, 456.0)                                        # the last item of the big piece
    arena = L.c8p_dev_alloc(eng.ctx, 8 * (n_small + n_big))
    side = L.c8p_dev_alloc(eng.ctx, 8 * 16)
    assert arena and side
    try:
        for rep in range(3):                                         # (whichever of the two streams the big gather lands on)
            eng._chk(L.c8p_copy_async(eng.ctx, C.c_void_p(arena + 8 * n_small), C.c_void_p(hs), 8 * 16))      # stale contents where the tail will be
            eng._chk(L.c8p_copy_wait(eng.ctx))
            p1 = (abi.Piece * 1)(abi.Piece(hs, 0, n_small, 1, abi.FMT_C32, 0.0, UNORDERED))
            p2 = (abi.Piece * 1)(abi.Piece(hb, n_small, n_big, 1, abi.FMT_C32, 0.0, UNORDERED))
            for k in range(1 + rep % 2):                             # an odd / even number of short gathers first: the long one alternates streams
                eng._chk(L.c8p_copy_gather(eng.ctx, C.c_void_p(arena), p1, 1))
            eng._chk(L.c8p_copy_gather(eng.ctx, C.c_void_p(arena), p2, 1))
            # at once: move the LAST 16 items the long gather writes (what the block fronts do with a partly received packet)
            eng._chk(L.c8p_copy_async(eng.ctx, C.c_void_p(side), C.c_void_p(arena + 8 * (n_small + n_big - 16)), 8 * 16))
            out = np.zeros(32, np.float32)
            eng._chk(L.c8p_copy_async(eng.ctx, C.c_void_p(out.ctypes.data), C.c_void_p(side), 8 * 16))
            eng._chk(L.c8p_copy_wait(eng.ctx))
            assert out[-2] == 123.0 and out[-1] == 456.0 and np.all(out[:-2] == 7.0), (rep, out)
    finally:
        L.c8p_dev_free(eng.ctx, C.c_void_p(arena)); L.c8p_dev_free(eng.ctx, C.c_void_p(side))
        L.c8p_host_free(C.c_void_p(hs)); L.c8p_host_free(C.c_void_p(hb))
        eng.close()


def test_a_mark_covers_the_twin_streams_earlier_gathers():
    """mark M1 after a LONG unordered gather, mark M2 after a short one on the other stream: waiting for M2 alone (what
    gather_lane::wait_range does: the newest overlapping mark) must wait for the long gather too"""
    eng = engine.Engine(device=0)
    L = eng.L
    n_small, n_big = 1 << 10, 24 << 20
    hs, a_s = pinned(L, n_small, 1.0)
    hb, a_b = pinned(L, n_big, 7.0)
    arena = L.c8p_dev_alloc(eng.ctx, 8 * (n_small + n_big))
    assert arena
    try:
        for rep in range(4):
            a_b[:] = float(10 + rep)
            p1 = (abi.Piece * 1)(abi.Piece(hs, 0, n_small, 1, abi.FMT_C32, 0.0, UNORDERED))
            p2 = (abi.Piece * 1)(abi.Piece(hb, n_small, n_big, 1, abi.FMT_C32, 0.0, UNORDERED))
            for k in range(rep % 2):
                eng._chk(L.c8p_copy_gather(eng.ctx, C.c_void_p(arena), p1, 1))
            eng._chk(L.c8p_copy_gather(eng.ctx, C.c_void_p(arena), p2, 1))
            m1 = L.c8p_copy_mark(eng.ctx)
            eng._chk(L.c8p_copy_gather(eng.ctx, C.c_void_p(arena), p1, 1))                                    # the other stream
            m2 = L.c8p_copy_mark(eng.ctx)
            assert m1 and m2 == m1 + 1
            eng._chk(L.c8p_copy_wait_mark(eng.ctx, m2))
            assert L.c8p_copy_mark_done(eng.ctx, m1) == 1            # at once: the earlier mark's gather — on the other stream — is done too
            # ... so overwriting its SOURCE now cannot change what arrived (a scheduler reusing the buffer)
            a_b[-64:] = -1.0
            out = np.zeros(32, np.float32)
            eng._chk(L.c8p_copy_async(eng.ctx, C.c_void_p(out.ctypes.data), C.c_void_p(arena + 8 * (n_small + n_big - 16)), 8 * 16))
            eng._chk(L.c8p_copy_wait(eng.ctx))
            assert np.all(out == float(10 + rep)), (rep, out[:4], out[-4:])
    finally:
        L.c8p_dev_free(eng.ctx, C.c_void_p(arena))
        L.c8p_host_free(C.c_void_p(hs)); L.c8p_host_free(C.c_void_p(hb))
        eng.close()


def test_copy_cus_option_is_taken_before_the_first_copy_only():
    """c8p_option(C8P_OPT_COPY_CUS): how many CUs the ctx's copy streams may use — what the block fronts set to 16; refused once the copy
    streams exist, for values out of range and for unknown options"""
    eng = engine.Engine(device=0)
    L = eng.L
    try:
        assert L.c8p_option(eng.ctx, 1, 16) == 0
        assert L.c8p_option(eng.ctx, 1, 300) == -22 and L.c8p_option(eng.ctx, 99, 1) == -22
        hs, a_s = pinned(L, 1024, 2.0)
        dst = L.c8p_dev_alloc(eng.ctx, 8 * 1024)
        p1 = (abi.Piece * 1)(abi.Piece(hs, 0, 1024, 1, abi.FMT_C32, 0.0, 0))
        eng._chk(L.c8p_copy_gather(eng.ctx, C.c_void_p(dst), p1, 1))
        out = np.zeros(2048, np.float32)
        eng._chk(L.c8p_copy_async(eng.ctx, C.c_void_p(out.ctypes.data), C.c_void_p(dst), 8 * 1024))
        eng._chk(L.c8p_copy_wait(eng.ctx))
        assert np.all(out == 2.0)
        assert L.c8p_option(eng.ctx, 1, 8) == -22           # too late: the streams exist
        L.c8p_dev_free(eng.ctx, C.c_void_p(dst)); L.c8p_host_free(C.c_void_p(hs))
    finally:
        eng.close()
