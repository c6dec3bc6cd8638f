// c8p_rawchain.h — raw 20 MS/s streams -> PDUs on the engine, as a pipeline (host side, plain C ABI).
//
// The reference's rx.grc wires  source -> presiso / trigger / sync -> signal -> demod -> decode  (examples/rx.grc:753-767:
// lib/trigger_impl.cc:59-117, lib/sync_impl.cc:62-192, lib/signal_impl.cc:68-204, then the demod / decode blocks); with
// everything resident on one MI355X the same chain is three engine calls per stretch of stream,
//     c8p_detect (presiso + trigger + sync)  ->  c8p_signal (consume = 1)  ->  c8p_rx (demod + decode),
// and this driver keeps three stretches in flight: i + 2 in c8p_detect while i + 1 is in c8p_signal and i in c8p_rx, each
// stage on its own engine context (own HIP streams) and host thread.  The callback sees every stretch's results in order.
//
// A c8p_rawchain handle owns the four engine contexts (their streams and arenas) for as long as it lives: create it once,
// reserve the arenas for the shape of the work (c8p_rawchain_cfg.reserve_*), and no allocation happens inside a run.
// Three ways to feed it:
//   c8p_rawchain_run     self-contained segments (device or host memory), nothing carried from one to the next
//   c8p_rawchain_stream  ONE resident capture worked through in segments, detection state + scan position carried across the cuts
//   c8p_rawchain_push / _round / _drain   LIVE ingest of K independent streams (K radios): increments of any size per stream,
//                        every round = one c8p_detect_multi over all streams' new items, one c8p_signal, one c8p_rx batch
#pragma once
#include <stdint.h>

#include "c8p_rx.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct c8p_rawchain c8p_rawchain;

typedef struct c8p_rawchain_cfg {
  uint32_t struct_size;        /* sizeof(c8p_rawchain_cfg): a caller built against another layout is refused               */
  int32_t  device;
  const char* lib;             /* path of libc8p_rx.so (NULL: the environment's C8P_RX_LIB, or "libc8p_rx.so")              */
  /* arenas sized at create (c8p_reserve on every context), so that the first segment costs what the others cost.  0 = grow on demand */
  uint64_t reserve_seg_items;  /* items of one segment / round (all streams together)                                       */
  uint32_t reserve_seg_events; /* sync events per segment                                                                   */
  uint32_t reserve_seg_packets;/* packets per segment ...                                                                   */
  uint32_t reserve_pkt_items;  /* ... and items of the longest one at the demod boundary (nsamp + 320)                      */
  int32_t  reserve_host;       /* 1: c8p_rawchain_run segments come from host memory                                        */
  int32_t  reserve_ant;        /* 1 or 2 antennas (0 = 1)                                                                   */
  /* live ingest; streams = 0: c8p_rawchain_push is not available */
  uint32_t streams;            /* K independent streams                                                                     */
  uint32_t stream_ant;         /* antennas per live stream: 0 / 1 = one (rx.grc: signal -> demod), 2 = two (rx2.grc: signal2 ->
                                  demod2; detection listens on antenna 0, c8p_rawchain_push2 delivers both)                 */
  uint64_t stream_round_items; /* most NEW items one stream may receive between two c8p_rawchain_round calls                */
  float    sc16_scale;         /* pushes flagged C8P_MEM_SC16: item = ((float)i * sc16_scale, (float)q * sc16_scale); 0 = 1 / 32767 (UHD's)  */
  float    sc8_scale;          /* pushes flagged C8P_MEM_SC8 (int8 pairs): 0 = 1 / 127                                                       */
} c8p_rawchain_cfg;

typedef struct c8p_rawchain_seg {
  const c8p_c32* raw;      /* device memory: the segment's items (antenna 0), resident until its callback has run */
  uint64_t n;              /* items                                                                               */
  uint64_t ant1_off;       /* 0: one antenna (signal -> demod).  Otherwise rx2.grc's chain (signal2 -> demod2): antenna 1's n items
                              start ant1_off items after raw (the same allocation); detection listens on antenna 0             */
  int32_t  mem;            /* C8P_MEM_DEVICE (1) or C8P_MEM_HOST (0; c8p_rawchain_run only, one antenna): the items come from host memory —
                              pinned (c8p_host_alloc) for the DMA rate — and cross PCIe once, inside c8p_detect; c8p_signal reads the
                              engine's device copy (C8P_BUF_DET_RAW)                                                             */
  int32_t  rsv;
} c8p_rawchain_seg;

typedef struct c8p_rawchain_out {
  int64_t  segment;        /* index of the segment (live ingest: of the round)                                    */
  int32_t  n_events;       /* sync events c8p_detect found                                                        */
  int32_t  n_packets;      /* packets the signal block's scan kept (what c8p_rx decoded)                          */
  const c8p_pkt_in*  pkts; /* [n_packets] their boundary descriptors (host)                                       */
  const c8p_pkt_out* outs; /* [n_packets] per-packet results (host): n_pdu, n_crc_fail, pdu_off, pdu_bytes        */
  c8p_ctx* rx;             /* the rx context with this segment selected: c8p_fetch(rx, C8P_BUF_PDU, ...) reads its PDU arena */
  const uint32_t* pkt_stream; /* [n_packets] which stream a packet came from (0 outside live ingest); packets are grouped by
                                 stream in ascending order and lie in stream order inside a group                  */
  const uint64_t* pkt_index;  /* [n_packets] item index of the packet's sync flag: in its stream (live ingest, counted from the first
                                 item ever pushed), in the capture (c8p_rawchain_stream), in the segment (c8p_rawchain_run)        */
} c8p_rawchain_out;

typedef void (*c8p_rawchain_cb)(void* user, const c8p_rawchain_out* out);

typedef struct c8p_rawchain_stage {     /* where one stage's host thread spent the run, in seconds                */
  double busy;             /* inside engine calls (they block until the stage's GPU work is known to the host)    */
  double wait_in;          /* blocked on the hand-over queue in front of it (detect: on the caller, live ingest)  */
  double wait_out;         /* blocked handing over (the next stage is the slower one); rx: inside the callback    */
} c8p_rawchain_stage;

typedef struct c8p_rawchain_stats {
  double  seconds;         /* wall time of the run                                                                */
  int64_t segments, items, events, packets, pdus;
  double  first_segment_s; /* from the start of the run to the first segment's results (pipeline fill, and whatever the first pass
                              through every context costs: with reserved arenas and warm contexts ~ three stage times)    */
  double  steady_s;        /* wall time over the last steady_segments segments (from the results of the one before them) */
  int64_t steady_segments, steady_items, steady_packets, steady_pdus;
  c8p_rawchain_stage detect, signal, rx;
  int64_t arena_growths;   /* device / pinned allocations made by the contexts during the run (0 once the arenas fit)    */
  int64_t arena_bytes;     /* device bytes the contexts hold after the run                                               */
  int64_t det_giveups;     /* k_detect_scan launches repeated with ticket numbering (a tile's look-back gave up), during the run */
  int64_t det_ticket_scans;
} c8p_rawchain_stats;

typedef struct c8p_rawchain_mark {      /* one segment's way through the stages: seconds since the start of the run */
  double det_begin, det_end, sig_begin, sig_end, rx_submit, rx_done;
} c8p_rawchain_mark;

/* err / errcap: message buffer of the caller, may be NULL.  Returns NULL on failure (no usable device, missing library or symbol, a
 * reserve that does not fit).  An engine with fewer than three slots (C8P_SLOTS = 1 / 2) works, with less overlap: the rx stage then
 * collects a segment before it submits the next, and the detect stage holds back until c8p_signal has read its copy of host items. */
c8p_rawchain* c8p_rawchain_create(const c8p_rawchain_cfg* cfg, char* err, int errcap);
void c8p_rawchain_destroy(c8p_rawchain* h);

/* Runs `n_seg` segments through the pipeline (each one a self-contained stretch of stream: no state is carried from one to the
 * next — cut a long capture with c8p_rawchain_stream when packets may straddle the cuts).  cb may be NULL.  Returns C8P_OK or a
 * C8P_E* code, with a message in err. */
int c8p_rawchain_run(c8p_rawchain* h, const c8p_rawchain_seg* segs, int64_t n_seg, c8p_rawchain_cb cb, void* user,
                     c8p_rawchain_stats* stats, char* err, int errcap);

/* ONE resident capture of n items (antenna 1, if any, ant1_off items behind antenna 0 in the same allocation), worked through in
 * segments of seg_items: the detection state (c8p_det_state at the commit point) and the signal block's scan position are carried
 * from one segment to the next, and a packet that straddles a cut is read from the capture wherever its items lie — the PDUs are
 * those of one c8p_detect / c8p_signal / c8p_rx pass over the whole capture (examples/rx_file.c), delivered segment by segment
 * while later segments are still being searched.  Flags that are still waiting for items when the capture ends yield nothing. */
int c8p_rawchain_stream(c8p_rawchain* h, const c8p_c32* raw, uint64_t n, uint64_t ant1_off, uint64_t seg_items,
                        c8p_rawchain_cb cb, void* user, c8p_rawchain_stats* stats, char* err, int errcap);

/* The marks of the last run / stream / live session, one per segment; returns how many there are (copies at most cap).  A live
 * session keeps a window — its last 4096 to 8192 rounds — and its steady-state figures are taken over that window. */
int64_t c8p_rawchain_timeline(c8p_rawchain* h, c8p_rawchain_mark* dst, int64_t cap);

/* ---- live ingest (c8p_rawchain_cfg.streams = K > 0, one antenna per stream) --------------------------------------------------
 * K independent streams, each what one radio delivers: its own trigger / sync state (c8p_det_state), commit point and signal-block
 * scan position live in the handle (c8p_rawchain_cfg.stream_ant = 2: two antennas per stream, c8p_rawchain_push2).  Items arrive in
 * increments of any size:
 *   c8p_rawchain_push   appends n items to one stream.  mem = C8P_MEM_DEVICE, C8P_MEM_PINNED (host memory the device can read:
 *                       c8p_host_alloc / hipHostMalloc / hipHostRegister — the caller vouches for it) or C8P_MEM_HOST (any host
 *                       memory: every piece's pointer is queried, ~4 us each, and pageable memory takes a slow staged copy).
 *                       mem | C8P_MEM_SC16: the items are int16 I/Q pairs as the radio sends them (4 bytes per item over the link
 *                       instead of 8; converted in the gather kernel with c8p_rawchain_cfg.sc16_scale, include/c8p_rx.h
 *                       C8P_FMT_SC16; device or pinned memory only); mem | C8P_MEM_SC8: int8 pairs, 2 bytes per item.  Nothing moves yet: the items must stay valid until the next
 *                       c8p_rawchain_round returns.  All pushes of a round are moved by ONE kernel that reads the host memory over
 *                       PCIe itself (c8p_copy_gather): 54-56 GB/s whatever the size of the increments (the DMA engines fall to
 *                       18 GB/s at 256 KB copies), and the DMA queues stay free for the stages' own descriptor / result copies
 *   c8p_rawchain_round  closes the round: everything pushed since the last one goes down the pipeline as ONE batch — one
 *                       c8p_detect_multi over all streams' new items, one c8p_signal with the scan rule per stream, one c8p_rx over
 *                       the packets of all streams; returns when the round's items are on the device (blocks while three rounds
 *                       are in flight).  The results arrive through the callback, on the pipeline's rx thread, round by round; a
 *                       packet whose items straddle rounds is delivered by the round that completes it.
 *   c8p_rawchain_round_async  the same without waiting for the copies: the pipeline's first stage waits for them instead, so the next
 *                       round's copies follow this round's without a gap on the link; the pushed items must then stay valid until
 *                       c8p_rawchain_ingested() (or c8p_rawchain_drain) has returned.  NOT a call that never blocks: a round of more
                       than C8P_GATHER_INLINE pushes goes through the engine's one piece table, and c8p_copy_gather waits for the
                       copy stream before it rewrites that table (include/c8p_rx.h) — with rounds back to back the call (and a push
                       that flushes 4096 pieces) blocks for what is left of the PREVIOUS round's transfer
 *   c8p_rawchain_drain  waits until every round has been delivered and ends the session (flags still waiting for items yield
 *                       nothing); the handle can start another one (all streams from the constructor state).
 * One thread calls push / round / drain.  The first push or round starts the session with the callback given to c8p_rawchain_live. */
int c8p_rawchain_live(c8p_rawchain* h, c8p_rawchain_cb cb, void* user);
int c8p_rawchain_push(c8p_rawchain* h, uint32_t stream, const void* items, uint64_t n, int32_t mem);
/* stream_ant = 2: the same n items of both antennas of one stream (they advance together) */
int c8p_rawchain_push2(c8p_rawchain* h, uint32_t stream, const void* ant0, const void* ant1, uint64_t n, int32_t mem);
/* m pushes in one call (one antenna per stream): streams[i], items[i], n[i] */
int c8p_rawchain_pushv(c8p_rawchain* h, int32_t m, const uint32_t* streams, const void* const* items, const uint64_t* n, int32_t mem);
int c8p_rawchain_round(c8p_rawchain* h);
int c8p_rawchain_round_async(c8p_rawchain* h);
int c8p_rawchain_ingested(c8p_rawchain* h);          /* every item pushed before the last round call is on the device */
int c8p_rawchain_drain(c8p_rawchain* h, c8p_rawchain_stats* stats);
const char* c8p_rawchain_error(c8p_rawchain* h);     /* message of the last failing call on this handle */

/* One-shot conveniences (create, run, destroy): what examples and small tools call.  lib / device as in c8p_rawchain_cfg. */
int c8p_rawchain_run_once(const char* lib, int device, const c8p_rawchain_seg* segs, int64_t n_seg, c8p_rawchain_cb cb, void* user,
                          c8p_rawchain_stats* stats, char* err, int errcap);
int c8p_rawchain_stream_once(const char* lib, int device, const c8p_c32* raw, uint64_t n, uint64_t ant1_off, uint64_t seg_items,
                             c8p_rawchain_cb cb, void* user, c8p_rawchain_stats* stats, char* err, int errcap);

#ifdef __cplusplus
}
#endif
