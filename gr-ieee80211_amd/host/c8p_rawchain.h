// c8p_rawchain.h — raw 20 MS/s stream -> PDUs on the engine, as a pipeline (host side, plain C ABI).
//
// The reference's rx.grc wires  source -> presiso / trigger / sync -> signal -> demod -> decode ; with everything resident on
// one MI355X the same chain is three engine calls per stream segment,
//     c8p_detect (presiso + trigger + sync)  ->  c8p_signal (consume = 1)  ->  c8p_rx (demod + decode),
// and this driver keeps three segments in flight: i + 2 in c8p_detect while i + 1 is in c8p_signal and i in c8p_rx, each
// stage on its own engine context (own HIP streams) and host thread.  The callback sees every segment's results in order.
#pragma once
#include <stdint.h>

#include "c8p_rx.h"

#ifdef __cplusplus
extern "C" {
#endif

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
  int64_t  segment;        /* index of the segment                                                                */
  int32_t  n_events;       /* sync events c8p_detect found                                                        */
  int32_t  n_packets;      /* packets the signal block's scan kept (what c8p_rx decoded)                          */
  const c8p_pkt_in*  pkts; /* [n_packets] their boundary descriptors (host)                                       */
  const c8p_pkt_out* outs; /* [n_packets] per-packet results (host): n_pdu, n_crc_fail, pdu_off, pdu_bytes        */
  c8p_ctx* rx;             /* the rx context with this segment selected: c8p_fetch(rx, C8P_BUF_PDU, ...) reads its PDU arena */
} c8p_rawchain_out;

typedef void (*c8p_rawchain_cb)(void* user, const c8p_rawchain_out* out);

typedef struct c8p_rawchain_stats {
  double  seconds;         /* wall time of the run                                                                */
  int64_t segments, items, events, packets, pdus;
} c8p_rawchain_stats;

/* Runs `n_seg` segments through the pipeline (each one a self-contained stretch of stream: no state is carried from one to the
 * next — cut a long capture with c8p_detect's commit points when packets may straddle the cuts).  `lib` = path of libc8p_rx.so
 * (NULL: the environment's C8P_RX_LIB, or "libc8p_rx.so").  cb may be NULL.  Returns C8P_OK or a C8P_E* code, with a message in err. */
int c8p_rawchain_run(const char* lib, int device, const c8p_rawchain_seg* segs, int64_t n_seg, c8p_rawchain_cb cb, void* user,
                     c8p_rawchain_stats* stats, char* err, int errcap);

/* ONE resident capture of n items (antenna 1, if any, ant1_off items behind antenna 0 in the same allocation), worked through in
 * segments of seg_items: the detection state (c8p_det_state at the commit point) and the signal block's scan position are carried
 * from one segment to the next, and a packet that straddles a cut is read from the capture wherever its items lie — the PDUs are
 * those of one c8p_detect / c8p_signal / c8p_rx pass over the whole capture (examples/rx_file.c), delivered segment by segment
 * while later segments are still being searched.  Flags that are still waiting for items when the capture ends yield nothing. */
int c8p_rawchain_stream(const char* lib, int device, const c8p_c32* raw, uint64_t n, uint64_t ant1_off, uint64_t seg_items,
                        c8p_rawchain_cb cb, void* user, c8p_rawchain_stats* stats, char* err, int errcap);

#ifdef __cplusplus
}
#endif
