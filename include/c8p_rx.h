/*
 * c8p_rx.h — C ABI of the MI355X-native 802.11 OFDM RX hot path (libc8p_rx.so).
 *
 * This is the drop-in boundary for the reference's demod / demod2 / decode blocks
 * (all citations are relative to the reference tree):
 *
 *   (entry points added later cite what they replace where they are declared: c8p_detect / c8p_detect_multi, c8p_tx, the caller-arena
 *   copies; include/../gr-ieee80211_amd/host/c8p_rawchain.h is the chain examples/rx.grc:753-767 as a pipeline)
 *   c8p_demod()   replaces  demod_impl::general_work   lib/demod_impl.cc:59-342
 *                 and       demod2_impl::general_work  lib/demod2_impl.cc:58-349
 *   c8p_decode()  replaces  decode_impl::general_work  lib/decode_impl.cc:60-162
 *   c8p_rx()      = c8p_demod() + c8p_decode() in one call; the LLR stream between them stays on chip (LDS) unless
 *                   c8p_cfg.keep_llr asks for it in C8P_BUF_LLR
 *   c8p_signal()  replaces  signal_impl::general_work  lib/signal_impl.cc:62-206   (the producer of the boundary;
 *                 and       signal2_impl::general_work lib/signal2_impl.cc          SURVEY.md §8f-1, optional)
 *
 * A "packet" is what one pass of the reference state machines handles: the items between
 * two tag sets on the stream.  The GNU Radio blocks stream packets one by one; this engine
 * takes a batch of them (descriptors + arenas) and processes all of them in parallel.
 *
 * Conventions
 *   - plain C types only, no torch / GNU Radio / HIP types in signatures;
 *   - every pointer is caller-owned; the engine never frees or keeps caller memory past
 *     c8p_sync(); engine-owned result arenas stay valid until their slot is reused — with the default 3 slots that is the
 *     3rd submit after theirs (c8p_ticket / c8p_select), with a caller-provided stream the next submit;
 *   - errors are negative return codes, never exceptions or abort();
 *   - one ctx per block instance / per thread; a ctx is not internally locked;
 *   - there is no CPU fallback: c8p_create() fails if no HIP device is usable.
 */
#ifndef C8P_RX_H
#define C8P_RX_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define C8P_ABI_VERSION 4   /* 3: c8p_sig_in.stream, c8p_signal_in.n_streams / scan_from_v, C8P_BUF_DET_RAW, c8p_detect_multi, c8p_reserve, c8p_counter, c8p_dev_*
                               4: C8P_FMT_F32 pieces, c8p_copy_gather towards host memory, c8p_copy_fence, c8p_fetch_gather, C8P_ENOTDEV */

/* return codes */
#define C8P_OK          0
#define C8P_EINVAL     (-22)
#define C8P_ENOMEM     (-12)
#define C8P_ENODEV     (-19)
#define C8P_EHIP       (-5)
#define C8P_ERANGE     (-34)
#define C8P_ENOTDEV    (-14)  /* c8p_copy_gather: a piece lies in memory the device cannot reach (pageable host memory) — the one
                                 refusal a caller answers with c8p_copy_async; every other bad argument stays C8P_EINVAL        */

/* where a caller buffer lives */
#define C8P_MEM_HOST    0
#define C8P_MEM_DEVICE  1
#define C8P_MEM_PINNED  2   /* host memory the device can read (c8p_host_alloc / hipHostMalloc / hipHostRegister): accepted where a call says
                               so (c8p_rawchain_push*); the caller vouches for it, the engine does not query the pointer              */
#define C8P_MEM_SC16    0x100 /* a flag on top of the kind, accepted by c8p_rawchain_push*: the items are int16 I/Q pairs (C8P_FMT_SC16) */
#define C8P_MEM_SC8     0x200 /* ... int8 I/Q pairs (C8P_FMT_SC8) */

/* packet formats: lib/cloud80211phy.h:35-40 */
#define C8P_F_L         0
#define C8P_F_HT        1
#define C8P_F_VHT       2
#define C8P_F_VHT_CHAN  20
/* coding rates: lib/cloud80211phy.h:46-49 */
#define C8P_CR_12 0
#define C8P_CR_23 1
#define C8P_CR_34 2
#define C8P_CR_56 3

/* limits: lib/decode_impl.h:35-36, lib/cloud80211phy.h:29-31 */
#define C8P_DECODE_B_MAX   4095
#define C8P_DECODE_T_MAX   32782
#define C8P_MAX_N_CBPSS    416
#define C8P_NDP_TOTAL      1024
#define C8P_NDP_BLOB_BYTES 1027

/* per-packet status after demod (which reference state ended the packet) */
#define C8P_ST_OK            0   /* tags written, LLRs (or the NDP dummy items) follow  */
#define C8P_ST_REJ_VHT       1   /* DEMOD_S_VHT sanity check failed  demod_impl.cc:159-170 */
#define C8P_ST_REJ_HT        2   /* DEMOD_S_HT sanity check failed   demod_impl.cc:186-198 */
#define C8P_ST_REJ_SHORT     3   /* fewer items on the port than the state machine needs  */
#define C8P_ST_REJ_UNSUP     4   /* reference behaviour undefined (stale state / unwritten output), see DESIGN.md */

typedef struct { float re, im; } c8p_c32;   /* gr_complex */

/*
 * One packet at the signal -> demod boundary.  Fields = the tags written by
 * signal_impl.cc:135-143 on the first item of the packet, plus where its items are.
 */
typedef struct c8p_pkt_in {
  uint64_t iq_off[2];   /* item offset of the packet's first item inside the iq arena, per input port */
  uint32_t n_items;     /* items present on each port for this packet (reference: nsamp + 320)         */
  uint32_t chan_off;    /* item offset of chan[64] (tag "chan", FFT order) inside the chan arena       */
  int32_t  lsig_mcs;    /* tag "mcs"   L-SIG rate index 0..7                                           */
  int32_t  lsig_len;    /* tag "len"   L-SIG LENGTH                                                    */
  int32_t  lsig_nsamp;  /* tag "nsamp"                                                                 */
  int32_t  seq;         /* tag "seq"                                                                   */
  float    cfo;         /* tag "cfo"                                                                   */
  float    snr;         /* tag "snr"                                                                   */
  float    rssi;        /* tag "rssi"                                                                  */
  uint8_t  n_ant;       /* 1: demod (1 port), 2: demod2 (2 ports)                                      */
  uint8_t  mupos;       /* demod::make(mupos, mugid): MU-MIMO user position 0/1                        */
  uint8_t  rsv[2];      /* rsv[0], written by c8p_signal only: C8P_SIG_NEED_MORE = the L-SIG is valid (lsig_* set) but  */
                        /* fewer than 224 + nsamp items were available: n_items = 0, the block waits (S_COPY)           */
} c8p_pkt_in;
#define C8P_SIG_NEED_MORE 1
#define C8P_SIG_SKIPPED   2   /* rsv[0], c8p_signal_in.consume only: n_items = 0, the block never looks at this flag            */
#define C8P_SIG_PENDING   3   /* rsv[0], c8p_signal_in.consume only: n_items = 0, behind (or at) a flag that waits for more items */

/*
 * demod -> decode tags (demod_impl.cc:224-253, demod2_impl.cc:230-250) plus the descriptor
 * fields the engine keeps for the data path.
 */
typedef struct c8p_pkt_tags {
  int32_t status;       /* C8P_ST_* ; tags below are valid only for C8P_ST_OK                          */
  int32_t format;       /* tag "format"                                                                */
  int32_t mcs;          /* tag "mcs"                                                                   */
  int32_t len;          /* tag "len"                                                                   */
  int32_t cr;           /* tag "cr"                                                                    */
  int32_t ampdu;        /* tag "ampdu"                                                                 */
  int32_t trellis;      /* tag "trellis"                                                               */
  int32_t total;        /* tag "total"  = nSym*nCBPS, or 1024 for an NDP                               */
  int32_t nsym, nss, nltf, ncbps, ncbpss, ndbps, nbpscs, nsymsamp;
  int32_t sumu;         /* VHT MU flag                                                                 */
  int32_t is_ndp;       /* nSym == 0: tag "mu2x1chan" present                                          */
  int32_t data_off;     /* item offset of the first data symbol inside the packet                      */
  int32_t seq;
  float   cfo, snr, rssi;   /* tags "cfo","snr","rssi" (pass-through)                                  */
  float   sssnr0, sssnr1;   /* tags "sssnr0"/"sssnr1" (VHT only)                                       */
  uint32_t rsv;
  uint64_t llr_off;     /* float offset of the packet's LLR items inside the LLR arena                 */
} c8p_pkt_tags;

/* decode result for one packet */
typedef struct c8p_pkt_out {
  int32_t  n_pdu;       /* PDUs published on port "out" for this packet (FCS-correct subframes / NDP report) */
  int32_t  n_crc_fail;  /* (sub)frames whose CRC-32 residue was wrong                                   */
  uint32_t pdu_bytes;   /* bytes used in this packet's PDU area: n_pdu records of {u16 len LE, len bytes} */
  uint32_t rsv;         /* diagnostic: passes of the segment-parallel traceback over this packet (1 = every segment started from the
                           right lane, i.e. every hint of the forward pass was right); 0 on the other decoder paths   */
  uint64_t pdu_off;     /* byte offset of the packet's PDU area inside the PDU arena                     */
  uint64_t bits_off;    /* byte offset of the packet's decoded bits (1 bit per byte) inside the bits arena */
} c8p_pkt_out;

/*
 * One sync event at the sync -> signal boundary: the item where the sync block raised its flag (L-LTF start + 24)
 * with the tags of sync_impl.cc:124-141 ("rad" = CFO in rad/sample, "snr", "rssi").
 */
typedef struct c8p_sig_in {
  uint64_t raw_off[2];  /* item offset of the sync item inside the raw arena, per antenna                     */
  uint32_t n_avail;     /* raw items available from the sync item on (a packet needs 224 + nsamp)             */
  float    cfo_rad;     /* tag "rad"                                                                          */
  float    snr, rssi;   /* tags "snr", "rssi" (pass-through)                                                  */
  uint8_t  n_ant;       /* 1: signal, 2: signal2                                                              */
  uint8_t  rsv[3];
  uint32_t stream;      /* c8p_signal_in.n_streams > 1: which of the independent streams raised this flag     */
} c8p_sig_in;

/*
 * One flagged item of the sync block's output stream with its tags (sync_impl.cc:111-141): what c8p_detect()
 * produces and what, per antenna, becomes a c8p_sig_in for c8p_signal().
 */
typedef struct c8p_sync_evt {
  uint64_t index;       /* item index of the sync flag in the raw stream (L-LTF start + ~16)                  */
  float    rad;         /* tag "rad": CFO in rad / sample (STF coarse + LTF fine)                             */
  float    snr;         /* tag "snr"                                                                          */
  float    rssi;        /* tag "rssi"                                                                         */
  uint32_t trig;        /* item index of the trigger that started this search (low 32 bits)                   */
} c8p_sync_evt;

/* Geometry of one transmitted packet (1 spatial stream, 20 MHz, long GI): the values the reference's modulation
 * descriptor holds after formatToModSu / procPktLenAggre (cloud80211phy.cc:2487-2524, tools/phy80211header.py:427-506). */
typedef struct c8p_tx_plan {
  int32_t nb, cr, ndbps, ncbps, nsd;  /* bits per subcarrier, C8P_CR_*, data / coded bits per symbol, data subcarriers */
  int32_t nsym;                       /* data symbols                                                               */
  int32_t psdu_len;                   /* PSDU bytes after VHT padding (= len for legacy / HT)                       */
  int32_t npad_eof, npad_octet;       /* VHT: EOF delimiters and zero octets appended to the A-MPDU                  */
  int32_t legacy_len;                 /* L-SIG LENGTH                                                               */
  int32_t n_pre;                      /* 80-sample symbols between L-SIG and the first data symbol                  */
  int32_t n_samp;                     /* samples of the packet                                                      */
} c8p_tx_plan;

/* One packet to transmit (1 spatial stream, 20 MHz, long GI): what pktgen -> encode -> modulation take from the
 * "packet" PDU (lib/pktgen_impl.cc, lib/encode_impl.cc:110-160), or tools/phy80211.py genFromMpdu / genFromAmpdu. */
typedef struct c8p_tx_pkt {
  uint64_t psdu_off;    /* byte offset of the PSDU inside the psdu arena: the MPDU (legacy, HT) or the A-MPDU (VHT)  */
  uint32_t len;         /* its length in bytes                                                                       */
  uint8_t  format;      /* C8P_F_L / C8P_F_HT / C8P_F_VHT, or C8P_F_VHT_MU: 2-user MU-MIMO — this struct describes user 0
                           and the next unused entry of c8p_tx_in.mu describes user 1 and the steering matrix            */
  uint8_t  mcs;         /* legacy 0-7, HT 0-7, VHT 0-8                                                               */
  uint8_t  scr_seed;    /* initial scrambler state, 1..127 (the reference toolbox uses 93)                           */
  uint8_t  flags;       /* C8P_TX_NO_WINDOW: do not halve the two samples at every field junction (the toolbox's
                           procConcat2Symbol does; the C++ chain only touches the legacy preamble, pad_impl.cc:68-71)   */
  float    mult;        /* amplitude multiplier (genFinalSig; tools/pktGenExample.py uses 12.0, x sqrt(2) for 2 streams) */
  uint32_t gap;         /* zero items inserted in the output stream before this packet                               */
  uint8_t  nss;         /* transmit chains / spatial streams: 0 or 1 = one, 2 = two (HT: mcs stays the per-stream rate 0-7,
                           the HT-SIG field carries mcs + 8; VHT with len = 0 and nss = 2: the NDP sounding packet)     */
  uint8_t  rsv[3];
} c8p_tx_pkt;

#define C8P_TX_NO_WINDOW 1
#define C8P_F_VHT_MU 3       /* c8p_tx only (C8P_F_VHT_MU of cloud80211phy.h:38): VHT MU-MIMO, 2 users x 1 stream, 2 chains */

/* Second user and steering of a C8P_F_VHT_MU packet: what encode2 / modulation2 take from the "mcs1", "len1", "gid" tags
 * and the C8P_F_VHT_BFQ message (lib/encode2_impl.cc:92-100, lib/modulation2_impl.cc:117-125), or genAmpduMu's
 * ampdu1 / mod1 / groupId / bfQ arguments (tools/phy80211.py:180). */
typedef struct c8p_tx_mu {
  uint64_t psdu_off;    /* user 1's A-MPDU inside the psdu arena                                                     */
  uint32_t len;         /* its length in bytes                                                                       */
  uint8_t  mcs;         /* user 1's VHT MCS 0-8 (user 0's is c8p_tx_pkt.mcs)                                         */
  uint8_t  group_id;    /* VHT-SIG-A group id, 1..62                                                                 */
  uint8_t  rsv[2];
  uint64_t q_off;       /* item offset of the packet's steering matrix inside c8p_tx_in.bfq: 256 complex64 =
                           [64 FFT bins, bin 0 = DC][q00 q01 q10 q11]; chain c sends q_c0 * user 0 + q_c1 * user 1
                           (the layout of d_vhtMuBfQ, cloud80211phy.cc:2932-2942)                                     */
} c8p_tx_mu;

typedef struct c8p_tx_desc {  /* per packet, in C8P_BUF_TX_DESC */
  uint64_t iq_off;      /* item offset of the packet's first sample inside C8P_BUF_TX_IQ (after its gap)             */
  uint64_t coded_off;   /* byte offset of its punctured coded bits (1 bit per byte) inside C8P_BUF_TX_CODED          */
  uint32_t n_samp;      /* samples                                                                                   */
  uint32_t n_coded;     /* coded bits = nsym * ncbps                                                                 */
  uint64_t coded_off1;  /* C8P_F_VHT_MU: user 1's coded bits                                                         */
  uint32_t n_coded1;
  uint32_t rsv;
} c8p_tx_desc;

typedef struct c8p_tx_in {
  int32_t n_pkt;
  int32_t mem;              /* C8P_MEM_* of psdu                                                         */
  const c8p_tx_pkt* pkts;   /* [n_pkt], host memory                                                      */
  const uint8_t* psdu;      /* PSDU byte arena                                                           */
  uint64_t n_psdu;
  uint32_t tail_gap;        /* zero items appended after the last packet                                 */
  uint32_t n_mu;            /* entries of mu (= number of C8P_F_VHT_MU packets in pkts)                  */
  const c8p_tx_mu* mu;      /* host memory; the j-th C8P_F_VHT_MU packet of pkts takes mu[j]             */
  const c8p_c32* bfq;       /* steering matrices, same memory kind as psdu                               */
  uint64_t n_bfq;           /* items                                                                     */
} c8p_tx_in;

typedef struct c8p_cfg {
  int32_t abi_version;      /* C8P_ABI_VERSION                                                         */
  int32_t device;           /* HIP device ordinal                                                      */
  int32_t profile;          /* 1: bracket every kernel launch with HIP events (c8p_kernel_time);
                               N > 1: only on every N-th submit (staggered per kernel) — sampling             */
  int32_t keep_bits;        /* 1: also store descrambled bits (v_unCodedBits) and v_scramBits          */
  void*   stream;           /* hipStream_t to launch on; NULL = engine-owned stream                    */
  int32_t keep_llr;         /* c8p_rx only.  0: the LLR stream stays on chip (the wave that demodulates a packet decodes it out of
                               LDS) and C8P_BUF_LLR is EMPTY after c8p_rx; 1: the LLR stream goes through C8P_BUF_LLR, as c8p_demod
                               always leaves it (other kernels; costs 0.47 GB of HBM traffic per 4096 cfg-2 packets); 2: the
                               on-chip kernels of 0, which additionally copy every symbol's LLRs out of LDS into C8P_BUF_LLR
                               before they are consumed — how the parity tests see the LLRs of the default path           */
} c8p_cfg;

typedef struct c8p_batch_in {
  int32_t n_pkt;
  int32_t mem;              /* C8P_MEM_* of the iq and chan arenas (both the same)                     */
  const c8p_pkt_in* pkts;   /* [n_pkt], ALWAYS host memory (the engine plans the arena layout from it on the host); to chain
                               device-resident C8P_BUF_SIG_PKT descriptors, c8p_fetch them first (48 bytes per packet)  */
  const c8p_c32*    iq;     /* iq arena                                                                */
  uint64_t          n_iq;   /* items in the iq arena                                                   */
  const c8p_c32*    chan;   /* chan arena                                                              */
  uint64_t          n_chan; /* items in the chan arena                                                 */
} c8p_batch_in;

typedef struct c8p_signal_in {
  int32_t n_evt;
  int32_t mem;              /* C8P_MEM_* of raw                                                        */
  const c8p_sig_in* evts;   /* [n_evt], host memory                                                    */
  const c8p_c32*    raw;    /* raw (not yet CFO-corrected) sample arena                                */
  uint64_t          n_raw;
  int32_t           consume; /* 0: every flag on its own (the caller applies the block's scan rule); the call only enqueues.
                                1: evts are the flags of ONE stream in stream order (raw_off[0] ascending) and the engine applies
                                signal_impl's S_TRIGGER scan (signal_impl.cc:76-111,154-160) to them: a flag inside the packet an
                                earlier flag copied, or inside the 80 items a failed L-SIG check skips, is never looked at
                                (C8P_SIG_SKIPPED); the first flag that has to wait for more items (fewer than 224, or
                                C8P_SIG_NEED_MORE) stops the scan, it and the flags behind it are C8P_SIG_PENDING.  Only the
                                packets the block copies are copied, back to back in C8P_BUF_SIG_IQ.  The call blocks until the
                                descriptors are known (the copies are only enqueued)                                          */
  int32_t           n_streams; /* consume = 1.  0 or 1: evts belong to one stream.  K > 1: to K independent streams (one signal block each:
                                 K instances of signal_impl's S_TRIGGER / S_DEMOD / S_COPY machine, lib/signal_impl.cc:68-204;
                                 c8p_sig_in.stream = 0 .. K-1), grouped by stream in ascending order and in stream order inside a group; the
                                 scan rule runs per stream, from scan_from_v[stream], and scan_next receives K positions            */
  c8p_pkt_in*       pkt_host; /* consume = 1: if not NULL, receives the n_evt descriptors (what C8P_BUF_SIG_PKT holds)         */
  uint64_t          scan_from; /* consume = 1: where the S_TRIGGER scan stands when it meets the first flag (raw_off[0] units): flags
                                 before it are C8P_SIG_SKIPPED — the position a previous call over the same stream left in *scan_next */
  uint64_t*         scan_next; /* consume = 1: if not NULL, receives the scan position after the last flag that was looked at
                                 (n_streams > 1: one per stream; a stream without flags keeps its scan_from_v value)             */
  const uint64_t*   scan_from_v; /* n_streams > 1: [n_streams] scan positions (NULL: all 0); scan_from is then ignored           */
} c8p_signal_in;

/* State of trigger_impl + sync_impl at a commit point of the stream: everything before it is final, nothing pending
 * refers to items before it.  Item positions are relative to the commit point. */
typedef struct c8p_det_state {
  int64_t run_start;    /* first item of the plateau (ac > 0.3) that is open at the commit point (< 0)          */
  int64_t prev_start;   /* item where the last 80-item countdown was armed (< 0)    trigger_impl.cc:86-91        */
  int64_t sync_pos;     /* first item sync's IDLE state will look at (> 0 while the 111 consumed items extend past the commit point) */
  int32_t in_run;       /* 1: a plateau is open (d_nPlateau > 0)                                                */
  int32_t run_fired;    /* 1: that plateau already armed its countdown (d_fPlateauEnd)                          */
  int32_t have_prev;    /* 1: prev_start is valid                                                               */
  float   run_max;      /* d_conjAc: maximum of ac over the open plateau                                        */
  c8p_c32 conj_avg;     /* d_conjMultiAvg: the last latched presiso output 1            sync_impl.cc:84-87       */
} c8p_det_state;

/* raw stream -> sync events: presiso.grc + trigger_impl + sync_impl (antenna 0 only, as in rx.grc / rx2.grc).
 * One-shot: n_hist = 0, state = NULL: the segment starts from the blocks' constructor state.
 * Streaming: raw = [n_hist items of history | new items]; the new items start at the commit point the previous call
 * returned, `state` is the state it returned, and n_hist >= C8P_DET_HIST items before it are supplied again (fewer only
 * at the very start of the stream).  Event indices are relative to raw[0]. */
#define C8P_DET_HIST 80   /* presiso looks back 16 + 63 items */
typedef struct c8p_detect_in {
  int32_t mem;              /* C8P_MEM_* of raw                                                        */
  int32_t keep;             /* 1: also keep presiso's two output streams (C8P_BUF_DET_AC, C8P_BUF_DET_CONJ), for tests:
                               the product path keeps them on the chip                                  */
  const c8p_c32* raw;       /* [n] raw samples of antenna 0                                            */
  uint64_t       n;
  uint64_t       n_hist;    /* leading items of raw that lie before the commit point                   */
  const c8p_det_state* state;   /* state at the commit point, NULL = constructor state                 */
  c8p_det_state* state_out;     /* may be NULL; receives the state at the new commit point             */
  uint64_t*      commit;        /* may be NULL; receives the new commit point (index into raw): triggers whose countdown or
                                   240-item search window runs past the end of raw are re-derived by the next call        */
  c8p_sync_evt*  evt_host;      /* may be NULL; receives the first evt_cap sync events (what C8P_BUF_DET_EVT holds; their number =
                                   its size / sizeof(c8p_sync_evt)): the call has them on the host anyway, this saves the caller a
                                   c8p_fetch — one device-to-host copy and one synchronisation per call                    */
  uint64_t       evt_cap;
} c8p_detect_in;

/* c8p_detect over K independent streams in ONE pass (one k_detect_scan launch: a tile belongs to a stream, the plateau look-back
 * stops at a stream's first tile): what K chains of presiso.grc + trigger + sync do to K radios — K instances of
 * examples/presiso.grc:37-229 -> trigger_impl::general_work (lib/trigger_impl.cc:59-117) -> sync_impl::general_work
 * (lib/sync_impl.cc:62-192), wired per radio as in examples/rx.grc:753-767; nothing crosses from one stream to another.  Every stream is a streaming
 * c8p_detect_in of its own (history, state at its commit point) whose items lie in one arena; the fields marked "out" are written
 * back.  A stream with no new items (n == n_hist) keeps its state.  C8P_BUF_DET_EVT holds the events of stream 0, then stream 1, ...
 * (each stream's in stream order, index relative to the stream's `off`); C8P_BUF_DET_TRIG holds each stream's trigger bytes at
 * trig_off. */
typedef struct c8p_det_stream {
  uint64_t off;             /* item offset of the stream's first supplied item (history included) inside raw            */
  uint64_t n;               /* items supplied: n_hist of history, then the new ones                                     */
  uint64_t n_hist;          /* as c8p_detect_in.n_hist                                                                   */
  uint64_t commit;          /* out: the new commit point (index relative to off)                                         */
  uint64_t trig_off;        /* out: byte offset of this stream's n - n_hist trigger bytes inside C8P_BUF_DET_TRIG        */
  uint32_t n_evt;           /* out: sync events of this stream                                                           */
  int32_t  has_state;       /* 0: the blocks' constructor state (`state` is ignored on the way in)                       */
  c8p_det_state state;      /* in: state at the commit point; out: state at the new commit point                         */
} c8p_det_stream;
typedef struct c8p_detect_multi_in {
  int32_t mem;              /* C8P_MEM_* of raw (host: every stream's items are uploaded to the same offsets of the engine's copy) */
  int32_t n_streams;
  const c8p_c32* raw;       /* the arena that holds every stream's items                                                */
  uint64_t       n_raw;     /* items in it                                                                              */
  c8p_det_stream* streams;  /* [n_streams], host memory, in / out                                                       */
  c8p_sync_evt*  evt_host;  /* as c8p_detect_in.evt_host: the events of all streams, stream after stream                    */
  uint64_t       evt_cap;
} c8p_detect_multi_in;

/* Arena sizes known in advance: c8p_reserve grows every slot's arenas to what submits of this shape need, so that no allocation
 * (hipMalloc / hipHostMalloc behind a stream synchronisation) happens inside a later submit.  Zero fields reserve nothing. */
typedef struct c8p_reserve_in {
  int32_t  rx_packets;      /* c8p_rx / c8p_demod / c8p_decode: packets per submit ...                                    */
  int32_t  rx_ant;          /* ... their ports (1 or 2) ...                                                               */
  uint32_t rx_items;        /* ... and items per packet and port (c8p_pkt_in.n_items of the longest one)                  */
  int32_t  rx_host;         /* 1: their iq / chan arenas come from host memory (the engine's staging copies are sized too) */
  int32_t  sig_events;      /* c8p_signal: flags per call                                                                 */
  int32_t  sig_host;        /* 1: raw comes from host memory ...                                                          */
  uint64_t sig_raw;         /* ... this many items of it                                                                  */
  uint64_t sig_items;       /* items of C8P_BUF_SIG_IQ: consume = 1: the kept packets (n_items per port, summed); consume = 0:
                               sum over the flags of (n_avail - 224 + 320) per port                                       */
  uint64_t det_items;       /* c8p_detect / c8p_detect_multi: items per call (history included, all streams)              */
  int32_t  det_host;        /* 1: they come from host memory                                                              */
  int32_t  det_streams;     /* c8p_detect_multi: streams per call (0 or 1: c8p_detect)                                    */
  int32_t  det_events;      /* accepted triggers per call (sync searches)                                                 */
  int32_t  det_keep;        /* 1: c8p_detect_in.keep                                                                      */
} c8p_reserve_in;

typedef struct c8p_decode_in {
  int32_t n_pkt;
  int32_t mem;              /* C8P_MEM_* of llr and mu2x1chan                                          */
  const c8p_pkt_tags* tags; /* [n_pkt], ALWAYS host memory (sizes are planned on the host); llr_off locates each packet's
                               `total` items                                                            */
  const float*   llr;       /* LLR arena (reference demod output stream layout)                        */
  uint64_t       n_llr;
  const c8p_c32* mu2x1chan; /* [n_pkt][128] tag "mu2x1chan" (only read for NDP packets); may be NULL   */
} c8p_decode_in;

/* result arenas (device memory, engine-owned) */
#define C8P_BUF_TAGS   0    /* c8p_pkt_tags[n_pkt]                    */
#define C8P_BUF_LLR    1    /* float, per packet at tags[p].llr_off (c8p_demod; c8p_rx only with keep_llr) */
#define C8P_BUF_OUT    2    /* c8p_pkt_out[n_pkt]                     */
#define C8P_BUF_PDU    3    /* bytes, per packet at out[p].pdu_off    */
#define C8P_BUF_BITS   4    /* bytes (keep_bits), descrambled bits    */
#define C8P_BUF_SBITS  5    /* bytes (keep_bits), bits before descrambling */
#define C8P_BUF_MUCHAN 6    /* c8p_c32[n_pkt][128] NDP time-domain LTF capture */
#define C8P_BUF_SIG_PKT  7  /* c8p_pkt_in[n_evt]: the boundary descriptors signal produced (n_items = 0: L-SIG check failed) */
#define C8P_BUF_SIG_IQ   8  /* CFO-corrected items, per packet at pkt.iq_off[a] (nsamp items + 320 zero pad)   */
#define C8P_BUF_SIG_CHAN 9  /* c8p_c32[n_evt][64] legacy channel estimates (tag "chan")                        */
#define C8P_BUF_DET_EVT  10 /* c8p_sync_evt[]: sync events of the last c8p_detect, in stream order                    */
#define C8P_BUF_DET_AC   11 /* float[n - n_hist] (keep = 1): presiso output 0 (autocorrelation magnitude / energy) of the new items  */
#define C8P_BUF_DET_TRIG 12 /* uint8[n - n_hist]: trigger output stream (bit 0x01 trigger, 0x02 latch conj; 0x04 = ac > 0.3, private) */
#define C8P_BUF_DET_CONJ 13 /* c8p_c32[n - n_hist] (keep = 1): presiso output 1                                       */
#define C8P_BUF_TX_IQ    14 /* c8p_c32[]: the transmitted stream(s), packets at tx_desc[p].iq_off, gaps are zeros; when any packet
                               has two chains the buffer holds chain 0 then chain 1 (each half = the whole stream length)       */
#define C8P_BUF_TX_DESC  15 /* c8p_tx_desc[n_pkt]                                                                              */
#define C8P_BUF_TX_CODED 16 /* uint8[]: punctured coded bits (encode's output), 1 bit per byte                                 */
#define C8P_BUF_DECSTAT  17 /* uint32[3 n_pkt], decoder diagnostics (tests, bench): [p] = 1 if the packet's LLRs failed the range check of the fast
                               Viterbi kernel (it was decoded by the clamp-exact kernel); [n_pkt + p] = number of 30-step trellis blocks the fast
                               kernel ran with the float compare because their path metrics were not provably positive; [2 n_pkt + p] = float
                               bits of the bound it was given: >= sum(max(0, -LLR)) over any 60 consecutive LLRs of the packet ([p] = 0 only) */
#define C8P_BUF_DET_RAW  18 /* c8p_c32[n]: c8p_detect's raw items on the device (history included) — the engine's copy when they came from host
                               memory, the caller's own pointer otherwise: what a following c8p_signal can read without a second upload */
#define C8P_BUF_COUNT    19

/* kernels, for c8p_kernel_time() */
#define C8P_K_LAYOUT   0    /* H2D upload of descriptors + arena layout (a copy, not a kernel) */
#define C8P_K_CLASSIFY 1
#define C8P_K_DEMOD    2
#define C8P_K_VITERBI  3
#define C8P_K_FINISH   4
#define C8P_K_SIGNAL   5
#define C8P_K_PRESISO  6    /* windowed autocorrelation + plateau / running-maximum scan, one pass (HBM-bound) */
#define C8P_K_TRIGGER  7    /* candidate lists of the tiles -> one list in stream order                       */
#define C8P_K_SYNCEVT  8    /* per-trigger LTF search + CFO              */
#define C8P_K_TXBITS   9    /* scramble + BCC + puncture + SIG fields   */
#define C8P_K_TXSYM    10   /* interleave + map + pilots + IFFT + GI    */
#define C8P_K_COUNT    11

typedef struct c8p_ctx c8p_ctx;

c8p_ctx* c8p_create(const c8p_cfg* cfg);
void     c8p_destroy(c8p_ctx* ctx);
const char* c8p_last_error(const c8p_ctx* ctx);

/* Enqueue on the ctx stream; return before the GPU work is done.  C8P_MEM_DEVICE inputs are read on the engine's streams:
 * unless the ctx was created on the caller's stream (c8p_cfg.stream), the caller must have completed the work that fills
 * them (stream / event synchronisation) before the call. */
int c8p_demod (c8p_ctx* ctx, const c8p_batch_in* in);   /* fills TAGS, LLR, MUCHAN              */
int c8p_decode(c8p_ctx* ctx, const c8p_decode_in* in);  /* fills OUT, PDU (BITS, SBITS)         */
int c8p_rx    (c8p_ctx* ctx, const c8p_batch_in* in);   /* fills everything but C8P_BUF_LLR (empty unless c8p_cfg.keep_llr) */
int c8p_signal(c8p_ctx* ctx, const c8p_signal_in* in);  /* fills SIG_PKT, SIG_IQ, SIG_CHAN (feed them to c8p_rx) */
int c8p_detect(c8p_ctx* ctx, const c8p_detect_in* in); /* fills DET_EVT, DET_TRIG (DET_AC, DET_CONJ); blocks until the events are known */
int c8p_detect_multi(c8p_ctx* ctx, const c8p_detect_multi_in* in); /* the same for K streams at once; fills DET_EVT, DET_TRIG, DET_RAW */
int c8p_reserve(c8p_ctx* ctx, const c8p_reserve_in* in);
int c8p_tx    (c8p_ctx* ctx, const c8p_tx_in* in);      /* fills TX_IQ, TX_DESC, TX_CODED (SURVEY §8f-4, 1 spatial stream) */
int c8p_tx_plan_of(int format, int mcs, int nss, int len, c8p_tx_plan* plan);   /* 1 if the combination is supported */
int c8p_tx_plan_mu(int mcs0, int len0, int mcs1, int len1, c8p_tx_plan* plan2); /* C8P_F_VHT_MU: plan2[0], plan2[1] = the users */
int c8p_sync  (c8p_ctx* ctx);

/* Submits rotate through C8P_SLOTS (default 3) sets of arenas, so the results of the last few submits stay resident.
 * c8p_ticket: id of the latest submit (> 0).  c8p_select: make c8p_buffer / c8p_fetch refer to that earlier submit
 * (C8P_ERANGE once its slot has been reused); the next submit selects itself.  Lets a caller upload batch i+1 while it
 * downloads batch i.  One slot only when the ctx runs on a caller-provided stream. */
int c8p_ticket(c8p_ctx* ctx);
int c8p_select(c8p_ctx* ctx, int ticket);

/* Device pointer / size in bytes of a result arena of the selected (by default: the last) submit. */
const void* c8p_buffer(c8p_ctx* ctx, int which, uint64_t* bytes);
/* Copy [offset, offset+bytes) of a result arena to host memory (synchronises the stream). */
int c8p_fetch(c8p_ctx* ctx, int which, void* dst, uint64_t offset, uint64_t bytes);

/* c8p_fetch without the synchronisation: enqueues the copy behind the selected submit; complete after c8p_wait(ticket) of that
 * submit or c8p_sync().  dst should come from c8p_host_alloc (pinned), otherwise the call may block like c8p_fetch; dst may
 * also be device memory (device-to-device: results gathered without touching the host). */
int c8p_fetch_async(c8p_ctx* ctx, int which, void* dst, uint64_t offset, uint64_t bytes);
/* Several c8p_fetch_async in ONE kernel launch behind the selected submit's kernels: piece i = bytes of result arena `which` from
 * `offset` to dst (pinned host memory — c8p_host_alloc — or device memory; offset, bytes and dst multiples of 4).  The shader cores
 * write the host memory over PCIe themselves: no copy-engine call per arena (on some boxes the first hipMemcpyAsync behind a submit's
 * kernels holds the calling thread for ~0.1 ms: what the block mirrors' per-batch downloads of tags, results and PDUs cost them).
 * At most C8P_GATHER_INLINE pieces.  Complete after c8p_wait(ticket) of that submit or c8p_sync(). */
#define C8P_BUF_FLAG (-1)   /* c8p_fetch_piece.which: not an arena — when every other piece of the call has landed, the 32-bit value `offset` is
                               written to dst (4 bytes of pinned host memory): a caller that polls that word needs no c8p_wait, and
                               may do so on ANOTHER thread than the one that submits on this ctx (the word is the whole interface)  */
typedef struct c8p_fetch_piece {
  int32_t  which;           /* C8P_BUF_*, or C8P_BUF_FLAG                 */
  uint32_t rsv;
  uint64_t offset, bytes;
  void*    dst;
} c8p_fetch_piece;
int c8p_fetch_gather(c8p_ctx* ctx, const c8p_fetch_piece* pieces, int32_t n);
/* Wait for one submit (uploads, kernels, the c8p_fetch_async copies behind it); C8P_ERANGE once its slot has been reused. */
int c8p_wait(c8p_ctx* ctx, int ticket);
/* 1: the submit behind `ticket` has finished (c8p_wait would return at once); 0: not yet; < 0: a C8P_E* code.  Never blocks. */
int c8p_done(c8p_ctx* ctx, int ticket);
/* Pinned host memory for staging arenas (what the block fronts use: lib/demod_impl.cc's scheduler buffers are pageable and
 * only valid during general_work, so the fronts copy packets into these).  NULL on failure. */
void* c8p_host_alloc(uint64_t bytes);
void  c8p_host_free(void* p);

/* Device memory for arenas a caller keeps across submits (host/c8p_rawchain.cc's stream arenas), and copies into / inside it that
 * are ordered on a copy stream of their own: c8p_copy_async enqueues one copy (either pointer may be host — pinned for an
 * asynchronous copy — or device memory), c8p_copy_spans moves n spans of items inside device memory with one launch,
 * c8p_copy_wait returns when everything enqueued so far has been done.  The engine's submits do NOT wait for these copies. */
typedef struct c8p_span { uint64_t dst_off, src_off, n; } c8p_span;      /* item (c8p_c32) offsets from dst / src, item count */
void* c8p_dev_alloc(c8p_ctx* ctx, uint64_t bytes);
void  c8p_dev_free(c8p_ctx* ctx, void* p);      /* waits for the ctx's copy streams first; ctx may be NULL when it has been destroyed */
int   c8p_copy_async(c8p_ctx* ctx, void* dst, const void* src, uint64_t bytes);
int   c8p_copy_spans(c8p_ctx* ctx, c8p_c32* dst, const c8p_c32* src, const c8p_span* spans, int32_t n);   /* spans: host memory */
int   c8p_copy_wait(c8p_ctx* ctx);
/* A mark on the copy stream: c8p_copy_mark returns a ticket (> 0) for "every copy enqueued so far"; c8p_copy_wait_mark(ticket) returns
 * when those copies are done — from any thread, also while another thread keeps enqueuing copies on the same ctx (the one exception
 * to "a ctx is not internally locked": marks and c8p_copy_async / c8p_copy_spans may be called concurrently).  The last 64 marks can
 * be waited for; an older ticket waits for a newer mark (never too little).  c8p_copy_mark_done(ticket) asks without waiting: 1 = those
 * copies are done, 0 = not yet, < 0 an error — how a caller that keeps several gathers in flight (the demod mirror's LLR output, written
 * ahead) collects the finished ones on its way past. */
/* n pieces gathered into dst by ONE kernel launch on the copy stream: piece i = n items from src (device memory, or host memory the
 * device can read: c8p_host_alloc / hipHostMalloc / hipHostRegister-ed) to dst + dst_off.  The shader cores read the host memory
 * over PCIe themselves, so no DMA engine is involved and the rate does not depend on the size of the pieces (54-57 GB/s alone;
 * measured in the live pipeline, 268 MB a round: 256 pieces of 1 MB 6.0 ms through this call, 17 ms through c8p_copy_async; 64 pieces of
 * 4 MB 6.0 against 6.3 ms).  A piece in pageable host memory is refused (C8P_ENOTDEV): use c8p_copy_async for it.
 * Either side may be the host: with src in device memory and dst (+ dst_off) in pinned host memory the same launch SCATTERS results
 * into a caller's buffers (the block mirrors write a batch's LLR items straight into the scheduler's output buffer with it).
 * Blocking behaviour.  Up to C8P_GATHER_INLINE pieces travel in the launch's own arguments and the call only enqueues.  A longer
 * list goes through ONE table in pinned memory: the call then first waits until every copy enqueued on the copy stream before it has
 * finished (the table may still be read) — a round's whole transfer time when rounds follow each other back to back; measured against
 * a ring of tables in round 4, the wait was the faster variant (the gaps it leaves are when the pipeline's small transfers get through
 * the link).  c8p_rawchain_push / _round_async inherit this: they can block for as long as the previous round's gather takes. */
#define C8P_GATHER_INLINE 96
typedef struct c8p_piece {
  const void* src;          /* n items: complex64 (8-byte aligned); fmt = C8P_FMT_SC16 / _SC8: int16 / int8 I/Q pairs (4 / 2-byte aligned) */
  uint64_t dst_off;         /* items                                                                                                   */
  uint64_t n;
  uint32_t trusted;         /* 1: the caller vouches that the device can read src (device memory, pinned host memory): no pointer query
                               — one hipPointerGetAttributes per piece costs ~4 us, a millisecond per round of 256 pushes               */
  uint32_t fmt;             /* C8P_FMT_* (one c8p_copy_gather call may mix them, except C8P_FMT_F32 with the others: dst_off units differ)     */
  float    scale;           /* C8P_FMT_SC16 / _SC8: item = ((float)i * scale, (float)q * scale), one float32 multiply each             */
  uint32_t rsv;             /* C8P_PIECE_TO_HOST, C8P_PIECE_UNORDERED or 0 (every piece of a call the same)                            */
} c8p_piece;
/* c8p_piece.rsv: the pieces of this call travel FROM the device TO host memory.  Such a call (up to C8P_GATHER_INLINE pieces) runs on a
 * copy stream of its own, so that a block that feeds the device and drains it at the same time (the demod mirror: samples down, LLR
 * items up) uses both directions of the link at once instead of queueing one transfer behind the other — each launch takes ~18 us
 * from enqueue to done whatever its size.  c8p_copy_mark / _wait_mark / _wait cover both streams. */
#define C8P_PIECE_TO_HOST 1u
/* c8p_piece.rsv, host -> device pieces (every piece of a call the same, up to C8P_GATHER_INLINE pieces): this call needs no order against the
 * ctx's other gathers — only marks, fences, waits and c8p_copy_async (which runs behind EVERY gather enqueued before it, on either
 * stream: the block fronts move a partly received packet out of an arena such gathers fill), which cover it.  A mark covers everything
 * enqueued before it on the copy stream, its twin and the device -> host stream, whichever earlier mark recorded it: waiting for the
 * newest of several marks is enough.  Such calls alternate between the copy stream and a twin of it: the
 * next gather is launched and ramps up while this one drains (a stream starts a kernel when the one before it has finished:
 * 1.4 MB gathers back to back 29.9 -> 27.8 us apiece, 46.8 -> 50.3 GB/s, tools/probes/link_duplex.py).  Without the flag two gathers of one
 * ctx run in the order they were enqueued (the second may overwrite what the first wrote). */
#define C8P_PIECE_UNORDERED 2u
/* C8P_FMT_SC16: what the radio puts on the wire.  The reference's flowgraph takes its stream from a UHD source with the default wire
 * format sc16 and CPU format fc32 (/root/reference/examples/rx.grc:689 `otw: ''`, :744 `type: fc32`): the host converts every sample
 * to float before the first block sees it.  Here the 4-byte samples cross PCIe and the gather kernel converts them on their way into
 * the arena — half the link bytes per item of a link-bound ingest; the arithmetic (int16 -> float32, one float32 multiply by the
 * caller's scale, UHD: 1/32767) is exact to reproduce on the host, so the stream the detector sees is bit for bit the converted one. */
#define C8P_FMT_C32   0
#define C8P_FMT_SC16  1
#define C8P_FMT_SC8   2   /* UHD's other wire format: int8 I, int8 Q (2-byte aligned), a quarter of the link bytes of complex64 */
#define C8P_FMT_F32   3   /* 4-byte items moved as they are (the float LLR stream between demod and decode): src 4-byte aligned, n and dst_off
                             count 4-byte items from (float*)dst */
int   c8p_copy_gather(c8p_ctx* ctx, c8p_c32* dst, const c8p_piece* pieces, int32_t n);           /* pieces: host memory */
uint64_t c8p_copy_mark(c8p_ctx* ctx);
int   c8p_copy_wait_mark(c8p_ctx* ctx, uint64_t mark);
int   c8p_copy_mark_done(c8p_ctx* ctx, uint64_t mark);
/* The NEXT submit of this ctx (c8p_demod / c8p_decode / c8p_rx) waits ON THE DEVICE for every copy enqueued on the copy stream so far:
 * how a caller that fills its own C8P_MEM_DEVICE arenas with c8p_copy_gather / c8p_copy_async hands them to a submit without
 * blocking a host thread in between (the block mirrors: pieces of the scheduler's pinned input buffer -> the batch's arena -> c8p_rx). */
int   c8p_copy_fence(c8p_ctx* ctx);

/* Counters of one ctx since c8p_create (monotonic unless stated otherwise). */
#define C8P_CNT_ARENA_GROWTHS    0   /* device / pinned allocations made by submits and c8p_reserve (each one sits behind a stream
                                        synchronisation: a submit that grows an arena is slow — warm up or c8p_reserve)          */
#define C8P_CNT_ARENA_BYTES      1   /* device bytes held by the arenas now (not monotonic)                                      */
#define C8P_CNT_HOST_BYTES       2   /* pinned host bytes held by the staging buffers now (not monotonic)                        */
#define C8P_CNT_DET_GIVEUPS      3   /* k_detect_scan launches in which a tile's look-back gave up waiting for a predecessor that
                                        had not started (the scan was then repeated with ticket numbering)                        */
#define C8P_CNT_DET_TICKET_SCANS 4   /* k_detect_scan launches with ticket numbering (repeats + C8P_DET_TICKETS = 1)              */
#define C8P_CNT_SLOTS            5   /* slots of this ctx (C8P_SLOTS; 1 on a caller-provided stream): results of that many submits stay resident */
#define C8P_CNT_SUBMITS          6
#define C8P_CNT_DET_TILE         7   /* a constant of the build: items per tile of the detection scan (the tests put plateaus on tile boundaries) */
#define C8P_CNT_COUNT            8
int64_t c8p_counter(c8p_ctx* ctx, int which);        /* < 0: C8P_EINVAL */

/* Per-ctx options a caller sets right after c8p_create (round 6; an added entry point, the structs and C8P_ABI_VERSION are unchanged:
 * a caller looks it up and goes without when an older library does not have it).  None changes results.
 *   C8P_OPT_COPY_CUS   compute units the ctx's copy streams (c8p_copy_gather, c8p_copy_async) may use, set before the first copy; the
 *                      environment's C8P_COPY_CUS still overrides it.  The default (32 for 8-byte items, 64 for sc16 / sc8) is tuned for
 *                      the live ingest's 268 MB rounds; a block front that gathers 2-3 MB per general_work call leaves more of the chip to
 *                      the chain's kernels with 16 (fused demod -> decode chain 704-721 k against 656-676 k packets/s, alternated on one
 *                      box: profiles/r6c/chain_copy_cus.txt), and that is what host/c8p_blocks.cc asks for.  0: no mask.            */
#define C8P_OPT_COPY_CUS 1
int c8p_option(c8p_ctx* ctx, int which, int64_t value);      /* C8P_EINVAL: unknown option, a value out of range, or too late */

/* profile=1: accumulated HIP-event time and launch count of one kernel since the last reset. */
int c8p_kernel_time(c8p_ctx* ctx, int kernel, double* total_ms, int64_t* launches);
int c8p_profile_reset(c8p_ctx* ctx);

/* Constant tables the engine generated (for tests): returns element count, copies at most cap ints. */
#define C8P_TAB_DEINT_L      0   /* + nbpsc index 0..3 (1,2,4,6)  in `sub`  */
#define C8P_TAB_DEINT_NL     1   /* + nbpsc index 0..4 (1,2,4,6,8)           */
#define C8P_TAB_DEINT_NL2    2
#define C8P_TAB_SIG_DEMAP_NL 3
#define C8P_TAB_DEINT_SIGB   4
#define C8P_TAB_SV_OUT       5
#define C8P_TAB_PILOT_P      6   /* as ints +1/-1 */
#define C8P_TAB_LTF_NL       7   /* as ints +1/-1/0 */
#define C8P_TAB_LTF_NL_VHT22 8
int c8p_table(int which, int sub, int32_t* dst, int cap);

/* Test hook: the device's scalar arithmetic primitives over n operand tuples (host arrays in / out), for comparison with the
 * reference's libm / libgcc forms on operands no packet exercises.
 *   C8P_SELFTEST_CABS  in = n x {re, im}                 out = n x |z|        std::abs(std::complex<float>) = hypotf
 *   C8P_SELFTEST_CDIV  in = n x {a.re, a.im, b.re, b.im} out = n x {re, im}   std::complex<float> division (libgcc __divsc3)
 *   C8P_SELFTEST_AC    in = n x {p.re, p.im, e}          out = n x ac         sqrtf(p.re^2 + p.im^2) / e  (presiso output 0) */
#define C8P_SELFTEST_CABS 0
#define C8P_SELFTEST_CDIV 1
#define C8P_SELFTEST_AC   2
int c8p_selftest(c8p_ctx* ctx, int which, const float* in, float* out, uint64_t n);

/* {ABI version, sizeof pkt_in, pkt_tags, pkt_out, cfg, batch_in, decode_in, sig_in, signal_in, sync_evt, detect_in, det_state, tx_pkt, tx_desc,
 * tx_in, tx_plan, tx_mu, det_stream, detect_multi_in, reserve_in, span, piece, fetch_piece} as compiled into the library. */
int c8p_abi_sizes(int32_t* dst, int cap);

#ifdef __cplusplus
}
#endif
#endif /* C8P_RX_H */
