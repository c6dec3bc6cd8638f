/* examples/rx_file.c — a recorded 20 MS/s complex64 stream to PDUs through the C ABI alone (plain C, no Python, no GNU Radio):
 *
 *     c8p_detect  (presiso + trigger + sync)        raw samples            -> sync events
 *     c8p_signal  (signal block)                     events + raw samples   -> packets at the demod boundary, on the device
 *     c8p_rx      (demod + decode)                   those packets          -> PDUs
 *
 * This is the rx.grc flowgraph of the reference (examples/rx.grc) with every block replaced by its engine entry point.
 *     cc -O2 -Iinclude examples/rx_file.c -o examples/rx_file -Lgr-ieee80211_amd -lc8p_rx -Wl,-rpath,$PWD/gr-ieee80211_amd
 *     examples/rx_file capture.c64           (prints one line per PDU; exit code 0)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "c8p_rx.h"

#define CHECK(c, call)                                                                      \
  do { int rc_ = (call); if (rc_ != C8P_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, c8p_last_error(c)); return 2; } } while (0)

int main(int argc, char** argv)
{
  if (argc < 2) { fprintf(stderr, "usage: %s capture.c64\n", argv[0]); return 1; }
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 1; }
  fseek(f, 0, SEEK_END);
  const long bytes = ftell(f);
  fseek(f, 0, SEEK_SET);
  const uint64_t n = (uint64_t)bytes / sizeof(c8p_c32);
  c8p_c32* raw = (c8p_c32*)malloc((size_t)n * sizeof(c8p_c32) + 16);
  if (!raw || fread(raw, sizeof(c8p_c32), (size_t)n, f) != (size_t)n) { fprintf(stderr, "read error\n"); return 1; }
  fclose(f);

  c8p_cfg cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.abi_version = C8P_ABI_VERSION;
  c8p_ctx* ctx = c8p_create(&cfg);
  if (!ctx) { fprintf(stderr, "c8p_create failed: no usable MI355X (there is no CPU fallback)\n"); return 2; }

  /* ---- packet detection over the whole capture (one segment: no history, no state in / out) */
  c8p_detect_in din;
  memset(&din, 0, sizeof(din));
  din.mem = C8P_MEM_HOST; din.raw = raw; din.n = n;
  CHECK(ctx, c8p_detect(ctx, &din));
  uint64_t evb = 0;
  (void)c8p_buffer(ctx, C8P_BUF_DET_EVT, &evb);
  const int n_evt = (int)(evb / sizeof(c8p_sync_evt));
  c8p_sync_evt* evts = (c8p_sync_evt*)calloc((size_t)n_evt + 1, sizeof(c8p_sync_evt));
  CHECK(ctx, c8p_fetch(ctx, C8P_BUF_DET_EVT, evts, 0, evb));
  printf("%llu samples, %d sync events\n", (unsigned long long)n, n_evt);

  /* ---- the signal block on every event */
  c8p_sig_in* sin_ = (c8p_sig_in*)calloc((size_t)n_evt + 1, sizeof(c8p_sig_in));
  for (int i = 0; i < n_evt; i++) {
    sin_[i].raw_off[0] = evts[i].index;
    sin_[i].n_avail = (uint32_t)(n - evts[i].index);
    sin_[i].cfo_rad = evts[i].rad; sin_[i].snr = evts[i].snr; sin_[i].rssi = evts[i].rssi;
    sin_[i].n_ant = 1;
  }
  c8p_signal_in sg;
  memset(&sg, 0, sizeof(sg));
  /* consume = 1: the engine applies the signal block's own scan rule (signal_impl.cc:76-111,154-160: sync flags inside a packet it
   * is copying are never seen, a failed L-SIG check skips 80 items) and copies only the packets the block would copy */
  c8p_pkt_in* pk = (c8p_pkt_in*)calloc((size_t)n_evt + 1, sizeof(c8p_pkt_in));
  sg.n_evt = n_evt; sg.mem = C8P_MEM_HOST; sg.evts = sin_; sg.raw = raw; sg.n_raw = n; sg.consume = 1; sg.pkt_host = pk;
  CHECK(ctx, c8p_signal(ctx, &sg));
  int n_pkt = 0;
  for (int i = 0; i < n_evt; i++)
    if (pk[i].n_items > 0) pk[n_pkt++] = pk[i];
  uint64_t iq_bytes = 0, ch_bytes = 0;
  const void* d_iq = c8p_buffer(ctx, C8P_BUF_SIG_IQ, &iq_bytes);        /* the packets stay on the device */
  const void* d_ch = c8p_buffer(ctx, C8P_BUF_SIG_CHAN, &ch_bytes);

  /* ---- demod + decode.  A second context: the result arenas of the signal call above must stay alive while c8p_rx reads them */
  c8p_ctx* rx = c8p_create(&cfg);
  if (!rx) { fprintf(stderr, "c8p_create failed\n"); return 2; }
  c8p_batch_in b;
  memset(&b, 0, sizeof(b));
  b.n_pkt = n_pkt; b.mem = C8P_MEM_DEVICE; b.pkts = pk;
  b.iq = (const c8p_c32*)d_iq; b.n_iq = iq_bytes / sizeof(c8p_c32);
  b.chan = (const c8p_c32*)d_ch; b.n_chan = ch_bytes / sizeof(c8p_c32);
  CHECK(rx, c8p_rx(rx, &b));
  CHECK(rx, c8p_sync(rx));
  c8p_pkt_out* outs = (c8p_pkt_out*)calloc((size_t)n_pkt + 1, sizeof(c8p_pkt_out));
  c8p_pkt_tags* tags = (c8p_pkt_tags*)calloc((size_t)n_pkt + 1, sizeof(c8p_pkt_tags));
  uint64_t pdu_bytes = 0;
  (void)c8p_buffer(rx, C8P_BUF_PDU, &pdu_bytes);
  uint8_t* pdu = (uint8_t*)malloc((size_t)pdu_bytes + 16);
  CHECK(rx, c8p_fetch(rx, C8P_BUF_OUT, outs, 0, (uint64_t)n_pkt * sizeof(c8p_pkt_out)));
  CHECK(rx, c8p_fetch(rx, C8P_BUF_TAGS, tags, 0, (uint64_t)n_pkt * sizeof(c8p_pkt_tags)));
  CHECK(rx, c8p_fetch(rx, C8P_BUF_PDU, pdu, 0, pdu_bytes));
  int n_pdu = 0;
  static const char* FMT[3] = {"legacy", "ht", "vht"};
  for (int p = 0; p < n_pkt; p++) {
    const uint8_t* r = pdu + outs[p].pdu_off;
    for (int k = 0; k < outs[p].n_pdu; k++) {           /* records {u16 n, blob}; blob = [format, len lo, len hi, MPDU.., mcs] */
      const unsigned nb = r[0] | ((unsigned)r[1] << 8);
      const uint8_t* blob = r + 2;
      if (blob[0] <= 2)
        printf("pdu %d: packet %d %s mcs %u, %u bytes, crc32 correct, first bytes %02x %02x %02x %02x\n", n_pdu, p, FMT[blob[0]],
               blob[nb - 1], nb - 4, blob[3], blob[4], blob[5], blob[6]);
      else
        printf("pdu %d: packet %d NDP channel report, %u bytes\n", n_pdu, p, nb);
      n_pdu++;
      r += 2 + nb;
    }
    if (outs[p].n_pdu == 0)
      printf("packet %d: status %d format %d mcs %d len %d, %d FCS failures\n", p, tags[p].status, tags[p].format, tags[p].mcs,
             tags[p].len, outs[p].n_crc_fail);
  }
  printf("%d packets, %d PDUs\n", n_pkt, n_pdu);
  c8p_destroy(rx);
  c8p_destroy(ctx);
  free(raw); free(evts); free(sin_); free(pk); free(outs); free(tags); free(pdu);
  return 0;
}
