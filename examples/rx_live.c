/* examples/rx_live.c — K live 20 MS/s streams to PDUs through the C ABI alone (plain C, no Python, no GNU Radio): what K copies of
 * the reference's rx.grc receive chain (examples/rx.grc:753-767: presiso / trigger / sync -> signal -> demod -> decode) do for K
 * radios, on one GPU through ONE host/c8p_rawchain handle.  Every capture file given on the command line plays one radio: its items
 * are pushed in increments, as a driver's receive callback would deliver them, and every round — everything pushed since the last
 * one, for all streams — goes down the pipeline as one detection pass, one signal call and one decode batch.
 *
 *     cc -O2 -Iinclude -Igr-ieee80211_amd/host examples/rx_live.c -o examples/rx_live -Lgr-ieee80211_amd -lc8p_blocks \
 *        -Wl,-rpath,$PWD/gr-ieee80211_amd
 *     examples/rx_live [-l libc8p_rx.so] [-r items_per_round] a.c64 b.c64 ...     (one line per PDU: stream, position, bytes)
 *     examples/rx_live -s 3.05e-5 a.sc16 b.sc16 ...     the files hold what the radios send — int16 I/Q pairs — and the gather kernel
 *                                                       converts ((float)v * scale): half the bytes over PCIe (C8P_MEM_SC16)
 */
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "c8p_rawchain.h"

struct sink { void* eng; int (*fetch)(c8p_ctx*, int, void*, uint64_t, uint64_t); const void* (*buffer)(c8p_ctx*, int, uint64_t*);
              long pdus, packets; int failed; };

/* runs on the pipeline's rx thread, round by round: the packets of all streams, grouped by stream, in stream order */
static void on_round(void* user, const c8p_rawchain_out* o)
{
  struct sink* s = (struct sink*)user;
  uint64_t nb = 0;
  (void)s->buffer(o->rx, C8P_BUF_PDU, &nb);
  uint8_t* pdu = (uint8_t*)malloc(nb ? nb : 1);
  if (nb && s->fetch(o->rx, C8P_BUF_PDU, pdu, 0, nb) != C8P_OK) { s->failed = 1; free(pdu); return; }
  for (int p = 0; p < o->n_packets; p++) {
    const c8p_pkt_out* r = &o->outs[p];
    const uint8_t* rec = pdu + r->pdu_off;
    for (int k = 0; k < r->n_pdu; k++) {             /* records {u16 len LE, blob = format, len lo, len hi, MPDU.., mcs} */
      const unsigned len = rec[0] | ((unsigned)rec[1] << 8);
      printf("round %lld stream %u item %llu: pdu %u bytes, format %u mcs %u, first bytes %02x %02x %02x %02x\n", (long long)o->segment,
             o->pkt_stream[p], (unsigned long long)o->pkt_index[p], len - 4, rec[2], rec[2 + len - 1], rec[5], rec[6], rec[7], rec[8]);
      rec += 2 + len;
      s->pdus++;
    }
    s->packets++;
  }
  free(pdu);
}

int main(int argc, char** argv)
{
  const char* lib = NULL;
  uint64_t per_round = 20000;
  float sc16 = 0.0f;                               /* > 0: the files are sc16, this is the scale */
  int a = 1;
  while (a + 1 < argc && argv[a][0] == '-') {
    if (!strcmp(argv[a], "-l")) lib = argv[a + 1];
    else if (!strcmp(argv[a], "-r")) per_round = strtoull(argv[a + 1], NULL, 10);
    else if (!strcmp(argv[a], "-s")) sc16 = strtof(argv[a + 1], NULL);
    else break;
    a += 2;
  }
  const int K = argc - a;
  if (K < 1 || per_round < 1) { fprintf(stderr, "usage: %s [-l libc8p_rx.so] [-r items_per_round] [-s sc16_scale] capture ...\n", argv[0]); return 1; }
  /* the engine library: the callback fetches the PDU arena through it (the same one the handle loads), and sc16 buffers come from its
   * c8p_host_alloc — the device reads them itself, so they must be pinned */
  struct sink sk;
  memset(&sk, 0, sizeof(sk));
  sk.eng = dlopen(lib ? lib : (getenv("C8P_RX_LIB") ? getenv("C8P_RX_LIB") : "libc8p_rx.so"), RTLD_NOW | RTLD_LOCAL);
  if (!sk.eng) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
  void* (*host_alloc)(uint64_t);
  void (*host_free)(void*);
  *(void**)&sk.fetch = dlsym(sk.eng, "c8p_fetch");
  *(void**)&sk.buffer = dlsym(sk.eng, "c8p_buffer");
  *(void**)&host_alloc = dlsym(sk.eng, "c8p_host_alloc");
  *(void**)&host_free = dlsym(sk.eng, "c8p_host_free");
  if (!sk.fetch || !sk.buffer || !host_alloc || !host_free) { fprintf(stderr, "c8p_fetch / c8p_buffer / c8p_host_alloc not exported\n"); return 2; }

  const size_t item = sc16 > 0.0f ? 4 : sizeof(c8p_c32);          /* bytes per item in the files and over the link */
  char** x = (char**)calloc((size_t)K, sizeof(*x));
  uint64_t* n = (uint64_t*)calloc((size_t)K, sizeof(*n));
  uint64_t* pos = (uint64_t*)calloc((size_t)K, sizeof(*pos));
  for (int s = 0; s < K; s++) {
    FILE* f = fopen(argv[a + s], "rb");
    if (!f) { perror(argv[a + s]); return 1; }
    fseek(f, 0, SEEK_END);
    n[s] = (uint64_t)ftell(f) / item;
    fseek(f, 0, SEEK_SET);
    x[s] = (char*)(sc16 > 0.0f ? host_alloc(n[s] * item + 16) : malloc((size_t)n[s] * item + 16));
    if (!x[s] || fread(x[s], item, (size_t)n[s], f) != (size_t)n[s]) { fprintf(stderr, "read error: %s\n", argv[a + s]); return 1; }
    fclose(f);
  }

  char err[512] = "";
  c8p_rawchain_cfg cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.struct_size = sizeof(cfg);
  cfg.lib = lib;                                   /* NULL: C8P_RX_LIB or "libc8p_rx.so" on the loader path */
  cfg.streams = (uint32_t)K;
  cfg.stream_round_items = per_round;
  cfg.sc16_scale = sc16;
  c8p_rawchain* h = c8p_rawchain_create(&cfg, err, (int)sizeof(err));
  if (!h) { fprintf(stderr, "c8p_rawchain_create: %s\n", err); return 2; }

  if (c8p_rawchain_live(h, on_round, &sk) != C8P_OK) { fprintf(stderr, "%s\n", c8p_rawchain_error(h)); return 2; }
  long rounds = 0;
  for (;;) {
    int any = 0;
    for (int s = 0; s < K; s++) {
      /* stream s delivers (s + 1) / K of a full increment per round: the radios are not in step */
      uint64_t m = per_round * (uint64_t)(s + 1) / (uint64_t)K;
      if (m > n[s] - pos[s]) m = n[s] - pos[s];
      if (!m) continue;
      if (c8p_rawchain_push(h, (uint32_t)s, x[s] + pos[s] * item, m, sc16 > 0.0f ? (C8P_MEM_PINNED | C8P_MEM_SC16) : C8P_MEM_HOST) != C8P_OK) { fprintf(stderr, "push: %s\n", c8p_rawchain_error(h)); return 2; }
      pos[s] += m;
      any = 1;
    }
    if (!any) break;
    if (c8p_rawchain_round(h) != C8P_OK) { fprintf(stderr, "round: %s\n", c8p_rawchain_error(h)); return 2; }
    rounds++;
  }
  c8p_rawchain_stats st;
  if (c8p_rawchain_drain(h, &st) != C8P_OK) { fprintf(stderr, "drain: %s\n", c8p_rawchain_error(h)); return 2; }
  c8p_rawchain_destroy(h);
  printf("%d streams, %ld rounds, %lld items, %lld sync events, %ld packets, %ld PDUs\n", K, rounds, (long long)st.items, (long long)st.events,
         sk.packets, sk.pdus);
  for (int s = 0; s < K; s++) { if (sc16 > 0.0f) host_free(x[s]); else free(x[s]); }
  free(x); free(n); free(pos);
  return sk.failed ? 2 : 0;
}
