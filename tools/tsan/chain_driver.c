/* tools/tsan/chain_driver.c — the demod -> decode block mirrors on two threads (c8ph_run_chain) from plain C, for tools/sanitize_tsan.sh.
 * Input: a file written by the script — per packet { u32 n_items, i32 mcs, i32 len, i32 nsamp, i32 seq, f32 cfo, f32 snr, f32 rssi,
 * f32 chan[128], f32 iq[2 * n_items] }.  fused = argv[3]. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

void* c8ph_create_ex(int kind, int mupos, int ifdebug, const char* lib, int max_batch, int flush_idle, int fused, char* errbuf, int errcap);
void c8ph_destroy(void* p);
void c8ph_push_items(void* p, int port, const void* data, long n_items);
void c8ph_push_tag_long(void* p, uint64_t off, const char* key, long v);
void c8ph_push_tag_float(void* p, uint64_t off, const char* key, float v);
void c8ph_push_tag_c32(void* p, uint64_t off, const char* key, const float* reim, int n);
long c8ph_run_chain(void* pa, void* pb, int chunk, int nout);
const char* c8ph_error(void* p);
long c8ph_msg_count(void* p);

int main(int argc, char** argv)
{
  if (argc < 4) { fprintf(stderr, "usage: %s libengine.so packets.bin fused(0|1|-1 = the defaults: fused_auto)\n", argv[0]); return 1; }
  char err[512] = "";
  void* dem = c8ph_create_ex(0, 0, 0, argv[1], 8, 1, atoi(argv[3]), err, (int)sizeof(err));
  void* dec = c8ph_create_ex(2, 0, 0, argv[1], 8, 1, atoi(argv[3]), err, (int)sizeof(err));
  if (!dem || !dec) { fprintf(stderr, "create: %s\n", err); return 2; }
  FILE* f = fopen(argv[2], "rb");
  if (!f) { perror(argv[2]); return 1; }
  uint64_t pos = 0;
  long n_pkt = 0;
  for (;;) {
    uint32_t n; int32_t q[4]; float t[3], chan[128];
    if (fread(&n, 4, 1, f) != 1) break;
    if (fread(q, 4, 4, f) != 4 || fread(t, 4, 3, f) != 3 || fread(chan, 4, 128, f) != 128) return 1;
    float* iq = (float*)malloc(8 * (size_t)n);
    if (fread(iq, 8, n, f) != n) return 1;
    c8ph_push_items(dem, 0, iq, (long)n);
    c8ph_push_tag_float(dem, pos, "cfo", t[0]); c8ph_push_tag_float(dem, pos, "snr", t[1]); c8ph_push_tag_float(dem, pos, "rssi", t[2]);
    c8ph_push_tag_long(dem, pos, "seq", q[3]); c8ph_push_tag_long(dem, pos, "mcs", q[0]); c8ph_push_tag_long(dem, pos, "len", q[1]);
    c8ph_push_tag_long(dem, pos, "nsamp", q[2]); c8ph_push_tag_c32(dem, pos, "chan", chan, 64);
    pos += n;
    n_pkt++;
    free(iq);
  }
  fclose(f);
  const long calls = c8ph_run_chain(dem, dec, 3000, 8192);
  if (calls < 0) { fprintf(stderr, "run_chain: %s | %s\n", c8ph_error(dem), c8ph_error(dec)); return 2; }
  printf("%ld packets in, %ld general_work calls, %ld PDUs (fused %s)\n", n_pkt, calls, c8ph_msg_count(dec), argv[3]);
  const int ok = c8ph_msg_count(dec) > 0;
  c8ph_destroy(dem); c8ph_destroy(dec);
  return ok ? 0 : 3;
}
