/* tools/tsan/blocks_driver.c — the scheduler harness of libc8p_blocks.so driven from plain C, for tools/sanitize_tsan.sh (python under
 * ThreadSanitizer is too slow): the rx_chain block over K capture files (equal lengths are cut to the shortest), with ragged calls. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

void* c8ph_create_rx_chain(int n_streams, int item_format, float scale, const char* lib, int items_per_round, int flush_idle, char* errbuf, int errcap);
void c8ph_destroy(void* p);
void c8ph_push_items(void* p, int port, const void* data, long n_items);
void c8ph_set_jitter(void* p, unsigned long long seed);
long c8ph_run(void* p, int chunk, int nout, int flush_at_end);
long c8ph_run_live(void* p, int chunk, int nout, int linger_ms);
const char* c8ph_error(void* p);
long c8ph_msg_count(void* p);

int main(int argc, char** argv)
{
  if (argc < 4) { fprintf(stderr, "usage: %s libengine.so items_per_round capture.c64 ...\n", argv[0]); return 1; }
  const int K = argc - 3;
  char err[512] = "";
  long total = 0;
  for (int pass = 0; pass < 2; pass++) {               /* pass 0: c8ph_run with jitter; pass 1: c8ph_run_live (the wake path) */
    void* h = c8ph_create_rx_chain(K, 0, 0.0f, argv[1], atoi(argv[2]), 1, err, (int)sizeof(err));
    if (!h) { fprintf(stderr, "create: %s\n", err); return 2; }
    long nmin = -1;
    char** buf = (char**)calloc((size_t)K, sizeof(*buf));
    for (int s = 0; s < K; s++) {
      FILE* f = fopen(argv[3 + s], "rb");
      if (!f) { perror(argv[3 + s]); return 1; }
      fseek(f, 0, SEEK_END);
      const long n = ftell(f) / 8;
      fseek(f, 0, SEEK_SET);
      buf[s] = (char*)malloc((size_t)n * 8 + 8);
      if (fread(buf[s], 8, (size_t)n, f) != (size_t)n) return 1;
      fclose(f);
      if (nmin < 0 || n < nmin) nmin = n;
    }
    for (int s = 0; s < K; s++) c8ph_push_items(h, s, buf[s], nmin);
    if (pass == 0) c8ph_set_jitter(h, 5);
    const long calls = pass == 0 ? c8ph_run(h, 2500, 4096, 1) : c8ph_run_live(h, 2500, 4096, 50);
    if (calls < 0) { fprintf(stderr, "run: %s\n", c8ph_error(h)); return 2; }
    printf("pass %d: %ld general_work calls, %ld PDUs\n", pass, calls, c8ph_msg_count(h));
    total += c8ph_msg_count(h);
    c8ph_destroy(h);
    for (int s = 0; s < K; s++) free(buf[s]);
    free(buf);
  }
  return total > 0 ? 0 : 3;
}
