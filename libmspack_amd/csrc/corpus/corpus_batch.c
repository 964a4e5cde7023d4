/* corpus_batch.c -- multi-threaded generation of batches of independent units (bench/test
 * infrastructure): plaintext from mspk_gen_plaintext, one LZX stream per unit. */
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include "corpus.h"

typedef struct {
  uint64_t base_seed; int kind, window_bits; size_t unit_bytes;
  const mspk_lzx_opts *opts;
  uint8_t *plain; uint8_t **tmp; size_t *tmp_len; uint64_t **tmp_fo;
  int first, last;
  uint64_t index0;               /* global index of unit 0 of this call (strong-scaling shards) */
} job_t;

static void *worker(void *arg) {
  job_t *j = (job_t *) arg;
  int u;
  size_t bound = mspk_lzx_bound(j->unit_bytes);
  for (u = j->first; u < j->last; u++) {
    uint8_t *p = j->plain + (size_t) u * j->unit_bytes;
    int reset = (int)((j->unit_bytes + 32767) / 32768);
    /* unit_seed = golden-ratio hash of (config<<32 | unit) */
    mspk_gen_plaintext(j->base_seed ^ (0x9E3779B97F4A7C15ull * (j->index0 + (uint64_t) u + 1ull)), j->kind, p, j->unit_bytes);
    j->tmp[u] = (uint8_t *) malloc(bound);
    j->tmp_fo[u] = (uint64_t *) calloc((size_t) reset + 1, sizeof(uint64_t));
    j->tmp_len[u] = mspk_lzx_encode(p, j->unit_bytes, j->window_bits, reset, j->opts, j->tmp[u], bound, j->tmp_fo[u]);
  }
  return NULL;
}

size_t mspk_corpus_lzx_units(uint64_t base_seed, int kind, int n_units, size_t unit_bytes,
                             int window_bits, const mspk_lzx_opts *opts, int n_threads,
                             uint8_t *plain, uint8_t *comp, size_t comp_cap,
                             uint64_t *comp_off, uint32_t *comp_len)
{
  return mspk_corpus_lzx_units_at(base_seed, 0, kind, n_units, unit_bytes, window_bits, opts, n_threads,
                                  plain, comp, comp_cap, comp_off, comp_len);
}

size_t mspk_corpus_lzx_units_at(uint64_t base_seed, uint64_t first_unit, int kind, int n_units, size_t unit_bytes,
                                int window_bits, const mspk_lzx_opts *opts, int n_threads,
                                uint8_t *plain, uint8_t *comp, size_t comp_cap,
                                uint64_t *comp_off, uint32_t *comp_len)
{
  return mspk_corpus_lzx_units_ft(base_seed, first_unit, kind, n_units, unit_bytes, window_bits, opts, n_threads,
                                  plain, comp, comp_cap, comp_off, comp_len, NULL);
}

size_t mspk_corpus_lzx_units_ft(uint64_t base_seed, uint64_t first_unit, int kind, int n_units, size_t unit_bytes,
                                int window_bits, const mspk_lzx_opts *opts, int n_threads,
                                uint8_t *plain, uint8_t *comp, size_t comp_cap,
                                uint64_t *comp_off, uint32_t *comp_len, uint64_t *tab_off)
{
  pthread_t th[256];
  job_t jobs[256];
  uint8_t **tmp = (uint8_t **) calloc((size_t) n_units, sizeof(*tmp));
  size_t *tmp_len = (size_t *) calloc((size_t) n_units, sizeof(*tmp_len));
  uint64_t **tmp_fo = (uint64_t **) calloc((size_t) n_units, sizeof(*tmp_fo));
  const size_t nfr = (unit_bytes + 32767) / 32768;
  size_t pos = 0;
  int t, u, ok = 1;
  if (n_threads < 1) n_threads = 1;
  if (n_threads > 256) n_threads = 256;
  if (n_threads > n_units) n_threads = n_units > 0 ? n_units : 1;
  for (t = 0; t < n_threads; t++) {
    jobs[t].base_seed = base_seed; jobs[t].kind = kind; jobs[t].window_bits = window_bits;
    jobs[t].unit_bytes = unit_bytes; jobs[t].opts = opts; jobs[t].plain = plain;
    jobs[t].tmp = tmp; jobs[t].tmp_len = tmp_len; jobs[t].tmp_fo = tmp_fo; jobs[t].index0 = first_unit;
    jobs[t].first = (int)((long long) n_units * t / n_threads);
    jobs[t].last  = (int)((long long) n_units * (t + 1) / n_threads);
    pthread_create(&th[t], NULL, worker, &jobs[t]);
  }
  for (t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
  for (u = 0; u < n_units; u++) {
    size_t l = tmp_len[u];
    pos = (pos + 15) & ~(size_t) 15;                 /* units start 16-byte aligned in the arena */
    if (l == 0 || pos + l + 4 > comp_cap) ok = 0;
    if (ok) { memcpy(comp + pos, tmp[u], l); comp_off[u] = pos; comp_len[u] = (uint32_t) l; pos += l + 4; }  /* >= 4 zero bytes follow every unit */
    if (ok && tab_off) {
      /* the unit's frame table, as a container would state it (CHM reset table / CFDATA block sizes): where
       * each 32 KiB frame starts in the compressed stream, uint32 offsets from the unit's first byte */
      size_t k;
      pos = (pos + 3) & ~(size_t) 3;
      if (pos + nfr * 4 > comp_cap) ok = 0;
      else {
        tab_off[u] = pos;
        for (k = 0; k < nfr; k++) { uint32_t v = (uint32_t) tmp_fo[u][k]; memcpy(comp + pos + 4 * k, &v, 4); }
        pos += nfr * 4;
      }
    }
    free(tmp[u]); free(tmp_fo[u]);
  }
  free(tmp); free(tmp_len); free(tmp_fo);
  return ok ? pos : 0;
}
