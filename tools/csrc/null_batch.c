/* scratch (host-path profiling on a machine without a GPU): a batch ABI that decodes NOTHING -- every unit is answered
 * "all bytes produced", the output arena is touched once (as the D2H copy would) and left zero.  Lets the C drivers'
 * own work (gather, checksums, arenas, slicing, sys->write) be timed and profiled alone.  Never part of the product. */
#include <string.h>
#include <time.h>
static double g_ms;
static double now_ms(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; }
#include "mspack_hip.h"
const char *mspack_hip_version(void) { return "null batch (profiling only)"; }
const char *mspack_hip_last_error(void) { return ""; }
int mspack_hip_device_count(void) { return 1; }
int mspack_hip_set_device(int d) { (void) d; return 0; }
int mspack_hip_decode_batch(mspack_hip_unit *units, size_t n, const void *in, size_t in_bytes, void *out, size_t out_bytes,
                            mspack_hip_result *res)
{
  size_t k;
  (void) in_bytes;
  { const double t0 = now_ms(); memset(out, 0, out_bytes); g_ms += now_ms() - t0; }      /* (first touch of the arena) */
  const double t1 = now_ms();
  for (k = 0; k < n; k++) {
    memset(&res[k], 0, sizeof(res[k])); res[k].out_len = res[k].good_len = units[k].out_len;
    if (units[k].kind == MSPACK_HIP_KIND_XORSUM) {        /* (what the device answers; not part of the drivers' time) */
      const unsigned char *d = (const unsigned char *) in + units[k].in_off;
      unsigned int w = units[k].in_len >> 2, sum = 0, v, tail = 0;
      while (w--) { memcpy(&v, d, 4); sum ^= v; d += 4; }
      switch (units[k].in_len & 3) { case 3: tail |= (unsigned int) *d++ << 16; /* fall through */ case 2: tail |= (unsigned int) *d++ << 8; /* fall through */ case 1: tail |= *d; }
      res[k].in_next = sum ^ tail; res[k].in_used = units[k].in_len;
    }
  }
  g_ms += now_ms() - t1;
  return 0;
}
int mspack_hip_decode_batch_multi(mspack_hip_unit *units, size_t n, const void *in, size_t in_bytes, void *out, size_t out_bytes,
                                  mspack_hip_result *res, int devices)
{ (void) devices; return mspack_hip_decode_batch(units, n, in, in_bytes, out, out_bytes, res); }
/* (no jobs: the drivers take the synchronous call) */
mspack_hip_job *mspack_hip_decode_batch_begin(mspack_hip_unit *units, size_t n, const void *in, size_t in_bytes, void *out, size_t out_bytes,
                                              mspack_hip_result *res)
{ (void) units; (void) n; (void) in; (void) in_bytes; (void) out; (void) out_bytes; (void) res; return 0; }
int mspack_hip_job_wait_unit(mspack_hip_job *job, size_t i) { (void) job; (void) i; return -1; }
int mspack_hip_job_end(mspack_hip_job *job) { (void) job; return -1; }
void mspack_hip_host_path_stats(double *ms4, int reset) { (void) reset; if (ms4) { ms4[0] = ms4[1] = ms4[3] = 0; ms4[2] = g_ms; } if (reset) g_ms = 0; }
int mspack_hip_pin(const void *p, size_t bytes) { (void) p; (void) bytes; return 1; }      /* (nothing to lock without a device) */
void mspack_hip_unpin(const void *p) { (void) p; }
void *mspack_hip_stage_alloc(size_t bytes) { (void) bytes; return 0; }                       /* (no device: the ordinary allocator) */
void mspack_hip_stage_free(void *p) { (void) p; }
