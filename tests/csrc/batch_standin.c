/* tests/csrc/batch_standin.c -- TEST INFRASTRUCTURE ONLY.  Never built into, linked with or loaded by
 * libmspack_hip.so; tests/conftest.py compiles it together with the host drivers' C files
 * (libmspack_amd/csrc/host/*.c) into tests/_build/libhostlogic_cpu.so so that the HOST logic of the
 * libmspack-compatible API (container parsing, unit gathering, skip-then-emit, error mapping, decoder
 * lifetime emulation) can be exercised by the `-m "not gpu"` suite on a machine without a GPU.
 *
 * It provides the batch ABI of include/mspack_hip.h (host-buffer entry points only) on top of the CPU
 * oracle (oracle/liboracle.so): one oracle call per unit.  Supported: LZX units (CAB folders, CHM reset
 * intervals, E8 origin, the reset log of MSPACK_HIP_UF_LZX_LOG), MSZIP and Quantum folder units, frame tables
 * (ignored: the oracle is serial), the feeder's failed read (MSPACK_HIP_UF_HARD_EOF), Quantum's good_len and marks
 * (MSPACK_HIP_UF_QTM_MARKS), LZX DELTA units (OAB
 * blocks with their reference data), LZSS and KWAJ LZH units, checksum units;
 * NOT MSZIP repair mode or KWAJ-framed MSZIP -- that makes the call fail, and the tests that need it run on the GPU.  The `-m gpu` parity tests never
 * see this file: they load the real library. */
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include "../../include/mspack_hip.h"
#include "../../oracle/oracle.h"

static char g_err[160] = "";
const char *mspack_standin_last_error(void) { return g_err; }
#ifndef STANDIN_NO_ABI
const char *mspack_hip_version(void) { return "host-logic stand-in (CPU oracle; tests only)"; }
const char *mspack_hip_last_error(void) { return g_err; }
int mspack_hip_device_count(void) { return 1; }
int mspack_hip_set_device(int d) { (void) d; return 0; }
#endif

/* units[order[0 .. n)] (order NULL: units[0 .. n)), one oracle call each -- also what tests/hostcheck puts in the place of a
 * kernel launch (STANDIN_NO_ABI: only this function, the batch ABI itself is then shim.hip's) */
int mspack_standin_decode_units(const mspack_hip_unit *units, const uint32_t *order, size_t n_units, const void *in, size_t in_bytes,
                                void *out, size_t out_bytes, mspack_hip_result *results)
{
  size_t i;
  for (i = 0; i < n_units; i++) {
    const size_t ui = order ? order[i] : i;
    const mspack_hip_unit *u = &units[ui];
    mspack_hip_result *r = &results[ui];
    oracle_result o;
    uint32_t qtm_good = 0; int have_qtm_good = 0;
    const uint8_t *src = (const uint8_t *) in + u->in_off;
    uint8_t *dst = (uint8_t *) out + u->out_off;
    memset(&o, 0, sizeof(o));
    memset(r, 0, sizeof(*r));
    if (u->in_off + u->in_len > in_bytes || u->out_off + u->out_len + (u->kind == MSPACK_HIP_KIND_MSZIP ? 32768u : 0u) > out_bytes) {
      snprintf(g_err, sizeof(g_err), "unit outside arena"); return -1;
    }
    /* the frame table is a hint for the GPU's frame-parallel parse; results do not depend on it */
    if (u->flags & ~(MSPACK_HIP_UF_FRAME_TABLE | MSPACK_HIP_UF_HARD_EOF | (u->kind == MSPACK_HIP_KIND_LZX ? MSPACK_HIP_UF_LZX_LOG : 0u) |
                     (u->kind == MSPACK_HIP_KIND_QUANTUM ? MSPACK_HIP_UF_QTM_MARKS : 0u))) { snprintf(g_err, sizeof(g_err), "stand-in: unit flags 0x%x unsupported", u->flags); return -1; }
    if (u->kind == 0) { r->err = 1; continue; }
    if (u->kind == MSPACK_HIP_KIND_XORSUM) {                     /* a CFDATA block's checksum (cabd.c:1462-1479) */
      if (u->out_len) { snprintf(g_err, sizeof(g_err), "a checksum unit has no output"); return -1; }
      r->in_used = u->in_len; r->in_next = oracle_cab_checksum(src, u->in_len, 0);
      continue;
    }
    oracle_set_hard_eof((u->flags & MSPACK_HIP_UF_HARD_EOF) != 0);
    switch (u->kind) {
    case MSPACK_HIP_KIND_LZX:
      oracle_lzx_decode(src, u->in_len, dst, u->out_len, u->out_len, u->out_len, u->window_bits, u->reset_frames,
                        u->e8_base, &o);
      if (u->flags & MSPACK_HIP_UF_LZX_LOG) {                 /* the unit's reset log (mspack_hip.h) */
        uint8_t *lg = dst + (((size_t) u->out_len + 32768 + 15) & ~(size_t) 15);
        uint32_t fr[256], cnt, k;
        if (u->out_off + (((size_t) u->out_len + 32768 + 15) & ~(size_t) 15) + 4 + 4 * (size_t) u->ref_len > out_bytes) {
          snprintf(g_err, sizeof(g_err), "unit's log outside arena"); return -1;
        }
        cnt = oracle_lzx_open_resets(fr, 256);
        memcpy(lg, &cnt, 4);
        for (k = 0; k < cnt && k < u->ref_len && k < 256; k++) memcpy(lg + 4 + 4 * (size_t) k, &fr[k], 4);
      }
      break;
    case MSPACK_HIP_KIND_MSZIP:
      /* (an MSZIP unit owns 32 KiB of room behind out_len: what its last block inflated to beyond the request lands there) */
      oracle_mszip_decode(src, u->in_len, dst, (size_t) u->out_len + 32768, u->out_len, 0, NULL, 0, NULL, &o);
      break;
    case MSPACK_HIP_KIND_QUANTUM:
      if ((u->flags & MSPACK_HIP_UF_QTM_MARKS) && u->ref_len) {  /* the unit's marks and their log (mspack_hip.h) */
        const size_t lo = ((size_t) u->out_len + 15) & ~(size_t) 15;
        if ((size_t) u->in_chunk * 4 + 4 * (size_t) u->ref_len > in_bytes || u->out_off + lo + 4 * (size_t) u->ref_len > out_bytes || (u->out_off & 3)) {
          snprintf(g_err, sizeof(g_err), "unit's marks or their log outside arena"); return -1;
        }
        oracle_qtm_set_marks((const uint32_t *)((const uint8_t *) in + (size_t) u->in_chunk * 4), u->ref_len, (uint32_t *)(dst + lo));
      }
      oracle_qtm_decode(src, u->in_len, dst, u->out_len, u->out_len, u->window_bits, &o);
      if (o.err != 0 && u->out_len) {
        /* good_len (mspack_hip.h): how far a SHORTER request would still have succeeded -- qtmd writes only when its window wraps
         * and at the end of a call that succeeds, so the bytes it handed over say nothing about it: bisect with the oracle */
        uint8_t *tmp = (uint8_t *) malloc((size_t) u->out_len + 64);
        uint32_t lo = 0, hi = u->out_len;                        /* decode(lo) succeeds, decode(hi) fails */
        while (tmp && hi - lo > 1) {
          const uint32_t mid = lo + (hi - lo) / 2;
          oracle_result t;
          memset(&t, 0, sizeof(t));
          oracle_qtm_decode(src, u->in_len, tmp, mid, mid, u->window_bits, &t);
          if (t.err == 0) lo = mid; else hi = mid;
        }
        if (tmp) { oracle_result t; memset(&t, 0, sizeof(t)); oracle_qtm_decode(src, u->in_len, dst, lo, lo, u->window_bits, &t); }   /* the good bytes, in place */
        free(tmp);
        qtm_good = lo; have_qtm_good = 1;
      }
      break;
    case MSPACK_HIP_KIND_LZX_DELTA:                       /* OAB blocks: reference data right below the unit's output */
      if (u->ref_len > u->out_off) { snprintf(g_err, sizeof(g_err), "unit's lower region outside arena"); return -1; }
      oracle_lzxd_decode(src, u->in_len, dst, u->out_len, u->out_len, u->out_len, u->window_bits, u->reset_frames,
                         u->e8_base, 1, dst - u->ref_len, u->ref_len, &o);
      break;
    case MSPACK_HIP_KIND_LZSS:                            /* out_len = room; the result says what the stream produced */
      if (u->out_off < 4096) { snprintf(g_err, sizeof(g_err), "unit's lower region outside arena"); return -1; }
      oracle_lzss_decode(src, u->in_len, u->window_bits, dst, u->out_len, &o);
      break;
    case MSPACK_HIP_KIND_KWAJ_LZH:
      if (u->out_off < 4096) { snprintf(g_err, sizeof(g_err), "unit's lower region outside arena"); return -1; }
      oracle_kwaj_lzh_decode(src, u->in_len, dst, u->out_len, &o);
      break;
    default:
      snprintf(g_err, sizeof(g_err), "stand-in: kind %d unsupported", u->kind); return -1;
    }
    oracle_set_hard_eof(0);
    r->err = o.err; r->flags = o.flags; r->out_len = (uint32_t) o.out_len; r->in_used = (uint32_t) o.in_used;
    r->good_len = have_qtm_good ? qtm_good : (uint32_t) o.out_len; r->in_next = (uint32_t) o.in_next;
  }
  return 0;
}

#ifndef STANDIN_NO_ABI
int mspack_hip_decode_batch(mspack_hip_unit *units, size_t n_units, const void *in, size_t in_bytes,
                            void *out, size_t out_bytes, mspack_hip_result *results)
{
  return mspack_standin_decode_units(units, NULL, n_units, in, in_bytes, out, out_bytes, results);
}

int mspack_hip_decode_batch_multi(mspack_hip_unit *units, size_t n_units, const void *in, size_t in_bytes,
                                  void *out, size_t out_bytes, mspack_hip_result *results, int n_devices)
{
  (void) n_devices;
  return mspack_hip_decode_batch(units, n_units, in, in_bytes, out, out_bytes, results);
}

/* jobs (mspack_hip.h): the stand-in's job is LAZY and hostile on purpose -- _begin() decodes nothing and fills the results with
 * 0xEE; _wait_unit(i) decodes unit i and the units before it (chunks finish in arena order: the real library promises no more);
 * _end() decodes the rest.  A driver that reads a result or a byte it has not waited for sees garbage here, in the CPU suite. */
static unsigned long g_jobs_begun, g_job_waits;
unsigned long mspack_standin_jobs_begun(void) { return g_jobs_begun; }
unsigned long mspack_standin_job_waits(void) { return g_job_waits; }
struct mspack_hip_job {
  mspack_hip_unit *units; size_t n, upto; const void *in; size_t in_bytes; void *out; size_t out_bytes; mspack_hip_result *res; int rc;
};
mspack_hip_job *mspack_hip_decode_batch_begin(mspack_hip_unit *units, size_t n_units, const void *in, size_t in_bytes,
                                              void *out, size_t out_bytes, mspack_hip_result *results)
{
  struct mspack_hip_job *j;
  const char *e = getenv("MSPACK_HIP_JOBS");
  if (e && atoi(e) == 0) return NULL;
  if (!(j = (struct mspack_hip_job *) malloc(sizeof(*j)))) return NULL;
  j->units = units; j->n = n_units; j->upto = 0; j->in = in; j->in_bytes = in_bytes; j->out = out; j->out_bytes = out_bytes;
  j->res = results; j->rc = 0;
  memset(results, 0xEE, n_units * sizeof(*results));
  g_jobs_begun++;
  return j;
}
static int job_advance(struct mspack_hip_job *j, size_t upto)
{
  while (!j->rc && j->upto < upto) {
    uint32_t one = (uint32_t) j->upto;
    j->rc = mspack_standin_decode_units(j->units, &one, 1, j->in, j->in_bytes, j->out, j->out_bytes, j->res);
    j->upto++;
  }
  return j->rc;
}
int mspack_hip_job_wait_unit(mspack_hip_job *job, size_t i) { g_job_waits++; return (!job || i >= job->n) ? -1 : job_advance(job, i + 1); }
int mspack_hip_job_end(mspack_hip_job *job)
{
  int rc;
  if (!job) return -1;
  rc = job_advance(job, job->n);
  free(job);
  return rc;
}

void mspack_hip_host_path_stats(double *ms4, int reset)
{
  (void) reset;
  if (ms4) ms4[0] = ms4[1] = ms4[2] = ms4[3] = 0.0;
}
int mspack_hip_pin(const void *p, size_t bytes) { (void) p; (void) bytes; return 1; }      /* (nothing to lock without a device) */
void mspack_hip_unpin(const void *p) { (void) p; }
void *mspack_hip_stage_alloc(size_t bytes) { (void) bytes; return 0; }                       /* (no device: the ordinary allocator) */
void mspack_hip_stage_free(void *p) { (void) p; }
#endif  /* !STANDIN_NO_ABI */
