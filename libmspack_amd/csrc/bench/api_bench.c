/* api_bench.c -- BENCH / TEST INFRASTRUCTURE (not the decode product): times the library through the API the north star
 * names -- mspack_create_cab_decompressor() / mspack_create_chm_decompressor() -> open() -> extract() of every file
 * (reference cabd.c:1075-1214, chmd.c:906-1041) -- with an in-memory mspack_system written in C, the way the reference's
 * own test/cabd_memory.c and test/cabd_md5.c:145 drive it.  No Python on the timed path: bench.py and the parity tests
 * hand over a container image and get the extracted bytes and the time split back.
 *
 * The mspack_system here: "in" is the container image (read-only), every other name opened for writing appends to the
 * caller's output buffer (files land one behind the other in extraction order), messages are counted.  Time spent inside
 * read / seek / write callbacks is accumulated, so that the harness can split a run into the driver's reads (gather),
 * the library's own phases (mspack_hip_host_path_stats: plan, H2D + launches, wait + D2H) and sys->write.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>
#include <time.h>
#include "../../../include/mspack.h"
#include "../../../include/mspack_hip.h"

typedef struct mspk_api_stats {
  double total_s;            /* create + open + every extract + close + destroy */
  double open_s;             /* open(): headers, file list (CHM: directory chunks) */
  double first_extract_s;    /* the first extract(): it gathers and decodes the batch */
  double read_s, write_s;    /* inside sys->read / seek, inside sys->write */
  double lib_plan_ms, lib_issue_ms, lib_drain_ms;   /* the host-buffer entry points' own phases, summed over their calls */
  unsigned long long bytes_out, bytes_read;
  unsigned int n_files, n_errors, n_messages, lib_calls;
  int first_error;
} mspk_api_stats;

struct mem_sys {
  struct mspack_system sys;
  const unsigned char *image; size_t image_len;
  unsigned char *out; size_t out_cap, out_len;
  double read_s, write_s;
  unsigned long long bytes_read;
  unsigned int n_messages;
};
struct mem_file { struct mem_sys *ms; int writing; size_t pos; };

static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double) ts.tv_sec + 1e-9 * (double) ts.tv_nsec; }

static struct mspack_file *ms_open(struct mspack_system *self, const char *filename, int mode)
{
  struct mem_sys *ms = (struct mem_sys *) self;
  struct mem_file *f;
  const int writing = mode == MSPACK_SYS_OPEN_WRITE || mode == MSPACK_SYS_OPEN_APPEND || mode == MSPACK_SYS_OPEN_UPDATE;
  if (!writing && strcmp(filename, "in") != 0) return NULL;
  if (!(f = (struct mem_file *) malloc(sizeof(*f)))) return NULL;
  f->ms = ms; f->writing = writing; f->pos = 0;
  return (struct mspack_file *) f;
}
static void ms_close(struct mspack_file *file) { free(file); }
static int ms_read(struct mspack_file *file, void *buffer, int bytes)
{
  struct mem_file *f = (struct mem_file *) file;
  struct mem_sys *ms = f->ms;
  const double t0 = now_s();
  size_t n;
  if (f->writing || bytes < 0) return -1;
  n = ms->image_len - f->pos;
  if (n > (size_t) bytes) n = (size_t) bytes;
  memcpy(buffer, ms->image + f->pos, n);
  f->pos += n; ms->bytes_read += n;
  ms->read_s += now_s() - t0;
  return (int) n;
}
static int ms_write(struct mspack_file *file, void *buffer, int bytes)
{
  struct mem_file *f = (struct mem_file *) file;
  struct mem_sys *ms = f->ms;
  const double t0 = now_s();
  if (!f->writing || bytes < 0) return -1;
  if (ms->out_len + (size_t) bytes > ms->out_cap) return -1;
  memcpy(ms->out + ms->out_len, buffer, (size_t) bytes);
  ms->out_len += (size_t) bytes;
  ms->write_s += now_s() - t0;
  return bytes;
}
static int ms_seek(struct mspack_file *file, off_t offset, int mode)
{
  struct mem_file *f = (struct mem_file *) file;
  off_t base = mode == MSPACK_SYS_SEEK_START ? 0 : (mode == MSPACK_SYS_SEEK_CUR ? (off_t) f->pos : (off_t) f->ms->image_len);
  if (f->writing) return -1;
  if (base + offset < 0 || (size_t)(base + offset) > f->ms->image_len) return -1;
  f->pos = (size_t)(base + offset);
  return 0;
}
static off_t ms_tell(struct mspack_file *file) { return (off_t)((struct mem_file *) file)->pos; }
/* messages: counted; formatted and kept (one line each) when the run asked for the log.  `file` may be NULL, so the
 * system is found through the one instance a run uses (the harness is single-threaded). */
static struct mem_sys *g_cur;
static char *g_log; static size_t g_log_cap, g_log_len;
static void log_line(const char *line)
{
  size_t n = strlen(line);
  if (!g_log || g_log_len + n + 2 > g_log_cap) return;
  memcpy(g_log + g_log_len, line, n); g_log_len += n;
  g_log[g_log_len++] = '\n'; g_log[g_log_len] = 0;
}
static void ms_message(struct mspack_file *file, const char *format, ...)
{
  char line[512];
  va_list ap;
  (void) file;
  if (g_cur) g_cur->n_messages++;
  if (!g_log) return;
  va_start(ap, format);
  vsnprintf(line, sizeof(line), format, ap);
  va_end(ap);
  log_line(line);
}
/* (alloc / free / copy: the library's own defaults, as in api.MemSystem -- an in-memory FILE layer, not an allocator.  With its
 *  default allocator the library may take its big staging arenas from page-locked memory it keeps, mspack_hip.h) */
extern struct mspack_system *mspack_default_system;

static void mem_sys_init(struct mem_sys *ms, const unsigned char *image, size_t image_len, unsigned char *out, size_t out_cap)
{
  memset(ms, 0, sizeof(*ms));
  ms->sys.open = ms_open; ms->sys.close = ms_close; ms->sys.read = ms_read; ms->sys.write = ms_write;
  ms->sys.seek = ms_seek; ms->sys.tell = ms_tell; ms->sys.message = ms_message;
  ms->sys.alloc = mspack_default_system->alloc; ms->sys.free = mspack_default_system->free; ms->sys.copy = mspack_default_system->copy;
  ms->sys.null_ptr = NULL;
  ms->image = image; ms->image_len = image_len; ms->out = out; ms->out_cap = out_cap;
  g_cur = ms;
}

static void finish(struct mem_sys *ms, mspk_api_stats *st, double t_start)
{
  double ph[4] = { 0, 0, 0, 0 };
  st->total_s = now_s() - t_start;
  st->read_s = ms->read_s; st->write_s = ms->write_s;
  st->bytes_out = ms->out_len; st->bytes_read = ms->bytes_read; st->n_messages = ms->n_messages;
  mspack_hip_host_path_stats(ph, 0);
  st->lib_plan_ms = ph[0]; st->lib_issue_ms = ph[1]; st->lib_drain_ms = ph[2]; st->lib_calls = (unsigned int) ph[3];
}

/* one cabinet image: open, extract every file in list order (the first extract() decodes every folder in one batch).
 * offsets (optional, n_files + 1 entries): where each file's bytes start in `out`. */
int mspk_api_bench_cab(const unsigned char *cab, size_t cab_len, unsigned char *out, size_t out_cap,
                       unsigned long long *offsets, unsigned int max_files, mspk_api_stats *st)
{
  struct mem_sys ms;
  struct mscab_decompressor *d;
  struct mscabd_cabinet *c;
  struct mscabd_file *f;
  double t0, t1;
  memset(st, 0, sizeof(*st));
  mem_sys_init(&ms, cab, cab_len, out, out_cap);
  mspack_hip_host_path_stats(NULL, 1);
  t0 = now_s();
  if (!(d = mspack_create_cab_decompressor(&ms.sys))) return -1;
  t1 = now_s();
  c = d->open(d, "in");
  st->open_s = now_s() - t1;
  if (!c) { st->first_error = d->last_error(d); mspack_destroy_cab_decompressor(d); return -2; }
  for (f = c->files; f; f = f->next) {
    const double e0 = now_s();
    int err;
    if (offsets && st->n_files < max_files) offsets[st->n_files] = ms.out_len;
    err = d->extract(d, f, "out");
    if (st->n_files == 0) st->first_extract_s = now_s() - e0;
    if (err) { if (!st->n_errors) st->first_error = err; st->n_errors++; }
    st->n_files++;
  }
  if (offsets && st->n_files < max_files) offsets[st->n_files] = ms.out_len;
  d->close(d, c);
  mspack_destroy_cab_decompressor(d);
  finish(&ms, st, t0);
  return 0;
}

/* one CHM image: open, extract every file of the directory in list order */
int mspk_api_bench_chm(const unsigned char *chm, size_t chm_len, unsigned char *out, size_t out_cap,
                       unsigned long long *offsets, unsigned int max_files, mspk_api_stats *st)
{
  struct mem_sys ms;
  struct mschm_decompressor *d;
  struct mschmd_header *h;
  struct mschmd_file *f;
  double t0, t1;
  memset(st, 0, sizeof(*st));
  mem_sys_init(&ms, chm, chm_len, out, out_cap);
  mspack_hip_host_path_stats(NULL, 1);
  t0 = now_s();
  if (!(d = mspack_create_chm_decompressor(&ms.sys))) return -1;
  t1 = now_s();
  h = d->open(d, "in");
  st->open_s = now_s() - t1;
  if (!h) { st->first_error = d->last_error(d); mspack_destroy_chm_decompressor(d); return -2; }
  for (f = h->files; f; f = f->next) {
    const double e0 = now_s();
    int err;
    if (offsets && st->n_files < max_files) offsets[st->n_files] = ms.out_len;
    err = d->extract(d, f, "out");
    if (st->n_files == 0) st->first_extract_s = now_s() - e0;
    if (err) { if (!st->n_errors) st->first_error = err; st->n_errors++; }
    st->n_files++;
  }
  if (offsets && st->n_files < max_files) offsets[st->n_files] = ms.out_len;
  d->close(d, h);
  mspack_destroy_chm_decompressor(d);
  finish(&ms, st, t0);
  return 0;
}

/* the files order[0..n) of one cabinet image extracted one after another with ONE decompressor (MSCABD_PARAM_FIXMSZIP /
 * _SALVAGE as given) -- the call sequence of oracle/ref_harness.c:refh_cab_extract, for the driver tests that compare error
 * codes, bytes and the MESSAGE LOG with the real reference's.  Every extract's bytes land at out_offs[i]; msgs receives the
 * formatted messages, one per line, with a '#extract i' line in front of every call. */
int mspk_api_cab_run(const unsigned char *cab, size_t cab_len, const int *order, int n_order, int fix_mszip, int salvage,
                     unsigned char *out, size_t out_cap, unsigned long long *out_offs, unsigned long long *out_lens, int *errs,
                     char *msgs, size_t msgs_cap)
{
  struct mem_sys ms;
  struct mscab_decompressor *d;
  struct mscabd_cabinet *c;
  int i;
  mem_sys_init(&ms, cab, cab_len, out, out_cap);
  g_log = msgs; g_log_cap = msgs_cap; g_log_len = 0;
  if (msgs && msgs_cap) msgs[0] = 0;
  if (!(d = mspack_create_cab_decompressor(&ms.sys))) { g_log = NULL; return -1; }
  d->set_param(d, MSCABD_PARAM_FIXMSZIP, fix_mszip);
  d->set_param(d, MSCABD_PARAM_SALVAGE, salvage);
  if (!(c = d->open(d, "in"))) { i = d->last_error(d); mspack_destroy_cab_decompressor(d); g_log = NULL; return i ? i : -2; }
  for (i = 0; i < n_order; i++) {
    struct mscabd_file *f = c->files;
    char mark[32];
    int k = order[i];
    size_t before = ms.out_len;
    while (f && k-- > 0) f = f->next;
    out_offs[i] = before; out_lens[i] = 0;
    if (!f) { errs[i] = MSPACK_ERR_ARGS; continue; }
    snprintf(mark, sizeof(mark), "#extract %d", i);
    log_line(mark);
    errs[i] = d->extract(d, f, "out");
    out_lens[i] = ms.out_len - before;
  }
  d->close(d, c);
  mspack_destroy_cab_decompressor(d);
  g_log = NULL;
  return 0;
}
