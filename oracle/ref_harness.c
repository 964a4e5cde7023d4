/* oracle/ref_harness.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Thin memory-to-memory drivers around the REAL reference codecs.  This file is compiled only
 * in the development container, against the headers and objects that live under
 * /root/reference (see oracle/Makefile target `ref`); the result goes to oracle/_ref/ and is
 * never linked into the product.  It lets tests (a) pin our CPU restatement (liboracle.so)
 * against the reference, (b) validate every synthetic corpus stream, and (c) time the true
 * reference CPU path as bench.py's cpu_baseline (kind "reference").
 *
 * Interfaces used: lzxd_init/lzxd_decompress/lzxd_free (lzx.h:146-214), mszipd_* (mszip.h:85-120),
 * qtmd_* (qtm.h:92-122), mspack_create_cab_decompressor / mspack_create_chm_decompressor
 * (mspack.h:522-558) through an in-memory struct mspack_system (mspack.h:285-455).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>
#include <stdint.h>
#include <time.h>
#include <pthread.h>

#include <system.h>   /* reference: mspack.h + macros */
#include <lzx.h>
#include <mszip.h>
#include <qtm.h>

/* ---- in-memory mspack_system ------------------------------------------------------------ */
#define MEMNAME_MAGIC 0x4d454d21u
struct memname {            /* what we pass as "filename" */
  unsigned magic;
  uint8_t *data;            /* read: source; write: destination (may be NULL = discard) */
  size_t   len;             /* read: size;   write: capacity */
  size_t   written;         /* write: bytes written (may exceed capacity; excess dropped) */
};
struct memfile {
  struct memname *mn;
  size_t pos;
  int writing;
};

static struct mspack_file *m_open(struct mspack_system *self, const char *filename, int mode) {
  struct memname *mn = (struct memname *) filename;
  struct memfile *f;
  (void) self;
  if (!mn || mn->magic != MEMNAME_MAGIC) return NULL;
  f = (struct memfile *) malloc(sizeof(*f));
  if (!f) return NULL;
  f->mn = mn; f->pos = 0; f->writing = (mode != MSPACK_SYS_OPEN_READ);
  if (f->writing) mn->written = 0;
  return (struct mspack_file *) f;
}
static void m_close(struct mspack_file *file) { free(file); }
static int m_read(struct mspack_file *file, void *buffer, int bytes) {
  struct memfile *f = (struct memfile *) file;
  size_t avail;
  if (!f || f->writing || bytes < 0) return -1;
  avail = f->mn->len - f->pos;
  if ((size_t) bytes > avail) bytes = (int) avail;
  memcpy(buffer, f->mn->data + f->pos, (size_t) bytes);
  f->pos += (size_t) bytes;
  return bytes;
}
static int m_write(struct mspack_file *file, void *buffer, int bytes) {
  struct memfile *f = (struct memfile *) file;
  if (!f || !f->writing || bytes < 0) return -1;
  if (f->mn->data && f->mn->written < f->mn->len) {
    size_t room = f->mn->len - f->mn->written;
    memcpy(f->mn->data + f->mn->written, buffer, (size_t) bytes < room ? (size_t) bytes : room);
  }
  f->mn->written += (size_t) bytes;
  return bytes;
}
static int m_seek(struct mspack_file *file, off_t offset, int mode) {
  struct memfile *f = (struct memfile *) file;
  off_t base;
  if (!f) return -1;
  switch (mode) {
  case MSPACK_SYS_SEEK_START: base = 0; break;
  case MSPACK_SYS_SEEK_CUR:   base = (off_t) f->pos; break;
  case MSPACK_SYS_SEEK_END:   base = (off_t) f->mn->len; break;
  default: return -1;
  }
  if (base + offset < 0 || (size_t)(base + offset) > f->mn->len) return -1;
  f->pos = (size_t)(base + offset);
  return 0;
}
static off_t m_tell(struct mspack_file *file) {
  struct memfile *f = (struct memfile *) file;
  return f ? (off_t) f->pos : 0;
}
/* what the library says through sys->message, formatted, one line each (driver tests compare our drivers' log with it;
 * single-threaded use only -- the timing entry points never produce messages) */
static char g_msgs[1 << 16];
static size_t g_msgs_len;
static void msg_append(const char *line) {
  size_t n = strlen(line);
  if (g_msgs_len + n + 2 > sizeof(g_msgs)) return;
  memcpy(g_msgs + g_msgs_len, line, n); g_msgs_len += n;
  g_msgs[g_msgs_len++] = '\n'; g_msgs[g_msgs_len] = 0;
}
/* ... and, per line, whether it came with a file handle ('H') or with NULL ('-'); the harness's own "#extract" marks are '#' */
static char g_msg_handles[1 << 12];
static size_t g_msg_handles_len;
static void handle_append(char c) { if (g_msg_handles_len + 2 < sizeof(g_msg_handles)) { g_msg_handles[g_msg_handles_len++] = c; g_msg_handles[g_msg_handles_len] = 0; } }
size_t refh_message_handles(char *buf, size_t cap) {
  size_t n = g_msg_handles_len < cap ? g_msg_handles_len : (cap ? cap - 1 : 0);
  if (cap) { memcpy(buf, g_msg_handles, n); buf[n] = 0; }
  g_msg_handles_len = 0; g_msg_handles[0] = 0;
  return n;
}
static void m_msg(struct mspack_file *file, const char *format, ...) {
  char line[512];
  va_list ap;
  handle_append(file ? 'H' : '-');
  va_start(ap, format);
  vsnprintf(line, sizeof(line), format, ap);
  va_end(ap);
  msg_append(line);
}
/* copy the log out (NUL-terminated, at most cap - 1 bytes) and clear it; returns its length */
size_t refh_messages(char *buf, size_t cap) {
  size_t n = g_msgs_len < cap ? g_msgs_len : (cap ? cap - 1 : 0);
  if (cap) { memcpy(buf, g_msgs, n); buf[n] = 0; }
  g_msgs_len = 0; g_msgs[0] = 0;
  return n;
}
/* Allocation: lzxd_init mallocs a fresh 2 MiB window per stream; at one stream per 64 KiB unit that
 * is an mmap/munmap pair per unit and, with hundreds of threads, mostly kernel time.  To give the CPU
 * baseline its best showing the harness recycles blocks per thread (a size-keyed free list behind
 * the mspack_system alloc/free hooks -- the reference code itself is untouched). */
#define CACHE_SLOTS 8
struct blk_hdr { size_t size; size_t pad; };
static __thread struct blk_hdr *blk_cache[CACHE_SLOTS];
static int zero_alloc;            /* tests: hand out zeroed blocks, so that reads of window bytes the
                                   * reference never wrote become comparable (oracle windows are zeroed) */
void refh_zero_alloc(int on) { zero_alloc = on; }
static void *m_alloc(struct mspack_system *self, size_t bytes) {
  struct blk_hdr *h;
  int i;
  (void) self;
  for (i = 0; i < CACHE_SLOTS; i++)
    if (blk_cache[i] && blk_cache[i]->size == bytes) {
      h = blk_cache[i]; blk_cache[i] = NULL;
      if (zero_alloc) memset(h + 1, 0, bytes);
      return h + 1;
    }
  h = (struct blk_hdr *) malloc(sizeof(*h) + bytes);
  if (!h) return NULL;
  h->size = bytes;
  if (zero_alloc) memset(h + 1, 0, bytes);
  return h + 1;
}
static void m_free(void *p) {
  struct blk_hdr *h;
  int i;
  if (!p) return;
  h = (struct blk_hdr *) p - 1;
  for (i = 0; i < CACHE_SLOTS; i++) if (!blk_cache[i]) { blk_cache[i] = h; return; }
  free(h);
}
static void m_copy(void *src, void *dest, size_t bytes) { memcpy(dest, src, bytes); }

static struct mspack_system mem_system = {
  &m_open, &m_close, &m_read, &m_write, &m_seek, &m_tell, &m_msg, &m_alloc, &m_free, &m_copy, NULL
};

/* ---- codec-level, memory to memory ------------------------------------------------------- */
/* All return the reference's MSPACK_ERR_* code; *written = bytes the codec wrote. */

int refh_lzx(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap, long long out_bytes,
             int window_bits, int reset_interval, long long output_length, size_t *written)
{
  struct memname src = { MEMNAME_MAGIC, (uint8_t *) in, in_len, 0 };
  struct memname dst = { MEMNAME_MAGIC, out, out_cap, 0 };
  struct mspack_file *fi = m_open(&mem_system, (const char *) &src, MSPACK_SYS_OPEN_READ);
  struct mspack_file *fo = m_open(&mem_system, (const char *) &dst, MSPACK_SYS_OPEN_WRITE);
  struct lzxd_stream *lzx = lzxd_init(&mem_system, fi, fo, window_bits, reset_interval, 4096,
                                      (off_t) output_length, 0);
  int err = MSPACK_ERR_ARGS;
  if (lzx) { err = lzxd_decompress(lzx, (off_t) out_bytes); lzxd_free(lzx); }
  if (written) *written = dst.written;
  m_close(fi); m_close(fo);
  return err;
}

/* LZX DELTA: lzxd_init(is_delta = 1) + lzxd_set_reference_data + lzxd_decompress (lzxd.c:274-382) */
int refh_lzxd(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap, long long out_bytes,
              int window_bits, int reset_interval, long long output_length,
              const uint8_t *ref, size_t ref_len, size_t *written)
{
  struct memname src = { MEMNAME_MAGIC, (uint8_t *) in, in_len, 0 };
  struct memname dst = { MEMNAME_MAGIC, out, out_cap, 0 };
  struct memname rsrc = { MEMNAME_MAGIC, (uint8_t *) ref, ref_len, 0 };
  struct mspack_file *fi = m_open(&mem_system, (const char *) &src, MSPACK_SYS_OPEN_READ);
  struct mspack_file *fo = m_open(&mem_system, (const char *) &dst, MSPACK_SYS_OPEN_WRITE);
  struct mspack_file *fr = m_open(&mem_system, (const char *) &rsrc, MSPACK_SYS_OPEN_READ);
  struct lzxd_stream *lzx = lzxd_init(&mem_system, fi, fo, window_bits, reset_interval, 4096,
                                      (off_t) output_length, 1);
  int err = MSPACK_ERR_ARGS;
  if (lzx) {
    err = ref_len ? lzxd_set_reference_data(lzx, &mem_system, fr, (unsigned int) ref_len) : MSPACK_ERR_OK;
    if (!err) err = lzxd_decompress(lzx, (off_t) out_bytes);
    lzxd_free(lzx);
  }
  if (written) *written = dst.written;
  m_close(fi); m_close(fo); m_close(fr);
  return err;
}

int refh_mszip(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap, long long out_bytes,
               int repair_mode, size_t *written)
{
  struct memname src = { MEMNAME_MAGIC, (uint8_t *) in, in_len, 0 };
  struct memname dst = { MEMNAME_MAGIC, out, out_cap, 0 };
  struct mspack_file *fi = m_open(&mem_system, (const char *) &src, MSPACK_SYS_OPEN_READ);
  struct mspack_file *fo = m_open(&mem_system, (const char *) &dst, MSPACK_SYS_OPEN_WRITE);
  struct mszipd_stream *zip = mszipd_init(&mem_system, fi, fo, repair_mode > 1 ? repair_mode : 4096, repair_mode);
  int err = MSPACK_ERR_ARGS;
  if (zip) { err = mszipd_decompress(zip, (off_t) out_bytes); mszipd_free(zip); }
  if (written) *written = dst.written;
  m_close(fi); m_close(fo);
  return err;
}

int refh_qtm(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap, long long out_bytes,
             int window_bits, size_t *written)
{
  struct memname src = { MEMNAME_MAGIC, (uint8_t *) in, in_len, 0 };
  struct memname dst = { MEMNAME_MAGIC, out, out_cap, 0 };
  struct mspack_file *fi = m_open(&mem_system, (const char *) &src, MSPACK_SYS_OPEN_READ);
  struct mspack_file *fo = m_open(&mem_system, (const char *) &dst, MSPACK_SYS_OPEN_WRITE);
  struct qtmd_stream *qtm = qtmd_init(&mem_system, fi, fo, window_bits, 4096);
  int err = MSPACK_ERR_ARGS;
  if (qtm) { err = qtmd_decompress(qtm, (off_t) out_bytes); qtmd_free(qtm); }
  if (written) *written = dst.written;
  m_close(fi); m_close(fo);
  return err;
}

/* what the real qtmd holds back when a request of `out_bytes` returns: o_end - o_ptr (qtmd.c:268-276 hands it to the next call's
 * output first) -- read from the stream's own state after ONE qtmd_decompress(out_bytes) */
int refh_qtm_carry(const uint8_t *in, size_t in_len, long long out_bytes, int window_bits, long long *carry)
{
  struct memname src = { MEMNAME_MAGIC, (uint8_t *) in, in_len, 0 };
  struct memname dst = { MEMNAME_MAGIC, NULL, 0, 0 };
  struct mspack_file *fi = m_open(&mem_system, (const char *) &src, MSPACK_SYS_OPEN_READ);
  struct mspack_file *fo = m_open(&mem_system, (const char *) &dst, MSPACK_SYS_OPEN_WRITE);
  struct qtmd_stream *qtm = qtmd_init(&mem_system, fi, fo, window_bits, 4096);
  int err = MSPACK_ERR_ARGS;
  *carry = -1;
  if (qtm) { err = qtmd_decompress(qtm, (off_t) out_bytes); *carry = (long long)(qtm->o_end - qtm->o_ptr); qtmd_free(qtm); }
  m_close(fi); m_close(fo);
  return err;
}

/* ---- container-level: CAB / CHM held in memory --------------------------------------------- */
/* Enumerate files: fills arrays (up to cap) and returns the number of files, or -err. */
int refh_cab_list(const uint8_t *cab, size_t cab_len, int cap,
                  unsigned *lengths, unsigned *offsets, int *comp_types, int *folder_ids,
                  char *names, int name_stride)
{
  struct memname src = { MEMNAME_MAGIC, (uint8_t *) cab, cab_len, 0 };
  struct mscab_decompressor *d = mspack_create_cab_decompressor(&mem_system);
  struct mscabd_cabinet *c;
  struct mscabd_file *f;
  struct mscabd_folder *fol;
  int n = 0;
  if (!d) return -MSPACK_ERR_NOMEMORY;
  c = d->open(d, (const char *) &src);
  if (!c) { n = -d->last_error(d); mspack_destroy_cab_decompressor(d); return n; }
  for (f = c->files; f; f = f->next, n++) {
    if (n < cap) {
      int fid = 0;
      for (fol = c->folders; fol && fol != f->folder; fol = fol->next) fid++;
      if (lengths) lengths[n] = f->length;
      if (offsets) offsets[n] = f->offset;
      if (comp_types) comp_types[n] = f->folder ? f->folder->comp_type : -1;
      if (folder_ids) folder_ids[n] = fid;
      if (names) { strncpy(names + (size_t) n * name_stride, f->filename, name_stride - 1);
                   names[(size_t) n * name_stride + name_stride - 1] = 0; }
    }
  }
  d->close(d, c);
  mspack_destroy_cab_decompressor(d);
  return n;
}

/* Extract the files listed in order[0..n_order) (indices into the file list) one after another
 * with ONE decompressor (so folder-state reuse behaves as in the reference).  errs[i] = code of
 * extract i; outputs are concatenated into out (each file at out_offs[i]). */
int refh_cab_extract(const uint8_t *cab, size_t cab_len, const int *order, int n_order,
                     uint8_t *out, size_t out_cap, size_t *out_offs, size_t *out_lens, int *errs,
                     int fix_mszip, int salvage)
{
  struct memname src = { MEMNAME_MAGIC, (uint8_t *) cab, cab_len, 0 };
  struct mscab_decompressor *d = mspack_create_cab_decompressor(&mem_system);
  struct mscabd_cabinet *c;
  size_t pos = 0;
  int i;
  if (!d) return MSPACK_ERR_NOMEMORY;
  d->set_param(d, MSCABD_PARAM_FIXMSZIP, fix_mszip);
  d->set_param(d, MSCABD_PARAM_SALVAGE, salvage);
  c = d->open(d, (const char *) &src);
  if (!c) { i = d->last_error(d); mspack_destroy_cab_decompressor(d); return i; }
  for (i = 0; i < n_order; i++) {
    struct mscabd_file *f = c->files;
    struct memname dst = { MEMNAME_MAGIC, out ? out + pos : NULL, out ? out_cap - pos : 0, 0 };
    int k = order[i];
    while (f && k-- > 0) f = f->next;
    if (!f) { errs[i] = MSPACK_ERR_ARGS; out_offs[i] = pos; out_lens[i] = 0; continue; }
    { char mark[32]; snprintf(mark, sizeof(mark), "#extract %d", i); msg_append(mark); handle_append('#'); }
    errs[i] = d->extract(d, f, (const char *) &dst);
    out_offs[i] = pos;
    out_lens[i] = dst.written;
    pos += dst.written < (out_cap - pos) ? dst.written : (out_cap - pos);
  }
  d->close(d, c);
  mspack_destroy_cab_decompressor(d);
  return MSPACK_ERR_OK;
}

/* ---- cabinet sets: open several cabinets, join them, list and extract (cabd.c:870-1064) ----------- */
/* ops = n_ops triples (op, a, b): op 0 = append(cab[a], cab[b]), 1 = prepend(cab[a], cab[b]); an index
 * of -1 passes NULL.  op_errs[i] = return code of op i.  Then the file list of cab[list_cab] is
 * reported (names, lengths, offsets, comp types, folder ordinals) and every listed file is extracted
 * in order with ONE decompressor.  Returns the number of files, or -err when a cabinet does not open. */
int refh_cabset(const uint8_t **cabs, const size_t *lens, int n_cabs, const int *ops, int n_ops, int *op_errs,
                int list_cab, int cap, unsigned *lengths, unsigned *offsets, int *comp_types, int *folder_ids,
                unsigned *folder_blocks, char *names, int name_stride,
                uint8_t *out, size_t out_cap, size_t *out_offs, size_t *out_lens, int *errs)
{
  struct memname src[16];
  struct mscabd_cabinet *c[16];
  struct mscab_decompressor *d = mspack_create_cab_decompressor(&mem_system);
  struct mscabd_file *f;
  struct mscabd_folder *fol;
  size_t pos = 0;
  int i, n = 0;
  if (!d || n_cabs > 16) return -MSPACK_ERR_ARGS;
  for (i = 0; i < n_cabs; i++) {
    src[i].magic = MEMNAME_MAGIC; src[i].data = (uint8_t *) cabs[i]; src[i].len = lens[i]; src[i].written = 0;
    c[i] = d->open(d, (const char *) &src[i]);
    if (!c[i]) { n = -d->last_error(d); mspack_destroy_cab_decompressor(d); return n ? n : -MSPACK_ERR_OPEN; }
  }
  for (i = 0; i < n_ops; i++) {
    struct mscabd_cabinet *a = ops[3 * i + 1] < 0 ? NULL : c[ops[3 * i + 1]];
    struct mscabd_cabinet *b = ops[3 * i + 2] < 0 ? NULL : c[ops[3 * i + 2]];
    op_errs[i] = ops[3 * i] ? d->prepend(d, a, b) : d->append(d, a, b);
  }
  for (f = c[list_cab]->files; f; f = f->next, n++) {
    if (n < cap) {
      int fid = 0;
      struct memname dst = { MEMNAME_MAGIC, out ? out + pos : NULL, out ? out_cap - pos : 0, 0 };
      for (fol = c[list_cab]->folders; fol && fol != f->folder; fol = fol->next) fid++;
      lengths[n] = f->length; offsets[n] = f->offset;
      comp_types[n] = f->folder ? f->folder->comp_type : -1;
      folder_ids[n] = fid;
      folder_blocks[n] = f->folder ? f->folder->num_blocks : 0;
      strncpy(names + (size_t) n * name_stride, f->filename, name_stride - 1);
      names[(size_t) n * name_stride + name_stride - 1] = 0;
      errs[n] = d->extract(d, f, (const char *) &dst);
      out_offs[n] = pos; out_lens[n] = dst.written;
      pos += dst.written < (out_cap - pos) ? dst.written : (out_cap - pos);
    }
  }
  /* cabinets that ended up joined are freed with the one they were joined to */
  for (i = 0; i < n_cabs; i++) {
    int j, dup = 0;
    struct mscabd_cabinet *w;
    if (!c[i]) continue;
    for (j = 0; j < i && !dup; j++) {
      if (!c[j]) continue;
      for (w = c[j]; w && !dup; w = w->prevcab) if (w == c[i]) dup = 1;
      for (w = c[j]; w && !dup; w = w->nextcab) if (w == c[i]) dup = 1;
    }
    if (dup) c[i] = NULL;
  }
  for (i = 0; i < n_cabs; i++) if (c[i]) d->close(d, c[i]);
  mspack_destroy_cab_decompressor(d);
  return n;
}

/* search(): base offsets, file counts and first file names of the cabinets found in a blob */
int refh_cab_search(const uint8_t *blob, size_t blob_len, int searchbuf, int cap, long long *base_offsets,
                    int *n_files, char *first_names, int name_stride)
{
  struct memname src = { MEMNAME_MAGIC, (uint8_t *) blob, blob_len, 0 };
  struct mscab_decompressor *d = mspack_create_cab_decompressor(&mem_system);
  struct mscabd_cabinet *c, *head;
  int n = 0;
  if (!d) return -MSPACK_ERR_NOMEMORY;
  if (searchbuf > 0) d->set_param(d, MSCABD_PARAM_SEARCHBUF, searchbuf);
  head = d->search(d, (const char *) &src);
  if (!head && d->last_error(d)) { n = -d->last_error(d); mspack_destroy_cab_decompressor(d); return n; }
  for (c = head; c; c = c->next, n++) {
    if (n < cap) {
      struct mscabd_file *f;
      int k = 0;
      for (f = c->files; f; f = f->next) k++;
      base_offsets[n] = (long long) c->base_offset; n_files[n] = k;
      strncpy(first_names + (size_t) n * name_stride, c->files ? c->files->filename : "", name_stride - 1);
      first_names[(size_t) n * name_stride + name_stride - 1] = 0;
    }
  }
  if (head) d->close(d, head);
  mspack_destroy_cab_decompressor(d);
  return n;
}

int refh_chm_list(const uint8_t *chm, size_t chm_len, int cap, long long *lengths,
                  long long *offsets, int *sections, char *names, int name_stride)
{
  struct memname src = { MEMNAME_MAGIC, (uint8_t *) chm, chm_len, 0 };
  struct mschm_decompressor *d = mspack_create_chm_decompressor(&mem_system);
  struct mschmd_header *h;
  struct mschmd_file *f;
  int n = 0;
  if (!d) return -MSPACK_ERR_NOMEMORY;
  h = d->open(d, (const char *) &src);
  if (!h) { n = -d->last_error(d); mspack_destroy_chm_decompressor(d); return n; }
  for (f = h->files; f; f = f->next, n++) {
    if (n < cap) {
      if (lengths) lengths[n] = (long long) f->length;
      if (offsets) offsets[n] = (long long) f->offset;
      if (sections) sections[n] = (int) f->section->id;
      if (names) { strncpy(names + (size_t) n * name_stride, f->filename, name_stride - 1);
                   names[(size_t) n * name_stride + name_stride - 1] = 0; }
    }
  }
  d->close(d, h);
  mspack_destroy_chm_decompressor(d);
  return n;
}

int refh_chm_extract(const uint8_t *chm, size_t chm_len, const int *order, int n_order,
                     uint8_t *out, size_t out_cap, size_t *out_offs, size_t *out_lens, int *errs)
{
  struct memname src = { MEMNAME_MAGIC, (uint8_t *) chm, chm_len, 0 };
  struct mschm_decompressor *d = mspack_create_chm_decompressor(&mem_system);
  struct mschmd_header *h;
  size_t pos = 0;
  int i;
  if (!d) return MSPACK_ERR_NOMEMORY;
  h = d->open(d, (const char *) &src);
  if (!h) { i = d->last_error(d); mspack_destroy_chm_decompressor(d); return i; }
  for (i = 0; i < n_order; i++) {
    struct mschmd_file *f = h->files;
    struct memname dst = { MEMNAME_MAGIC, out ? out + pos : NULL, out ? out_cap - pos : 0, 0 };
    int k = order[i];
    while (f && k-- > 0) f = f->next;
    if (!f) { errs[i] = MSPACK_ERR_ARGS; out_offs[i] = pos; out_lens[i] = 0; continue; }
    { char mark[32]; snprintf(mark, sizeof(mark), "#extract %d", i); msg_append(mark); handle_append('#'); }
    errs[i] = d->extract(d, f, (const char *) &dst);
    out_offs[i] = pos;
    out_lens[i] = dst.written;
    pos += dst.written < (out_cap - pos) ? dst.written : (out_cap - pos);
  }
  d->close(d, h);
  mspack_destroy_chm_decompressor(d);
  return MSPACK_ERR_OK;
}

/* fast_open + fast_find for a list of NUL-separated names: errs[i] = return code, and (section id or
 * -1 when nothing was found, offset, length) as the reference filled them in (chmd.c:543-640) */
int refh_chm_find(const uint8_t *chm, size_t chm_len, const char *names, int n_names,
                  int *errs, int *sections, long long *offsets, long long *lengths)
{
  struct memname src = { MEMNAME_MAGIC, (uint8_t *) chm, chm_len, 0 };
  struct mschm_decompressor *d = mspack_create_chm_decompressor(&mem_system);
  struct mschmd_header *h;
  int i;
  if (!d) return MSPACK_ERR_NOMEMORY;
  h = d->fast_open(d, (const char *) &src);
  if (!h) { i = d->last_error(d); mspack_destroy_chm_decompressor(d); return i ? i : MSPACK_ERR_OPEN; }
  for (i = 0; i < n_names; i++) {
    struct mschmd_file r;
    errs[i] = d->fast_find(d, h, names, &r, (int) sizeof(r));
    sections[i] = r.section ? (int) r.section->id : -1;
    offsets[i] = (long long) r.offset; lengths[i] = (long long) r.length;
    names += strlen(names) + 1;
  }
  d->close(d, h);
  mspack_destroy_chm_decompressor(d);
  return MSPACK_ERR_OK;
}

/* OAB files through the reference's msoab_decompressor (oabd.c:103-382); base == NULL: full file */
int refh_oab(const uint8_t *in, size_t in_len, const uint8_t *base, size_t base_len,
             uint8_t *out, size_t out_cap, size_t *written, int decompbuf)
{
  struct memname src = { MEMNAME_MAGIC, (uint8_t *) in, in_len, 0 };
  struct memname bsrc = { MEMNAME_MAGIC, (uint8_t *) base, base_len, 0 };
  struct memname dst = { MEMNAME_MAGIC, out, out_cap, 0 };
  struct msoab_decompressor *d = mspack_create_oab_decompressor(&mem_system);
  int err;
  if (!d) return MSPACK_ERR_NOMEMORY;
  if (decompbuf > 0) d->set_param(d, MSOABD_PARAM_DECOMPBUF, decompbuf);
  err = base ? d->decompress_incremental(d, (const char *) &src, (const char *) &bsrc, (const char *) &dst)
             : d->decompress(d, (const char *) &src, (const char *) &dst);
  if (written) *written = dst.written;
  mspack_destroy_oab_decompressor(d);
  return err;
}

/* SZDD and KWAJ files through the reference's decompressors (szddd.c, kwajd.c): decompress(input, output).
 * kind 0 = SZDD, 1 = KWAJ.  For KWAJ the header fields are reported too. */
int refh_szdd_kwaj(int kind, const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap, size_t *written,
                   int *comp_type, long long *length, char *filename /* >= 16 bytes */, int *open_err)
{
  struct memname src = { MEMNAME_MAGIC, (uint8_t *) in, in_len, 0 };
  struct memname dst = { MEMNAME_MAGIC, out, out_cap, 0 };
  int err;
  *open_err = 0; if (comp_type) *comp_type = -1; if (length) *length = -1; if (filename) filename[0] = 0;
  if (kind == 0) {
    struct msszdd_decompressor *d = mspack_create_szdd_decompressor(&mem_system);
    struct msszddd_header *h = d->open(d, (const char *) &src);
    if (!h) { *open_err = d->last_error(d); mspack_destroy_szdd_decompressor(d); return *open_err; }
    if (comp_type) *comp_type = h->format;
    if (length) *length = (long long) h->length;
    if (filename) { filename[0] = h->missing_char; filename[1] = 0; }
    err = d->extract(d, h, (const char *) &dst);
    d->close(d, h);
    mspack_destroy_szdd_decompressor(d);
  }
  else {
    struct mskwaj_decompressor *d = mspack_create_kwaj_decompressor(&mem_system);
    struct mskwajd_header *h = d->open(d, (const char *) &src);
    if (!h) { *open_err = d->last_error(d); mspack_destroy_kwaj_decompressor(d); return *open_err; }
    if (comp_type) *comp_type = h->comp_type;
    if (length) *length = (long long) h->length;
    if (filename && h->filename) { strncpy(filename, h->filename, 15); filename[15] = 0; }
    err = d->extract(d, h, (const char *) &dst);
    d->close(d, h);
    mspack_destroy_kwaj_decompressor(d);
  }
  if (written) *written = dst.written;
  return err;
}

/* ---- timing: the reference codec over a batch of independent units, T threads -------------- */
struct bench_job {
  int kind;                       /* 0 = LZX, 1 = MSZIP, 2 = Quantum */
  const uint8_t *in_base;
  const unsigned long long *in_off;
  const unsigned *in_len;
  const unsigned *out_len;
  int window_bits, reset_frames;
  int first, last;                /* unit range [first,last) */
  int reps;
  uint8_t *scratch;               /* >= max out_len */
  size_t scratch_cap;
  unsigned long long bytes_out;
  int errors;
  pthread_barrier_t *start, *stop;
};

static void *bench_worker(void *arg) {
  struct bench_job *j = (struct bench_job *) arg;
  int u, r;
  pthread_barrier_wait(j->start);            /* all threads exist and are warm before the clock starts */
  for (r = 0; r < j->reps; r++) {
    for (u = j->first; u < j->last; u++) {
      size_t w = 0; int err;
      const uint8_t *in = j->in_base + j->in_off[u];
      if (j->kind == 0)
        err = refh_lzx(in, j->in_len[u], j->scratch, j->scratch_cap, j->out_len[u], j->window_bits,
                       j->reset_frames, j->out_len[u], &w);
      else if (j->kind == 1)
        err = refh_mszip(in, j->in_len[u], j->scratch, j->scratch_cap, j->out_len[u], 0, &w);
      else
        err = refh_qtm(in, j->in_len[u], j->scratch, j->scratch_cap, j->out_len[u], j->window_bits, &w);
      if (err != MSPACK_ERR_OK || w != j->out_len[u]) j->errors++;
      j->bytes_out += w;
    }
  }
  pthread_barrier_wait(j->stop);
  return NULL;
}

/* Every thread owns one decompressor at a time (libmspack's threading contract, mspack.h:122-156) and
 * decodes its slice of the units `reps` times.  The clock runs from the moment all threads are
 * released until all are done.  Returns wall seconds; *bytes_out = total decoded bytes. */
double refh_bench(int kind, const uint8_t *in_base, const unsigned long long *in_off,
                  const unsigned *in_len, const unsigned *out_len, int n_units,
                  int window_bits, int reset_frames, int n_threads, int reps,
                  unsigned long long *bytes_out, int *errors)
{
  pthread_t *th;
  struct bench_job *jobs;
  pthread_barrier_t start, stop;
  struct timespec t0, t1;
  size_t cap = 0;
  int t, u;
  if (n_threads < 1) n_threads = 1;
  if (n_threads > n_units) n_threads = n_units;
  if (reps < 1) reps = 1;
  for (u = 0; u < n_units; u++) if (out_len[u] > cap) cap = out_len[u];
  th = (pthread_t *) calloc((size_t) n_threads, sizeof(*th));
  jobs = (struct bench_job *) calloc((size_t) n_threads, sizeof(*jobs));
  pthread_barrier_init(&start, NULL, (unsigned) n_threads + 1);
  pthread_barrier_init(&stop, NULL, (unsigned) n_threads + 1);
  for (t = 0; t < n_threads; t++) {
    jobs[t].kind = kind; jobs[t].in_base = in_base; jobs[t].in_off = in_off;
    jobs[t].in_len = in_len; jobs[t].out_len = out_len;
    jobs[t].window_bits = window_bits; jobs[t].reset_frames = reset_frames; jobs[t].reps = reps;
    jobs[t].first = (int)((long long) n_units * t / n_threads);
    jobs[t].last  = (int)((long long) n_units * (t + 1) / n_threads);
    jobs[t].scratch = (uint8_t *) malloc(cap + 64); jobs[t].scratch_cap = cap;
    jobs[t].start = &start; jobs[t].stop = &stop;
    pthread_create(&th[t], NULL, bench_worker, &jobs[t]);
  }
  pthread_barrier_wait(&start);
  clock_gettime(CLOCK_MONOTONIC, &t0);
  pthread_barrier_wait(&stop);
  clock_gettime(CLOCK_MONOTONIC, &t1);
  for (t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
  *bytes_out = 0; *errors = 0;
  for (t = 0; t < n_threads; t++) {
    *bytes_out += jobs[t].bytes_out; *errors += jobs[t].errors; free(jobs[t].scratch);
  }
  pthread_barrier_destroy(&start); pthread_barrier_destroy(&stop);
  free(th); free(jobs);
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

const char *refh_version(void) { return "libmspack reference (oracle/_ref), built from /root/reference"; }
