/* chmd.c -- CHM driver of the libmspack-compatible API (include/mspack.h), on the GPU batch decoder.
 *
 * Mirrors the behaviour of the reference's chmd.c for open / fast_open / fast_find / extract:
 *   ITSF + header sections + PMGL directory parsing and its error codes .. chmd.c:254-532
 *   section-0 (stored) files: seek + copy ................................. chmd.c:958-987
 *   ControlData / ResetTable / SpanInfo -> window, reset interval, length .. chmd.c:1072-1315
 *   lzxd lifetime: a decoder lives from the reset point of the first file it was asked for until an
 *   error or a backwards request (chmd.c:989-1040); errors of any frame it has to cross are reported
 * but NOT its control flow: every LZX reset interval (chmd.c:1147-1149) is an independent unit, so
 * the first section-1 extract() decodes ALL intervals of the CHM in one GPU batch and later
 * extract() calls are slices of that result.  The reference's decoder lifetime is emulated on top
 * (virtual start / position), because it decides (a) which frames' errors a request sees and (b) the
 * origin of the E8 translation (lzxd.c:712 uses bytes since lzxd_init): intervals whose header asks
 * for E8 are re-decoded with the origin the reference would have had.
 * fast_find follows the reference: PMGI index descent, quick-reference binary
 * search inside a chunk, UTF-8 case-folded name order (chmd.c:543-898).
 */
#include <stdlib.h>
#include <stdio.h>
#include "host_common.h"

#define FRAME 32768

static const char content_name[]  = "::DataSpace/Storage/MSCompressed/Content";
static const char control_name[]  = "::DataSpace/Storage/MSCompressed/ControlData";
static const char spaninfo_name[] = "::DataSpace/Storage/MSCompressed/SpanInfo";
static const char rtable_name[]   = "::DataSpace/Storage/MSCompressed/Transform/"
                                    "{7FC28940-9D31-11D0-9B27-00A0C91E9C7C}/InstanceData/ResetTable";

struct chm_p {
  struct mschmd_header base;
  /* decoded section 1 */
  int sec1_state;                 /* 0 = not set up, 1 = ready, <0 = -(setup error)                 */
  int window_bits, use_table;
  off_t interval_bytes, padded_len, stream_len;
  unsigned int n_intervals;
  unsigned char *arena; size_t arena_len;     /* the Content stream (+ a few following bytes)      */
  uint64_t *ioff;                 /* compressed offset of every interval                            */
  unsigned char *dec;             /* padded_len decoded bytes, E8 origin = each interval's own start */
  mspack_hip_result *ires;        /* per interval                                                   */
};
struct chmd_p {
  struct mschm_decompressor base;
  struct mspack_system *system;
  int error;
  /* emulated lzxd lifetime (chmd.c:989-1040) */
  struct chm_p *v_chm; off_t v_init, v_offset; int v_alive;
};

static off_t read_encint(const unsigned char **p, const unsigned char *end, int *err) {
  off_t v = 0; unsigned char c = 0x80; int i = 0;
  while ((c & 0x80) && (i++ < 9)) {
    if (*p >= end) { *err = 1; return 0; }
    c = *(*p)++;
    v = (v << 7) | (c & 0x7F);
  }
  if (i == 9 && (c & 0x80)) { *err = 1; return 0; }
  return v;
}

/* ---- headers (reference chmd.c:254-532) ------------------------------------------------------------- */
static const unsigned char itsf_guids[32] = {
  0x10, 0xFD, 0x01, 0x7C, 0xAA, 0x7B, 0xD0, 0x11, 0x9E, 0x0C, 0x00, 0xA0, 0xC9, 0x22, 0xE6, 0xEC,
  0x11, 0xFD, 0x01, 0x7C, 0xAA, 0x7B, 0xD0, 0x11, 0x9E, 0x0C, 0x00, 0xA0, 0xC9, 0x22, 0xE6, 0xEC
};

static int read_headers(struct mspack_system *sys, struct mspack_file *fh, struct mschmd_header *chm, int entire)
{
  unsigned char buf[0x54], *chunk;
  struct mschmd_file *tail = NULL;
  off_t off_hs0, filelen;
  unsigned int n, errors = 0;

  chm->files = NULL; chm->sysfiles = NULL; chm->chunk_cache = NULL;
  chm->sec0.base.chm = chm; chm->sec0.base.id = 0;
  chm->sec1.base.chm = chm; chm->sec1.base.id = 1;
  chm->sec1.content = chm->sec1.control = chm->sec1.spaninfo = chm->sec1.rtable = NULL;

  if (sys->read(fh, buf, 0x38) != 0x38) return MSPACK_ERR_READ;
  if (rd_le32(buf) != 0x46535449u) return MSPACK_ERR_SIGNATURE;
  if (memcmp(buf + 0x18, itsf_guids, 32) != 0) return MSPACK_ERR_SIGNATURE;
  chm->version = rd_le32(buf + 4);
  chm->timestamp = rd_be32(buf + 0x10);
  chm->language = rd_le32(buf + 0x14);
  if (chm->version > 3) sys->message(fh, "WARNING; CHM version > 3");
  if (sys->read(fh, buf, 0x28) != 0x28) return MSPACK_ERR_READ;
  off_hs0 = (off_t) rd_le64(buf);
  chm->dir_offset = (off_t) rd_le64(buf + 0x10);
  chm->sec0.offset = (off_t) rd_le64(buf + 0x20);
  if (sys->seek(fh, off_hs0, MSPACK_SYS_SEEK_START)) return MSPACK_ERR_SEEK;
  if (sys->read(fh, buf, 0x18) != 0x18) return MSPACK_ERR_READ;
  chm->length = (off_t) rd_le64(buf + 8);
  if (!mspack_sys_filelen(sys, fh, &filelen)) {
    if (chm->length > filelen) sys->message(fh, "WARNING; file possibly truncated by %lld bytes", (long long)(chm->length - filelen));
    else if (chm->length < filelen) sys->message(fh, "WARNING; possible %lld extra bytes at end of file", (long long)(filelen - chm->length));
  }
  if (sys->seek(fh, chm->dir_offset, MSPACK_SYS_SEEK_START)) return MSPACK_ERR_SEEK;
  if (sys->read(fh, buf, 0x54) != 0x54) return MSPACK_ERR_READ;
  chm->dir_offset = sys->tell(fh);
  chm->chunk_size = rd_le32(buf + 0x10);
  chm->density = rd_le32(buf + 0x14);
  chm->depth = rd_le32(buf + 0x18);
  chm->index_root = rd_le32(buf + 0x1C);
  chm->first_pmgl = rd_le32(buf + 0x20);
  chm->last_pmgl = rd_le32(buf + 0x24);
  chm->num_chunks = rd_le32(buf + 0x2C);
  if (chm->version < 3) chm->sec0.offset = chm->dir_offset + ((off_t) chm->chunk_size * chm->num_chunks);
  if (chm->sec0.offset > chm->length) return MSPACK_ERR_DATAFORMAT;
  if (chm->chunk_size < 0x14 + 2) return MSPACK_ERR_DATAFORMAT;
  if (chm->num_chunks == 0) return MSPACK_ERR_DATAFORMAT;
  if (chm->num_chunks > 100000) return MSPACK_ERR_DATAFORMAT;
  if (chm->chunk_size > 8192) return MSPACK_ERR_DATAFORMAT;
  if ((off_t) chm->chunk_size * (off_t) chm->num_chunks > chm->length) return MSPACK_ERR_DATAFORMAT;
  if (chm->chunk_size != 4096) sys->message(fh, "WARNING; chunk size is not 4096");
  if (chm->first_pmgl != 0) sys->message(fh, "WARNING; first PMGL chunk is not zero");
  if (chm->first_pmgl > chm->last_pmgl) return MSPACK_ERR_DATAFORMAT;
  if (chm->index_root != 0xFFFFFFFFu && chm->index_root >= chm->num_chunks) return MSPACK_ERR_DATAFORMAT;
  if (!entire) return MSPACK_ERR_OK;

  if (chm->first_pmgl != 0 && sys->seek(fh, (off_t) chm->first_pmgl * (off_t) chm->chunk_size, MSPACK_SYS_SEEK_CUR))
    return MSPACK_ERR_SEEK;
  n = chm->last_pmgl - chm->first_pmgl + 1;
  if (!(chunk = (unsigned char *) sys->alloc(sys, chm->chunk_size))) return MSPACK_ERR_NOMEMORY;
  while (n--) {
    const unsigned char *p, *end;
    int entries, err = 0;
    if (sys->read(fh, chunk, (int) chm->chunk_size) != (int) chm->chunk_size) { sys->free(chunk); return MSPACK_ERR_READ; }
    if (rd_le32(chunk) != 0x4C474D50u) continue;                          /* PMGL only */
    if (rd_le32(chunk + 4) < 2) sys->message(fh, "WARNING; PMGL quickref area is too small");
    if (rd_le32(chunk + 4) > chm->chunk_size - 0x14) sys->message(fh, "WARNING; PMGL quickref area is too large");
    p = chunk + 0x14; end = chunk + chm->chunk_size - 2;
    entries = (int) rd_le16(end);
    while (entries--) {
      unsigned int name_len, section;
      const unsigned char *name;
      off_t offset, length;
      struct mschmd_file *fi;
      name_len = (unsigned int) read_encint(&p, end, &err);
      if (err || name_len > (unsigned int)(end - p)) break;
      name = p; p += name_len;
      section = (unsigned int) read_encint(&p, end, &err);
      offset = read_encint(&p, end, &err);
      length = read_encint(&p, end, &err);
      if (err) break;
      if (name_len < 2 || !name[0] || !name[1]) continue;
      if (offset == 0 && length == 0 && name[name_len - 1] == '/') continue;
      if (section > 1) { sys->message(fh, "invalid section number '%u'.", section); continue; }
      if (!(fi = (struct mschmd_file *) sys->alloc(sys, sizeof(*fi) + name_len + 1))) { sys->free(chunk); return MSPACK_ERR_NOMEMORY; }
      fi->next = NULL;
      fi->filename = (char *) &fi[1];
      fi->section = section ? (struct mschmd_section *) &chm->sec1 : (struct mschmd_section *) &chm->sec0;
      fi->offset = offset; fi->length = length;
      sys->copy((void *) name, fi->filename, name_len);
      fi->filename[name_len] = 0;
      if (name[0] == ':' && name[1] == ':') {
        if (name_len == 40 && !memcmp(name, content_name, 40)) chm->sec1.content = fi;
        else if (name_len == 44 && !memcmp(name, control_name, 44)) chm->sec1.control = fi;
        else if (name_len == 41 && !memcmp(name, spaninfo_name, 41)) chm->sec1.spaninfo = fi;
        else if (name_len == 105 && !memcmp(name, rtable_name, 105)) chm->sec1.rtable = fi;
        fi->next = chm->sysfiles; chm->sysfiles = fi;
      }
      else { if (tail) tail->next = fi; else chm->files = fi; tail = fi; }
    }
    if (entries >= 0) errors++;
  }
  sys->free(chunk);
  return errors ? MSPACK_ERR_DATAFORMAT : MSPACK_ERR_OK;
}

static void free_sec1(struct mspack_system *sys, struct chm_p *c) {
  sys->free(c->arena); sys->free(c->ioff); sys->free(c->dec); sys->free(c->ires);
  c->arena = NULL; c->ioff = NULL; c->dec = NULL; c->ires = NULL; c->sec1_state = 0;
}

static void chmd_close(struct mschm_decompressor *base, struct mschmd_header *chm)
{
  struct chmd_p *self = (struct chmd_p *) base;
  struct mspack_system *sys;
  struct mschmd_file *fi, *nfi;
  unsigned int i;
  if (!self) return;
  sys = self->system;
  self->error = MSPACK_ERR_OK;
  if (!chm) return;
  for (fi = chm->files; fi; fi = nfi) { nfi = fi->next; sys->free(fi); }
  for (fi = chm->sysfiles; fi; fi = nfi) { nfi = fi->next; sys->free(fi); }
  if (self->v_chm == (struct chm_p *) chm) { self->v_chm = NULL; self->v_alive = 0; }
  if (chm->chunk_cache) {
    for (i = 0; i < chm->num_chunks; i++) sys->free(chm->chunk_cache[i]);
    sys->free(chm->chunk_cache);
  }
  free_sec1(sys, (struct chm_p *) chm);
  sys->free(chm);
}

static struct mschmd_header *open_common(struct mschm_decompressor *base, const char *filename, int entire)
{
  struct chmd_p *self = (struct chmd_p *) base;
  struct mspack_system *sys;
  struct mspack_file *fh;
  struct chm_p *c = NULL;
  if (!self) return NULL;
  sys = self->system;
  if (!(fh = sys->open(sys, filename, MSPACK_SYS_OPEN_READ))) { self->error = MSPACK_ERR_OPEN; return NULL; }
  if ((c = (struct chm_p *) sys->alloc(sys, sizeof(*c)))) {
    int err;
    memset(c, 0, sizeof(*c));
    c->base.filename = filename;
    err = read_headers(sys, fh, &c->base, entire);
    if (err) {
      /* like the reference: a badly encoded directory that yielded SOME entries is returned with a
       * warning and no error (chmd.c:166-176) */
      if (err == MSPACK_ERR_DATAFORMAT && (c->base.files || c->base.sysfiles)) {
        sys->message(fh, "WARNING; contents are corrupt");
        err = MSPACK_ERR_OK;
      }
      else { chmd_close(base, &c->base); c = NULL; }
    }
    self->error = err;
  }
  else self->error = MSPACK_ERR_NOMEMORY;
  sys->close(fh);
  return (struct mschmd_header *) c;
}
static struct mschmd_header *chmd_open(struct mschm_decompressor *b, const char *f) { return open_common(b, f, 1); }
static struct mschmd_header *chmd_fast_open(struct mschm_decompressor *b, const char *f) { return open_common(b, f, 0); }

/* ---- fast_find (reference chmd.c:543-898) --------------------------------------------------------------
 * Descends the PMGI index (when the header names an index root) to the PMGL chunk that can hold the
 * name, or walks the PMGL chain otherwise.  Inside a chunk: binary search over the quick-reference
 * entries (one per 1 + 2^density directory entries), then a linear scan of that group.  Names compare
 * as UTF-8 code points, case-insensitively through towlower(), lengths breaking ties -- the same
 * order the directory is sorted in.  Chunks are cached in chm->chunk_cache. */
#include <wctype.h>

/* one UTF-8 character; never reads past e, does not check continuation bytes, lets some overlong
 * forms through (chmd.c:861-880) */
static int utf8_next(const unsigned char **s, const unsigned char *e) {
  const unsigned char *p = *s;
  unsigned int x = *p++;
  int c;
  if (x < 0x80) c = (int) x;
  else if (x >= 0xC2 && x < 0xE0 && p < e) { c = (int)((x & 0x1F) << 6 | (p[0] & 0x3F)); p += 1; }
  else if (x >= 0xE0 && x < 0xF0 && p + 1 < e) { c = (int)((x & 0x0F) << 12 | (p[0] & 0x3F) << 6 | (p[1] & 0x3F)); p += 2; }
  else if (x >= 0xF0 && x <= 0xF5 && p + 2 < e) {
    c = (int)((x & 0x07) << 18 | (p[0] & 0x3F) << 12 | (p[1] & 0x3F) << 6 | (p[2] & 0x3F));
    if (c > 0x10FFFF) c = 0xFFFD;
    p += 3;
  }
  else c = 0xFFFD;
  *s = p;
  return c;
}

static int name_compare(const char *s1, const char *s2, int l1, int l2) {
  const unsigned char *p1 = (const unsigned char *) s1, *p2 = (const unsigned char *) s2;
  const unsigned char *e1 = p1 + l1, *e2 = p2 + l2;
  while (p1 < e1 && p2 < e2) {
    int c1 = utf8_next(&p1, e1), c2 = utf8_next(&p2, e2);
    if (c1 == c2) continue;
    c1 = (int) towlower((wint_t) c1); c2 = (int) towlower((wint_t) c2);
    if (c1 != c2) return c1 - c2;
  }
  return l1 - l2;
}

static unsigned char *read_chunk(struct chmd_p *self, struct mschmd_header *chm, struct mspack_file *fh, unsigned int n)
{
  struct mspack_system *sys = self->system;
  unsigned char *buf;
  if (n >= chm->num_chunks) return NULL;
  if (!chm->chunk_cache) {
    size_t size = sizeof(unsigned char *) * chm->num_chunks;
    if (!(chm->chunk_cache = (unsigned char **) sys->alloc(sys, size))) { self->error = MSPACK_ERR_NOMEMORY; return NULL; }
    memset(chm->chunk_cache, 0, size);
  }
  if (chm->chunk_cache[n]) return chm->chunk_cache[n];
  if (!(buf = (unsigned char *) sys->alloc(sys, chm->chunk_size))) { self->error = MSPACK_ERR_NOMEMORY; return NULL; }
  if (sys->seek(fh, chm->dir_offset + (off_t) n * (off_t) chm->chunk_size, MSPACK_SYS_SEEK_START)) {
    self->error = MSPACK_ERR_SEEK; sys->free(buf); return NULL;
  }
  if (sys->read(fh, buf, (int) chm->chunk_size) != (int) chm->chunk_size) {
    self->error = MSPACK_ERR_READ; sys->free(buf); return NULL;
  }
  if (!(buf[0] == 'P' && buf[1] == 'M' && buf[2] == 'G' && (buf[3] == 'L' || buf[3] == 'I'))) {
    self->error = MSPACK_ERR_SEEK; sys->free(buf); return NULL;      /* the reference's code for it */
  }
  return chm->chunk_cache[n] = buf;
}

static void skip_encint(const unsigned char **p, const unsigned char *end) {
  while (*p < end && (*(*p)++ & 0x80)) ;
}

/* -1 = malformed chunk, 0 = the name is not here, 1 = found: *res points at the entry's data (PMGL:
 * section, offset, length; PMGI: the child chunk number of the last entry not above the name) */
static int search_chunk(struct mschmd_header *chm, const unsigned char *chunk, const char *filename,
                        const unsigned char **res, const unsigned char **res_end)
{
  const int is_pmgl = (chunk[3] == 'L');
  const unsigned int entries_off = is_pmgl ? 0x14u : 8u;
  const unsigned int fname_len = (unsigned int) strlen(filename);
  const unsigned int qr_size = rd_le32(chunk + 4);
  const unsigned char *start = chunk + chm->chunk_size - 2;      /* entry count; quick refs grow down from here */
  const unsigned char *end = chunk + chm->chunk_size - qr_size;
  unsigned int num_entries = rd_le16(start), qr_density = 1u + (1u << chm->density), qr_entries, name_len;
  const unsigned char *p;
  int cmp = 0, err = 0;

  qr_entries = (num_entries + qr_density - 1) / qr_density;
  if (num_entries == 0) return -1;
  if (qr_size > chm->chunk_size) return -1;
  *res_end = end;
  if ((int) qr_entries * 2 > (int)(start - end)) qr_entries = 0;  /* more quick refs than room: do without */

  if (qr_entries > 0) {
    unsigned int L = 0, R = qr_entries - 1, M;
    do {
      M = (L + R) >> 1;
      p = chunk + entries_off + (M ? rd_le16(start - (M << 1)) : 0);
      name_len = (unsigned int) read_encint(&p, end, &err);
      if (err || name_len > (unsigned int)(end - p)) return -1;
      cmp = name_compare(filename, (const char *) p, (int) fname_len, (int) name_len);
      if (cmp == 0) break;
      else if (cmp < 0) { if (M) R = M - 1; else return 0; }
      else L = M + 1;
    } while (L <= R);
    M = (L + R) >> 1;
    if (cmp == 0) { *res = p + name_len; return 1; }
    p = chunk + entries_off + (M ? rd_le16(start - (M << 1)) : 0);
    num_entries -= M * qr_density;
    if (num_entries > qr_density) num_entries = qr_density;
  }
  else p = chunk + entries_off;

  *res = NULL;
  while (num_entries-- > 0) {
    name_len = (unsigned int) read_encint(&p, end, &err);
    if (err || name_len > (unsigned int)(end - p)) return -1;
    cmp = name_compare(filename, (const char *) p, (int) fname_len, (int) name_len);
    p += name_len;
    if (cmp == 0) { *res = p; return 1; }
    if (cmp < 0) break;
    if (is_pmgl) { skip_encint(&p, end); skip_encint(&p, end); skip_encint(&p, end); }
    else { *res = p; skip_encint(&p, end); }
  }
  return is_pmgl ? 0 : (*res ? 1 : 0);
}

static int chmd_fast_find(struct mschm_decompressor *base, struct mschmd_header *chm, const char *filename,
                          struct mschmd_file *f_ptr, int f_size)
{
  struct chmd_p *self = (struct chmd_p *) base;
  struct mspack_system *sys;
  struct mspack_file *fh;
  const unsigned char *chunk, *p = NULL, *end = NULL;
  int err = MSPACK_ERR_OK, result = -1, e = 0;
  unsigned int n;
  if (!self || !chm || !f_ptr || f_size != (int) sizeof(struct mschmd_file)) return MSPACK_ERR_ARGS;
  sys = self->system;
  memset(f_ptr, 0, sizeof(*f_ptr));
  if (!(fh = sys->open(sys, chm->filename, MSPACK_SYS_OPEN_READ))) return MSPACK_ERR_OPEN;

  if (chm->index_root < chm->num_chunks) {
    n = chm->index_root;
    for (;;) {
      if (!(chunk = read_chunk(self, chm, fh, n))) { sys->close(fh); return self->error; }
      if ((result = search_chunk(chm, chunk, filename, &p, &end)) <= 0) break;
      if (chunk[3] == 'L') break;
      n = (unsigned int) read_encint(&p, end, &e);
      if (e) { sys->close(fh); return self->error = MSPACK_ERR_DATAFORMAT; }
    }
  }
  else {
    for (n = chm->first_pmgl; n <= chm->last_pmgl; n = rd_le32(chunk + 0x10)) {
      if (!(chunk = read_chunk(self, chm, fh, n))) { err = self->error; break; }
      if ((result = search_chunk(chm, chunk, filename, &p, &end)) > 0) break;
      if (n == rd_le32(chunk + 0x10)) break;              /* a chunk that names itself as its successor */
    }
  }
  if (result > 0) {
    unsigned int sec = (unsigned int) read_encint(&p, end, &e);
    f_ptr->section = sec == 0 ? (struct mschmd_section *) &chm->sec0 : (struct mschmd_section *) &chm->sec1;
    f_ptr->offset = read_encint(&p, end, &e);
    f_ptr->length = read_encint(&p, end, &e);
    if (e) { sys->close(fh); return self->error = MSPACK_ERR_DATAFORMAT; }
  }
  else if (result < 0) err = MSPACK_ERR_DATAFORMAT;
  sys->close(fh);
  return self->error = err;
}

static int find_sys_file(struct chmd_p *self, struct mschmd_sec_mscompressed *sec, struct mschmd_file **f_ptr, const char *name)
{
  struct mspack_system *sys = self->system;
  struct mschmd_file result;
  if (*f_ptr) return MSPACK_ERR_OK;
  if (chmd_fast_find(&self->base, sec->base.chm, name, &result, (int) sizeof(result)) || !result.section)
    return MSPACK_ERR_DATAFORMAT;
  if (!(*f_ptr = (struct mschmd_file *) sys->alloc(sys, sizeof(result)))) return MSPACK_ERR_NOMEMORY;
  **f_ptr = result;
  (*f_ptr)->filename = (char *) name;
  (*f_ptr)->next = sec->base.chm->sysfiles;
  sec->base.chm->sysfiles = *f_ptr;
  return MSPACK_ERR_OK;
}

static unsigned char *read_sec0_file(struct chmd_p *self, struct mspack_file *fh, struct mschmd_file *file, int *err)
{
  struct mspack_system *sys = self->system;
  unsigned char *data;
  int len;
  if (!file || !file->section || file->section->id != 0) { *err = MSPACK_ERR_DATAFORMAT; return NULL; }
  len = (int) file->length;
  if (!(data = (unsigned char *) sys->alloc(sys, (size_t) len + 1))) { *err = MSPACK_ERR_NOMEMORY; return NULL; }
  if (sys->seek(fh, file->section->chm->sec0.offset + file->offset, MSPACK_SYS_SEEK_START)) { *err = MSPACK_ERR_SEEK; sys->free(data); return NULL; }
  if (sys->read(fh, data, len) != len) { *err = MSPACK_ERR_READ; sys->free(data); return NULL; }
  return data;
}

/* ---- section 1: set up the interval table and decode everything in one batch ------------------------- */
static int decode_intervals(struct chmd_p *self, struct chm_p *c, unsigned int first, unsigned int count,
                            int32_t e8_shift, unsigned char *out, mspack_hip_result *res)
{
  /* interval i is unit i-first; its E8 origin is shifted by e8_shift bytes (0 = its own start) */
  struct mspack_system *sys = self->system;
  mspack_hip_unit *units = (mspack_hip_unit *) sys->alloc(sys, (size_t) count * sizeof(*units));
  unsigned int k;
  int rc;
  if (!units) return MSPACK_ERR_NOMEMORY;
  memset(units, 0, (size_t) count * sizeof(*units));
  for (k = 0; k < count; k++) {
    uint64_t off = c->ioff[first + k];
    units[k].in_off = off;
    units[k].in_len = (uint32_t)((uint64_t) c->arena_len > off ? (uint64_t) c->arena_len - off : 0);
    units[k].out_off = (uint64_t) k * (uint64_t) c->interval_bytes;
    units[k].out_len = (uint32_t) c->interval_bytes;
    units[k].kind = MSPACK_HIP_KIND_LZX;
    units[k].window_bits = (uint8_t) c->window_bits;
    units[k].reset_frames = (uint16_t)(c->interval_bytes / FRAME);
    units[k].e8_base = e8_shift + (int32_t)((int64_t) k * c->interval_bytes);
  }
  if (!c->use_table) {            /* no usable reset table: one serial unit from offset 0 (chmd.c:1159-1166) */
    units[0].out_len = (uint32_t) c->stream_len;
    units[0].e8_base = 0;
  }
  rc = mspack_hip_decode_batch(units, count, c->arena, c->arena_len + 64, out,
                               (size_t)(c->use_table ? (off_t) count * c->interval_bytes : c->stream_len) + 64, res);
  sys->free(units);
  if (rc) { sys->message(NULL, "GPU batch decode failed: %s", mspack_hip_last_error()); return MSPACK_ERR_DECRUNCH; }
  return MSPACK_ERR_OK;
}

static int setup_sec1(struct chmd_p *self, struct chm_p *c, struct mspack_file *fh)
{
  struct mspack_system *sys = self->system;
  struct mschmd_sec_mscompressed *sec = &c->base.sec1;
  unsigned char *data;
  int err = MSPACK_ERR_OK;
  unsigned int version, wsize, frames_per;
  off_t reset_interval, total = 0;

  if ((err = find_sys_file(self, sec, &sec->content, content_name))) return err;
  if ((err = find_sys_file(self, sec, &sec->control, control_name))) return err;
  if (sec->control->length != 0x1C) return MSPACK_ERR_DATAFORMAT;
  if (!(data = read_sec0_file(self, fh, sec->control, &err))) return err;
  if (rd_le32(data + 4) != 0x43585A4Cu) { sys->free(data); return MSPACK_ERR_SIGNATURE; }
  version = rd_le32(data + 8);
  reset_interval = (off_t) rd_le32(data + 0x0C);
  wsize = rd_le32(data + 0x10);
  sys->free(data);
  if (version == 2) { reset_interval *= FRAME; wsize *= FRAME; }
  else if (version != 1) return MSPACK_ERR_DATAFORMAT;
  switch (wsize) {
  case 0x008000: c->window_bits = 15; break; case 0x010000: c->window_bits = 16; break;
  case 0x020000: c->window_bits = 17; break; case 0x040000: c->window_bits = 18; break;
  case 0x080000: c->window_bits = 19; break; case 0x100000: c->window_bits = 20; break;
  case 0x200000: c->window_bits = 21; break;
  default: return MSPACK_ERR_DATAFORMAT;
  }
  if (reset_interval == 0 || reset_interval % FRAME) return MSPACK_ERR_DATAFORMAT;
  if (reset_interval / FRAME > 65535) return MSPACK_ERR_DATAFORMAT;
  c->interval_bytes = reset_interval;
  frames_per = (unsigned int)(reset_interval / FRAME);

  /* the compressed stream: Content, plus the few bytes the reference could read past it */
  {
    off_t start = c->base.sec0.offset + sec->content->offset, flen = 0, avail;
    size_t want;
    if (sec->content->section->id != 0) return MSPACK_ERR_DATAFORMAT;
    if (mspack_sys_filelen(sys, fh, &flen)) return MSPACK_ERR_SEEK;
    avail = flen > start ? flen - start : 0;
    want = (size_t) sec->content->length + 64;
    if ((off_t) want > avail) want = (size_t) avail;
    if (!(c->arena = (unsigned char *) sys->alloc(sys, want + 128))) return MSPACK_ERR_NOMEMORY;
    memset(c->arena, 0, want + 128);
    if (sys->seek(fh, start, MSPACK_SYS_SEEK_START)) return MSPACK_ERR_SEEK;
    if (want && sys->read(fh, c->arena, (int) want) != (int) want) return MSPACK_ERR_READ;
    c->arena_len = want;
  }

  /* reset table: every interval's compressed offset (chmd.c:1195-1267) */
  c->use_table = 0;
  if (!find_sys_file(self, sec, &sec->rtable, rtable_name) && sec->rtable->length >= 0x28 &&
      sec->rtable->length <= 1000000 && (data = read_sec0_file(self, fh, sec->rtable, &err))) {
    unsigned int nent = rd_le32(data + 4), esz = rd_le32(data + 8), toff = rd_le32(data + 0x0C);
    if (rd_le32(data + 0x20) == FRAME && (esz == 4 || esz == 8)) {
      unsigned int ni, k, ok = 1;
      total = (off_t) rd_le64(data + 0x10);
      c->padded_len = (total + reset_interval - 1) & -reset_interval;      /* chmd.c:1153-1157 */
      ni = (unsigned int)(c->padded_len / reset_interval);
      if (ni && (c->ioff = (uint64_t *) sys->alloc(sys, (size_t) ni * sizeof(uint64_t)))) {
        for (k = 0; k < ni; k++) {
          unsigned int entry = k * frames_per;
          uint64_t pos = (uint64_t) toff + (uint64_t) entry * esz;
          if (entry >= nent || pos > (uint64_t) sec->rtable->length - esz) { ok = 0; break; }
          c->ioff[k] = (esz == 4) ? rd_le32(data + pos) : (uint64_t) rd_le64(data + pos);
        }
        if (ok) { c->use_table = 1; c->n_intervals = ni; }
        else { sys->free(c->ioff); c->ioff = NULL; }
      }
    }
    sys->free(data);
  }
  if (!c->use_table) {
    /* fall back to SpanInfo: one stream from offset 0 (chmd.c:1159-1166, 1275-1315) */
    if (find_sys_file(self, sec, &sec->spaninfo, spaninfo_name)) return MSPACK_ERR_DATAFORMAT;
    if (sec->spaninfo->length != 8) return MSPACK_ERR_DATAFORMAT;
    if (!(data = read_sec0_file(self, fh, sec->spaninfo, &err))) return err;
    total = (off_t) rd_le64(data);
    sys->free(data);
    if (total <= 0 || total > 0xFFFF0000LL) return MSPACK_ERR_DATAFORMAT;
    if (!(c->ioff = (uint64_t *) sys->alloc(sys, sizeof(uint64_t)))) return MSPACK_ERR_NOMEMORY;
    c->ioff[0] = 0; c->n_intervals = 1; c->padded_len = total;
  }
  c->stream_len = c->use_table ? c->padded_len : total;

  /* decode every interval of the CHM in one batch */
  if (!(c->dec = (unsigned char *) sys->alloc(sys, (size_t) c->stream_len + 128))) return MSPACK_ERR_NOMEMORY;
  if (!(c->ires = (mspack_hip_result *) sys->alloc(sys, (size_t) c->n_intervals * sizeof(mspack_hip_result)))) return MSPACK_ERR_NOMEMORY;
  return decode_intervals(self, c, 0, c->n_intervals, 0, c->dec, c->ires);
}

/* how far can a decoder that starts at `from` (an interval start) produce bytes without error?
 * returns MSPACK_ERR_OK if [from, end] (incl. the look-ahead frame, lzxd.c:419) decodes, else the
 * error; *good = first byte position that is not available */
static int range_status(struct chm_p *c, off_t from, off_t end, off_t *good)
{
  off_t need_frame = end / FRAME;                          /* last frame index the reference decodes */
  if (!c->use_table) {
    mspack_hip_result *r = &c->ires[0];
    off_t nframes = (c->stream_len + FRAME - 1) / FRAME;
    *good = r->good_len;
    if (r->err == MSPACK_ERR_OK) return (end <= c->stream_len) ? MSPACK_ERR_OK : MSPACK_ERR_DECRUNCH;
    if ((off_t) r->good_len >= c->stream_len) return (need_frame < nframes) ? MSPACK_ERR_OK : r->err;
    return (need_frame < (off_t)(r->good_len / FRAME)) ? MSPACK_ERR_OK : r->err;
  }
  else {
    unsigned int fper = (unsigned int)(c->interval_bytes / FRAME);
    unsigned int i0 = (unsigned int)(from / c->interval_bytes), i;
    off_t last_needed = need_frame / fper;                  /* interval holding the last needed frame */
    for (i = i0; i < c->n_intervals && (off_t) i <= last_needed; i++) {
      mspack_hip_result *r = &c->ires[i];
      off_t base = (off_t) i * c->interval_bytes;
      if (r->err != MSPACK_ERR_OK && !(r->flags & MSPACK_HIP_F_LOOKAHEAD_READ)) {
        /* frames of this interval before the failing one are fine */
        off_t good_frames = base / FRAME + r->good_len / FRAME;
        *good = base + r->good_len;
        if (need_frame < good_frames) return MSPACK_ERR_OK;
        return r->err;
      }
    }
    *good = c->padded_len;
    if (need_frame >= c->padded_len / FRAME) {
      /* the request ends exactly at the end of the stream: the reference still starts one more
       * (empty) frame, which only fails if the input is exhausted there */
      mspack_hip_result *r = &c->ires[c->n_intervals - 1];
      if (end > c->padded_len) return MSPACK_ERR_DECRUNCH;
      if (r->flags & MSPACK_HIP_F_LOOKAHEAD_READ) return MSPACK_ERR_READ;
    }
    return MSPACK_ERR_OK;
  }
}

static int chmd_extract(struct mschm_decompressor *base, struct mschmd_file *file, const char *filename)
{
  struct chmd_p *self = (struct chmd_p *) base;
  struct mspack_system *sys;
  struct chm_p *c;
  struct mspack_file *fh, *infh = NULL;

  if (!self) return MSPACK_ERR_ARGS;
  if (!file || !file->section) return self->error = MSPACK_ERR_ARGS;
  sys = self->system;
  c = (struct chm_p *) file->section->chm;

  if (!(infh = sys->open(sys, c->base.filename, MSPACK_SYS_OPEN_READ))) return self->error = MSPACK_ERR_OPEN;
  if (self->v_chm != c) { self->v_chm = c; self->v_alive = 0; self->v_offset = 0; }
  if (!(fh = sys->open(sys, filename, MSPACK_SYS_OPEN_WRITE))) { sys->close(infh); return self->error = MSPACK_ERR_OPEN; }
  if (!file->length) { sys->close(fh); sys->close(infh); return self->error = MSPACK_ERR_OK; }
  self->error = MSPACK_ERR_OK;

  if (file->section->id == 0) {
    if (sys->seek(infh, c->base.sec0.offset + file->offset, MSPACK_SYS_SEEK_START)) self->error = MSPACK_ERR_SEEK;
    else {
      unsigned char buf[512];
      off_t length = file->length, maxlen = c->base.length - sys->tell(infh);
      if (length > maxlen) sys->message(fh, "WARNING; file is %lld bytes longer than CHM file", (long long)(length - maxlen));
      while (length > 0) {
        int run = length > (off_t) sizeof(buf) ? (int) sizeof(buf) : (int) length;
        if (sys->read(infh, buf, run) != run) { self->error = MSPACK_ERR_READ; break; }
        if (sys->write(fh, buf, run) != run) { self->error = MSPACK_ERR_WRITE; break; }
        length -= run;
      }
    }
  }
  else {
    int err = MSPACK_ERR_OK;
    if (c->sec1_state == 0) {
      err = setup_sec1(self, c, infh);
      if (err) { free_sec1(sys, c); c->sec1_state = -err; } else c->sec1_state = 1;
    }
    else if (c->sec1_state < 0) err = -c->sec1_state;
    if (!err) {
      /* emulate the reference decoder's lifetime (chmd.c:993-999): restart at the file's reset
       * point when there is no live decoder or the request goes backwards */
      off_t start, end, good = 0, length = file->length, maxlen;
      if (!self->v_alive || file->offset < self->v_offset) {
        self->v_init = c->use_table ? (file->offset / c->interval_bytes) * c->interval_bytes : 0;
        self->v_offset = self->v_init;
        self->v_alive = 1;
      }
      start = self->v_offset;
      if (file->offset > c->stream_len) err = MSPACK_ERR_DECRUNCH;               /* chmd.c:1002-1005 */
      else {
        maxlen = c->stream_len - file->offset;
        if (length > maxlen) {
          sys->message(fh, "WARNING; file is %lld bytes longer than compressed section", (long long)(length - maxlen));
          length = maxlen + 1;                        /* decodes what exists, then errors out */
        }
        /* skip phase, then emit phase */
        if (file->offset > start) err = range_status(c, start, file->offset, &good);
        if (!err) {
          const unsigned char *src = c->dec;
          unsigned char *tmp = NULL;
          off_t have;
          end = file->offset + length;
          err = range_status(c, start, end, &good);
          have = (good > file->offset) ? good - file->offset : 0;
          if (have > length) have = length;
          if (file->offset + have > c->stream_len) have = c->stream_len - file->offset;
          /* E8: intervals whose header enables the translation depend on where the reference's
           * decoder was initialised; re-decode those with that origin */
          if (c->use_table && have > 0) {
            unsigned int i0 = (unsigned int)(file->offset / c->interval_bytes);
            unsigned int i1 = (unsigned int)((file->offset + have - 1) / c->interval_bytes), i;
            int any = 0;
            for (i = i0; i <= i1; i++)
              if ((c->ires[i].flags & MSPACK_HIP_F_E8_APPLIED) && (off_t) i * c->interval_bytes != self->v_init) any = 1;
            if (any) {
              unsigned int cnt = i1 - i0 + 1;
              mspack_hip_result *r2 = (mspack_hip_result *) sys->alloc(sys, cnt * sizeof(*r2));
              tmp = (unsigned char *) sys->alloc(sys, (size_t) cnt * (size_t) c->interval_bytes + 128);
              if (!r2 || !tmp) err = MSPACK_ERR_NOMEMORY;
              else {
                int e2 = decode_intervals(self, c, i0, cnt, (int32_t)((off_t) i0 * c->interval_bytes - self->v_init), tmp, r2);
                if (e2) err = e2;
                else src = tmp - (off_t) i0 * c->interval_bytes;
              }
              sys->free(r2);
            }
          }
          if (have > 0 && err != MSPACK_ERR_NOMEMORY) {
            if (write_slice(sys, fh, src + file->offset, (size_t) have) != MSPACK_ERR_OK) err = MSPACK_ERR_WRITE;
          }
          sys->free(tmp);
        }
      }
      if (err) self->v_alive = 0;                       /* chmd.c:1036-1040 */
      else self->v_offset = file->offset + length;
    }
    self->error = err;
  }
  sys->close(fh);
  sys->close(infh);
  return self->error;
}

static int chmd_error(struct mschm_decompressor *base) {
  struct chmd_p *self = (struct chmd_p *) base;
  return self ? self->error : MSPACK_ERR_ARGS;
}

struct mschm_decompressor *mspack_create_chm_decompressor(struct mspack_system *sys)
{
  struct chmd_p *self;
  if (!sys) sys = mspack_default_system;
  if (!mspack_valid_system(sys)) return NULL;
  if (!(self = (struct chmd_p *) sys->alloc(sys, sizeof(*self)))) return NULL;
  memset(self, 0, sizeof(*self));
  self->base.open = &chmd_open;
  self->base.close = &chmd_close;
  self->base.extract = &chmd_extract;
  self->base.last_error = &chmd_error;
  self->base.fast_open = &chmd_fast_open;
  self->base.fast_find = &chmd_fast_find;
  self->system = sys;
  self->error = MSPACK_ERR_OK;
  return &self->base;
}

void mspack_destroy_chm_decompressor(struct mschm_decompressor *base)
{
  struct chmd_p *self = (struct chmd_p *) base;
  if (self) self->system->free(self);
}
