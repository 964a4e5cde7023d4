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

struct chm_chunk {                /* a run of reset intervals decoded in one GPU batch */
  unsigned char *buf;             /* decoded bytes (E8 origin 0), or NULL when evicted               */
  unsigned long stamp;            /* LRU clock                                                      */
  int res_valid;                  /* ires[] of these intervals is filled in                         */
  /* the chunk's batch while it is still running (mspack_hip.h: jobs): interval first + i is unit i; of buf and ires[] only what
   * chunk_wait() has covered may be read, and neither may be freed before chunk_settle() */
  mspack_hip_job *job;
  mspack_hip_unit *job_units;
  unsigned int job_waited;        /* units [0, job_waited) have come through                        */
};
struct chm_p {
  struct mschmd_header base;
  /* section 1 */
  int sec1_state;                 /* 0 = not set up, 1 = ready, <0 = -(setup error)                 */
  int window_bits;
  unsigned int fper;              /* frames per reset interval                                      */
  off_t interval_bytes, padded_len;
  unsigned int n_intervals;       /* padded_len / interval_bytes                                    */
  unsigned int n_fast;            /* intervals [0, n_fast) have a reset-table entry                 */
  int span_err; off_t span_len;   /* SpanInfo: the fallback stream length (chmd.c:1159-1166)        */
  unsigned char *arena; size_t arena_len;     /* the CHM file from the start of Content to its end  */
  int arena_pinned;                           /* page-locked (mspack_hip_pin) until free_sec1                */
  off_t content_start;            /* file offset of the Content stream                              */
  uint64_t *ioff;                 /* reset-table entry (compressed offset) of intervals [0, n_fast)  */
  size_t ftab_off;                /* arena offset of the per-frame tables (0: none), n_fast * fper uint32: every
                                     frame's compressed offset from its interval's start (MSPACK_HIP_UF_FRAME_TABLE) */
  size_t arena_room;              /* bytes of the arena buffer that belong to the batch decoder's input arena */
  mspack_hip_result *ires;        /* stand-alone result per interval                                */
  struct chm_chunk *chunks; unsigned int n_chunks, chunk_int; unsigned long stamp;
  /* the serial span of the virtual decoder (damaged / table-less files) */
  int s_valid, s_mode; off_t s_init, s_cover; unsigned char *s_buf; mspack_hip_result s_res;
  unsigned int s_log_n, s_log_cap;     /* the serial span's reset log (MSPACK_HIP_UF_LZX_LOG): frames, counted from s_init, whose
                                          reset found a block open; it lies in s_buf behind the span (s_log) */
  const unsigned char *s_log;
};
struct vdec {                     /* the reference's lzxd instance, replayed (chmd.c:989-1040)       */
  int alive, mode, serial, seek_pending;   /* mode 0: created at a reset-table entry, 1: at offset 0 with SpanInfo */
  off_t init, length, offset;     /* creation point, stream length it was created with, position    */
  off_t dlen;                     /* the reference's d->length (chmd.c:1176): what extract() checks offsets and lengths against.
                                     == length, except for a decoder created exactly AT the stream's stated end: lzxd_init then
                                     gets an output length of 0, which means "not known" (lzxd.c:419, 458-461) -- it decodes
                                     whatever frames follow, for as long as the input lasts */
  off_t decoded_end;              /* end of the last frame it has decoded: [offset, decoded_end) is stored
                                     up in its window and handed out without decoding (lzxd.c:397-408)  */
  uint64_t in_off;                /* compressed offset it was created at                            */
};
struct chmd_p {
  struct mschm_decompressor base;
  struct mspack_system *system;
  int error;
  struct chm_p *v_chm; struct vdec v;
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
  /* the reference keeps ONE bad-ENCINT flag for the whole listing (chmd.c:262, never cleared): after the first badly encoded
   * integer every later chunk ends at its first entry -- nothing more is listed (found by tools/fuzz_chmdir_cpu.py) */
  int err = 0;

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
    int entries;
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

static void free_sec1(struct mspack_system *sys, struct chm_p *c);

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
  if (self->v_chm == (struct chm_p *) chm) { self->v_chm = NULL; self->v.alive = 0; }
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

/* ---- names ------------------------------------------------------------------------------------------------
 * Directory names are UTF-8 and the directory is sorted by their code points after simple case folding.  The decoder
 * below has to accept exactly what the reference's accepts (chmd.c:861-880: lead bytes C2..DF / E0..EF / F0..F5, no
 * check of the continuation bytes, U+FFFD for everything else and for a sequence the name is too short for), or a
 * damaged name would be looked for in another place than the reference looks. */
struct cp_cursor { const unsigned char *at, *stop; };

static unsigned int cp_take(struct cp_cursor *cur)
{
  const unsigned int lead = *cur->at++;
  unsigned int more, cp, i;
  if (lead < 0x80u) return lead;
  if (lead >= 0xC2u && lead <= 0xDFu) { more = 1; cp = lead & 0x1Fu; }
  else if (lead >= 0xE0u && lead <= 0xEFu) { more = 2; cp = lead & 0x0Fu; }
  else if (lead >= 0xF0u && lead <= 0xF5u) { more = 3; cp = lead & 0x07u; }
  else return 0xFFFDu;
  if ((size_t)(cur->stop - cur->at) < more) return 0xFFFDu;     /* cut short: one byte consumed */
  for (i = 0; i < more; i++) cp = (cp << 6) | (*cur->at++ & 0x3Fu);
  return cp > 0x10FFFFu ? 0xFFFDu : cp;
}

/* < 0, 0, > 0: `want` sorts before / equals / sorts after `have`.  Code points that differ are compared after
 * towlower(); when one name is a prefix of the other the shorter one sorts first (byte lengths). */
static int name_order(const char *want, int want_len, const unsigned char *have, int have_len)
{
  struct cp_cursor a, b;
  a.at = (const unsigned char *) want; a.stop = a.at + want_len;
  b.at = have; b.stop = have + have_len;
  while (a.at < a.stop && b.at < b.stop) {
    const unsigned int x = cp_take(&a), y = cp_take(&b);
    if (x != y) {
      const int fx = (int) towlower((wint_t) x), fy = (int) towlower((wint_t) y);
      if (fx != fy) return fx - fy;
    }
  }
  return want_len - have_len;
}

/* ---- directory chunks --------------------------------------------------------------------------------------
 * A chunk (chm->chunk_size bytes): "PMGL" + free-space length at 4 (+ prev / next chunk numbers at
 * 0xC / 0x10, entries from 0x14) or "PMGI" + free-space length at 4 (entries from 8).  The entries are packed from the
 * front -- ENCINT name length, name, then for PMGL three ENCINTs (section, offset, length), for PMGI one (the child
 * chunk) -- and the chunk ENDS with a 16-bit entry count preceded by the quick-reference table growing downwards: the
 * i-th word below the count is the offset, from the first entry, of entry i * (1 + 2^density). */
static unsigned char *chunk_get(struct chmd_p *self, struct mschmd_header *chm, struct mspack_file *fh, unsigned int n)
{
  struct mspack_system *sys = self->system;
  const size_t bytes = chm->chunk_size;
  unsigned char *data;
  int failed = 0;
  if (n >= chm->num_chunks) return NULL;
  if (!chm->chunk_cache) {
    unsigned char **slots = (unsigned char **) sys->alloc(sys, sizeof(unsigned char *) * chm->num_chunks);
    unsigned int i;
    if (!slots) { self->error = MSPACK_ERR_NOMEMORY; return NULL; }
    for (i = 0; i < chm->num_chunks; i++) slots[i] = NULL;
    chm->chunk_cache = slots;
  }
  if (chm->chunk_cache[n]) return chm->chunk_cache[n];
  if (!(data = (unsigned char *) sys->alloc(sys, bytes))) { self->error = MSPACK_ERR_NOMEMORY; return NULL; }
  if (sys->seek(fh, chm->dir_offset + (off_t) n * (off_t) bytes, MSPACK_SYS_SEEK_START)) failed = MSPACK_ERR_SEEK;
  else if (sys->read(fh, data, (int) bytes) != (int) bytes) failed = MSPACK_ERR_READ;
  else if (memcmp(data, "PMG", 3) != 0 || (data[3] != 'L' && data[3] != 'I')) failed = MSPACK_ERR_SEEK;   /* (the reference's code for a bad signature, chmd.c:676-680) */
  if (failed) { self->error = failed; sys->free(data); return NULL; }
  chm->chunk_cache[n] = data;
  return data;
}

struct chunk_view {
  const unsigned char *first;        /* the first entry */
  const unsigned char *limit;        /* end of the entry area = chunk end - free-space length */
  const unsigned char *count_at;     /* the 16-bit entry count; quick-reference word i sits 2 * i bytes below it */
  unsigned int n_entries, n_quick, per_quick;
  int leaf;                          /* PMGL */
};
/* one entry: its name and what follows it; 0 when the entry does not fit the entry area */
static int entry_name(const struct chunk_view *cv, const unsigned char *at, const unsigned char **name, unsigned int *len)
{
  int bad = 0;
  const unsigned int l = (unsigned int) read_encint(&at, cv->limit, &bad);     /* (32 bits of it, as the reference keeps) */
  if (bad || l > (unsigned int)(cv->limit - at)) return 0;
  *name = at; *len = l;
  return 1;
}
static const unsigned char *quick_entry(const struct chunk_view *cv, unsigned int i)
{
  return cv->first + (i ? rd_le16(cv->count_at - 2u * i) : 0u);
}
static const unsigned char *pass_encints(const unsigned char *at, const unsigned char *limit, int count)
{
  while (count-- > 0) while (at < limit && (*at++ & 0x80)) ;
  return at;
}

/* -1 = malformed chunk, 0 = the name is not here, 1 = found: *res points at the entry's data (PMGL: section, offset,
 * length; PMGI: the child chunk number of the last entry not above the name), *res_end = end of the entry area.
 * The quick-reference probes are the reference's (chmd.c:705-800: the middle of a closed interval, rounded down), so
 * that a chunk whose quick-reference words or entries are damaged is judged by the same entries. */
static int search_chunk(struct mschmd_header *chm, const unsigned char *chunk, const char *filename,
                        const unsigned char **res, const unsigned char **res_end)
{
  struct chunk_view cv;
  const int want_len = (int) strlen(filename);
  const unsigned int free_len = rd_le32(chunk + 4);
  const unsigned char *at, *name, *below = NULL;
  unsigned int len, left, lo, hi;

  cv.leaf = chunk[3] == 'L';
  cv.first = chunk + (cv.leaf ? 0x14 : 8);
  cv.count_at = chunk + chm->chunk_size - 2;
  cv.limit = chunk + chm->chunk_size - free_len;
  cv.n_entries = rd_le16(cv.count_at);
  cv.per_quick = 1u + (1u << chm->density);
  if (cv.n_entries == 0 || free_len > chm->chunk_size) return -1;
  *res_end = cv.limit;
  cv.n_quick = (cv.n_entries + cv.per_quick - 1u) / cv.per_quick;
  if ((int)(2u * cv.n_quick) > (int)(cv.count_at - cv.limit)) cv.n_quick = 0;     /* a table that cannot fit: do without */

  /* which group of per_quick entries?  [lo, hi): the quick entries not yet known to sort before (below lo) or after
   * (from hi on) the name */
  at = cv.first; left = cv.n_entries;
  for (lo = 0, hi = cv.n_quick; lo < hi; ) {
    const unsigned int mid = (lo + hi - 1u) >> 1;
    int o;
    at = quick_entry(&cv, mid);
    if (!entry_name(&cv, at, &name, &len)) return -1;
    o = name_order(filename, want_len, name, (int) len);
    if (o == 0) { *res = name + len; return 1; }
    if (o > 0) lo = mid + 1u;
    else if (mid == 0u) return 0;                               /* sorts before the chunk's first entry */
    else hi = mid;
  }
  if (cv.n_quick) {
    /* lo - 1 = the last quick entry that sorts before the name: its group is the only one that can hold it */
    const unsigned int g = lo - 1u;
    at = quick_entry(&cv, g);
    left = cv.n_entries - g * cv.per_quick;
    if (left > cv.per_quick) left = cv.per_quick;
  }
  for (; left > 0; left--) {
    int o;
    if (!entry_name(&cv, at, &name, &len)) return -1;
    o = name_order(filename, want_len, name, (int) len);
    at = name + len;
    if (o == 0) { *res = at; return 1; }
    if (o < 0) break;
    if (!cv.leaf) below = at;                                   /* (an index chunk: descend below the last name not above ours) */
    at = pass_encints(at, cv.limit, cv.leaf ? 3 : 1);
  }
  *res = below;
  return below ? 1 : 0;
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
      if (!(chunk = chunk_get(self, chm, fh, n))) { sys->close(fh); return self->error; }
      if ((result = search_chunk(chm, chunk, filename, &p, &end)) <= 0) break;
      if (chunk[3] == 'L') break;
      n = (unsigned int) read_encint(&p, end, &e);
      if (e) { sys->close(fh); return self->error = MSPACK_ERR_DATAFORMAT; }
    }
  }
  else {
    for (n = chm->first_pmgl; n <= chm->last_pmgl; n = rd_le32(chunk + 0x10)) {
      if (!(chunk = chunk_get(self, chm, fh, n))) { err = self->error; break; }
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

/* ---- section 1 (reference chmd.c:989-1041, 1072-1315) ---------------------------------------------------
 *
 * What the reference does: extract() keeps ONE lzxd instance alive.  It is (re)created when there is none
 * or the request lies behind its position: at the reset point of the file's interval if the reset table
 * has an entry for it (stream length = UncompLen padded to the interval), else at offset 0 with SpanInfo's
 * length.  A request is lzxd_decompress(skip) + lzxd_decompress(length); every decompress call decodes
 * whole frames up to and INCLUDING the frame that holds the first byte after the request (lzxd.c:419), and
 * any error kills the instance.
 *
 * What we do: every reset interval with a table entry is an independent unit ("fast" results: decoded in
 * GPU batches of CHM_CHUNK_BYTES, buffers kept LRU within the cache budget, per-interval results kept
 * for good).  A virtual decoder (struct vdec) replays the reference's instance on top of those results.
 * An interval's stand-alone result IS the live decoder's result when the decoder was created AT it.  For
 * an interval the live decoder crosses INTO, that only holds if the stand-alone decode is clean (an
 * error there may be a match that reaches before the interval: legal for a decoder that has that
 * history, lzxd.c:622-634), the table entry is where the previous interval's bits really end
 * (result.in_next) and no block spans the reset (MSPACK_HIP_F_BLOCK_OPEN, lzxd.c:424-431).  Otherwise --
 * damaged or odd files only -- the virtual decoder goes SERIAL: one unit from its creation point on, which
 * is the reference's own control flow, exact by construction.  E8: the translation origin is the creation
 * point (lzxd.c:712); fast results use origin 0, so intervals that applied E8 are decoded again with the
 * shifted origin when the decoder was created elsewhere. */
#define CHM_CHUNK_BYTES ((off_t) 64 << 20)
#define CHM_SERIAL_SLACK 16                 /* serial spans are decoded this many intervals past the need */

static int hip_batch(mspack_hip_unit *units, size_t n, const void *in, size_t in_bytes, void *out, size_t out_bytes,
                     mspack_hip_result *res)
{
  int dv = mspack_hip_default_devices();
  return dv > 1 ? mspack_hip_decode_batch_multi(units, n, in, in_bytes, out, out_bytes, res, dv)
                : mspack_hip_decode_batch(units, n, in, in_bytes, out, out_bytes, res);
}

/* decode intervals [first, first+count) stand-alone; interval i's E8 origin is i*interval - e8_origin */
static int decode_intervals(struct chmd_p *self, struct chm_p *c, unsigned int first, unsigned int count,
                            off_t e8_origin, unsigned char *out, mspack_hip_result *res, struct chm_chunk *as_job)
{
  struct mspack_system *sys = self->system;
  mspack_hip_unit *units = (mspack_hip_unit *) sys->alloc(sys, (size_t) count * sizeof(*units));
  unsigned int k;
  int rc;
  if (!units) return MSPACK_ERR_NOMEMORY;
  memset(units, 0, (size_t) count * sizeof(*units));
  for (k = 0; k < count; k++) {
    uint64_t off = c->ioff[first + k];
    if (off > (uint64_t) c->arena_len) off = c->arena_len;        /* an entry beyond the file: nothing to read */
    units[k].in_off = off;
    units[k].in_len = (uint32_t)((uint64_t) c->arena_len - off > 0xFFFFFFF0u ? 0xFFFFFFF0u : (uint64_t) c->arena_len - off);
    units[k].out_off = (uint64_t) k * (uint64_t) c->interval_bytes;
    units[k].out_len = (uint32_t) c->interval_bytes;
    {
      /* the stream ends where the (padded) UncompLen says: an interval that holds the end is as long as what is left of it --
       * for an interval size that is no power of two the reference's padding (& -interval, chmd.c:1153-1157) is no multiple of
       * it.  (Intervals wholly behind the stated end are only ever decoded by a decoder created AT it: unknown length, §vdec) */
      const off_t at = (off_t)(first + k) * c->interval_bytes;
      if (at < c->padded_len && c->padded_len - at < c->interval_bytes) units[k].out_len = (uint32_t)(c->padded_len - at);
    }
    units[k].kind = MSPACK_HIP_KIND_LZX;
    units[k].window_bits = (uint8_t) c->window_bits;
    units[k].reset_frames = (uint16_t) c->fper;
    units[k].e8_base = (int32_t)((off_t)(first + k) * c->interval_bytes - e8_origin);
    if (c->ftab_off) {
      units[k].flags |= MSPACK_HIP_UF_FRAME_TABLE;
      units[k].in_chunk = (uint32_t)((c->ftab_off + (size_t)(first + k) * c->fper * 4) / 4);
    }
  }
  /* a chunk's batch runs as a job when it can: extract() of the first file returns when the file's intervals are through,
   * and the caller writes it while the rest is decoded and copied back (one device: a sharded batch is synchronous) */
  if (as_job && count > 1 && mspack_hip_default_devices() <= 1 &&
      (as_job->job = mspack_hip_decode_batch_begin(units, count, c->arena, c->arena_room, out, (size_t) count * (size_t) c->interval_bytes + 64, res))) {
    as_job->job_units = units; as_job->job_waited = 0;
    return MSPACK_ERR_OK;
  }
  rc = hip_batch(units, count, c->arena, c->arena_room, out, (size_t) count * (size_t) c->interval_bytes + 64, res);
  sys->free(units);
  if (rc) { sys->message(NULL, "GPU batch decode failed: %s", mspack_hip_last_error()); return MSPACK_ERR_DECRUNCH; }
  return MSPACK_ERR_OK;
}

/* the chunk's batch, if it is still running, to its end: buf and ires[] are the driver's again (0, or the batch's failure) */
static int chunk_settle(struct mspack_system *sys, struct chm_chunk *ch)
{
  int rc = 0;
  if (ch->job) {
    rc = mspack_hip_job_end(ch->job);
    ch->job = NULL;
    sys->free(ch->job_units); ch->job_units = NULL;
  }
  return rc;
}

static void free_sec1(struct mspack_system *sys, struct chm_p *c) {
  unsigned int i;
  if (c->chunks) for (i = 0; i < c->n_chunks; i++) { (void) chunk_settle(sys, &c->chunks[i]); mspack_arena_free(sys, c->chunks[i].buf); }
  if (c->arena_pinned) { mspack_hip_unpin(c->arena); c->arena_pinned = 0; }
  sys->free(c->chunks); mspack_arena_free(sys, c->arena); sys->free(c->ioff); sys->free(c->ires); mspack_arena_free(sys, c->s_buf);
  c->chunks = NULL; c->arena = NULL; c->ioff = NULL; c->ires = NULL; c->s_buf = NULL; c->sec1_state = 0;
  c->n_chunks = 0; c->s_valid = 0;
}

/* ControlData, the compressed stream, ResetTable and SpanInfo: everything chmd_init_decomp (chmd.c:1072-1188)
 * reads, parsed once.  Returns the error the reference's init would return for EVERY file. */
static int setup_sec1(struct chmd_p *self, struct chm_p *c, struct mspack_file *fh)
{
  struct mspack_system *sys = self->system;
  struct mschmd_sec_mscompressed *sec = &c->base.sec1;
  unsigned char *data;
  int err = MSPACK_ERR_OK;
  unsigned int version, wsize;
  off_t reset_interval;
  size_t arena_alloc = 0;            /* bytes behind c->arena */

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
  /* the reference computes in `int` (chmd.c:1075,1114-1121): an interval beyond 2^31 wraps there */
  reset_interval = (off_t)(int)(unsigned int) reset_interval;
  if (reset_interval == 0 || reset_interval % FRAME) return MSPACK_ERR_DATAFORMAT;
  /* (an interval that wrapped negative passes the reference's checks and ends in lzxd_init(reset_interval < 0) == NULL:
   *  MSPACK_ERR_NOMEMORY, lzxd.c:297-300, chmd.c:1183-1187) */
  if (reset_interval < 0) return MSPACK_ERR_NOMEMORY;
  if (reset_interval / FRAME > 65535) return MSPACK_ERR_DATAFORMAT;                          /* ours: unit field width */
  c->interval_bytes = reset_interval;
  c->fper = (unsigned int)(reset_interval / FRAME);

  /* the compressed stream: from the start of Content to the end of the FILE -- the reference's decoder
   * reads on from wherever it is (sys->read on the CHM itself), not just inside Content */
  {
    off_t start = c->base.sec0.offset + sec->content->offset, flen = 0, avail;
    size_t want;
    if (sec->content->section->id != 0) return MSPACK_ERR_DATAFORMAT;
    if (mspack_sys_filelen(sys, fh, &flen)) return MSPACK_ERR_SEEK;
    avail = (start >= 0 && flen > start) ? flen - start : 0;
    want = (size_t) avail;
    /* (room behind the stream: 128 zero bytes, then one uint32 per frame the reset table can describe) */
    {
      /* the frame table below holds n_fast * fper entries, n_fast <= table bytes / (fper * entry size) + 1 intervals:
       * at most one uint32 per 4 table bytes plus one interval's worth (fper comes from ControlData: up to 65535) */
      size_t extra = 0;
      if (!find_sys_file(self, sec, &sec->rtable, rtable_name) && sec->rtable->length >= 0x28 && sec->rtable->length <= 1000000)
        extra = (size_t) sec->rtable->length + (size_t) c->fper * 4u;
      arena_alloc = want + 128 + extra + 16;
      if (!(c->arena = (unsigned char *) mspack_arena_alloc(sys, arena_alloc))) return MSPACK_ERR_NOMEMORY;
      memset(c->arena + want, 0, arena_alloc - want);         /* (the stream itself is read over the rest) */
      c->arena_room = want + 64;
    }
    c->ftab_off = 0;
    if (want) {
      if (sys->seek(fh, start, MSPACK_SYS_SEEK_START)) return MSPACK_ERR_SEEK;
      if (sys->read(fh, c->arena, (int) want) != (int) want) return MSPACK_ERR_READ;
    }
    c->arena_len = want;
    c->content_start = start;
  }

  /* reset table (read_reset_table, chmd.c:1195-1267): which intervals have a usable entry */
  c->n_fast = 0; c->n_intervals = 0; c->padded_len = 0;
  if (!find_sys_file(self, sec, &sec->rtable, rtable_name) && sec->rtable->length >= 0x28 &&
      sec->rtable->length <= 1000000 && (data = read_sec0_file(self, fh, sec->rtable, &err))) {
    unsigned int nent = rd_le32(data + 4), esz = rd_le32(data + 8), toff = rd_le32(data + 0x0C);
    if (rd_le32(data + 0x20) == FRAME && (esz == 4 || esz == 8)) {
      off_t total = (off_t) rd_le64(data + 0x10);
      uint64_t ni, nt;
      c->padded_len = (total + reset_interval - 1) & -reset_interval;      /* chmd.c:1153-1157 */
      /* (intervals that hold bytes of the padded length: for an interval that is no power of two the reference's rounding
       *  above does not give a multiple of it) */
      ni = c->padded_len > 0 ? (uint64_t)((c->padded_len + reset_interval - 1) / reset_interval) : 0;
      if (ni > 0x7FFFFFFFu / c->fper) ni = 0x7FFFFFFFu / c->fper;
      /* entries exist for the intervals whose first frame index is below NumEntries and inside the table's file -- whatever
       * UncompLen says: the reference looks an entry up BEFORE it looks at the length (chmd.c:1146-1157), so a file that lies
       * behind a dishonest UncompLen is answered from ITS interval (and then refused: lzxd_init with a negative length ->
       * MSPACK_ERR_NOMEMORY, or "offset beyond the length" -> MSPACK_ERR_DECRUNCH), not from SpanInfo */
      nt = (uint64_t) nent / c->fper + 1u;
      { const uint64_t by_file = (uint64_t) sec->rtable->length / esz / c->fper + 1u; if (nt > by_file) nt = by_file; }
      if (nt > 0x7FFFFFFFu / c->fper) nt = 0x7FFFFFFFu / c->fper;
      if (nt && (c->ioff = (uint64_t *) sys->alloc(sys, (size_t) nt * sizeof(uint64_t)))) {
        unsigned int k;
        for (k = 0; k < nt; k++) {
          unsigned int entry = k * c->fper;
          unsigned int pos = toff + entry * esz;                              /* unsigned wrap as in chmd.c:1237 */
          if (entry >= nent || (off_t) pos > sec->rtable->length - (off_t) esz) break;
          c->ioff[k] = (esz == 4) ? rd_le32(data + pos) : (uint64_t) rd_le64(data + pos);
        }
        c->n_fast = k;
        c->n_intervals = (unsigned int) ni;
        /* the frames' offsets inside every interval, for the frame-parallel parse: a hint, never trusted */
        if (c->fper >= 2 && c->n_fast &&
            ((c->arena_len + 128 + 3) & ~(size_t) 3) + (size_t) c->n_fast * c->fper * 4u <= arena_alloc) {   /* (never without room) */
          const size_t base = (c->arena_len + 128 + 3) & ~(size_t) 3;
          uint32_t *tab = (uint32_t *)(c->arena + base);
          unsigned int j;
          int ok = 1;
          for (k = 0; k < c->n_fast && ok; k++)
            for (j = 0; j < c->fper; j++) {
              unsigned int entry = k * c->fper + j;
              unsigned int pos = toff + entry * esz;
              uint64_t v;
              if (entry >= nent || (off_t) pos > sec->rtable->length - (off_t) esz) { tab[entry] = 0xFFFFFFFFu; continue; }
              v = (esz == 4) ? rd_le32(data + pos) : (uint64_t) rd_le64(data + pos);
              tab[entry] = (v >= c->ioff[k] && v - c->ioff[k] < 0xFFFFFFF0u) ? (uint32_t)(v - c->ioff[k]) : 0xFFFFFFFFu;
            }
          c->ftab_off = base;
          c->arena_room = base + (size_t) c->n_fast * c->fper * 4;
        }
      }
    }
    sys->free(data);
  }
  /* SpanInfo (read_spaninfo, chmd.c:1275-1315): the fallback's stream length, or the error it ends in */
  c->span_err = MSPACK_ERR_OK; c->span_len = 0;
  if (find_sys_file(self, sec, &sec->spaninfo, spaninfo_name)) c->span_err = MSPACK_ERR_DATAFORMAT;
  else if (sec->spaninfo->length != 8) c->span_err = MSPACK_ERR_DATAFORMAT;
  else if (!(data = read_sec0_file(self, fh, sec->spaninfo, &err))) c->span_err = err;
  else {
    c->span_len = (off_t) rd_le64(data);
    sys->free(data);
    if (c->span_len <= 0) c->span_err = MSPACK_ERR_DATAFORMAT;
    else if (c->span_len > 0xFFFF0000LL) c->span_err = MSPACK_ERR_DATAFORMAT;   /* ours: a unit's out_len is 32 bits */
  }

  /* every batch of this CHM reads the arena: page-locked once, its copies to the device are plain DMA (advice only) */
  if (!c->arena_pinned && arena_alloc >= ((size_t) 4 << 20) && !mspack_arena_is_locked(c->arena)) c->arena_pinned = mspack_hip_pin(c->arena, mspack_arena_room(arena_alloc)) == 0;

  /* fast-result bookkeeping */
  if (c->n_fast) {
    unsigned int i;
    c->chunk_int = (unsigned int)(CHM_CHUNK_BYTES / c->interval_bytes); if (!c->chunk_int) c->chunk_int = 1;
    c->n_chunks = (c->n_fast + c->chunk_int - 1) / c->chunk_int;
    if (!(c->chunks = (struct chm_chunk *) sys->alloc(sys, (size_t) c->n_chunks * sizeof(*c->chunks)))) return MSPACK_ERR_NOMEMORY;
    for (i = 0; i < c->n_chunks; i++) {
      c->chunks[i].buf = NULL; c->chunks[i].stamp = 0; c->chunks[i].res_valid = 0;
      c->chunks[i].job = NULL; c->chunks[i].job_units = NULL; c->chunks[i].job_waited = 0;
    }
    if (!(c->ires = (mspack_hip_result *) sys->alloc(sys, (size_t) c->n_fast * sizeof(mspack_hip_result)))) return MSPACK_ERR_NOMEMORY;
  }
  return MSPACK_ERR_OK;
}

/* a chunk whose batch is still running: wait until intervals up to k_to (of this chunk) have come through */
static int chunk_wait(struct chmd_p *self, struct chm_chunk *ch, unsigned int first, unsigned int count, unsigned int k_to)
{
  struct mspack_system *sys = self->system;
  unsigned int upto = k_to - first + 1;
  if (!ch->job) return MSPACK_ERR_OK;
  if (upto > count) upto = count;
  while (ch->job_waited < upto) {
    if (mspack_hip_job_wait_unit(ch->job, ch->job_waited)) {
      /* the batch failed: what decode_intervals says of a failed call -- nothing of the chunk is kept */
      (void) chunk_settle(sys, ch);
      sys->message(NULL, "GPU batch decode failed: %s", mspack_hip_last_error());
      mspack_arena_free(sys, ch->buf); ch->buf = NULL; ch->res_valid = 0;
      return MSPACK_ERR_DECRUNCH;
    }
    ch->job_waited++;
  }
  if (ch->job_waited >= count && chunk_settle(sys, ch)) {
    sys->message(NULL, "GPU batch decode failed: %s", mspack_hip_last_error());
    mspack_arena_free(sys, ch->buf); ch->buf = NULL; ch->res_valid = 0;
    return MSPACK_ERR_DECRUNCH;
  }
  return MSPACK_ERR_OK;
}

/* fast results (and, if `need_buf`, the decoded bytes) of the chunk that holds interval k -- of its intervals up to k_to */
static int ensure_chunk_to(struct chmd_p *self, struct chm_p *c, unsigned int k, unsigned int k_to, int need_buf, struct chm_chunk **out)
{
  struct mspack_system *sys = self->system;
  unsigned int ci = k / c->chunk_int, first = ci * c->chunk_int, i;
  unsigned int count = (first + c->chunk_int <= c->n_fast) ? c->chunk_int : c->n_fast - first;
  struct chm_chunk *ch = &c->chunks[ci];
  int err;
  if (out) *out = ch;
  ch->stamp = ++c->stamp;
  if (ch->buf || (ch->res_valid && !need_buf)) return chunk_wait(self, ch, first, count, k_to);
  /* stay inside the cache budget: drop the least recently used buffers */
  {
    size_t budget = (size_t) mspack_hip_cache_mb() << 20, each = (size_t) c->chunk_int * (size_t) c->interval_bytes;
    for (;;) {
      size_t used = 0; struct chm_chunk *lru = NULL;
      for (i = 0; i < c->n_chunks; i++)
        if (c->chunks[i].buf) { used += each; if (&c->chunks[i] != ch && (!lru || c->chunks[i].stamp < lru->stamp)) lru = &c->chunks[i]; }
      if (used + each <= budget || !lru) break;
      if (chunk_settle(sys, lru)) lru->res_valid = 0;      /* (a batch that failed behind what was asked of it: decoded again when needed) */
      mspack_arena_free(sys, lru->buf); lru->buf = NULL;
    }
  }
  if (!(ch->buf = (unsigned char *) mspack_arena_alloc(sys, (size_t) count * (size_t) c->interval_bytes + 128))) return MSPACK_ERR_NOMEMORY;
  /* (a chunk whose bytes will not be kept -- beyond the cache budget -- is decoded synchronously: its buffer goes at once) */
  {
    const int keep = need_buf || count * (size_t) c->interval_bytes <= ((size_t) mspack_hip_cache_mb() << 20);
    err = decode_intervals(self, c, first, count, 0, ch->buf, &c->ires[first], keep ? ch : NULL);
    if (err) { mspack_arena_free(sys, ch->buf); ch->buf = NULL; return err; }
    ch->res_valid = 1;
    if (!keep) { mspack_arena_free(sys, ch->buf); ch->buf = NULL; }
  }
  return chunk_wait(self, ch, first, count, k_to);
}
static int ensure_chunk(struct chmd_p *self, struct chm_p *c, unsigned int k, int need_buf, struct chm_chunk **out)
{
  return ensure_chunk_to(self, c, k, k, need_buf, out);
}

static int interval_clean(const mspack_hip_result *r) {
  return r->err == MSPACK_ERR_OK || (r->err == MSPACK_ERR_READ && (r->flags & MSPACK_HIP_F_LOOKAHEAD_READ));
}

/* the serial span of the virtual decoder: ONE unit from its creation point covering at least `need` bytes */
static int ensure_serial(struct chmd_p *self, struct chm_p *c, off_t need)
{
  struct mspack_system *sys = self->system;
  struct vdec *v = &self->v;
  off_t full = v->length - v->init, cover;
  mspack_hip_unit u;
  uint64_t off = v->in_off;
  unsigned int log_cap;
  size_t log_off;
  if (need > full) need = full;
  if (c->s_valid && c->s_init == v->init && c->s_mode == v->mode && c->s_cover >= need) return MSPACK_ERR_OK;
  cover = need + (off_t) CHM_SERIAL_SLACK * c->interval_bytes;
  cover = (cover + FRAME - 1) & ~(off_t)(FRAME - 1);
  if (cover >= full) cover = full;
  if (cover > 0xFFFF0000LL) return MSPACK_ERR_DATAFORMAT;                /* ours: 32-bit unit length */
  mspack_arena_free(sys, c->s_buf); c->s_buf = NULL; c->s_valid = 0;
  /* (one log entry per reset point the span can reach, the look-ahead frame's included) */
  log_cap = (unsigned int)(cover / c->interval_bytes) + 2u;
  log_off = ((size_t) cover + 32768 + 15) & ~(size_t) 15;
  if (!(c->s_buf = (unsigned char *) mspack_arena_alloc(sys, log_off + 4 + 4 * (size_t) log_cap + 128))) return MSPACK_ERR_NOMEMORY;
  memset(&u, 0, sizeof(u));
  if (off > (uint64_t) c->arena_len) off = c->arena_len;
  u.in_off = off;
  u.in_len = (uint32_t)((uint64_t) c->arena_len - off > 0xFFFFFFF0u ? 0xFFFFFFF0u : (uint64_t) c->arena_len - off);
  u.out_off = 0; u.out_len = (uint32_t) cover;
  u.kind = MSPACK_HIP_KIND_LZX; u.window_bits = (uint8_t) c->window_bits; u.reset_frames = (uint16_t) c->fper;
  u.e8_base = 0;
  u.flags = MSPACK_HIP_UF_LZX_LOG; u.ref_len = log_cap;
  if (mspack_hip_decode_batch(&u, 1, c->arena, c->arena_len + 64, c->s_buf, log_off + 4 + 4 * (size_t) log_cap + 64, &c->s_res)) {
    sys->message(NULL, "GPU batch decode failed: %s", mspack_hip_last_error());
    return MSPACK_ERR_DECRUNCH;
  }
  c->s_log = c->s_buf + log_off + 4; c->s_log_cap = log_cap;
  c->s_log_n = rd_le32(c->s_buf + log_off); if (c->s_log_n > log_cap) c->s_log_n = log_cap;
  c->s_valid = 1; c->s_init = v->init; c->s_mode = v->mode; c->s_cover = cover;
  return MSPACK_ERR_OK;
}

static int vdec_decode(struct chmd_p *self, struct chm_p *c, off_t A, off_t B, off_t *good);

/* One lzxd_decompress(B - A) of the virtual decoder standing at A.  Returns the reference's code; *good =
 * first byte position that was not produced (>= B on success). */
static int vdec_phase(struct chmd_p *self, struct chm_p *c, off_t A, off_t B, off_t *good)
{
  struct vdec *v = &self->v;
  off_t fe = B / FRAME;                         /* last frame the call decodes (lzxd.c:419), absolute index */
  int err;
  *good = B;
  (void) A;
  /* bytes of the frame decoded last are stored up: a call they satisfy decodes nothing (lzxd.c:397-408) */
  if (B <= v->decoded_end) return MSPACK_ERR_OK;
  err = vdec_decode(self, c, v->decoded_end, B, good);
  if (!err) { v->decoded_end = (fe + 1) * FRAME; if (v->decoded_end > v->length) v->decoded_end = v->length; }
  return err;
}

/* the frames from position A (a frame start the decoder has not decoded yet) up to and INCLUDING the frame
 * that holds B -- a call that has to decode always goes one frame past an end on a frame boundary (lzxd.c:419) */
static int vdec_decode(struct chmd_p *self, struct chm_p *c, off_t A, off_t B, off_t *good)
{
  struct vdec *v = &self->v;
  off_t fe = B / FRAME;
  int err;
  if (!v->serial) {
    unsigned int k0 = (unsigned int)(v->init / c->interval_bytes), k, k_hi;
    off_t total_frames = v->length / FRAME;     /* table mode: the length is padded to whole intervals */
    off_t kh = fe / c->fper;
    /* (the intervals of THIS decoder's stream: c->n_intervals, but for the decoder of unknown length, struct vdec) */
    const unsigned int nv = (unsigned int)((v->length + c->interval_bytes - 1) / c->interval_bytes);
    k_hi = (kh >= (off_t) nv) ? nv - 1 : (unsigned int) kh;
    for (k = (unsigned int)(A / c->interval_bytes); k <= k_hi && (off_t) k * c->interval_bytes < v->length; k++) {
      if (k >= c->n_fast) { v->serial = 1; break; }
      if ((err = ensure_chunk(self, c, k, 0, NULL))) return err;
      if (k > k0) {
        const mspack_hip_result *p;
        if ((err = ensure_chunk(self, c, k - 1, 0, NULL))) return err;
        p = &c->ires[k - 1];
        if (!interval_clean(&c->ires[k]) || (p->flags & MSPACK_HIP_F_BLOCK_OPEN) ||
            c->ioff[k] != c->ioff[k - 1] + p->in_next) { v->serial = 1; break; }
      }
      else if (!interval_clean(&c->ires[k])) {
        off_t g = (off_t) k * c->interval_bytes + c->ires[k].good_len;
        if (fe >= g / FRAME) { *good = g; return c->ires[k].err; }
      }
    }
    if (!v->serial) {
      if (fe >= total_frames) {
        /* the request reaches the end of the stream: the reference still enters one more (empty) frame,
         * at a reset point, and reads the header bit there -- which fails only if the input is exhausted */
        if (nv && nv <= c->n_fast) {
          if ((err = ensure_chunk(self, c, nv - 1, 0, NULL))) return err;
          if (c->ires[nv - 1].flags & MSPACK_HIP_F_LOOKAHEAD_READ) { *good = v->length; return MSPACK_ERR_READ; }
        }
        if (B > v->length) { *good = v->length; return MSPACK_ERR_DECRUNCH; }         /* lzxd.c:758-761 */
      }
      return MSPACK_ERR_OK;
    }
  }
  /* serial: the decoder's own unit, positions relative to its creation point */
  {
    off_t full = v->length - v->init, need = (fe + 1) * FRAME - v->init, nframes, rfe = fe - v->init / FRAME, g;
    const mspack_hip_result *r = &c->s_res;
    if ((err = ensure_serial(self, c, need))) return err;
    nframes = (c->s_cover + FRAME - 1) / FRAME;
    {
      /* "invalid reset interval": said once per lzxd_decompress call that enters -- decodes or fails in -- a frame whose
       * reset found a block still open (lzxd.c:423-431; the warning comes before anything of that frame is read) */
      const off_t ra = (A - v->init) / FRAME;
      off_t last = rfe;
      unsigned int i;
      if (r->err && !((off_t) r->good_len >= full) && (off_t)(r->good_len / FRAME) < last) last = (off_t)(r->good_len / FRAME);
      for (i = 0; i < c->s_log_n; i++) {
        const off_t f = (off_t) rd_le32(c->s_log + 4 * (size_t) i);
        if (f >= ra && f <= last) {
          self->system->message(NULL, "WARNING; invalid reset interval detected during LZX decompression");
          break;
        }
      }
    }
    if (c->s_cover < full) {
      /* a partial span is good for the frames it holds completely */
      if ((off_t) r->good_len / FRAME > rfe) return MSPACK_ERR_OK;
      g = v->init + r->good_len; *good = g;
      return r->err ? r->err : MSPACK_ERR_DECRUNCH;
    }
    if (r->err == MSPACK_ERR_OK) {
      if (B > v->length) { *good = v->length; return MSPACK_ERR_DECRUNCH; }
      return MSPACK_ERR_OK;
    }
    if ((off_t) r->good_len >= full) {                 /* everything came out; the look-ahead failed */
      if (rfe < nframes) return MSPACK_ERR_OK;
      *good = v->length; return r->err;
    }
    if (rfe < (off_t)(r->good_len / FRAME)) return MSPACK_ERR_OK;
    *good = v->init + r->good_len;
    return r->err;
  }
}

/* write [from, to) of the uncompressed stream as the virtual decoder produced it */
static int vdec_emit(struct chmd_p *self, struct chm_p *c, struct mspack_file *fh, off_t from, off_t to)
{
  struct mspack_system *sys = self->system;
  struct vdec *v = &self->v;
  if (to <= from) return MSPACK_ERR_OK;
  if (v->serial) return write_slice(sys, fh, c->s_buf + (from - v->init), (size_t)(to - from));
  while (from < to) {
    unsigned int k = (unsigned int)(from / c->interval_bytes), ci = k / c->chunk_int;
    unsigned int cfirst = ci * c->chunk_int, clast = cfirst + c->chunk_int - 1;   /* this chunk's intervals */
    off_t cend = ((off_t) clast + 1) * c->interval_bytes, stop = to < cend ? to : cend;
    unsigned int k1 = (unsigned int)((stop - 1) / c->interval_bytes), i;
    struct chm_chunk *ch;
    int err, shifted = 0;
    if ((err = ensure_chunk_to(self, c, k, k1, 1, &ch))) return err;
    /* E8: fast results have origin 0; the reference's origin is where its decoder was created (lzxd.c:712) */
    if (v->init != 0)
      for (i = k; i <= k1; i++) if (c->ires[i].flags & MSPACK_HIP_F_E8_APPLIED) shifted = 1;
    if (shifted) {
      unsigned int cnt = k1 - k + 1;
      mspack_hip_result *r2 = (mspack_hip_result *) sys->alloc(sys, cnt * sizeof(*r2));
      unsigned char *tmp = (unsigned char *) mspack_arena_alloc(sys, (size_t) cnt * (size_t) c->interval_bytes + 128);
      err = (!r2 || !tmp) ? MSPACK_ERR_NOMEMORY : decode_intervals(self, c, k, cnt, v->init, tmp, r2, NULL);
      if (!err) err = write_slice(sys, fh, tmp + (from - (off_t) k * c->interval_bytes), (size_t)(stop - from));
      sys->free(r2); mspack_arena_free(sys, tmp);
      if (err) return err;
    }
    else if ((err = write_slice(sys, fh, ch->buf + (from - (off_t) cfirst * c->interval_bytes), (size_t)(stop - from)))) return err;
    from = stop;
  }
  return MSPACK_ERR_OK;
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
  if (self->v_chm != c) { self->v_chm = c; self->v.alive = 0; }
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
    struct vdec *v = &self->v;
    int err = MSPACK_ERR_OK;
    if (c->sec1_state == 0) {
      err = setup_sec1(self, c, infh);
      if (err) { free_sec1(sys, c); c->sec1_state = -err; } else c->sec1_state = 1;
    }
    else if (c->sec1_state < 0) err = -c->sec1_state;

    /* (re)create the decoder: none alive, or the request lies behind it (chmd.c:993-999) */
    if (!err && (!v->alive || file->offset < v->offset)) {
      off_t k0 = file->offset / c->interval_bytes;
      v->alive = 0;
      if (file->offset >= 0 && k0 < (off_t) c->n_fast) {
        v->mode = 0; v->init = k0 * c->interval_bytes; v->length = v->dlen = c->padded_len; v->in_off = c->ioff[k0];
        if (v->init > v->dlen) err = MSPACK_ERR_NOMEMORY;            /* lzxd_init(output_length < 0) == NULL (lzxd.c:297-300, chmd.c:1183-1187) */
        else if (v->init == v->dlen) v->length = (off_t) c->n_fast * c->interval_bytes;    /* output_length 0: every interval the table has */
      }
      else if (c->span_err) err = c->span_err;                              /* chmd.c:1159-1166 */
      else { v->mode = 1; v->init = 0; v->length = v->dlen = c->span_len; v->in_off = 0; }
      if (!err) { v->alive = 1; v->serial = (v->mode == 1); v->offset = v->decoded_end = v->init; v->seek_pending = 1; }
    }
    if (!err) {
      if (file->offset > v->dlen) err = MSPACK_ERR_DECRUNCH;                /* chmd.c:1002-1005; the decoder lives on */
      else if (v->seek_pending && sys->seek(infh, c->content_start + (off_t) v->in_off, MSPACK_SYS_SEEK_START))
        err = MSPACK_ERR_SEEK;                                              /* chmd.c:1008-1011; it lives on, too */
      else {
        off_t length = file->length, maxlen = v->dlen - file->offset, good = 0;
        v->seek_pending = 0;
        if (file->offset > v->offset) err = vdec_phase(self, c, v->offset, file->offset, &good);   /* skip */
        if (!err) {
          int werr;
          if (length > maxlen) {
            sys->message(fh, "WARNING; file is %lld bytes longer than compressed section", (long long)(length - maxlen));
            length = maxlen + 1;                        /* decodes what exists, then errors out */
          }
          err = vdec_phase(self, c, file->offset, file->offset + length, &good);                    /* emit */
          if (good > file->offset + length) good = file->offset + length;
          if (err != MSPACK_ERR_NOMEMORY && good > file->offset) {
            if (good > v->length) good = v->length;
            if ((werr = vdec_emit(self, c, fh, file->offset, good))) err = werr;
          }
        }
        if (err) v->alive = 0;                          /* chmd.c:1036-1040 */
        else v->offset = file->offset + length;
      }
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
