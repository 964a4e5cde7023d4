/* cabd.c -- CAB driver of the libmspack-compatible API (include/mspack.h), on the GPU batch decoder.
 *
 * Mirrors the behaviour of the reference's cabd.c for the path open -> extract:
 *   header / folder / file parsing and its error codes ...... cabd.c:317-628
 *   extract() argument checks, in the reference's order ....... cabd.c:1075-1125
 *   CFDATA reader: sizes, reserve areas, checksum .............. cabd.c:1362-1479
 *   codec dispatch by comp_type, Quantum 0xFF trailer, LZX total length .. cabd.c:1226-1269,1327-1340
 *   READ errors of a codec are reported as the feeder's error .. cabd.c:1198,1205
 * but NOT its control flow: the reference streams one folder through one codec instance and
 * re-decodes from block 0 whenever an earlier file is requested (cabd.c:1142-1146).  Here the first
 * extract() that touches a cabinet gathers EVERY folder of that cabinet into one batch (one unit per
 * folder), decodes it on the GPU with mspack_hip_decode_batch[_multi], keeps the decoded folders,
 * and every extract() is then a slice of that result.  A unit's result tells how far it decoded
 * without error; files that end inside that prefix succeed exactly as they do in the reference
 * (which never decodes further than asked), later ones return the unit's error.
 * Not implemented yet: cabinet sets (append/prepend, split blocks).
 */
#include <stdlib.h>
#include <stdio.h>
#include "host_common.h"

#define CAB_BLOCKMAX   32768u
#define CAB_INPUTMAX   (CAB_BLOCKMAX + 6144u)
#define CAB_INPUTMAX_SALVAGE 65535u
#define CAB_LENGTHMAX  (CAB_BLOCKMAX * 65535u)

struct cab_p;
struct folder_p {
  struct mscabd_folder base;
  struct cab_p *cab;
  off_t data_offset;                  /* first CFDATA of this folder                               */
  struct mscabd_file *merge_prev, *merge_next;
  /* decoded state */
  int decoded;
  unsigned char *dec;                 /* decoded bytes (good prefix valid)                         */
  unsigned int total;                 /* sum of the blocks' uncompressed sizes                     */
  unsigned int good_len;              /* bytes that decode without error                           */
  unsigned int n_frames_good;         /* LZX: complete frames in good_len                          */
  int dec_err;                        /* MSPACK_ERR_* of the unit                                  */
  int read_err;                       /* what the feeder would have reported for ERR_READ          */
  int hard_eof;                       /* the block chain ended with a read failure, not cleanly    */
  unsigned int res_flags;
};
struct cab_p {
  struct mscabd_cabinet base;
  int block_resv;
};
struct cabd_p {
  struct mscab_decompressor base;
  struct mspack_system *system;
  int error, read_error;
  int searchbuf_size, fix_mszip, buf_size, salvage;
  int devices, cache_mb;
};

/* ---- small helpers -------------------------------------------------------------------------------- */
static unsigned int cab_checksum(const unsigned char *data, unsigned int bytes, unsigned int cksum) {
  unsigned int n = bytes >> 2, tail = 0;
  while (n--) { cksum ^= rd_le32(data); data += 4; }
  switch (bytes & 3) {
  case 3: tail |= (unsigned int) *data++ << 16; /* fall through */
  case 2: tail |= (unsigned int) *data++ << 8;  /* fall through */
  case 1: tail |= *data;
  }
  return cksum ^ tail;
}

static char *read_cstring(struct mspack_system *sys, struct mspack_file *fh, int permit_empty, int *error) {
  off_t base = sys->tell(fh);
  char buf[256], *str;
  int len = sys->read(fh, buf, 256), i;
  if (len <= 0) { *error = MSPACK_ERR_READ; return NULL; }
  for (i = 0; i < len && buf[i]; i++) ;
  if (i == len || (i == 0 && !permit_empty)) { *error = MSPACK_ERR_DATAFORMAT; return NULL; }
  if (sys->seek(fh, base + (off_t)(i + 1), MSPACK_SYS_SEEK_START)) { *error = MSPACK_ERR_SEEK; return NULL; }
  if (!(str = (char *) sys->alloc(sys, (size_t) i + 1))) { *error = MSPACK_ERR_NOMEMORY; return NULL; }
  sys->copy(buf, str, (size_t) i + 1);
  *error = MSPACK_ERR_OK;
  return str;
}

static void free_folder_cache(struct mspack_system *sys, struct folder_p *f) {
  sys->free(f->dec); f->dec = NULL; f->decoded = 0;
}

/* ---- headers (reference cabd.c:317-628) ------------------------------------------------------------- */
static int read_files(struct mspack_system *sys, struct mspack_file *fh, struct cab_p *cab,
                      int num_folders, int num_files, int salvage)
{
  struct mscabd_file *tail = NULL;
  unsigned char buf[16];
  int i, err;
  for (i = 0; i < num_files; i++) {
    struct mscabd_file *f;
    unsigned int fidx, x;
    if (sys->read(fh, buf, 16) != 16) return MSPACK_ERR_READ;
    if (!(f = (struct mscabd_file *) sys->alloc(sys, sizeof(*f)))) return MSPACK_ERR_NOMEMORY;
    f->next = NULL;
    f->length = rd_le32(buf);
    f->offset = rd_le32(buf + 4);
    fidx = rd_le16(buf + 8);
    f->attribs = (int) rd_le16(buf + 14);
    f->folder = NULL;
    if (fidx < 0xFFFD) {
      if ((int) fidx < num_folders) {
        struct mscabd_folder *fo = cab->base.folders;
        while (fidx-- && fo) fo = fo->next;
        f->folder = fo;
      }
    }
    else {
      struct folder_p *fp;
      if (fidx == 0xFFFE || fidx == 0xFFFF) {            /* continued to next: lives in the last folder */
        struct mscabd_folder *fo = cab->base.folders;
        while (fo->next) fo = fo->next;
        f->folder = fo; fp = (struct folder_p *) fo;
        if (!fp->merge_next) fp->merge_next = f;
      }
      if (fidx == 0xFFFD || fidx == 0xFFFF) {            /* continued from previous: first folder       */
        f->folder = cab->base.folders; fp = (struct folder_p *) f->folder;
        if (!fp->merge_prev) fp->merge_prev = f;
      }
    }
    x = rd_le16(buf + 12);
    f->time_h = (char)(x >> 11); f->time_m = (char)((x >> 5) & 0x3F); f->time_s = (char)((x << 1) & 0x3E);
    x = rd_le16(buf + 10);
    f->date_d = (char)(x & 0x1F); f->date_m = (char)((x >> 5) & 0xF); f->date_y = (int)(x >> 9) + 1980;
    f->filename = read_cstring(sys, fh, 0, &err);
    if (err || !f->folder) {
      sys->free(f->filename); sys->free(f);
      if (salvage) continue;
      return err ? err : MSPACK_ERR_DATAFORMAT;
    }
    if (tail) tail->next = f; else cab->base.files = f;
    tail = f;
  }
  return MSPACK_ERR_OK;
}

static int read_headers(struct mspack_system *sys, struct mspack_file *fh, struct cab_p *cab,
                        off_t offset, int salvage, int quiet)
{
  unsigned char buf[64];
  int num_folders, num_files, folder_resv = 0, i, err;
  struct folder_p *fol = NULL, *tail = NULL;
  off_t files_off_hdr, files_off_real;

  cab->base.next = NULL; cab->base.files = NULL; cab->base.folders = NULL;
  cab->base.prevcab = cab->base.nextcab = NULL;
  cab->base.prevname = cab->base.nextname = cab->base.previnfo = cab->base.nextinfo = NULL;
  cab->base.base_offset = offset;
  cab->block_resv = 0;

  if (sys->seek(fh, offset, MSPACK_SYS_SEEK_START)) return MSPACK_ERR_SEEK;
  if (sys->read(fh, buf, 0x24) != 0x24) return MSPACK_ERR_READ;
  if (rd_le32(buf) != 0x4643534Du) return MSPACK_ERR_SIGNATURE;
  cab->base.length = rd_le32(buf + 0x08);
  cab->base.set_id = (unsigned short) rd_le16(buf + 0x20);
  cab->base.set_index = (unsigned short) rd_le16(buf + 0x22);
  files_off_hdr = (off_t) rd_le32(buf + 0x10);
  num_folders = (int) rd_le16(buf + 0x1A);
  if (num_folders == 0) { if (!quiet) sys->message(fh, "no folders in cabinet."); return MSPACK_ERR_DATAFORMAT; }
  num_files = (int) rd_le16(buf + 0x1C);
  if (num_files == 0) { if (!quiet) sys->message(fh, "no files in cabinet."); return MSPACK_ERR_DATAFORMAT; }
  if (buf[0x19] != 1 && buf[0x18] != 3) { if (!quiet) sys->message(fh, "WARNING; cabinet version is not 1.3"); }
  cab->base.flags = (int) rd_le16(buf + 0x1E);
  cab->base.header_resv = 0;
  if (cab->base.flags & MSCAB_HDR_RESV) {
    if (sys->read(fh, buf, 4) != 4) return MSPACK_ERR_READ;
    cab->base.header_resv = (unsigned short) rd_le16(buf);
    folder_resv = buf[2];
    cab->block_resv = buf[3];
    if (cab->base.header_resv > 60000 && !quiet) sys->message(fh, "WARNING; reserved header > 60000.");
    if (cab->base.header_resv && sys->seek(fh, (off_t) cab->base.header_resv, MSPACK_SYS_SEEK_CUR)) return MSPACK_ERR_SEEK;
  }
  if (cab->base.flags & MSCAB_HDR_PREVCAB) {
    cab->base.prevname = read_cstring(sys, fh, 0, &err); if (err) return err;
    cab->base.previnfo = read_cstring(sys, fh, 1, &err); if (err) return err;
  }
  if (cab->base.flags & MSCAB_HDR_NEXTCAB) {
    cab->base.nextname = read_cstring(sys, fh, 0, &err); if (err) return err;
    cab->base.nextinfo = read_cstring(sys, fh, 1, &err); if (err) return err;
  }
  for (i = 0; i < num_folders; i++) {
    if (sys->read(fh, buf, 8) != 8) return MSPACK_ERR_READ;
    if (folder_resv && sys->seek(fh, (off_t) folder_resv, MSPACK_SYS_SEEK_CUR)) return MSPACK_ERR_SEEK;
    if (!(fol = (struct folder_p *) sys->alloc(sys, sizeof(*fol)))) return MSPACK_ERR_NOMEMORY;
    memset(fol, 0, sizeof(*fol));
    fol->base.comp_type = (int) rd_le16(buf + 6);
    fol->base.num_blocks = rd_le16(buf + 4);
    fol->cab = cab;
    fol->data_offset = offset + (off_t) rd_le32(buf);
    if (tail) tail->base.next = &fol->base; else cab->base.folders = &fol->base;
    tail = fol;
  }
  files_off_real = sys->tell(fh) - cab->base.base_offset;
  err = read_files(sys, fh, cab, num_folders, num_files, salvage);
  if (files_off_real != files_off_hdr) {
    if (!quiet) sys->message(fh, "WARNING; atypical files offset in header");
    if (salvage && files_off_hdr < (off_t) cab->base.length &&
        !sys->seek(fh, files_off_hdr + cab->base.base_offset, MSPACK_SYS_SEEK_START)) {
      struct mscabd_file *first = cab->base.files, *second;
      int err2 = read_files(sys, fh, cab, num_folders, num_files, salvage);
      second = cab->base.files;
      if (first && first != second) {
        struct mscabd_file *e = first;
        while (e->next) e = e->next;
        e->next = second; cab->base.files = first;
      }
      err = err ? err : err2;
    }
  }
  if (err) {
    if (salvage && cab->base.files) { if (!quiet) sys->message(fh, "WARNING; ignoring error %d while salvaging", err); }
    else return err;
  }
  if (!cab->base.files) return MSPACK_ERR_DATAFORMAT;
  return MSPACK_ERR_OK;
}

/* ---- public methods ------------------------------------------------------------------------------------ */
static void cabd_close(struct mscab_decompressor *base, struct mscabd_cabinet *origcab)
{
  struct cabd_p *self = (struct cabd_p *) base;
  struct mspack_system *sys;
  if (!self) return;
  sys = self->system;
  self->error = MSPACK_ERR_OK;
  while (origcab) {
    struct mscabd_file *fi, *nfi;
    struct mscabd_folder *fo, *nfo;
    struct mscabd_cabinet *nextc = origcab->next;
    for (fi = origcab->files; fi; fi = nfi) { nfi = fi->next; sys->free(fi->filename); sys->free(fi); }
    for (fo = origcab->folders; fo; fo = nfo) {
      nfo = fo->next;
      free_folder_cache(sys, (struct folder_p *) fo);
      sys->free(fo);
    }
    sys->free(origcab->prevname); sys->free(origcab->nextname);
    sys->free(origcab->previnfo); sys->free(origcab->nextinfo);
    sys->free(origcab);
    origcab = nextc;
  }
}

static struct mscabd_cabinet *cabd_open(struct mscab_decompressor *base, const char *filename)
{
  struct cabd_p *self = (struct cabd_p *) base;
  struct mspack_system *sys;
  struct mspack_file *fh;
  struct cab_p *cab = NULL;
  if (!self) return NULL;
  sys = self->system;
  if (!(fh = sys->open(sys, filename, MSPACK_SYS_OPEN_READ))) { self->error = MSPACK_ERR_OPEN; return NULL; }
  if ((cab = (struct cab_p *) sys->alloc(sys, sizeof(*cab)))) {
    int err;
    memset(cab, 0, sizeof(*cab));
    cab->base.filename = filename;
    err = read_headers(sys, fh, cab, (off_t) 0, self->salvage, 0);
    if (err) { cabd_close(base, &cab->base); cab = NULL; }
    self->error = err;
  }
  else self->error = MSPACK_ERR_NOMEMORY;
  sys->close(fh);
  return (struct mscabd_cabinet *) cab;
}

/* search: scan a file for embedded cabinets (reference cabd.c:656-868, simplified: every "MSCF"
 * whose header parses is returned, in file order) */
static struct mscabd_cabinet *cabd_search(struct mscab_decompressor *base, const char *filename)
{
  struct cabd_p *self = (struct cabd_p *) base;
  struct mspack_system *sys;
  struct mspack_file *fh;
  struct mscabd_cabinet *head = NULL, *tail = NULL;
  unsigned char *buf;
  off_t flen = 0, pos = 0;
  int bufsz;
  if (!self) return NULL;
  sys = self->system;
  bufsz = self->searchbuf_size;
  if (!(buf = (unsigned char *) sys->alloc(sys, (size_t) bufsz + 4))) { self->error = MSPACK_ERR_NOMEMORY; return NULL; }
  if (!(fh = sys->open(sys, filename, MSPACK_SYS_OPEN_READ))) { sys->free(buf); self->error = MSPACK_ERR_OPEN; return NULL; }
  self->error = MSPACK_ERR_OK;
  if (mspack_sys_filelen(sys, fh, &flen) == MSPACK_ERR_OK) {
    while (pos < flen) {
      int n, i;
      off_t skip_to = -1;
      if (sys->seek(fh, pos, MSPACK_SYS_SEEK_START)) { self->error = MSPACK_ERR_SEEK; break; }
      n = sys->read(fh, buf, bufsz);
      if (n < 4) break;
      for (i = 0; i + 4 <= n; i++) {
        if (buf[i] == 'M' && buf[i + 1] == 'S' && buf[i + 2] == 'C' && buf[i + 3] == 'F') {
          struct cab_p *cab = (struct cab_p *) sys->alloc(sys, sizeof(*cab));
          if (!cab) { self->error = MSPACK_ERR_NOMEMORY; break; }
          memset(cab, 0, sizeof(*cab));
          cab->base.filename = filename;
          if (read_headers(sys, fh, cab, pos + i, self->salvage, 1) == MSPACK_ERR_OK) {
            if (tail) tail->next = &cab->base; else head = &cab->base;
            tail = &cab->base;
            skip_to = pos + i + (off_t)(cab->base.length ? cab->base.length : 4);
            break;
          }
          cabd_close(base, &cab->base);
          self->error = MSPACK_ERR_OK;
        }
      }
      if (self->error) break;
      if (skip_to >= 0) pos = skip_to;
      else pos += (n >= 4) ? (n - 3) : n;
    }
  }
  sys->close(fh);
  sys->free(buf);
  return head;
}

static int cabd_merge_unsupported(struct mscab_decompressor *base, struct mscabd_cabinet *a, struct mscabd_cabinet *b)
{
  struct cabd_p *self = (struct cabd_p *) base;
  (void) a; (void) b;
  if (!self) return MSPACK_ERR_ARGS;
  self->system->message(NULL, "cabinet sets (append/prepend) are not supported by this build yet");
  return self->error = MSPACK_ERR_ARGS;
}

/* ---- gather + batch decode ------------------------------------------------------------------------------ */
struct gathered {
  struct folder_p *fol;
  unsigned char *stream; size_t len, cap;      /* codec input: payloads (+0xFF per block for Quantum) */
  unsigned int total;                           /* sum of uncompressed sizes of the blocks read     */
  int read_err; int hard_eof;
};

/* walk the CFDATA chain of one folder (reference cabd.c:1283-1345 + 1362-1459) */
static int gather_folder(struct cabd_p *self, struct mspack_file *fh, struct gathered *g)
{
  struct mspack_system *sys = self->system;
  struct folder_p *fol = g->fol;
  const int method = fol->base.comp_type & 0x0F;
  const int ignore_cksum = self->salvage || (self->fix_mszip && method == MSCAB_COMP_MSZIP);
  const int ignore_size = self->salvage;
  unsigned int b;
  g->len = 0; g->total = 0; g->read_err = MSPACK_ERR_OK; g->hard_eof = 0;
  g->cap = (size_t) fol->base.num_blocks * 1024 + 65536;
  if (!(g->stream = (unsigned char *) sys->alloc(sys, g->cap + 64))) return MSPACK_ERR_NOMEMORY;
  if (sys->seek(fh, fol->data_offset, MSPACK_SYS_SEEK_START)) { g->read_err = MSPACK_ERR_SEEK; g->hard_eof = 1; return MSPACK_ERR_OK; }
  for (b = 0; b < fol->base.num_blocks; b++) {
    unsigned char hdr[8];
    unsigned int len, ulen, cksum;
    int err = MSPACK_ERR_OK;
    if (sys->read(fh, hdr, 8) != 8) err = MSPACK_ERR_READ;
    else if (fol->cab->block_resv && sys->seek(fh, (off_t) fol->cab->block_resv, MSPACK_SYS_SEEK_CUR)) err = MSPACK_ERR_SEEK;
    if (!err) {
      len = rd_le16(hdr + 4); ulen = rd_le16(hdr + 6);
      if (len > CAB_INPUTMAX && (!ignore_size || len > CAB_INPUTMAX_SALVAGE)) err = MSPACK_ERR_DATAFORMAT;
      else if (ulen > CAB_BLOCKMAX && !ignore_size) err = MSPACK_ERR_DATAFORMAT;
    }
    if (!err) {
      if (g->len + len + 1 > g->cap) {
        size_t ncap = (g->cap + len + 1) * 2;
        unsigned char *n = (unsigned char *) sys->alloc(sys, ncap + 64);
        if (!n) { sys->free(g->stream); g->stream = NULL; return MSPACK_ERR_NOMEMORY; }
        sys->copy(g->stream, n, g->len); sys->free(g->stream); g->stream = n; g->cap = ncap;
      }
      if (sys->read(fh, g->stream + g->len, (int) len) != (int) len) err = MSPACK_ERR_READ;
    }
    if (!err && (cksum = rd_le32(hdr))) {
      unsigned int sum = cab_checksum(g->stream + g->len, len, 0);
      if (cab_checksum(hdr + 4, 4, sum) != cksum) {
        if (!ignore_cksum) err = MSPACK_ERR_CHECKSUM;
        else sys->message(fh, "WARNING; bad block checksum found");
      }
    }
    if (!err && ulen == 0) {
      /* a block split over the next cabinet of a set: sets are not supported yet */
      sys->message(fh, "WARNING; ran out of cabinets in set. Are any missing?");
      err = MSPACK_ERR_DATAFORMAT;
    }
    if (err) { g->read_err = err; g->hard_eof = 1; break; }
    g->len += len;
    if (method == MSCAB_COMP_QUANTUM) g->stream[g->len++] = 0xFF;
    g->total += ulen;
  }
  if (!g->hard_eof) g->read_err = self->salvage ? MSPACK_ERR_OK : MSPACK_ERR_DATAFORMAT;  /* ran out of blocks */
  else {
    /* the codec pulls buf_size bytes per read; a read that reaches the bad block fails as a whole,
     * so everything from the start of that read on is lost (cabd.c:1297-1324) */
    size_t q = (size_t)((self->buf_size + 1) & ~1);
    g->len -= g->len % q;
  }
  memset(g->stream + g->len, 0, 64);
  return MSPACK_ERR_OK;
}

/* decode every not-yet-decoded folder of `cab` (budget permitting, `want` always) in ONE batch */
static int decode_cabinet(struct cabd_p *self, struct cab_p *cab, struct folder_p *want)
{
  struct mspack_system *sys = self->system;
  struct mspack_file *fh;
  struct mscabd_folder *fo;
  struct gathered *gs;
  mspack_hip_unit *units;
  mspack_hip_result *res;
  unsigned char *in_arena = NULL, *out_arena = NULL;
  size_t n = 0, k, in_bytes = 0, out_bytes = 0, budget = (size_t) self->cache_mb << 20, used = 0;
  int err = MSPACK_ERR_OK, rc;

  for (fo = cab->base.folders; fo; fo = fo->next) n++;
  gs = (struct gathered *) sys->alloc(sys, n * sizeof(*gs));
  units = (mspack_hip_unit *) sys->alloc(sys, n * sizeof(*units));
  res = (mspack_hip_result *) sys->alloc(sys, n * sizeof(*res));
  if (!gs || !units || !res) { sys->free(gs); sys->free(units); sys->free(res); return MSPACK_ERR_NOMEMORY; }
  if (!(fh = sys->open(sys, cab->base.filename, MSPACK_SYS_OPEN_READ))) {
    sys->free(gs); sys->free(units); sys->free(res); return MSPACK_ERR_OPEN;
  }
  n = 0;
  for (fo = cab->base.folders; fo; fo = fo->next) {
    struct folder_p *fp = (struct folder_p *) fo;
    size_t est = (size_t) fo->num_blocks * CAB_BLOCKMAX;
    if (fp->decoded) continue;
    if (fp != want && used + est > budget) continue;
    used += est;
    gs[n].fol = fp;
    if ((err = gather_folder(self, fh, &gs[n]))) break;
    n++;
  }
  sys->close(fh);
  if (err) { for (k = 0; k < n; k++) sys->free(gs[k].stream); sys->free(gs); sys->free(units); sys->free(res); return err; }

  /* lay the units out in two arenas */
  memset(units, 0, n * sizeof(*units));
  for (k = 0; k < n; k++) {
    struct folder_p *fp = gs[k].fol;
    int method = fp->base.comp_type & 0x0F;
    in_bytes = (in_bytes + 15) & ~(size_t) 15;
    units[k].in_off = in_bytes; units[k].in_len = (uint32_t) gs[k].len;
    in_bytes += gs[k].len;
    units[k].out_off = out_bytes; units[k].out_len = gs[k].total;
    out_bytes += ((size_t) gs[k].total + 32768 + 15) & ~(size_t) 15;
    units[k].kind = (uint8_t) method;
    units[k].window_bits = (uint8_t)((fp->base.comp_type >> 8) & 0x1F);
    units[k].reset_frames = 0; units[k].e8_base = 0;
    units[k].flags = (gs[k].hard_eof ? MSPACK_HIP_UF_HARD_EOF : 0) |
                     ((self->fix_mszip && method == MSCAB_COMP_MSZIP) ? MSPACK_HIP_UF_MSZIP_REPAIR : 0);
  }
  in_arena = (unsigned char *) sys->alloc(sys, in_bytes + 64);
  out_arena = (unsigned char *) sys->alloc(sys, out_bytes + 64);
  if (!in_arena || !out_arena) err = MSPACK_ERR_NOMEMORY;
  else {
    size_t nhip = 0;
    memset(in_arena, 0, in_bytes + 64);
    for (k = 0; k < n; k++) sys->copy(gs[k].stream, in_arena + units[k].in_off, gs[k].len);
    /* stored folders need no codec: their payloads ARE the data (cabd.c:1505-1556) */
    for (k = 0; k < n; k++) if (units[k].kind >= 1 && units[k].kind <= 3) nhip++;
    memset(res, 0, n * sizeof(*res));
    if (nhip) {
      /* kinds other than 1..3 are answered with MSPACK_ERR_ARGS by the kernels; fix them up below */
      rc = (self->devices > 1)
        ? mspack_hip_decode_batch_multi(units, n, in_arena, in_bytes + 64, out_arena, out_bytes + 64, res, self->devices)
        : mspack_hip_decode_batch(units, n, in_arena, in_bytes + 64, out_arena, out_bytes + 64, res);
      if (rc) {
        sys->message(NULL, "GPU batch decode failed: %s", mspack_hip_last_error());
        err = MSPACK_ERR_DECRUNCH;
      }
    }
  }
  if (!err) {
    for (k = 0; k < n; k++) {
      struct folder_p *fp = gs[k].fol;
      int method = fp->base.comp_type & 0x0F;
      fp->total = gs[k].total; fp->read_err = gs[k].read_err; fp->hard_eof = gs[k].hard_eof;
      if (!(fp->dec = (unsigned char *) sys->alloc(sys, (size_t) gs[k].total + 1))) { err = MSPACK_ERR_NOMEMORY; break; }
      if (method == MSCAB_COMP_NONE) {
        size_t m = gs[k].len < gs[k].total ? gs[k].len : gs[k].total;
        sys->copy(gs[k].stream, fp->dec, m);
        fp->good_len = (unsigned int) m;
        fp->dec_err = (m == gs[k].total && !gs[k].hard_eof) ? MSPACK_ERR_OK : MSPACK_ERR_READ;
        fp->res_flags = 0;
      }
      else if (method >= 1 && method <= 3) {
        unsigned int g = res[k].good_len > gs[k].total ? gs[k].total : res[k].good_len;
        sys->copy(out_arena + units[k].out_off, fp->dec, g);
        fp->good_len = g; fp->dec_err = res[k].err; fp->res_flags = res[k].flags;
      }
      else { fp->good_len = 0; fp->dec_err = MSPACK_ERR_DATAFORMAT; fp->res_flags = 0; }   /* cabd.c:1254 */
      fp->n_frames_good = fp->good_len / CAB_BLOCKMAX;
      fp->decoded = 1;
    }
  }
  for (k = 0; k < n; k++) sys->free(gs[k].stream);
  sys->free(gs); sys->free(units); sys->free(res); sys->free(in_arena); sys->free(out_arena);
  return err;
}

/* can the reference produce bytes [0, end) of this folder, and with which error if not? */
static int folder_status(struct folder_p *fp, unsigned int end, int read_error)
{
  int method = fp->base.comp_type & 0x0F, ok;
  if (fp->dec_err == MSPACK_ERR_OK && end > fp->total) {
    /* the request goes past everything the blocks hold.  After a failed block read the codec's
     * next refill fails (-> the feeder's error); after a clean end LZX knows the stream length and
     * gives up with DECRUNCH (lzxd.c:458-461,758-761), the others run into the end of input */
    if (fp->hard_eof) return read_error;
    return (method == MSCAB_COMP_LZX) ? MSPACK_ERR_DECRUNCH : read_error;
  }
  if (method == MSCAB_COMP_LZX) {
    /* lzxd decodes every frame up to and including frame end/32768 (one frame of look-ahead when
     * `end` is a frame multiple, lzxd.c:419) */
    unsigned int need_f = end / CAB_BLOCKMAX, nframes = (fp->total + CAB_BLOCKMAX - 1) / CAB_BLOCKMAX;
    if (fp->dec_err == MSPACK_ERR_OK) ok = 1;
    else if (fp->good_len >= fp->total) ok = (need_f < nframes);          /* only the look-ahead failed */
    else ok = (need_f < fp->n_frames_good);
  }
  else ok = (end <= fp->good_len);
  if (ok) return MSPACK_ERR_OK;
  if (fp->dec_err == MSPACK_ERR_OK) return read_error;
  return (fp->dec_err == MSPACK_ERR_READ) ? read_error : fp->dec_err;
}

static int cabd_extract(struct mscab_decompressor *base, struct mscabd_file *file, const char *filename)
{
  struct cabd_p *self = (struct cabd_p *) base;
  struct mspack_system *sys;
  struct mspack_file *fh;
  struct folder_p *fol;
  unsigned int filelen;

  if (!self) return MSPACK_ERR_ARGS;
  if (!file) return self->error = MSPACK_ERR_ARGS;
  sys = self->system;
  fol = (struct folder_p *) file->folder;

  /* the reference's argument checks, in its order (cabd.c:1090-1125) */
  if (file->offset > CAB_LENGTHMAX) return self->error = MSPACK_ERR_DATAFORMAT;
  filelen = file->length;
  if (filelen > (CAB_LENGTHMAX - file->offset)) {
    if (self->salvage) filelen = CAB_LENGTHMAX - file->offset;
    else return self->error = MSPACK_ERR_DATAFORMAT;
  }
  if (!fol || fol->merge_prev) {
    sys->message(NULL, "ERROR; file \"%s\" cannot be extracted, cabinet set is incomplete", file->filename);
    return self->error = MSPACK_ERR_DECRUNCH;
  }
  if (!self->salvage) {
    unsigned int maxlen = fol->base.num_blocks * CAB_BLOCKMAX;
    if (file->offset > maxlen || filelen > (maxlen - file->offset)) {
      sys->message(NULL, "ERROR; file \"%s\" cannot be extracted, cabinet set is incomplete", file->filename);
      return self->error = MSPACK_ERR_DECRUNCH;
    }
  }
  switch (fol->base.comp_type & 0x0F) {
  case MSCAB_COMP_NONE: case MSCAB_COMP_MSZIP: case MSCAB_COMP_QUANTUM: case MSCAB_COMP_LZX: break;
  default: return self->error = MSPACK_ERR_DATAFORMAT;                   /* cabd.c:1254 */
  }
  {
    int wb = (fol->base.comp_type >> 8) & 0x1F, m = fol->base.comp_type & 0x0F;
    /* lzxd_init / qtmd_init refuse these windows -> MSPACK_ERR_NOMEMORY (cabd.c:1256) */
    if ((m == MSCAB_COMP_LZX && (wb < 15 || wb > 21)) || (m == MSCAB_COMP_QUANTUM && (wb < 10 || wb > 21)))
      return self->error = MSPACK_ERR_NOMEMORY;
  }

  if (!fol->decoded) {
    int err = decode_cabinet(self, fol->cab, fol);
    if (err) return self->error = err;
  }
  self->read_error = fol->read_err;

  if (!(fh = sys->open(sys, filename, MSPACK_SYS_OPEN_WRITE))) return self->error = MSPACK_ERR_OPEN;
  self->error = MSPACK_ERR_OK;
  if (filelen) {
    /* skip phase: getting to file->offset must itself be error free (cabd.c:1195-1199) */
    int err = file->offset ? folder_status(fol, file->offset, self->read_error) : MSPACK_ERR_OK;
    if (!err) {
      unsigned int end = file->offset + filelen;
      unsigned int have = fol->good_len > file->offset ? fol->good_len - file->offset : 0;
      if (have > filelen) have = filelen;
      err = folder_status(fol, end, self->read_error);
      if (write_slice(sys, fh, fol->dec + file->offset, have) != MSPACK_ERR_OK) err = MSPACK_ERR_WRITE;
    }
    self->error = err;
  }
  sys->close(fh);
  return self->error;
}

static int cabd_param(struct mscab_decompressor *base, int param, int value)
{
  struct cabd_p *self = (struct cabd_p *) base;
  if (!self) return MSPACK_ERR_ARGS;
  switch (param) {
  case MSCABD_PARAM_SEARCHBUF: if (value < 4) return MSPACK_ERR_ARGS; self->searchbuf_size = value; break;
  case MSCABD_PARAM_FIXMSZIP:  self->fix_mszip = value; break;
  case MSCABD_PARAM_DECOMPBUF: if (value < 4) return MSPACK_ERR_ARGS; self->buf_size = value; break;
  case MSCABD_PARAM_SALVAGE:   self->salvage = value; break;
  case MSCABD_PARAM_HIP_DEVICES:  if (value < 1) return MSPACK_ERR_ARGS; self->devices = value; break;
  case MSCABD_PARAM_HIP_CACHE_MB: if (value < 1) return MSPACK_ERR_ARGS; self->cache_mb = value; break;
  default: return MSPACK_ERR_ARGS;
  }
  return MSPACK_ERR_OK;
}

static int cabd_error(struct mscab_decompressor *base) {
  struct cabd_p *self = (struct cabd_p *) base;
  return self ? self->error : MSPACK_ERR_ARGS;
}

struct mscab_decompressor *mspack_create_cab_decompressor(struct mspack_system *sys)
{
  struct cabd_p *self;
  if (!sys) sys = mspack_default_system;
  if (!mspack_valid_system(sys)) return NULL;
  if (!(self = (struct cabd_p *) sys->alloc(sys, sizeof(*self)))) return NULL;
  self->base.open = &cabd_open;
  self->base.close = &cabd_close;
  self->base.search = &cabd_search;
  self->base.extract = &cabd_extract;
  self->base.prepend = &cabd_merge_unsupported;
  self->base.append = &cabd_merge_unsupported;
  self->base.set_param = &cabd_param;
  self->base.last_error = &cabd_error;
  self->system = sys;
  self->error = MSPACK_ERR_OK; self->read_error = MSPACK_ERR_OK;
  self->searchbuf_size = 32768; self->fix_mszip = 0; self->buf_size = 4096; self->salvage = 0;
  self->devices = 1; self->cache_mb = 2048;
  return &self->base;
}

void mspack_destroy_cab_decompressor(struct mscab_decompressor *base)
{
  struct cabd_p *self = (struct cabd_p *) base;
  if (self) self->system->free(self);
}
