/* szdd_kwaj.c -- SZDD and KWAJ drivers of the libmspack-compatible API (include/mspack.h), on the GPU
 * batch decoder.
 *
 * Mirrors the reference's szddd.c and kwajd.c:
 *   SZDD headers (normal "SZDD\x88\xF0\x27\x33" 'A' missing-char length, QBasic "SZ \x88\xF0\x27\x33\xD1"
 *   length) and their error codes ............................................ szddd.c:140-170
 *   KWAJ header, optional fields (length, two unknown fields, 8.3 name parts, extra text) and their
 *   error codes .............................................................. kwajd.c:155-250
 *   open() keeps the input file open until close() (szddd.c:72-107, kwajd.c:93-147)
 *   extract(): seek to the data, open the output, decode, close the output .... szddd.c:177-214, kwajd.c:257-331
 *   methods: SZDD = LZSS (EXPAND or QBASIC start position); KWAJ 0 copy, 1 XOR 0xFF, 2 LZSS (QBASIC),
 *   3 LZH, 4 MSZIP blocks framed by 16-bit lengths
 * The stored methods (KWAJ 0/1) are copies and run on the host like the reference's; everything coded is
 * ONE unit for the batch decoder (MSPACK_HIP_KIND_LZSS / _KWAJ_LZH / MSZIP with MSPACK_HIP_UF_MSZIP_KWAJ).
 * None of these streams carries a length the decoder uses: the unit gets room for the largest possible
 * expansion (LZSS: 18 bytes per 2 input bytes; LZH: 17 bytes per input byte), MSZIP grows on demand.
 */
#include <stdlib.h>
#include <stdio.h>
#include "host_common.h"

#define SZDD_INPUT 2048
#define KWAJ_INPUT 2048

struct szdd_hdr_p { struct msszddd_header base; struct mspack_file *fh; };
struct kwaj_hdr_p { struct mskwajd_header base; struct mspack_file *fh; };
struct szdd_p { struct msszdd_decompressor base; struct mspack_system *system; int error; };
struct kwaj_p { struct mskwaj_decompressor base; struct mspack_system *system; int error; };

/* read everything from the current position to the end of the file; -1 = read error */
static long slurp(struct mspack_system *sys, struct mspack_file *fh, unsigned char **data)
{
  size_t cap = 65536, n = 0;
  unsigned char *buf = (unsigned char *) sys->alloc(sys, cap + 64);
  if (!buf) return -2;
  for (;;) {
    int r;
    if (n + 32768 > cap) {
      unsigned char *nb = (unsigned char *) sys->alloc(sys, cap * 2 + 64);
      if (!nb) { sys->free(buf); return -2; }
      sys->copy(buf, nb, n); sys->free(buf); buf = nb; cap *= 2;
    }
    r = sys->read(fh, buf + n, 32768);
    if (r < 0) { sys->free(buf); return -1; }
    if (r == 0) break;
    n += (size_t) r;
  }
  memset(buf + n, 0, 64);
  *data = buf;
  return (long) n;
}

/* decode ONE coded stream on the GPU and write what it produced */
static int run_unit(struct mspack_system *sys, struct mspack_file *outfh, const unsigned char *in, size_t in_len,
                    int kind, int mode, unsigned int unit_flags, size_t room)
{
  for (;;) {
    mspack_hip_unit u;
    mspack_hip_result r;
    unsigned char *out;
    const size_t below = (kind == MSPACK_HIP_KIND_MSZIP) ? 0 : 4096, slack = (kind == MSPACK_HIP_KIND_MSZIP) ? 32768 : 0;
    int rc, err;
    if (room > 0xFFFF0000u - 65536u) return MSPACK_ERR_NOMEMORY;
    if (!(out = (unsigned char *) sys->alloc(sys, below + room + slack + 64))) return MSPACK_ERR_NOMEMORY;
    memset(&u, 0, sizeof(u)); memset(&r, 0, sizeof(r));
    u.in_off = 0; u.in_len = (uint32_t) in_len;
    u.out_off = below; u.out_len = (uint32_t) room;
    u.kind = (uint8_t) kind; u.window_bits = (uint8_t) mode; u.flags = unit_flags;
    rc = mspack_hip_decode_batch(&u, 1, in, in_len + 48, out, below + room + slack + 64, &r);
    if (rc) {
      sys->message(NULL, "GPU batch decode failed: %s", mspack_hip_last_error());
      sys->free(out);
      return MSPACK_ERR_DECRUNCH;
    }
    if (r.flags & MSPACK_HIP_F_OUT_FULL) { sys->free(out); room *= 4; continue; }     /* MSZIP: more room */
    err = r.err;
    if (write_slice(sys, outfh, out + below, r.out_len > room ? room : r.out_len) && !err) err = MSPACK_ERR_WRITE;
    sys->free(out);
    return err;
  }
}

/* ---- SZDD --------------------------------------------------------------------------------------------- */
static const unsigned char sig_expand[8] = { 0x53, 0x5A, 0x44, 0x44, 0x88, 0xF0, 0x27, 0x33 };
static const unsigned char sig_qbasic[8] = { 0x53, 0x5A, 0x20, 0x88, 0xF0, 0x27, 0x33, 0xD1 };

static int szdd_headers(struct mspack_system *sys, struct mspack_file *fh, struct msszddd_header *hdr)
{
  unsigned char buf[8];
  if (sys->read(fh, buf, 8) != 8) return MSPACK_ERR_READ;
  if (!memcmp(buf, sig_expand, 8)) {
    hdr->format = MSSZDD_FMT_NORMAL;
    if (sys->read(fh, buf, 6) != 6) return MSPACK_ERR_READ;
    if (buf[0] != 0x41) return MSPACK_ERR_DATAFORMAT;
    hdr->missing_char = (char) buf[1];
    hdr->length = (off_t) rd_le32(buf + 2);
  }
  else if (!memcmp(buf, sig_qbasic, 8)) {
    hdr->format = MSSZDD_FMT_QBASIC;
    if (sys->read(fh, buf, 4) != 4) return MSPACK_ERR_READ;
    hdr->missing_char = '\0';
    hdr->length = (off_t) rd_le32(buf);
  }
  else return MSPACK_ERR_SIGNATURE;
  return MSPACK_ERR_OK;
}

static struct msszddd_header *szdd_open(struct msszdd_decompressor *base, const char *filename)
{
  struct szdd_p *self = (struct szdd_p *) base;
  struct mspack_system *sys;
  struct szdd_hdr_p *hdr;
  struct mspack_file *fh;
  if (!self) return NULL;
  sys = self->system;
  fh = sys->open(sys, filename, MSPACK_SYS_OPEN_READ);
  hdr = (struct szdd_hdr_p *) sys->alloc(sys, sizeof(*hdr));
  if (fh && hdr) { hdr->fh = fh; self->error = szdd_headers(sys, fh, &hdr->base); }
  else { if (!fh) self->error = MSPACK_ERR_OPEN; if (!hdr) self->error = MSPACK_ERR_NOMEMORY; }
  if (self->error) { if (fh) sys->close(fh); sys->free(hdr); hdr = NULL; }
  return (struct msszddd_header *) hdr;
}

static void szdd_close(struct msszdd_decompressor *base, struct msszddd_header *hdr)
{
  struct szdd_p *self = (struct szdd_p *) base;
  if (!self || !self->system || !hdr) return;
  self->system->close(((struct szdd_hdr_p *) hdr)->fh);
  self->system->free(hdr);
  self->error = MSPACK_ERR_OK;
}

static int szdd_extract(struct msszdd_decompressor *base, struct msszddd_header *hdr, const char *filename)
{
  struct szdd_p *self = (struct szdd_p *) base;
  struct mspack_system *sys;
  struct mspack_file *fh, *outfh;
  unsigned char *data = NULL;
  long n;
  if (!self) return MSPACK_ERR_ARGS;
  if (!hdr) return self->error = MSPACK_ERR_ARGS;
  sys = self->system;
  fh = ((struct szdd_hdr_p *) hdr)->fh;
  if (sys->seek(fh, (off_t)(hdr->format == MSSZDD_FMT_NORMAL ? 14 : 12), MSPACK_SYS_SEEK_START)) return self->error = MSPACK_ERR_SEEK;
  if (!(outfh = sys->open(sys, filename, MSPACK_SYS_OPEN_WRITE))) return self->error = MSPACK_ERR_OPEN;
  n = slurp(sys, fh, &data);
  if (n == -2) self->error = MSPACK_ERR_NOMEMORY;
  else if (n == -1) self->error = MSPACK_ERR_READ;        /* (the reference fails at the first bad read) */
  else {
    self->error = run_unit(sys, outfh, data, (size_t) n, MSPACK_HIP_KIND_LZSS,
                           hdr->format == MSSZDD_FMT_NORMAL ? 0 : 2, 0, (size_t) n * 9 + 64);
    sys->free(data);
  }
  sys->close(outfh);
  return self->error;
}

static int szdd_decompress(struct msszdd_decompressor *base, const char *input, const char *output)
{
  struct szdd_p *self = (struct szdd_p *) base;
  struct msszddd_header *hdr;
  int error;
  if (!self) return MSPACK_ERR_ARGS;
  if (!(hdr = szdd_open(base, input))) return self->error;
  error = szdd_extract(base, hdr, output);
  szdd_close(base, hdr);
  return self->error = error;
}

static int szdd_error(struct msszdd_decompressor *base) {
  struct szdd_p *self = (struct szdd_p *) base;
  return self ? self->error : MSPACK_ERR_ARGS;
}

struct msszdd_decompressor *mspack_create_szdd_decompressor(struct mspack_system *sys)
{
  struct szdd_p *self;
  if (!sys) sys = mspack_default_system;
  if (!mspack_valid_system(sys)) return NULL;
  if (!(self = (struct szdd_p *) sys->alloc(sys, sizeof(*self)))) return NULL;
  self->base.open = &szdd_open; self->base.close = &szdd_close; self->base.extract = &szdd_extract;
  self->base.decompress = &szdd_decompress; self->base.last_error = &szdd_error;
  self->system = sys; self->error = MSPACK_ERR_OK;
  return &self->base;
}
void mspack_destroy_szdd_decompressor(struct msszdd_decompressor *base) {
  struct szdd_p *self = (struct szdd_p *) base;
  if (self) self->system->free(self);
}

/* ---- KWAJ --------------------------------------------------------------------------------------------- */
/* one 8.3 name part: up to `max` bytes of a NUL-terminated string (kwajd.c:212-238) */
static int kwaj_name_part(struct mspack_system *sys, struct mspack_file *fh, char **fn, int max)
{
  unsigned char buf[16];
  int len, i;
  if ((len = sys->read(fh, buf, max)) < 2) return MSPACK_ERR_READ;
  for (i = 0; i < len; i++) if (!(*(*fn)++ = (char) buf[i])) break;
  if (i == max && buf[max - 1] != '\0') return MSPACK_ERR_DATAFORMAT;
  if (sys->seek(fh, (off_t)(i + 1 - len), MSPACK_SYS_SEEK_CUR)) return MSPACK_ERR_SEEK;
  (*fn)--;                                              /* drop the terminator */
  return MSPACK_ERR_OK;
}

static int kwaj_headers(struct mspack_system *sys, struct mspack_file *fh, struct mskwajd_header *hdr)
{
  unsigned char buf[16];
  int i, err;
  hdr->filename = NULL; hdr->extra = NULL;
  if (sys->read(fh, buf, 14) != 14) return MSPACK_ERR_READ;
  if (rd_le32(buf) != 0x4A41574Bu || rd_le32(buf + 4) != 0xD127F088u) return MSPACK_ERR_SIGNATURE;
  hdr->comp_type = (unsigned short) rd_le16(buf + 8);
  hdr->data_offset = (off_t) rd_le16(buf + 10);
  hdr->headers = (int) rd_le16(buf + 12);
  hdr->length = 0; hdr->extra_length = 0;
  if (hdr->headers & MSKWAJ_HDR_HASLENGTH) {
    if (sys->read(fh, buf, 4) != 4) return MSPACK_ERR_READ;
    hdr->length = (off_t) rd_le32(buf);
  }
  if (hdr->headers & MSKWAJ_HDR_HASUNKNOWN1) { if (sys->read(fh, buf, 2) != 2) return MSPACK_ERR_READ; }
  if (hdr->headers & MSKWAJ_HDR_HASUNKNOWN2) {
    if (sys->read(fh, buf, 2) != 2) return MSPACK_ERR_READ;
    i = (int) rd_le16(buf);
    if (sys->seek(fh, (off_t) i, MSPACK_SYS_SEEK_CUR)) return MSPACK_ERR_SEEK;
  }
  if (hdr->headers & (MSKWAJ_HDR_HASFILENAME | MSKWAJ_HDR_HASFILEEXT)) {
    char *fn = (char *) sys->alloc(sys, 13);
    if (!(hdr->filename = fn)) return MSPACK_ERR_NOMEMORY;
    if (hdr->headers & MSKWAJ_HDR_HASFILENAME) { if ((err = kwaj_name_part(sys, fh, &fn, 9))) return err; }
    if (hdr->headers & MSKWAJ_HDR_HASFILEEXT) {
      *fn++ = '.';
      if ((err = kwaj_name_part(sys, fh, &fn, 4))) return err;
    }
    *fn = '\0';
  }
  if (hdr->headers & MSKWAJ_HDR_HASEXTRATEXT) {
    if (sys->read(fh, buf, 2) != 2) return MSPACK_ERR_READ;
    i = (int) rd_le16(buf);
    if (!(hdr->extra = (char *) sys->alloc(sys, (size_t) i + 1))) return MSPACK_ERR_NOMEMORY;
    if (sys->read(fh, hdr->extra, i) != i) return MSPACK_ERR_READ;
    hdr->extra[i] = '\0';
    hdr->extra_length = (unsigned short) i;
  }
  return MSPACK_ERR_OK;
}

static void kwaj_close(struct mskwaj_decompressor *base, struct mskwajd_header *hdr)
{
  struct kwaj_p *self = (struct kwaj_p *) base;
  if (!self || !self->system || !hdr) return;
  self->system->close(((struct kwaj_hdr_p *) hdr)->fh);
  self->system->free(hdr->filename); self->system->free(hdr->extra);
  self->system->free(hdr);
  self->error = MSPACK_ERR_OK;
}

static struct mskwajd_header *kwaj_open(struct mskwaj_decompressor *base, const char *filename)
{
  struct kwaj_p *self = (struct kwaj_p *) base;
  struct mspack_system *sys;
  struct kwaj_hdr_p *hdr;
  struct mspack_file *fh;
  if (!self) return NULL;
  sys = self->system;
  /* like kwajd_open (kwajd.c:95-123): no file, no header object */
  if (!(fh = sys->open(sys, filename, MSPACK_SYS_OPEN_READ))) { self->error = MSPACK_ERR_OPEN; return NULL; }
  hdr = (struct kwaj_hdr_p *) sys->alloc(sys, sizeof(*hdr));
  if (hdr) {
    memset(hdr, 0, sizeof(*hdr));                 /* filename / extra are freed on every error path */
    hdr->fh = fh; self->error = kwaj_headers(sys, fh, &hdr->base);
  }
  else self->error = MSPACK_ERR_NOMEMORY;
  if (self->error) {
    if (fh) sys->close(fh);
    if (hdr) { sys->free(hdr->base.filename); sys->free(hdr->base.extra); }
    sys->free(hdr);
    hdr = NULL;
  }
  return (struct mskwajd_header *) hdr;
}

static int kwaj_extract(struct mskwaj_decompressor *base, struct mskwajd_header *hdr, const char *filename)
{
  struct kwaj_p *self = (struct kwaj_p *) base;
  struct mspack_system *sys;
  struct mspack_file *fh, *outfh;
  if (!self) return MSPACK_ERR_ARGS;
  if (!hdr) return self->error = MSPACK_ERR_ARGS;
  sys = self->system;
  fh = ((struct kwaj_hdr_p *) hdr)->fh;
  if (sys->seek(fh, hdr->data_offset, MSPACK_SYS_SEEK_START)) return self->error = MSPACK_ERR_SEEK;
  if (!(outfh = sys->open(sys, filename, MSPACK_SYS_OPEN_WRITE))) return self->error = MSPACK_ERR_OPEN;
  self->error = MSPACK_ERR_OK;
  if (hdr->comp_type == MSKWAJ_COMP_NONE || hdr->comp_type == MSKWAJ_COMP_XOR) {
    unsigned char *buf = (unsigned char *) sys->alloc(sys, KWAJ_INPUT);
    if (buf) {
      int rd, i;
      while ((rd = sys->read(fh, buf, KWAJ_INPUT)) > 0) {
        if (hdr->comp_type == MSKWAJ_COMP_XOR) for (i = 0; i < rd; i++) buf[i] ^= 0xFF;
        if (sys->write(outfh, buf, rd) != rd) { self->error = MSPACK_ERR_WRITE; break; }
      }
      if (rd < 0) self->error = MSPACK_ERR_READ;
      sys->free(buf);
    }
    else self->error = MSPACK_ERR_NOMEMORY;
  }
  else if (hdr->comp_type == MSKWAJ_COMP_SZDD || hdr->comp_type == MSKWAJ_COMP_LZH || hdr->comp_type == MSKWAJ_COMP_MSZIP) {
    unsigned char *data = NULL;
    long n = slurp(sys, fh, &data);
    if (n == -2) self->error = MSPACK_ERR_NOMEMORY;
    else if (n == -1) self->error = MSPACK_ERR_READ;
    else {
      if (hdr->comp_type == MSKWAJ_COMP_SZDD)
        self->error = run_unit(sys, outfh, data, (size_t) n, MSPACK_HIP_KIND_LZSS, 2, 0, (size_t) n * 9 + 64);
      else if (hdr->comp_type == MSKWAJ_COMP_LZH)
        self->error = run_unit(sys, outfh, data, (size_t) n, MSPACK_HIP_KIND_KWAJ_LZH, 0, 0, (size_t) n * 18 + 4096);
      else
        self->error = run_unit(sys, outfh, data, (size_t) n, MSPACK_HIP_KIND_MSZIP, 0, MSPACK_HIP_UF_MSZIP_KWAJ,
                               ((size_t) n * 8 + 65536 + 32767) & ~(size_t) 32767);
      sys->free(data);
    }
  }
  else self->error = MSPACK_ERR_DATAFORMAT;
  sys->close(outfh);
  return self->error;
}

static int kwaj_decompress(struct mskwaj_decompressor *base, const char *input, const char *output)
{
  struct kwaj_p *self = (struct kwaj_p *) base;
  struct mskwajd_header *hdr;
  int error;
  if (!self) return MSPACK_ERR_ARGS;
  if (!(hdr = kwaj_open(base, input))) return self->error;
  error = kwaj_extract(base, hdr, output);
  kwaj_close(base, hdr);
  return self->error = error;
}

static int kwaj_error(struct mskwaj_decompressor *base) {
  struct kwaj_p *self = (struct kwaj_p *) base;
  return self ? self->error : MSPACK_ERR_ARGS;
}

struct mskwaj_decompressor *mspack_create_kwaj_decompressor(struct mspack_system *sys)
{
  struct kwaj_p *self;
  if (!sys) sys = mspack_default_system;
  if (!mspack_valid_system(sys)) return NULL;
  if (!(self = (struct kwaj_p *) sys->alloc(sys, sizeof(*self)))) return NULL;
  self->base.open = &kwaj_open; self->base.close = &kwaj_close; self->base.extract = &kwaj_extract;
  self->base.decompress = &kwaj_decompress; self->base.last_error = &kwaj_error;
  self->system = sys; self->error = MSPACK_ERR_OK;
  return &self->base;
}
void mspack_destroy_kwaj_decompressor(struct mskwaj_decompressor *base) {
  struct kwaj_p *self = (struct kwaj_p *) base;
  if (self) self->system->free(self);
}
