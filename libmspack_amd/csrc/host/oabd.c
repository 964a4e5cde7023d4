/* oabd.c -- OAB (Offline Address Book, ".LZX") driver of the libmspack-compatible API, on the GPU batch
 * decoder.
 *
 * Mirrors the behaviour of the reference's oabd.c:
 *   full files ....... header (version 3.1, block_max, target_size), then blocks of 16-byte header
 *                      (flags, compressed size, uncompressed size, CRC) + data .... oabd.c:103-237
 *   patch files ...... header (version 3.2), blocks (patch size, target size, source size, CRC); every
 *                      block is an LZX DELTA stream whose reference data is the next `source size`
 *                      bytes of the base file ..................................... oabd.c:239-382
 *   window per block . 17 bits, grown until it holds the block (plus the rounded-up source) .. :191-194,:329-334
 *   CRC .............. reflected CRC-32 (edb88320) from 0xffffffff, NOT inverted at the end,
 *                      over the bytes written .................................... oabd.c:88-100, crc32.h
 * but not its control flow: every compressed block is its own lzxd_init (oabd.c:199, :340), i.e. an
 * independent unit, so ALL blocks of a file are decoded in ONE GPU batch (MSPACK_HIP_KIND_LZX_DELTA) and
 * the file is then replayed in order: each block's bytes are written, its error (if any) returned at the
 * point where the reference would have returned it.
 */
#include <stdlib.h>
#include <stdio.h>
#include "host_common.h"

struct oabd_p {
  struct msoab_decompressor base;
  struct mspack_system *system;
  int buf_size;
};

struct oab_blk {
  unsigned int csize, dsize, ssize, crc;
  int compressed;
  int window_bits;
  size_t in_pos, in_have;             /* compressed / stored bytes in the arena; how many the file held */
  int ref_err;                        /* patch: error while fetching the reference data              */
  size_t unit;                        /* index into the unit table (compressed blocks)                */
};

static unsigned int crc_table[256];
static void crc_init(void) {
  unsigned int i, k;
  if (crc_table[1]) return;
  for (i = 0; i < 256; i++) {
    unsigned int c = i;
    for (k = 0; k < 8; k++) c = (c & 1) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1);
    crc_table[i] = c;
  }
}
static unsigned int crc_update(unsigned int c, const unsigned char *p, size_t n) {
  while (n--) c = crc_table[(c ^ *p++) & 0xFF] ^ (c >> 8);
  return c;
}

/* read up to `want` bytes; returns the count (short at end of file), -1 on a read error */
static long read_upto(struct mspack_system *sys, struct mspack_file *fh, unsigned char *dst, size_t want) {
  size_t got = 0;
  while (got < want) {
    int run = (want - got) > (1u << 20) ? (1 << 20) : (int)(want - got);
    int r = sys->read(fh, dst + got, run);
    if (r < 0) return -1;
    if (r == 0) break;
    got += (size_t) r;
  }
  return (long) got;
}

static int oab_run(struct oabd_p *self, const char *input, const char *base, const char *output, int patch)
{
  struct mspack_system *sys = self->system;
  struct mspack_file *infh = NULL, *basefh = NULL, *outfh = NULL;
  unsigned char hdr[0x1c], bh[16];
  unsigned int block_max, target_size;
  struct oab_blk *blks = NULL;
  size_t n_blks = 0, cap_blks = 0, n_units = 0, in_bytes = 0, out_bytes = 0, k;
  unsigned char *in_arena = NULL, *out_arena = NULL;
  size_t in_cap = 0;
  mspack_hip_unit *units = NULL;
  mspack_hip_result *res = NULL;
  int ret = MSPACK_ERR_OK, tail_err = MSPACK_ERR_OK;
  const unsigned int hsize = patch ? 0x1c : 0x10;

  crc_init();
  if (!(infh = sys->open(sys, input, MSPACK_SYS_OPEN_READ))) return MSPACK_ERR_OPEN;
  if (sys->read(infh, hdr, (int) hsize) != (int) hsize) { ret = MSPACK_ERR_READ; goto out; }
  if (rd_le32(hdr) != 3 || rd_le32(hdr + 4) != (patch ? 2u : 1u)) { ret = MSPACK_ERR_SIGNATURE; goto out; }
  block_max = rd_le32(hdr + 8);
  target_size = rd_le32(hdr + (patch ? 0x10 : 0x0c));
  if (patch) {
    if (block_max < 16) block_max = 16;                     /* oabd.c:283-285 */
    if (!(basefh = sys->open(sys, base, MSPACK_SYS_OPEN_READ))) { ret = MSPACK_ERR_OPEN; goto out; }
  }
  if (!(outfh = sys->open(sys, output, MSPACK_SYS_OPEN_WRITE))) { ret = MSPACK_ERR_OPEN; goto out; }

  /* pass 1: walk the block headers in file order, gather every block's bytes.  The first structural
   * problem ends the walk; it is reported once the blocks before it have been replayed. */
  while (target_size) {
    struct oab_blk b;
    long got;
    size_t room, want;
    memset(&b, 0, sizeof(b));
    if (sys->read(infh, bh, 16) != 16) { tail_err = MSPACK_ERR_READ; break; }
    if (patch) {
      b.csize = rd_le32(bh); b.dsize = rd_le32(bh + 4); b.ssize = rd_le32(bh + 8); b.crc = rd_le32(bh + 12);
      b.compressed = 1;
      if (b.dsize > block_max || b.dsize > target_size || b.ssize > block_max) { tail_err = MSPACK_ERR_DATAFORMAT; break; }
    }
    else {
      unsigned int flags = rd_le32(bh);
      b.csize = rd_le32(bh + 4); b.dsize = rd_le32(bh + 8); b.crc = rd_le32(bh + 12);
      if (b.dsize > block_max || b.dsize > target_size || flags > 1) { tail_err = MSPACK_ERR_DATAFORMAT; break; }
      b.compressed = (int) flags;
      if (!flags && b.dsize != b.csize) { tail_err = MSPACK_ERR_DATAFORMAT; break; }
    }
    if (b.compressed) {
      unsigned int wsz = patch ? (((b.ssize + 32767u) & ~32767u) + b.dsize) : b.dsize;
      b.window_bits = 17;
      while (b.window_bits < 25 && (1u << b.window_bits) < wsz) b.window_bits++;
    }
    /* the block's bytes (compressed stream, or the stored data) */
    in_bytes = (in_bytes + 15) & ~(size_t) 15;
    /* csize is untrusted: never reserve more than the file still holds (a 32-byte file must not be able to
     * ask for gigabytes; the reference streams with window-sized memory) */
    {
      off_t flen = 0, here = sys->tell(infh);
      want = (size_t) b.csize;
      if (!mspack_sys_filelen(sys, infh, &flen) && flen >= here && (off_t) want > flen - here) want = (size_t)(flen - here);
      room = want + 64;
    }
    if (in_bytes + room > in_cap) {
      size_t ncap = (in_bytes + room) * 2 + 65536;
      unsigned char *n = (unsigned char *) sys->alloc(sys, ncap);
      if (!n) { ret = MSPACK_ERR_NOMEMORY; goto out; }
      if (in_arena) { sys->copy(in_arena, n, in_bytes); sys->free(in_arena); }
      in_arena = n; in_cap = ncap;
    }
    got = read_upto(sys, infh, in_arena + in_bytes, want);
    if (got < 0) { tail_err = MSPACK_ERR_READ; break; }
    b.in_pos = in_bytes; b.in_have = (size_t) got;
    memset(in_arena + in_bytes + got, 0, 64);
    in_bytes += (size_t) got + 16;
    if (b.compressed) b.unit = n_units++;
    if (n_blks == cap_blks) {
      size_t nc = cap_blks ? cap_blks * 2 : 64;
      struct oab_blk *n = (struct oab_blk *) sys->alloc(sys, nc * sizeof(*n));
      if (!n) { ret = MSPACK_ERR_NOMEMORY; goto out; }
      if (blks) { sys->copy(blks, n, n_blks * sizeof(*n)); sys->free(blks); }
      blks = n; cap_blks = nc;
    }
    blks[n_blks++] = b;
    target_size -= b.dsize;
    if ((size_t) got < b.csize) break;                       /* the file ends inside this block */
  }

  /* the unit table: output regions back to back, each preceded by room for its reference data */
  if (n_units) {
    units = (mspack_hip_unit *) sys->alloc(sys, n_units * sizeof(*units));
    res = (mspack_hip_result *) sys->alloc(sys, n_units * sizeof(*res));
    if (!units || !res) { ret = MSPACK_ERR_NOMEMORY; goto out; }
    memset(units, 0, n_units * sizeof(*units)); memset(res, 0, n_units * sizeof(*res));
    for (k = 0; k < n_blks; k++) {
      struct oab_blk *b = &blks[k];
      mspack_hip_unit *u;
      if (!b->compressed) continue;
      u = &units[b->unit];
      out_bytes = (out_bytes + 15) & ~(size_t) 15;
      out_bytes += ((size_t) b->ssize + 15) & ~(size_t) 15;
      u->in_off = b->in_pos; u->in_len = (uint32_t) b->in_have;
      u->out_off = out_bytes; u->out_len = b->dsize;
      u->kind = MSPACK_HIP_KIND_LZX_DELTA; u->window_bits = (uint8_t) b->window_bits;
      u->ref_len = b->ssize;
      out_bytes += b->dsize;
    }
    if (!(out_arena = (unsigned char *) sys->alloc(sys, out_bytes + 64))) { ret = MSPACK_ERR_NOMEMORY; goto out; }
    /* patch: each block's reference data is the next `source size` bytes of the base file (oabd.c:346) */
    if (patch) {
      int base_dead = 0;
      for (k = 0; k < n_blks; k++) {
        struct oab_blk *b = &blks[k];
        mspack_hip_unit *u = &units[b->unit];
        if (b->ssize == 0) continue;
        if (base_dead) { b->ref_err = MSPACK_ERR_READ; continue; }
        if (read_upto(sys, basefh, out_arena + u->out_off - b->ssize, b->ssize) != (long) b->ssize) {
          b->ref_err = MSPACK_ERR_READ; base_dead = 1;
        }
      }
    }
    {
      /* blocks whose reference data could not be read and empty blocks are kept out of the batch */
      size_t nsel = 0;
      mspack_hip_unit *sel = (mspack_hip_unit *) sys->alloc(sys, n_units * sizeof(*sel));
      mspack_hip_result *rsel = (mspack_hip_result *) sys->alloc(sys, n_units * sizeof(*rsel));
      size_t *map = (size_t *) sys->alloc(sys, n_units * sizeof(*map));
      if (!sel || !rsel || !map) { sys->free(sel); sys->free(rsel); sys->free(map); ret = MSPACK_ERR_NOMEMORY; goto out; }
      for (k = 0; k < n_blks; k++) {
        struct oab_blk *b = &blks[k];
        if (!b->compressed || b->ref_err || b->dsize == 0) continue;
        map[nsel] = b->unit; sel[nsel++] = units[b->unit];
      }
      if (nsel) {
        int rc = mspack_hip_decode_batch(sel, nsel, in_arena, in_bytes + 48, out_arena, out_bytes + 64, rsel);
        if (rc) {
          sys->message(NULL, "GPU batch decode failed: %s", mspack_hip_last_error());
          sys->free(sel); sys->free(rsel); sys->free(map);
          ret = MSPACK_ERR_DECRUNCH; goto out;
        }
        for (k = 0; k < nsel; k++) res[map[k]] = rsel[k];
      }
      sys->free(sel); sys->free(rsel); sys->free(map);
    }
  }

  /* pass 2: replay the file in order */
  for (k = 0; k < n_blks && !ret; k++) {
    struct oab_blk *b = &blks[k];
    if (!b->compressed) {
      /* copy_fh (oabd.c:384-403): buf_size pieces; a piece the file does not fully hold is not written */
      size_t whole = b->in_have;
      if (whole < b->dsize) whole -= whole % (size_t) self->buf_size;
      if (write_slice(sys, outfh, in_arena + b->in_pos, whole < b->dsize ? whole : b->dsize)) ret = MSPACK_ERR_WRITE;
      else if (b->in_have < b->dsize) ret = MSPACK_ERR_READ;
      continue;
    }
    if (b->ref_err) { ret = b->ref_err; break; }                            /* lzxd_set_reference_data */
    {
      const mspack_hip_unit *u = &units[b->unit];
      const mspack_hip_result *r = &res[b->unit];
      unsigned int n = b->dsize ? r->out_len : 0;
      int err = b->dsize ? r->err : MSPACK_ERR_OK;
      if (n > b->dsize) n = b->dsize;
      if (write_slice(sys, outfh, out_arena + u->out_off, n)) { ret = MSPACK_ERR_WRITE; break; }
      if (err) { ret = err; break; }
      if (b->in_have < b->csize) { ret = MSPACK_ERR_READ; break; }            /* the padding skip fails */
      if (crc_update(0xffffffffu, out_arena + u->out_off, n) != b->crc) { ret = MSPACK_ERR_CHECKSUM; break; }
    }
  }
  if (!ret) ret = tail_err;

out:
  if (outfh) sys->close(outfh);
  if (basefh) sys->close(basefh);
  if (infh) sys->close(infh);
  sys->free(blks); sys->free(in_arena); sys->free(out_arena); sys->free(units); sys->free(res);
  return ret;
}

static int oabd_decompress(struct msoab_decompressor *base, const char *input, const char *output) {
  struct oabd_p *self = (struct oabd_p *) base;
  if (!self) return MSPACK_ERR_ARGS;
  return oab_run(self, input, NULL, output, 0);
}
static int oabd_decompress_incremental(struct msoab_decompressor *base, const char *input, const char *basefile,
                                       const char *output) {
  struct oabd_p *self = (struct oabd_p *) base;
  if (!self) return MSPACK_ERR_ARGS;
  return oab_run(self, input, basefile, output, 1);
}
static int oabd_param(struct msoab_decompressor *base, int param, int value) {
  struct oabd_p *self = (struct oabd_p *) base;
  if (self && param == MSOABD_PARAM_DECOMPBUF && value >= 16) { self->buf_size = value; return MSPACK_ERR_OK; }
  return MSPACK_ERR_ARGS;
}

struct msoab_decompressor *mspack_create_oab_decompressor(struct mspack_system *sys)
{
  struct oabd_p *self;
  if (!sys) sys = mspack_default_system;
  if (!mspack_valid_system(sys)) return NULL;
  if (!(self = (struct oabd_p *) sys->alloc(sys, sizeof(*self)))) return NULL;
  self->base.decompress = &oabd_decompress;
  self->base.decompress_incremental = &oabd_decompress_incremental;
  self->base.set_param = &oabd_param;
  self->system = sys;
  self->buf_size = 4096;
  return &self->base;
}

void mspack_destroy_oab_decompressor(struct msoab_decompressor *base)
{
  struct oabd_p *self = (struct oabd_p *) base;
  if (self) self->system->free(self);
}
