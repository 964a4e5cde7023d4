/* oracle/mszip_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * CPU restatement of the reference MSZIP folder decoder ("CK"-framed raw deflate, one frame per
 * <=32 KiB block, 32 KiB history that is never cleared between blocks).
 * Follows (by behaviour):
 *   bit reader .............. libmspack/mspack/readbits.h:133-214 with mszipd.c:19-27 (LSB first, bytewise)
 *   static tables ........... mszipd.c:46-73 (RFC 1951 sec. 3.2.5 / 3.2.7; generated here)
 *   dynamic header .......... mszipd.c:91-151
 *   inflate ................. mszipd.c:154-316
 *   block loop / CK scan .... mszipd.c:377-460
 */
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include "oracle.h"
extern __thread int oracle_hard_eof_;   /* lzx_oracle.c: oracle_set_hard_eof() */
#include "oracle_huff.h"

#define FRAME 32768u

typedef struct {
  const uint8_t *in; size_t in_len, pos;
  uint32_t bb; int bl;
  int err;                      /* MSPACK_ERR_READ when the input ran dry */
  /* What the reference's stream STRUCT holds, as opposed to the decoder's local copies (readbits.h:
   * 119-131): STORE_BITS writes all four fields, and every refill (read_input, readbits.h:184-214)
   * rewinds the struct's i_ptr to the start of the new input chunk.  Repair mode restarts from these
   * after a failed block (mszipd.c:404), so they are observable. */
  size_t bufsz, s_iptr;
  uint32_t s_bb; int s_bl;
} zbits_t;

static int z_byte(zbits_t *b, unsigned *v) {
  if (b->pos <= b->in_len && (b->pos % b->bufsz == 0 || b->pos == b->in_len)) b->s_iptr = b->pos;   /* refill point */
  if (b->pos < b->in_len) { *v = b->in[b->pos++]; return 0; }
  if (!oracle_hard_eof_ && b->pos < b->in_len + 2) { b->pos++; *v = 0; return 0; }   /* (a FAILED read -- sys->read < 0 -- fabricates nothing: readbits.h:196-198) */
  b->err = ORC_READ; return 1;
}
static void z_store(zbits_t *b) { b->s_iptr = b->pos; b->s_bb = b->bb; b->s_bl = b->bl; }
static void z_restore(zbits_t *b) { b->pos = b->s_iptr; b->bb = b->s_bb; b->bl = b->s_bl; }
static int z_ensure(zbits_t *b, int n) {
  while (b->bl < n) { unsigned v; if (z_byte(b, &v)) return 1; b->bb |= v << b->bl; b->bl += 8; }
  return 0;
}
#define ZPEEK(b, n) ((b)->bb & ((1u << (n)) - 1))
#define ZDROP(b, n) do { (b)->bb >>= (n); (b)->bl -= (n); } while (0)
static int z_bits(zbits_t *b, int n, unsigned *v) {
  if (z_ensure(b, n)) return 1;
  *v = n ? ZPEEK(b, n) : 0; ZDROP(b, n); return 0;
}
static unsigned rev16(unsigned v) {
  v = ((v & 0x5555) << 1) | ((v >> 1) & 0x5555);
  v = ((v & 0x3333) << 2) | ((v >> 2) & 0x3333);
  v = ((v & 0x0F0F) << 4) | ((v >> 4) & 0x0F0F);
  return ((v & 0x00FF) << 8) | ((v >> 8) & 0x00FF);
}
/* negative = inflate error class (-> ERR_DECRUNCH), positive = mspack error (ERR_READ) */
#define INF_ERR (-1)
static int z_sym(zbits_t *b, const oh_table *t, int ensure_bits, int *sym) {
  int l;
  if (z_ensure(b, ensure_bits)) return ORC_READ;
  *sym = oh_decode(t, rev16(b->bb & 0xFFFF), &l);
  if (*sym < 0) return INF_ERR;
  ZDROP(b, l);
  return 0;
}

typedef struct {
  zbits_t b;
  uint8_t window[FRAME];
  uint32_t wpos, bytes_output;
  uint8_t lit_len[288], dist_len[32];
  oh_table lit_t, dist_t, bl_t;
} zip_t;

static uint16_t len_base[29], dist_base[30];
static uint8_t  len_extra[29], dist_extra[30];
static const uint8_t clen_order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
static void init_tables(void) {
  int i; unsigned base;
  if (len_base[0]) return;
  for (i = 0, base = 3; i < 28; i++) {            /* RFC 1951 3.2.5 length codes 257..284 */
    int e = (i < 8) ? 0 : (i - 4) / 4;
    len_base[i] = (uint16_t) base; len_extra[i] = (uint8_t) e; base += 1u << e;
  }
  len_base[28] = 258; len_extra[28] = 0;
  for (i = 0, base = 1; i < 30; i++) {            /* distance codes 0..29 */
    int e = (i < 4) ? 0 : (i - 2) / 2;
    dist_base[i] = (uint16_t) base; dist_extra[i] = (uint8_t) e; base += 1u << e;
  }
}

static int flush(zip_t *z, uint32_t n) {
  z->bytes_output += n;
  return z->bytes_output > FRAME;
}
#define PUT(z, byte) do { (z)->window[(z)->wpos++] = (uint8_t)(byte); \
    if ((z)->wpos == FRAME) { if (flush((z), FRAME)) return INF_ERR; (z)->wpos = 0; } } while (0)

static int read_dynamic(zip_t *z) {
  uint8_t bl_len[19], lens[288 + 32];
  unsigned nlit, ndist, nbl, i, run, code, last = 0, v;
  zbits_t *b = &z->b;
  if (z_bits(b, 5, &nlit) || z_bits(b, 5, &ndist) || z_bits(b, 4, &nbl)) return ORC_READ;
  nlit += 257; ndist += 1; nbl += 4;
  if (nlit > 288 || ndist > 32) return INF_ERR;
  memset(bl_len, 0, sizeof(bl_len));
  for (i = 0; i < nbl; i++) { if (z_bits(b, 3, &v)) return ORC_READ; bl_len[clen_order[i]] = (uint8_t) v; }
  if (oh_build(&z->bl_t, bl_len, 19, 7)) return INF_ERR;
  for (i = 0; i < nlit + ndist; i++) {
    int s, r;
    if ((r = z_sym(b, &z->bl_t, 7, &s))) return r;
    code = (unsigned) s;
    if (code < 16) { lens[i] = (uint8_t)(last = code); continue; }
    switch (code) {
    case 16: if (z_bits(b, 2, &run)) return ORC_READ; run += 3;  code = last; break;
    case 17: if (z_bits(b, 3, &run)) return ORC_READ; run += 3;  code = 0;    break;
    case 18: if (z_bits(b, 7, &run)) return ORC_READ; run += 11; code = 0;    break;
    default: return INF_ERR;
    }
    if (i + run > nlit + ndist) return INF_ERR;
    while (run--) lens[i++] = (uint8_t) code;
    i--;
  }
  memset(z->lit_len, 0, 288); memcpy(z->lit_len, lens, nlit);
  memset(z->dist_len, 0, 32); memcpy(z->dist_len, lens + nlit, ndist);
  return 0;
}

static int inflate_block_stream(zip_t *z) {
  zbits_t *b = &z->b;
  unsigned last_block, type, v;
  do {
    if (z_bits(b, 1, &last_block) || z_bits(b, 2, &type)) return ORC_READ;
    if (type == 0) {
      uint8_t hdr[4]; unsigned i, length, ncomp;
      ZDROP(b, b->bl & 7);
      for (i = 0; b->bl >= 8; i++) {                                /* mszipd.c:176-181 */
        if (i == 4) return INF_ERR;
        hdr[i] = (uint8_t) ZPEEK(b, 8); ZDROP(b, 8);
      }
      if (b->bl != 0) return INF_ERR;
      while (i < 4) { if (z_byte(b, &v)) return ORC_READ; hdr[i++] = (uint8_t) v; }
      length = hdr[0] | (hdr[1] << 8); ncomp = hdr[2] | (hdr[3] << 8);
      if (length != (~ncomp & 0xFFFF)) return INF_ERR;
      while (length--) { if (z_byte(b, &v)) return ORC_READ; PUT(z, v); }
    }
    else if (type == 1 || type == 2) {
      if (type == 1) {
        unsigned i = 0;
        while (i < 144) z->lit_len[i++] = 8;
        while (i < 256) z->lit_len[i++] = 9;
        while (i < 280) z->lit_len[i++] = 7;
        while (i < 288) z->lit_len[i++] = 8;
        memset(z->dist_len, 5, 32);
      }
      else { int r; z_store(b); r = read_dynamic(z); if (r) return r; z_store(b); }   /* mszipd.c:223,149 */
      if (oh_build(&z->lit_t, z->lit_len, 288, 9)) return INF_ERR;
      if (oh_build(&z->dist_t, z->dist_len, 32, 6)) return INF_ERR;
      for (;;) {
        int sym, r;
        if ((r = z_sym(b, &z->lit_t, 16, &sym))) return r;
        if (sym < 256) { PUT(z, sym); continue; }
        if (sym == 256) break;
        {
          unsigned code = (unsigned) sym - 257, length, dist, mpos, e;
          if (code >= 29) return INF_ERR;
          if (z_bits(b, len_extra[code], &e)) return ORC_READ;
          length = len_base[code] + e;
          if ((r = z_sym(b, &z->dist_t, 16, &sym))) return r;
          if (sym >= 30) return INF_ERR;
          if (z_bits(b, dist_extra[sym], &e)) return ORC_READ;
          dist = dist_base[sym] + e;
          if (getenv("ORACLE_ZIP_TRACE")) fprintf(stderr, "zip match: pos %u len %u dist %u\n", (unsigned) z->wpos, length, dist);
          mpos = ((dist > z->wpos) ? FRAME : 0) + z->wpos - dist;   /* mszipd.c:267-268 */
          while (length--) { uint8_t c = z->window[mpos++]; mpos &= FRAME - 1; PUT(z, c); }
        }
      }
    }
    else return INF_ERR;
  } while (!last_block);
  if (z->wpos) { if (flush(z, z->wpos)) return INF_ERR; }
  z_store(b);                                                       /* mszipd.c:312 */
  return 0;
}

int oracle_mszip_decode(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap,
                        uint64_t out_bytes, int repair_mode, uint32_t *block_lens, int cap,
                        int *n_blocks, oracle_result *res)
{
  zip_t *z = (zip_t *) calloc(1, sizeof(*z));
  uint64_t written = 0, remaining = out_bytes;
  int err = ORC_OK, nb = 0;
  uint32_t leftover = 0;

  memset(res, 0, sizeof(*res));
  init_tables();
  z->b.in = in; z->b.in_len = in_len;
  z->b.bufsz = (repair_mode > 1) ? (size_t) repair_mode : 4096;     /* >1: the feeder's buffer size */
  while (remaining > 0) {
    unsigned v; int state = 0, r;
    uint32_t n;
    leftover = 0;
    z_restore(&z->b);                                               /* mszipd.c:404 */
    ZDROP(&z->b, z->b.bl & 7);
    do {                                                            /* mszipd.c:406-414 */
      if (z_bits(&z->b, 8, &v)) { err = ORC_READ; goto done; }
      if (v == 'C') state = 1;
      else if (state == 1 && v == 'K') state = 2;
      else state = 0;
    } while (state != 2);
    z->wpos = 0; z->bytes_output = 0;
    z_store(&z->b);                                                 /* mszipd.c:419 */
    r = inflate_block_stream(z);
    if (r) {
      if (repair_mode) {                                            /* mszipd.c:422-433 */
        if (z->bytes_output == 0 && z->wpos > 0) flush(z, z->wpos);
        if (z->bytes_output < FRAME) memset(z->window + z->bytes_output, 0, FRAME - z->bytes_output);
        z->bytes_output = FRAME;
      }
      else { err = (r > 0) ? r : ORC_DECRUNCH; goto done; }
    }
    if (block_lens && nb < cap) block_lens[nb] = z->bytes_output;
    nb++;
    n = (remaining < z->bytes_output) ? (uint32_t) remaining : z->bytes_output;
    if (out && written < out_cap) {
      size_t room = out_cap - (size_t) written;
      memcpy(out + written, z->window, n < room ? n : room);
    }
    written += n;
    if (r > 0 && repair_mode) { err = r; goto done; }
    remaining -= n;
    leftover = z->bytes_output - n;
    if (leftover && out && written + leftover <= out_cap) memcpy(out + written, z->window + n, leftover);   /* (room behind the request: the rest of the block) */
  }
done:
  if (n_blocks) *n_blocks = nb;
  res->err = err; res->out_len = written; res->in_used = z->b.pos;
  /* what the last block inflated to BEYOND the request: mszipd keeps those bytes (o_ptr .. o_end, mszipd.c:386-392, 440-452) and
   * hands them to the next call -- a block is as long as its deflate stream, whatever the CFDATA header said (the batch ABI
   * reports it in mspack_hip_result.in_next, unused for MSZIP otherwise; `out` holds the bytes when there is room) */
  res->in_next = (err == 0) ? leftover : 0;
  free(z);
  return err;
}
