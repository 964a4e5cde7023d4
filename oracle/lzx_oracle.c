/* oracle/lzx_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * CPU restatement of the reference LZX stream decoder, buffer to buffer.
 * Follows (by behaviour, not by text):
 *   bit reader ............ libmspack/mspack/readbits.h:133-214 with lzxd.c:83-92 (LE16 words, MSB first)
 *   code-length reader .... lzxd.c:138-183
 *   static slot tables .... lzxd.c:209-255 (computed here from the formula in lzxd.c:202-207)
 *   reset / init .......... lzxd.c:257-346
 *   frame / block loop .... lzxd.c:419-756
 *   E8 translation ........ lzxd.c:706-736
 * The window is kept exactly like the reference's (a 2^window_bits ring that is never cleared),
 * except that it starts zero-filled where the reference's is uninitialised malloc memory.
 */
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include "oracle.h"
__thread int oracle_hard_eof_;          /* oracle_set_hard_eof(): the feeder's read FAILS at in_len (cabd.c:1322-1324) instead of ending */
void oracle_set_hard_eof(int on) { oracle_hard_eof_ = on; }
#include "oracle_huff.h"

#define FRAME 32768u
#define NUM_CHARS 256
#define MAIN_MAXSYMS (NUM_CHARS + 290 * 8)
#define LEN_MAXSYMS 250
#define SAFETY 64

enum { BT_INVALID = 0, BT_VERBATIM = 1, BT_ALIGNED = 2, BT_UNCOMPRESSED = 3 };

typedef struct {
  const uint8_t *in; size_t in_len, pos;   /* pos == the reference's i_ptr offset */
  uint32_t bb; int bl;                     /* bit_buffer (MSB-aligned), bits_left */
  int err;
} bits_t;

static int rd_byte(bits_t *b, unsigned *v) {
  /* readbits.h:192-214: at EOF two zero bytes are fabricated once; after that ERR_READ */
  if (b->pos < b->in_len) { *v = b->in[b->pos++]; return 0; }
  if (!oracle_hard_eof_ && b->pos < b->in_len + 2) { b->pos++; *v = 0; return 0; }   /* (a FAILED read -- sys->read < 0 -- fabricates nothing: readbits.h:196-198) */
  b->err = ORC_READ; return 1;
}
static int ensure(bits_t *b, int n) {
  while (b->bl < n) {
    unsigned b0, b1;
    if (rd_byte(b, &b0) || rd_byte(b, &b1)) return 1;
    b->bb |= ((b1 << 8) | b0) << (32 - 16 - b->bl);
    b->bl += 16;
  }
  return 0;
}
#define PEEK(b, n)   ((b)->bb >> (32 - (n)))
#define DROP(b, n)   do { (b)->bb <<= (n); (b)->bl -= (n); } while (0)
static int rd_bits(bits_t *b, int n, unsigned *v) {
  if (ensure(b, n)) return 1;
  *v = PEEK(b, n); DROP(b, n); return 0;
}
static int rd_sym(bits_t *b, const oh_table *t, const uint8_t *lens, int *sym) {
  int l;
  if (ensure(b, 16)) return 1;
  *sym = oh_decode(t, PEEK(b, 16), &l);
  if (*sym < 0) { b->err = ORC_DECRUNCH; return 1; }
  (void) lens;
  DROP(b, l);
  return 0;
}

typedef struct {
  bits_t b;
  uint8_t *win; uint32_t wsize, wpos, frame_posn;
  uint32_t frame, reset_frames, num_offsets;
  uint64_t offset, length;          /* bytes written so far; overall length (0 = unknown) */
  uint32_t R0, R1, R2, block_length, block_remaining;
  int32_t  intel_filesize;
  int block_type, header_read, intel_started, length_empty;
  int is_delta; uint32_t ref_size;   /* LZX DELTA (lzxd.c:288-293, 348-382) */
  uint8_t pre_len[20 + SAFETY], main_len[MAIN_MAXSYMS + SAFETY], len_len[LEN_MAXSYMS + SAFETY],
          ali_len[8 + SAFETY];
  oh_table pre_t, main_t, len_t, ali_t;
} lzx_t;

static uint32_t slot_base[290];
static uint8_t  slot_extra[290];
static const uint16_t slots_for_bits[11] = { 30, 32, 34, 36, 38, 42, 50, 66, 98, 162, 290 };
static void init_slots(void) {
  int i; uint32_t base = 0;
  if (slot_base[1]) return;
  for (i = 0; i < 290; i++) {
    int e = (i < 4) ? 0 : (i < 36 ? (i / 2) - 1 : 17);
    slot_base[i] = base; slot_extra[i] = (uint8_t) e;
    base += 1u << e;
  }
}

static void reset_state(lzx_t *z) {
  z->R0 = z->R1 = z->R2 = 1;
  z->header_read = 0; z->block_remaining = 0; z->block_type = BT_INVALID;
  memset(z->main_len, 0, MAIN_MAXSYMS);
  memset(z->len_len, 0, LEN_MAXSYMS);
}

/* lzxd.c:138-183; run lengths are NOT clipped to `last` (the SAFETY area absorbs them) */
static int read_lens(lzx_t *z, uint8_t *lens, unsigned first, unsigned last) {
  unsigned x, y; int s;
  for (x = 0; x < 20; x++) { if (rd_bits(&z->b, 4, &y)) return 1; z->pre_len[x] = (uint8_t) y; }
  if (oh_build(&z->pre_t, z->pre_len, 20, 6)) { z->b.err = ORC_DECRUNCH; return 1; }
  for (x = first; x < last; ) {
    if (rd_sym(&z->b, &z->pre_t, z->pre_len, &s)) return 1;
    if (s == 17) { if (rd_bits(&z->b, 4, &y)) return 1; y += 4;  while (y--) lens[x++] = 0; }
    else if (s == 18) { if (rd_bits(&z->b, 5, &y)) return 1; y += 20; while (y--) lens[x++] = 0; }
    else if (s == 19) {
      int v;
      if (rd_bits(&z->b, 1, &y)) return 1;
      y += 4;
      if (rd_sym(&z->b, &z->pre_t, z->pre_len, &s)) return 1;
      v = (int) lens[x] - s; if (v < 0) v += 17;
      while (y--) lens[x++] = (uint8_t) v;
    }
    else { int v = (int) lens[x] - s; if (v < 0) v += 17; lens[x++] = (uint8_t) v; }
  }
  return 0;
}

static int block_header(lzx_t *z) {
  unsigned t, hi, lo, i, v;
  bits_t *b = &z->b;
  if (z->block_type == BT_UNCOMPRESSED && (z->block_length & 1)) {   /* lzxd.c:469-474 */
    if (rd_byte(b, &v)) return 1;
  }
  if (rd_bits(b, 3, &t) || rd_bits(b, 16, &hi) || rd_bits(b, 8, &lo)) return 1;
  z->block_type = (int) t;
  z->block_remaining = z->block_length = (hi << 8) | lo;
  if (getenv("ORACLE_LZX_TRACE")) fprintf(stderr, "lzx block: frame %u type %d length %u\n", (unsigned) z->frame, z->block_type, (unsigned) z->block_length);
  switch (z->block_type) {
  case BT_ALIGNED:
    for (i = 0; i < 8; i++) { if (rd_bits(b, 3, &v)) return 1; z->ali_len[i] = (uint8_t) v; }
    if (oh_build(&z->ali_t, z->ali_len, 8, 7)) { b->err = ORC_DECRUNCH; return 1; }
    /* fall through */
  case BT_VERBATIM:
    if (read_lens(z, z->main_len, 0, 256)) return 1;
    if (read_lens(z, z->main_len, 256, NUM_CHARS + z->num_offsets)) return 1;
    if (oh_build(&z->main_t, z->main_len, MAIN_MAXSYMS, 12)) { b->err = ORC_DECRUNCH; return 1; }
    if (z->main_len[0xE8] != 0) z->intel_started = 1;
    if (read_lens(z, z->len_len, 0, 249)) return 1;
    z->length_empty = 0;
    if (oh_build(&z->len_t, z->len_len, LEN_MAXSYMS, 12)) {           /* lzxd.c:111-125 */
      for (i = 0; i < LEN_MAXSYMS; i++) if (z->len_len[i]) { b->err = ORC_DECRUNCH; return 1; }
      z->length_empty = 1;
    }
    break;
  case BT_UNCOMPRESSED: {
    uint8_t buf[12];
    z->intel_started = 1;
    if (b->bl == 0) { if (ensure(b, 16)) return 1; }               /* lzxd.c:506-507 */
    b->bl = 0; b->bb = 0;
    for (i = 0; i < 12; i++) { if (rd_byte(b, &v)) return 1; buf[i] = (uint8_t) v; }
    z->R0 = buf[0] | (buf[1] << 8) | (buf[2]  << 16) | ((uint32_t) buf[3]  << 24);
    z->R1 = buf[4] | (buf[5] << 8) | (buf[6]  << 16) | ((uint32_t) buf[7]  << 24);
    z->R2 = buf[8] | (buf[9] << 8) | (buf[10] << 16) | ((uint32_t) buf[11] << 24);
    break; }
  default:
    b->err = ORC_DECRUNCH; return 1;
  }
  return 0;
}

/* decode `run` bytes of a verbatim/aligned block; returns the signed remainder (<=0) or
 * INT32_MIN on error */
static int32_t decode_run(lzx_t *z, int32_t run) {
  bits_t *b = &z->b;
  uint8_t *win = z->win;
  while (run > 0) {
    int sym;
    if (rd_sym(b, &z->main_t, z->main_len, &sym)) return INT32_MIN;
    if (sym < NUM_CHARS) { win[z->wpos++] = (uint8_t) sym; run--; continue; }
    {
      uint32_t m = (uint32_t)(sym - NUM_CHARS), slot = m >> 3, off, i, j;
      int len = (int)(m & 7);
      uint8_t *dst, *src;
      if (len == 7) {
        int foot;
        if (z->length_empty) { b->err = ORC_DECRUNCH; return INT32_MIN; }
        if (rd_sym(b, &z->len_t, z->len_len, &foot)) return INT32_MIN;
        len += foot;
      }
      len += 2;
      if (slot == 0) off = z->R0;
      else if (slot == 1) { off = z->R1; z->R1 = z->R0; z->R0 = off; }
      else if (slot == 2) { off = z->R2; z->R2 = z->R0; z->R0 = off; }
      else {
        int extra = slot_extra[slot];
        unsigned v;
        off = slot_base[slot] - 2;
        if (extra >= 3 && z->block_type == BT_ALIGNED) {
          int a;
          if (extra > 3) { if (rd_bits(b, extra - 3, &v)) return INT32_MIN; off += v << 3; }
          if (rd_sym(b, &z->ali_t, z->ali_len, &a)) return INT32_MIN;
          off += (uint32_t) a;
        }
        else if (extra) { if (rd_bits(b, extra, &v)) return INT32_MIN; off += v; }
        z->R2 = z->R1; z->R1 = z->R0; z->R0 = off;
      }
      if (len == 257 && z->is_delta) {                               /* lzxd.c:588-611 */
        unsigned p, x = 0;
        if (ensure(b, 3)) return INT32_MIN;
        p = PEEK(b, 3);
        if ((p & 4) == 0)      { DROP(b, 1); if (rd_bits(b, 8, &x)) return INT32_MIN; }
        else if ((p >> 1) == 2) { DROP(b, 2); if (rd_bits(b, 10, &x)) return INT32_MIN; x += 0x100; }
        else if (p == 6)        { DROP(b, 3); if (rd_bits(b, 12, &x)) return INT32_MIN; x += 0x500; }
        else                    { DROP(b, 3); if (rd_bits(b, 15, &x)) return INT32_MIN; }
        len += (int) x;
      }
      if (z->wpos + (uint32_t) len > z->wsize) { b->err = ORC_DECRUNCH; return INT32_MIN; }
#ifdef ORACLE_MATCH_HOOK      /* analysis tools only (tools/analysis): every match as the decoder applies it */
      ORACLE_MATCH_HOOK(z->wpos, (uint32_t) len, off, slot);
#endif
      dst = &win[z->wpos]; i = (uint32_t) len;
      if (off > z->wpos) {                                           /* lzxd.c:622-642 */
        if ((uint64_t) off > z->offset && (off - z->wpos) > z->ref_size) { b->err = ORC_DECRUNCH; return INT32_MIN; }
        j = off - z->wpos;
        if (j > z->wsize) { b->err = ORC_DECRUNCH; return INT32_MIN; }
        src = &win[z->wsize - j];
        if (j < i) { i -= j; while (j-- > 0) *dst++ = *src++; src = win; }
        while (i-- > 0) *dst++ = *src++;
      }
      else { src = dst - off; while (i-- > 0) *dst++ = *src++; }
      run -= len; z->wpos += (uint32_t) len;
    }
  }
  return run;
}

/* the reset points of the LAST decode call of this thread that found a block open (lzxd.c:423-431) */
#define ORC_OPEN_MAX 256
static __thread uint32_t g_open_n, g_open_frames[ORC_OPEN_MAX];
uint32_t oracle_lzx_open_resets(uint32_t *frames, uint32_t cap) {
  uint32_t i;
  for (i = 0; i < g_open_n && i < cap && i < ORC_OPEN_MAX; i++) frames[i] = g_open_frames[i];
  return g_open_n;
}

int oracle_lzx_decode(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap,
                      uint64_t out_bytes, uint64_t length, int window_bits, int reset_frames,
                      int32_t e8_base, oracle_result *res)
{
  return oracle_lzxd_decode(in, in_len, out, out_cap, out_bytes, length, window_bits, reset_frames, e8_base,
                            0, NULL, 0, res);
}

/* is_delta != 0: LZX DELTA (window 2^17..2^25, a 16-bit chunk size in front of every frame, match
 * lengths extended beyond 257, `ref` = reference data preloaded at the end of the window:
 * lzxd.c:288-293, 348-382, 440-444, 588-611, 622-634) */
int oracle_lzxd_decode(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap,
                       uint64_t out_bytes, uint64_t length, int window_bits, int reset_frames,
                       int32_t e8_base, int is_delta, const uint8_t *ref, size_t ref_len, oracle_result *res)
{
  lzx_t *z;
  uint64_t written = 0, remaining = out_bytes, in_next = 0;
  uint32_t end_frame, flags = 0;
  int err = ORC_OK;

  g_open_n = 0;
  memset(res, 0, sizeof(*res));
  if (is_delta ? (window_bits < 17 || window_bits > 25) : (window_bits < 15 || window_bits > 21)) { res->err = ORC_ARGS; return ORC_ARGS; }
  if (reset_frames < 0 || (ref_len && !is_delta) || ref_len > ((size_t) 1 << window_bits)) { res->err = ORC_ARGS; return ORC_ARGS; }
  init_slots();
  z = (lzx_t *) calloc(1, sizeof(*z));
  z->win = (uint8_t *) calloc(1, (size_t) 1 << window_bits);
  z->b.in = in; z->b.in_len = in_len;
  z->wsize = 1u << window_bits;
  z->reset_frames = (uint32_t) reset_frames;
  z->length = length;
  z->num_offsets = (uint32_t) slots_for_bits[window_bits - 15] << 3;
  z->is_delta = is_delta; z->ref_size = (uint32_t) ref_len;
  if (ref_len) memcpy(z->win + z->wsize - ref_len, ref, ref_len);
  reset_state(z);
  if (out_bytes == 0) goto done;

  end_frame = (uint32_t)((z->offset + out_bytes) / FRAME) + 1;       /* lzxd.c:419 */
  while (z->frame < end_frame) {
    uint32_t frame_size;
    int32_t todo;
    unsigned v, hi, lo;
    const uint8_t *fsrc;
    uint8_t e8buf[FRAME];

    if (z->reset_frames && (z->frame % z->reset_frames) == 0) {
      /* a block still open at a reset point: the reference says "WARNING; invalid reset interval detected during LZX
       * decompression" (once per call) and decodes on (lzxd.c:423-431).  Which frames: oracle_lzx_open_resets() */
      if (z->block_remaining) { if (g_open_n < ORC_OPEN_MAX) g_open_frames[g_open_n] = z->frame; g_open_n++; }
      reset_state(z);
    }
    if (z->is_delta) { if (ensure(&z->b, 16)) goto fail; DROP(&z->b, 16); }     /* chunk size, lzxd.c:440-444 */
    if (!z->header_read) {
      hi = lo = 0;
      if (rd_bits(&z->b, 1, &v)) goto fail;
      if (v) { if (rd_bits(&z->b, 16, &hi) || rd_bits(&z->b, 16, &lo)) goto fail; }
      z->intel_filesize = (int32_t)((hi << 16) | lo);
      if (z->intel_filesize) flags |= ORC_F_INTEL_HEADER;
      z->header_read = 1;
    }
    frame_size = FRAME;
    if (z->length && (z->length - z->offset) < (uint64_t) frame_size)
      frame_size = (uint32_t)(z->length - z->offset);

    todo = (int32_t)(z->frame_posn + frame_size - z->wpos);
    while (todo > 0) {
      int32_t run;
      if (z->block_remaining == 0) { if (block_header(z)) goto fail; }
      run = (int32_t) z->block_remaining;
      if (run > todo) run = todo;
      todo -= run; z->block_remaining -= (uint32_t) run;
      switch (z->block_type) {
      case BT_ALIGNED: case BT_VERBATIM:
        run = decode_run(z, run);
        if (run == INT32_MIN) goto fail;
        break;
      case BT_UNCOMPRESSED: {
        uint8_t *dst = &z->win[z->wpos];
        z->wpos += (uint32_t) run;
        while (run > 0) { if (rd_byte(&z->b, &v)) goto fail; *dst++ = (uint8_t) v; run--; }
        break; }
      default: z->b.err = ORC_DECRUNCH; goto fail;
      }
      if (run < 0) {                                                 /* lzxd.c:678-685 */
        if ((uint32_t)(-run) > z->block_remaining) { z->b.err = ORC_DECRUNCH; goto fail; }
        z->block_remaining -= (uint32_t)(-run);
      }
    }
    if ((z->wpos - z->frame_posn) != frame_size) { z->b.err = ORC_DECRUNCH; goto fail; }

    if (z->b.bl > 0) { if (ensure(&z->b, 16)) goto fail; }          /* lzxd.c:695-697 */
    if (z->b.bl & 15) DROP(&z->b, z->b.bl & 15);
    if (frame_size) {                     /* diagnostics for callers that chain units (CHM reset intervals) */
      in_next = z->b.pos - (uint64_t)(z->b.bl >> 3);
      flags = z->block_remaining ? (flags | ORC_F_BLOCK_OPEN) : (flags & ~ORC_F_BLOCK_OPEN);
    }

    fsrc = &z->win[z->frame_posn];
    if (z->intel_started && z->intel_filesize && z->frame < 32768 && frame_size > 10) {
      uint8_t *d = e8buf, *dend = e8buf + frame_size - 10;
      int32_t curpos = (int32_t)((uint32_t) e8_base + (uint32_t) z->offset);
      int32_t filesize = z->intel_filesize;
      memcpy(e8buf, fsrc, frame_size);
      while (d < dend) {
        int32_t abs_off, rel;
        if (*d++ != 0xE8) { curpos++; continue; }
        abs_off = (int32_t)(d[0] | (d[1] << 8) | (d[2] << 16) | ((uint32_t) d[3] << 24));
        if (abs_off >= -curpos && abs_off < filesize) {
          rel = (abs_off >= 0) ? abs_off - curpos : abs_off + filesize;
          d[0] = (uint8_t) rel; d[1] = (uint8_t)(rel >> 8);
          d[2] = (uint8_t)(rel >> 16); d[3] = (uint8_t)(rel >> 24);
        }
        d += 4; curpos += 5;
      }
      fsrc = e8buf;
      flags |= ORC_F_E8_APPLIED;
    }
    {
      uint32_t n = (remaining < (uint64_t) frame_size) ? (uint32_t) remaining : frame_size;
      if (out && written < out_cap) {
        size_t room = out_cap - (size_t) written;
        memcpy(out + written, fsrc, n < room ? n : room);
      }
      written += n; z->offset += n; remaining -= n;
    }
    z->frame_posn += frame_size; z->frame++;
    if (z->wpos == z->wsize) z->wpos = 0;
    if (z->frame_posn == z->wsize) z->frame_posn = 0;
    continue;
fail:
    err = z->b.err ? z->b.err : ORC_DECRUNCH;
    /* the look-ahead frame (lzxd.c:419) may run dry after every requested byte is out */
    if (err == ORC_READ && remaining == 0) flags |= ORC_F_LOOKAHEAD_READ;
    break;
  }
  if (err == ORC_OK && remaining) err = ORC_DECRUNCH;                /* lzxd.c:758-761 */
done:
  res->err = err; res->flags = flags; res->out_len = written; res->in_used = z->b.pos; res->in_next = in_next;
  free(z->win); free(z);
  return err;
}

int oracle_huff_accepts(const uint8_t *lens, int nsyms, int tablebits) {
  return oh_accepts(lens, nsyms, tablebits) ? 0 : 1;
}
