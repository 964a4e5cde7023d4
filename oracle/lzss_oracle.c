/* oracle/lzss_oracle.c -- TEST INFRASTRUCTURE ONLY: CPU restatement of the two small single-stream codecs
 * of SZDD / KWAJ files, buffer to buffer.
 *   LZSS (SZDD, KWAJ method 2, MS Help) ... libmspack/mspack/lzssd.c:36-91: 4096-byte ring filled with
 *       spaces, first write position 4096-16 (QBASIC: 4096-18), control bytes LSB first (MSHELP: inverted),
 *       1 = literal, 0 = (ring position, length 3..18); decoding simply ends where the input ends.
 *   KWAJ LZH (method 3) .................. libmspack/mspack/kwajd.c:432-563: five Huffman trees (lengths in
 *       one of four encodings), MSB-first bits fed a byte at a time, a literal-run / match state, the same
 *       ring; the stream has no end marker: past the end the reader feeds zero bytes and the decoder stops
 *       at the first symbol or field that used one of those bits (kwajd.c:392-410, 548-563).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"
#include "oracle_huff.h"

int oracle_lzss_decode(const uint8_t *in, size_t in_len, int mode, uint8_t *out, size_t out_cap, oracle_result *res)
{
  uint8_t win[4096];
  size_t ip = 0, op = 0;
  unsigned pos, invert;
  memset(res, 0, sizeof(*res));
  if (mode < 0 || mode > 2) { res->err = ORC_ARGS; return ORC_ARGS; }
  memset(win, 0x20, sizeof(win));
  pos = 4096 - (mode == 2 ? 18 : 16);
  invert = (mode == 1) ? 0xFF : 0;
  for (;;) {
    unsigned c, i;
    if (ip >= in_len) break;
    c = in[ip++] ^ invert;
    for (i = 1; i & 0xFF; i <<= 1) {
      if (c & i) {
        if (ip >= in_len) goto done;
        win[pos] = in[ip++];
        if (op < out_cap) out[op] = win[pos];
        op++; pos = (pos + 1) & 4095;
      }
      else {
        unsigned mpos, len;
        if (ip >= in_len) goto done;
        mpos = in[ip++];
        if (ip >= in_len) goto done;
        mpos |= (in[ip] & 0xF0u) << 4;
        len = (in[ip++] & 0x0F) + 3;
        while (len--) {
          win[pos] = win[mpos];
          if (op < out_cap) out[op] = win[pos];
          op++; pos = (pos + 1) & 4095; mpos = (mpos + 1) & 4095;
        }
      }
    }
  }
done:
  res->err = ORC_OK; res->out_len = op; res->in_used = ip;
  return ORC_OK;
}

/* ---- KWAJ LZH ------------------------------------------------------------------------------------------- */
typedef struct {
  const uint8_t *in; size_t in_len, ip;
  uint32_t bb; int bl;              /* MSB-first bit buffer */
  int input_end;                    /* 0, or 8 * (zero bytes fed after the end of the input) */
} lzh_bits;

static void lzh_ensure(lzh_bits *b, int n) {               /* kwajd.c:374-381, 548-563 */
  while (b->bl < n) {
    unsigned byte = 0;
    if (b->ip < b->in_len) byte = b->in[b->ip++];
    else b->input_end += 8;
    b->bb |= byte << (32 - 8 - b->bl);
    b->bl += 8;
  }
}
/* returns 1 when the read used bits from beyond the end (the decoder then stops with OK) */
static int lzh_bits_safe(lzh_bits *b, int n, unsigned *v) {
  lzh_ensure(b, n);
  *v = n ? (b->bb >> (32 - n)) : 0;
  b->bb <<= n; b->bl -= n;
  return b->input_end && b->bl < b->input_end;
}
static int lzh_sym_safe(lzh_bits *b, const oh_table *t, int *sym, int *bad) {
  int l;
  lzh_ensure(b, 16);
  *sym = oh_decode(t, b->bb >> 16, &l);
  if (*sym < 0) { *bad = 1; return 1; }
  b->bb <<= l; b->bl -= l;
  return b->input_end && b->bl < b->input_end;
}

static int lzh_read_lens(lzh_bits *b, unsigned type, unsigned n, uint8_t *lens) {   /* kwajd.c:497-546 */
  unsigned i, c, sel;
  switch (type) {
  case 0:
    c = (n == 16) ? 4 : (n == 32) ? 5 : (n == 64) ? 6 : (n == 256) ? 8 : 0;
    for (i = 0; i < n; i++) lens[i] = (uint8_t) c;
    break;
  case 1:
    if (lzh_bits_safe(b, 4, &c)) return 1;
    lens[0] = (uint8_t) c;
    for (i = 1; i < n; i++) {
      if (lzh_bits_safe(b, 1, &sel)) return 1;
      if (sel == 0) lens[i] = (uint8_t) c;
      else {
        if (lzh_bits_safe(b, 1, &sel)) return 1;
        if (sel == 0) lens[i] = (uint8_t) ++c;
        else { if (lzh_bits_safe(b, 4, &c)) return 1; lens[i] = (uint8_t) c; }
      }
    }
    break;
  case 2:
    if (lzh_bits_safe(b, 4, &c)) return 1;
    lens[0] = (uint8_t) c;
    for (i = 1; i < n; i++) {
      if (lzh_bits_safe(b, 2, &sel)) return 1;
      if (sel == 3) { if (lzh_bits_safe(b, 4, &c)) return 1; }
      else c = (unsigned)(c + sel - 1);
      lens[i] = (uint8_t) c;
    }
    break;
  case 3:
    for (i = 0; i < n; i++) { if (lzh_bits_safe(b, 4, &c)) return 1; lens[i] = (uint8_t) c; }
    break;
  }
  return 0;
}

int oracle_kwaj_lzh_decode(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap, oracle_result *res)
{
  static const unsigned nsyms[5] = { 16, 16, 32, 64, 256 };
  uint8_t win[4096], lens[5][256 + 64];
  oh_table *tabs = (oh_table *) calloc(5, sizeof(oh_table));
  lzh_bits b;
  unsigned types[6], i, pos = 0, lit_run = 0, v;
  size_t op = 0;
  int err = ORC_OK, bad = 0, sym;

  memset(res, 0, sizeof(*res));
  memset(&b, 0, sizeof(b)); b.in = in; b.in_len = in_len;
  memset(win, 0x20, sizeof(win));
  memset(lens, 0, sizeof(lens));
  for (i = 0; i < 6; i++) { if (lzh_bits_safe(&b, 4, &types[i])) goto done; }
  for (i = 0; i < 5; i++) {
    if (lzh_read_lens(&b, types[i], nsyms[i], lens[i])) goto done;
    if (oh_build(&tabs[i], lens[i], (int) nsyms[i], 9)) { err = ORC_DATAFORMAT; goto done; }
  }
  while (!b.input_end) {
    int len;
    if (lzh_sym_safe(&b, &tabs[lit_run ? 1 : 0], &len, &bad)) break;
    if (len > 0) {
      unsigned offset;
      len += 2; lit_run = 0;
      if (lzh_sym_safe(&b, &tabs[3], &sym, &bad)) break;
      offset = (unsigned) sym << 6;
      if (lzh_bits_safe(&b, 6, &v)) break;
      offset |= v;
      while (len-- > 0) {
        win[pos] = win[(pos + 4096 - offset) & 4095];
        if (op < out_cap) out[op] = win[pos];
        op++; pos = (pos + 1) & 4095;
      }
    }
    else {
      if (lzh_sym_safe(&b, &tabs[2], &len, &bad)) break;
      len++;
      lit_run = (len == 32) ? 0 : 1;
      while (len-- > 0) {
        if (lzh_sym_safe(&b, &tabs[4], &sym, &bad)) goto out_loop;
        win[pos] = (uint8_t) sym;
        if (op < out_cap) out[op] = win[pos];
        op++; pos = (pos + 1) & 4095;
      }
    }
  }
out_loop:
  if (bad) err = ORC_DATAFORMAT;
done:
  res->err = err; res->out_len = op; res->in_used = b.ip;
  free(tabs);
  return err;
}
