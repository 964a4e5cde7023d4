/* qtm_enc.c -- a Quantum encoder for the synthetic corpora (test/bench infrastructure).
 *
 * The reference has no Quantum compressor (libmspack/mspack/qtmc.c is a stub); this one was written
 * against what qtmd.c accepts:
 *   nine adaptive models, mirrored update rule ...... qtmd.c:125-182,242-251
 *   16-bit carry-less arithmetic coder ............... inverse of GET_SYMBOL, qtmd.c:92-123
 *   selector / literal / match symbols, slot tables .. qtmd.c:307-350, 66-82
 *   per-frame restart H=0xFFFF,L=0, 16-bit C ......... qtmd.c:292-295
 *   frame end: byte align, bytes until 0xFF .......... qtmd.c:430-442 (the 0xFF is added by cabd.c:1330)
 * Raw extra bits are not arithmetic-coded: the decoder's C register runs 16 bits ahead, so they sit
 * in the stream after arithmetic-code bit number 16 + (renormalisation shifts so far in the frame).
 */
#include <stdlib.h>
#include <string.h>
#include "corpus.h"

#define FRAME 32768u
#define HBITS 15

typedef struct { uint16_t sym, cumfreq; } msym_t;
typedef struct { int shiftsleft, entries; msym_t syms[65]; } model_t;

static void model_init(model_t *m, int start, int len) {
  int i;
  m->shiftsleft = 4; m->entries = len;
  for (i = 0; i <= len; i++) { m->syms[i].sym = (uint16_t)(start + i); m->syms[i].cumfreq = (uint16_t)(len - i); }
}
static void model_update(model_t *m) {
  int i, j;
  if (--m->shiftsleft) {
    for (i = m->entries - 1; i >= 0; i--) {
      m->syms[i].cumfreq >>= 1;
      if (m->syms[i].cumfreq <= m->syms[i + 1].cumfreq) m->syms[i].cumfreq = (uint16_t)(m->syms[i + 1].cumfreq + 1);
    }
    return;
  }
  m->shiftsleft = 50;
  for (i = 0; i < m->entries; i++) {
    m->syms[i].cumfreq = (uint16_t)(m->syms[i].cumfreq - m->syms[i + 1].cumfreq);
    m->syms[i].cumfreq++;
    m->syms[i].cumfreq >>= 1;
  }
  for (i = 0; i < m->entries - 1; i++)
    for (j = i + 1; j < m->entries; j++)
      if (m->syms[i].cumfreq < m->syms[j].cumfreq) { msym_t t = m->syms[i]; m->syms[i] = m->syms[j]; m->syms[j] = t; }
  for (i = m->entries - 1; i >= 0; i--) m->syms[i].cumfreq = (uint16_t)(m->syms[i].cumfreq + m->syms[i + 1].cumfreq);
}

typedef struct { uint32_t at; uint32_t val; uint8_t nbits; } raw_t;
typedef struct {
  uint8_t *abits; size_t nabits, cap_abits;
  raw_t *raws; size_t nraws, cap_raws;
  uint32_t H, L, pending, shifts;
} frame_t;

static void abit(frame_t *f, int b) {
  if (f->nabits == f->cap_abits) { f->cap_abits = f->cap_abits ? f->cap_abits * 2 : 1 << 16; f->abits = (uint8_t *) realloc(f->abits, f->cap_abits); }
  f->abits[f->nabits++] = (uint8_t) b;
}
static void abit_plus_pending(frame_t *f, int b) {
  abit(f, b);
  while (f->pending) { abit(f, !b); f->pending--; }
}
static void put_raw(frame_t *f, uint32_t val, int nbits) {
  if (!nbits) return;
  if (f->nraws == f->cap_raws) { f->cap_raws = f->cap_raws ? f->cap_raws * 2 : 4096; f->raws = (raw_t *) realloc(f->raws, f->cap_raws * sizeof(raw_t)); }
  f->raws[f->nraws].at = f->shifts; f->raws[f->nraws].val = val; f->raws[f->nraws].nbits = (uint8_t) nbits; f->nraws++;
}
static void encode_symbol(frame_t *f, model_t *m, int sym) {
  int k = 0, i;
  uint32_t range, tot, H = f->H, L = f->L;
  while (m->syms[k].sym != sym) k++;
  range = (H - L) + 1; tot = m->syms[0].cumfreq;
  H = L + (m->syms[k].cumfreq * range) / tot - 1;
  L = L + (m->syms[k + 1].cumfreq * range) / tot;
  H &= 0xFFFF; L &= 0xFFFF;
  i = k + 1;
  do { m->syms[--i].cumfreq += 8; } while (i > 0);
  if (m->syms[0].cumfreq > 3800) model_update(m);
  for (;;) {
    if ((L & 0x8000) != (H & 0x8000)) {
      if ((L & 0x4000) && !(H & 0x4000)) { f->pending++; L &= 0x3FFF; H |= 0x4000; }
      else break;
    }
    else abit_plus_pending(f, (int)(L >> 15) & 1);
    L = (L << 1) & 0xFFFF; H = ((H << 1) | 1) & 0xFFFF;
    f->shifts++;
  }
  f->H = H; f->L = L;
}

static uint32_t pos_base[42]; static uint8_t pos_extra[42], len_base[27], len_extra[27];
static void init_tables(void) {
  unsigned i, off;
  if (pos_base[1]) return;
  for (i = 0, off = 0; i < 42; i++) { pos_base[i] = off; pos_extra[i] = (uint8_t)(((i < 2) ? 0 : (i - 2)) >> 1); off += 1u << pos_extra[i]; }
  for (i = 0, off = 0; i < 26; i++) { len_base[i] = (uint8_t) off; len_extra[i] = (uint8_t)(((i < 2) ? 0 : (i - 2)) >> 2); off += 1u << len_extra[i]; }
  len_base[26] = 254; len_extra[26] = 0;
}
static int pos_slot(uint32_t v) { int s = 41; while (pos_base[s] > v) s--; return s; }
static int len_slot(uint32_t v) { int s = 26; while (len_base[s] > v) s--; return s; }

static inline uint32_t hash3(const uint8_t *p) {
  return (((uint32_t) p[0] << 16 | (uint32_t) p[1] << 8 | p[2]) * 2654435761u) >> (32 - HBITS);
}

size_t mspk_qtm_bound(size_t n) { return n + n / 4 + (n / FRAME + 2) * 64 + 1024; }

size_t mspk_qtm_encode(const uint8_t *src, size_t n, int window_bits, int chain_depth,
                       uint8_t *dst, size_t dst_cap, uint32_t *frame_size)
{
  model_t m0, m1, m2, m3, m4, m5, m6, m6l, m7;
  frame_t f;
  int32_t *head, *prev;
  size_t p = 0, outn = 0, nframes = (n + FRAME - 1) / FRAME, fi;
  uint32_t wsize, max4, max5, max6;
  int wb2 = window_bits * 2, n4, n5;

  if (window_bits < 10 || window_bits > 21) return 0;
  init_tables();
  if (chain_depth <= 0) chain_depth = 16;
  wsize = 1u << window_bits;
  n4 = wb2 > 24 ? 24 : wb2; n5 = wb2 > 36 ? 36 : wb2;
  model_init(&m0, 0, 64); model_init(&m1, 64, 64); model_init(&m2, 128, 64); model_init(&m3, 192, 64);
  model_init(&m4, 0, n4); model_init(&m5, 0, n5); model_init(&m6, 0, wb2); model_init(&m6l, 0, 27); model_init(&m7, 0, 7);
  /* largest offset each selector can express (offset-1 < base[last]+2^extra[last]), capped by the window */
  max4 = pos_base[n4 - 1] + (1u << pos_extra[n4 - 1]); if (max4 > wsize) max4 = wsize;
  max5 = pos_base[n5 - 1] + (1u << pos_extra[n5 - 1]); if (max5 > wsize) max5 = wsize;
  max6 = wsize;
  memset(&f, 0, sizeof(f));
  head = (int32_t *) calloc((size_t) 1 << HBITS, sizeof(int32_t));
  prev = (int32_t *) malloc(sizeof(int32_t) * (n + 8));

  for (fi = 0; fi < nframes; fi++) {
    size_t fend = (fi + 1) * FRAME < n ? (fi + 1) * FRAME : n, i, ri, total_bits, bitpos;
    uint8_t *o; size_t obytes;
    f.nabits = 0; f.nraws = 0; f.H = 0xFFFF; f.L = 0; f.pending = 0; f.shifts = 0;
    while (p < fend) {
      int best = 0, depth = chain_depth, maxl = (int)((fend - p) < 259 ? (fend - p) : 259);
      uint32_t boff = 0;
      if (p + 2 < n && maxl >= 3) {
        int32_t c = head[hash3(src + p)];
        while (c > 0 && depth-- > 0) {
          size_t q = (size_t)(c - 1);
          uint32_t o2 = (uint32_t)(p - q);
          int l = 0;
          if (o2 > max6) break;
          while (l < maxl && src[q + l] == src[p + l]) l++;
          if (l > best && ((l >= 5) || (l == 4 && o2 <= max5) || (l == 3 && o2 <= max4 && o2 < 2048))) { best = l; boff = o2; }
          if (best >= maxl) break;
          c = prev[c - 1];
        }
      }
      if (best >= 3) {
        int s, e; uint32_t v = boff - 1;
        size_t q;
        if (best == 3) { encode_symbol(&f, &m7, 4); s = pos_slot(v); encode_symbol(&f, &m4, s); put_raw(&f, v - pos_base[s], pos_extra[s]); }
        else if (best == 4) { encode_symbol(&f, &m7, 5); s = pos_slot(v); encode_symbol(&f, &m5, s); put_raw(&f, v - pos_base[s], pos_extra[s]); }
        else {
          encode_symbol(&f, &m7, 6);
          e = len_slot((uint32_t)(best - 5)); encode_symbol(&f, &m6l, e); put_raw(&f, (uint32_t)(best - 5) - len_base[e], len_extra[e]);
          s = pos_slot(v); encode_symbol(&f, &m6, s); put_raw(&f, v - pos_base[s], pos_extra[s]);
        }
        for (q = p; q < p + (size_t) best; q++) if (q + 2 < n) { uint32_t h = hash3(src + q); prev[q] = head[h]; head[h] = (int32_t) q + 1; }
        p += (size_t) best;
      }
      else {
        uint8_t c = src[p];
        encode_symbol(&f, &m7, c >> 6);
        encode_symbol(&f, c < 64 ? &m0 : c < 128 ? &m1 : c < 192 ? &m2 : &m3, c);
        if (p + 2 < n) { uint32_t h = hash3(src + p); prev[p] = head[h]; head[h] = (int32_t) p + 1; }
        p++;
      }
    }
    /* terminate the arithmetic code, then make sure the decoder's 16-bit look-ahead is covered */
    f.pending++;
    abit_plus_pending(&f, f.L >= 0x4000 ? 1 : 0);
    while (f.nabits < 16u + f.shifts) abit(&f, 0);
    /* splice: raw chunk recorded at shift count s goes in front of arithmetic bit 16+s */
    total_bits = f.nabits;
    for (ri = 0; ri < f.nraws; ri++) total_bits += f.raws[ri].nbits;
    obytes = (total_bits + 7) / 8;
    if (outn + obytes > dst_cap) { outn = 0; break; }
    o = dst + outn; memset(o, 0, obytes);
    bitpos = 0; ri = 0;
    for (i = 0; i <= f.nabits; i++) {
      while (ri < f.nraws && 16u + f.raws[ri].at == i) {
        int b;
        for (b = f.raws[ri].nbits - 1; b >= 0; b--, bitpos++)
          if ((f.raws[ri].val >> b) & 1) o[bitpos >> 3] |= (uint8_t)(0x80 >> (bitpos & 7));
        ri++;
      }
      if (i < f.nabits) { if (f.abits[i]) o[bitpos >> 3] |= (uint8_t)(0x80 >> (bitpos & 7)); bitpos++; }
    }
    if (frame_size) frame_size[fi] = (uint32_t) obytes;
    outn += obytes;
  }
  free(head); free(prev); free(f.abits); free(f.raws);
  return outn;
}
