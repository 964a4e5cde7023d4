/* lzx_enc.c -- an LZX encoder for the synthetic corpora (test/bench infrastructure).
 *
 * The reference has no LZX compressor (libmspack/mspack/lzxc.c is a stub), so this one was
 * written against what lzxd.c accepts:
 *   interval header (1 bit [+32]) .......... lzxd.c:447-453
 *   block header: 3-bit type, 24-bit size .. lzxd.c:477-479
 *   aligned tree 8x3 bits, pretree-coded main/length trees (delta mod 17 vs. previous block,
 *   run codes 17/18/19) ..................... lzxd.c:484-499,138-183
 *   uncompressed block padding + R0-R2 ..... lzxd.c:501-517,469-474
 *   main symbol = 256 + slot*8 + min(len-2,7), length tree for len-9, slots 0..2 = R0..R2 LRU,
 *   extra bits / aligned low 3 bits ......... lzxd.c:542-586
 *   16-bit re-alignment after every frame .. lzxd.c:695-697
 *   E8 call translation (inverse applied here) lzxd.c:706-736
 * Matches never cross a frame or block end (lzxd.c:678-693 rejects that).
 */
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include "corpus.h"
#include "huff_enc.h"

#define FRAME 32768u
#define MINM 2
#define MAXM 257
#define HBITS 15
#define MAXSLOTS 290
#define MAXM_DELTA 32768
#define MAINSYMS (256 + MAXSLOTS * 8)

static const uint16_t slots_for_bits[11] = { 30, 32, 34, 36, 38, 42, 50, 66, 98, 162, 290 };
static uint32_t slot_base[MAXSLOTS + 1];
static uint8_t  slot_extra[MAXSLOTS + 1];
static void init_slots_once(void) {
  uint32_t base = 0; int i;
  for (i = 0; i <= MAXSLOTS; i++) {
    int e = (i < 4) ? 0 : (i < 36 ? (i / 2) - 1 : 17);
    slot_base[i] = base; slot_extra[i] = (uint8_t) e; base += 1u << e;
  }
}
/* (the batch generators encode on several threads: the first call of each used to fill the tables unsynchronised -- the same
 * values from every thread, but a data race all the same: tests/hostcheck under ThreadSanitizer) */
static pthread_once_t slots_once = PTHREAD_ONCE_INIT;
static void init_slots(void) { pthread_once(&slots_once, init_slots_once); }
static int slot_of(uint32_t formatted) {
  int lo = 0, hi = MAXSLOTS - 1;      /* (callers never pass offsets beyond their window) */
  while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (slot_base[mid] <= formatted) lo = mid; else hi = mid - 1; }
  return lo;
}

/* ---- bit writer: MSB-first into little-endian 16-bit words (lzxd.c:85-91) -------------------------- */
typedef struct { uint8_t *p; size_t cap, n; uint64_t acc; int nbits; int overflow; } bw_t;
static void bw_word(bw_t *w, unsigned v) {
  if (w->n + 2 > w->cap) { w->overflow = 1; return; }
  w->p[w->n++] = (uint8_t) v; w->p[w->n++] = (uint8_t)(v >> 8);
}
static void bw_put(bw_t *w, uint32_t v, int n) {
  if (n == 0) return;
  w->acc = (w->acc << n) | (v & ((n >= 32) ? 0xFFFFFFFFu : ((1u << n) - 1)));
  w->nbits += n;
  while (w->nbits >= 16) { bw_word(w, (unsigned)(w->acc >> (w->nbits - 16)) & 0xFFFF); w->nbits -= 16; }
}
static void bw_align(bw_t *w) { if (w->nbits) bw_put(w, 0, 16 - w->nbits); }
static void bw_byte(bw_t *w, unsigned v) {          /* only while word-aligned */
  if (w->n + 1 > w->cap) { w->overflow = 1; return; }
  w->p[w->n++] = (uint8_t) v;
}

typedef struct { uint32_t off; uint16_t len; uint8_t rslot; uint8_t lit; } tok_t;

/* encoder statistics (diagnostics for DESIGN.md / bench): totals since process start */
unsigned long long mspk_lzx_stat_tokens = 0, mspk_lzx_stat_literals = 0, mspk_lzx_stat_match_bytes = 0;
unsigned long long mspk_lzx_stat_offhist[24];   /* matches by floor(log2(offset)) */
unsigned long long mspk_lzx_stat_lenhist[10];   /* matches by length: 2,3,4,5-8,9-16,17-32,33-64,65-128,129+ */

typedef struct {
  const uint8_t *src;            /* (possibly E8-pretranslated) plaintext                      */
  size_t istart, iend;           /* current interval                                            */
  uint32_t wmax;                 /* largest legal offset                                        */
  int32_t *head, *prev;          /* hash chains, positions relative to istart (+1)              */
  uint32_t R[3];
  int depth, lazy, use_rep;
  int maxm;                      /* longest match: 257, or 32768 in DELTA streams              */
  size_t bias;                   /* DELTA: bytes of reference data in front of the plaintext    */
} mf_t;

static inline uint32_t hash3(const uint8_t *p) {
  return (((uint32_t) p[0] << 16 | (uint32_t) p[1] << 8 | p[2]) * 2654435761u) >> (32 - HBITS);
}
static inline void mf_insert(mf_t *m, size_t p) {
  if (p + 2 < m->iend) {
    uint32_t h = hash3(m->src + p);
    m->prev[p - m->istart] = m->head[h];
    m->head[h] = (int32_t)(p - m->istart) + 1;
  }
}
static inline int match_len(const uint8_t *a, const uint8_t *b, int maxl) {
  int l = 0;
  while (l < maxl && a[l] == b[l]) l++;
  return l;
}
/* best match at p (length capped at maxl); returns length (0 = none), *off, *rslot (3 = explicit) */
static int mf_find(mf_t *m, size_t p, int maxl, uint32_t *off, int *rslot) {
  const uint8_t *s = m->src;
  int best = 0, k, depth = m->depth;
  uint32_t boff = 0; int bslot = 3;
  size_t avail = p - m->istart;
  if (maxl < MINM) return 0;
  if (m->use_rep) {
    for (k = 0; k < 3; k++) {
      uint32_t o = m->R[k];
      if (o >= 1 && o <= avail && o <= m->wmax) {
        int l = match_len(s + p, s + p - o, maxl);
        if (l >= MINM && l > best) { best = l; boff = o; bslot = k; }
      }
    }
  }
  if (p + 2 < m->iend && maxl >= 3) {
    int32_t c = m->head[hash3(s + p)];
    while (c > 0 && depth-- > 0) {
      size_t q = m->istart + (size_t)(c - 1);
      uint32_t o = (uint32_t)(p - q);
      if (o > m->wmax) break;
      if (best == 0 || s[q + best] == s[p + best]) {
        int l = match_len(s + p, s + q, maxl);
        /* an explicit offset must beat a repeat match by 2+, and 3-byte matches must be near */
        if (l >= 3 && l > best + (bslot < 3 ? 1 : 0) && (l > 3 || o < 4096)) { best = l; boff = o; bslot = 3; }
      }
      if (best >= maxl) break;
      c = m->prev[c - 1];
    }
  }
  *off = boff; *rslot = bslot;
  return best;
}
static void r_update(uint32_t *R, int rslot, uint32_t off) {
  if (rslot == 0) return;
  if (rslot == 1) { R[1] = R[0]; R[0] = off; }
  else if (rslot == 2) { R[2] = R[0]; R[0] = off; }
  else { R[2] = R[1]; R[1] = R[0]; R[0] = off; }
}

/* tokenise [bstart,bend): never across a 32 KiB frame end */
static size_t parse_block(mf_t *m, size_t bstart, size_t bend, tok_t *toks) {
  size_t p = bstart, nt = 0;
  while (p < bend) {
    size_t fend = ((p - m->bias) / FRAME + 1) * FRAME + m->bias;
    size_t lim = bend < fend ? bend : fend;
    int maxl = (int)((lim - p) < (size_t) m->maxm ? (lim - p) : (size_t) m->maxm);
    uint32_t off = 0; int rs = 3;
    int len = mf_find(m, p, maxl, &off, &rs);
    if (len >= MINM && m->lazy && len < 40 && p + 1 < lim) {
      uint32_t off2; int rs2;
      int maxl2 = (int)((lim - p - 1) < (size_t) m->maxm ? (lim - p - 1) : (size_t) m->maxm);
      int len2;
      mf_insert(m, p);
      len2 = mf_find(m, p + 1, maxl2, &off2, &rs2);
      if (len2 > len + (rs < 3 ? 1 : 0)) {           /* defer: emit a literal now */
        toks[nt].len = 0; toks[nt].lit = m->src[p]; toks[nt].off = 0; toks[nt].rslot = 0; nt++;
        p++;
        continue;
      }
      toks[nt].len = (uint16_t) len; toks[nt].off = off; toks[nt].rslot = (uint8_t) rs; toks[nt].lit = 0; nt++;
      r_update(m->R, rs, off);
      { size_t q; for (q = p + 1; q < p + (size_t) len; q++) mf_insert(m, q); }
      p += (size_t) len;
      continue;
    }
    if (len >= MINM) {
      size_t q;
      toks[nt].len = (uint16_t) len; toks[nt].off = off; toks[nt].rslot = (uint8_t) rs; toks[nt].lit = 0; nt++;
      r_update(m->R, rs, off);
      for (q = p; q < p + (size_t) len; q++) mf_insert(m, q);
      p += (size_t) len;
    }
    else {
      toks[nt].len = 0; toks[nt].lit = m->src[p]; toks[nt].off = 0; toks[nt].rslot = 0; nt++;
      mf_insert(m, p);
      p++;
    }
  }
  return nt;
}

/* ---- code-length transmission (lzxd.c:138-183) ------------------------------------------------------- */
typedef struct { uint8_t sym, extra_bits, sym2; uint16_t extra; } plsym_t;

static void write_lens(bw_t *w, const uint8_t *prev, const uint8_t *cur, int first, int last)
{
  plsym_t seq[MAINSYMS + 64];
  uint32_t freq[20];
  uint8_t plen[20];
  uint16_t pcode[20];
  int ns = 0, x = first, i;
  memset(freq, 0, sizeof(freq));
  while (x < last) {
    int run = 1;
    while (x + run < last && cur[x + run] == cur[x]) run++;
    if (cur[x] == 0 && run >= 20) {
      int r = run > 51 ? 51 : run;
      seq[ns].sym = 18; seq[ns].extra_bits = 5; seq[ns].extra = (uint16_t)(r - 20); freq[18]++; ns++; x += r;
    }
    else if (cur[x] == 0 && run >= 4) {
      int r = run > 19 ? 19 : run;
      seq[ns].sym = 17; seq[ns].extra_bits = 4; seq[ns].extra = (uint16_t)(r - 4); freq[17]++; ns++; x += r;
    }
    else if (run >= 4) {
      int r = run > 5 ? 5 : run;
      int z = ((int) prev[x] - (int) cur[x] + 17) % 17;
      seq[ns].sym = 19; seq[ns].extra_bits = 1; seq[ns].extra = (uint16_t)(r - 4); seq[ns].sym2 = (uint8_t) z;
      freq[19]++; freq[z]++; ns++; x += r;
    }
    else {
      int z = ((int) prev[x] - (int) cur[x] + 17) % 17;
      seq[ns].sym = (uint8_t) z; seq[ns].extra_bits = 0; seq[ns].extra = 0; freq[z]++; ns++; x++;
    }
  }
  he_build_lengths(freq, 20, 15, plen);
  he_assign_codes(plen, 20, pcode);
  for (i = 0; i < 20; i++) bw_put(w, plen[i], 4);
  for (i = 0; i < ns; i++) {
    bw_put(w, pcode[seq[i].sym], plen[seq[i].sym]);
    if (seq[i].extra_bits) bw_put(w, seq[i].extra, seq[i].extra_bits);
    if (seq[i].sym == 19) bw_put(w, pcode[seq[i].sym2], plen[seq[i].sym2]);
  }
}

typedef struct {
  uint8_t main_len[MAINSYMS + 8], len_len[256];   /* lengths of the previous block (delta base) */
} lens_state_t;

/* DELTA: the 16-bit chunk size in front of every frame; the decoder skips it (lzxd.c:440-444) */
static void put_chunk_size(bw_t *w, size_t pos, size_t total) {
  size_t left = total - pos;
  bw_put(w, (uint32_t)(left < FRAME ? left : FRAME) & 0xFFFF, 16);
}

static void emit_compressed_block(bw_t *w, lens_state_t *ls, const tok_t *toks, size_t nt,
                                  uint32_t block_bytes, int num_main, int want_type,
                                  size_t pos, uint64_t *frame_off, int delta, size_t total)
{
  uint32_t fmain[MAINSYMS], flen[256], fali[8];
  uint8_t main_len[MAINSYMS + 8], len_len[256], ali_len[8];
  uint16_t main_code[MAINSYMS], len_code[256], ali_code[8];
  uint64_t ali_count = 0, cost_ali, cost_verb;
  size_t i;
  int type, k;

  memset(fmain, 0, sizeof(fmain)); memset(flen, 0, sizeof(flen)); memset(fali, 0, sizeof(fali));
  for (i = 0; i < nt; i++) {
    const tok_t *t = &toks[i];
    if (t->len == 0) { fmain[t->lit]++; continue; }
    {
      int lh = (t->len > MAXM ? MAXM : t->len) - MINM, slot;
      if (t->rslot < 3) slot = t->rslot;
      else {
        uint32_t f = t->off + 2;
        slot = slot_of(f);
        if (slot_extra[slot] >= 3) { fali[(f - slot_base[slot]) & 7]++; ali_count++; }
      }
      fmain[256 + (slot << 3) + (lh < 7 ? lh : 7)]++;
      if (lh >= 7) flen[lh - 7]++;
    }
  }
  he_build_lengths(fali, 8, 7, ali_len);
  for (k = 0, cost_ali = 24; k < 8; k++) cost_ali += (uint64_t) fali[k] * ali_len[k];
  cost_verb = ali_count * 3;
  type = (want_type == 1 || want_type == 2) ? want_type : (cost_ali < cost_verb ? 2 : 1);
  if (type == 2) {          /* aligned tree must be a complete code over 3-bit lengths */
    int used = 0;
    for (k = 0; k < 8; k++) used += (ali_len[k] != 0);
    if (used < 2) for (k = 0; k < 8; k++) ali_len[k] = 3;
    he_assign_codes(ali_len, 8, ali_code);
  }
  memset(main_len, 0, sizeof(main_len)); memset(len_len, 0, sizeof(len_len));
  he_build_lengths(fmain, num_main, 16, main_len);
  he_build_lengths(flen, 249, 16, len_len);
  he_assign_codes(main_len, num_main, main_code);
  he_assign_codes(len_len, 249, len_code);

  bw_put(w, (uint32_t) type, 3);
  bw_put(w, block_bytes >> 8, 16); bw_put(w, block_bytes & 0xFF, 8);
  if (type == 2) for (k = 0; k < 8; k++) bw_put(w, ali_len[k], 3);
  write_lens(w, ls->main_len, main_len, 0, 256);
  write_lens(w, ls->main_len, main_len, 256, num_main);
  write_lens(w, ls->len_len, len_len, 0, 249);
  memcpy(ls->main_len, main_len, (size_t) num_main);
  memcpy(ls->len_len, len_len, 249);

  for (i = 0; i < nt; i++) {
    const tok_t *t = &toks[i];
    if (t->len == 0) { bw_put(w, main_code[t->lit], main_len[t->lit]); pos++; }
    else {
      int lh = (t->len > MAXM ? MAXM : t->len) - MINM, slot, ms;
      uint32_t f = 0;
      if (t->rslot < 3) slot = t->rslot;
      else { f = t->off + 2; slot = slot_of(f); }
      ms = 256 + (slot << 3) + (lh < 7 ? lh : 7);
      bw_put(w, main_code[ms], main_len[ms]);
      if (lh >= 7) bw_put(w, len_code[lh - 7], len_len[lh - 7]);
      if (slot >= 3) {
        int e = slot_extra[slot];
        uint32_t x = f - slot_base[slot];
        if (type == 2 && e >= 3) {
          if (e > 3) bw_put(w, x >> 3, e - 3);
          bw_put(w, ali_code[x & 7], ali_len[x & 7]);
        }
        else if (e) {
          if (e > 16) { bw_put(w, x >> 16, e - 16); bw_put(w, x & 0xFFFF, 16); }
          else bw_put(w, x, e);
        }
      }
      if (delta && t->len >= MAXM) {     /* length 257 announces an extension (lzxd.c:588-611) */
        uint32_t x = (uint32_t) t->len - MAXM;
        if (x < 0x100) { bw_put(w, 0, 1); bw_put(w, x, 8); }
        else if (x < 0x100 + 0x400) { bw_put(w, 2, 2); bw_put(w, x - 0x100, 10); }
        else if (x < 0x500 + 0x1000) { bw_put(w, 6, 3); bw_put(w, x - 0x500, 12); }
        else { bw_put(w, 7, 3); bw_put(w, x, 15); }
      }
      pos += t->len;
    }
    if ((pos % FRAME) == 0) {            /* frame complete: re-align to 16 bits (lzxd.c:695-697) */
      bw_align(w);
      if (frame_off) frame_off[pos / FRAME] = w->n;
      if (delta && pos < total) put_chunk_size(w, pos, total);
    }
  }
}

/* inverse of the decoder's E8 translation for one frame (lzxd.c:706-736) */
static void e8_pretranslate(uint8_t *frame, uint32_t frame_size, int32_t curpos, int32_t filesize)
{
  uint32_t i = 0;
  if (frame_size <= 10) return;
  while (i < frame_size - 10) {
    if (frame[i++] != 0xE8) { curpos++; continue; }
    {
      int64_t v = (int32_t)(frame[i] | (frame[i + 1] << 8) | (frame[i + 2] << 16) | ((uint32_t) frame[i + 3] << 24));
      if (v >= -(int64_t) curpos && v < (int64_t) filesize) {
        int32_t s = (v < (int64_t) filesize - curpos) ? (int32_t)(v + curpos) : (int32_t)(v - filesize);
        frame[i] = (uint8_t) s; frame[i + 1] = (uint8_t)(s >> 8);
        frame[i + 2] = (uint8_t)(s >> 16); frame[i + 3] = (uint8_t)(s >> 24);
      }
    }
    i += 4; curpos += 5;
  }
}

size_t mspk_lzx_bound(size_t n) { return n + n / 8 + (n / FRAME + 2) * 512 + 4096; }

size_t mspk_lzx_encode(const uint8_t *src_in, size_t n, int window_bits, int reset_frames,
                       const mspk_lzx_opts *opts_in, uint8_t *dst, size_t dst_cap,
                       uint64_t *frame_off)
{
  mspk_lzx_opts o;
  bw_t w;
  mf_t m;
  lens_state_t *ls;
  tok_t *toks;
  uint8_t *src, *srcbuf;
  size_t interval_bytes, istart, nframes = (n + FRAME - 1) / FRAME, fi, blk_no = 0, rl;
  int num_main;

  memset(&o, 0, sizeof(o));
  if (opts_in) o = *opts_in; else { o.use_repeats = 1; o.lazy = 1; }
  if (o.block_size <= 0) o.block_size = (int) FRAME;
  if (o.chain_depth <= 0) o.chain_depth = 24;
  if (o.delta ? (window_bits < 17 || window_bits > 25 || reset_frames != 0 || o.intel_filesize)
              : (window_bits < 15 || window_bits > 21 || o.ref_len)) return 0;
  rl = o.delta ? o.ref_len : 0;
  if (o.delta && rl + n > ((size_t) 1 << window_bits)) return 0;   /* reference data + output share the window */
  init_slots();
  num_main = 256 + (slots_for_bits[window_bits - 15] << 3);

  /* the match finder sees [reference data | plaintext]: a source inside the reference data is an
   * offset larger than the window position, which is exactly what the decoder resolves against the
   * end of its window (lzxd.c:618-642) */
  srcbuf = (uint8_t *) malloc(rl + n + 16);
  if (rl) memcpy(srcbuf, o.ref, rl);
  src = srcbuf + rl;
  memcpy(src, src_in, n);
  if (o.intel_filesize) {
    for (fi = 0; fi < nframes && fi < 32768; fi++) {
      size_t fs = fi * FRAME, fl = (n - fs) < FRAME ? (n - fs) : FRAME;
      e8_pretranslate(src + fs, (uint32_t) fl, (int32_t)((uint32_t) o.e8_base + (uint32_t) fs), o.intel_filesize);
    }
  }
  memset(&w, 0, sizeof(w)); w.p = dst; w.cap = dst_cap;
  memset(&m, 0, sizeof(m));
  m.src = src; m.depth = o.chain_depth; m.lazy = o.lazy; m.use_rep = o.use_repeats;
  m.maxm = o.delta ? MAXM_DELTA : MAXM;
  m.wmax = (1u << window_bits) - 3;
  m.head = (int32_t *) malloc(sizeof(int32_t) << HBITS);
  interval_bytes = reset_frames > 0 ? (size_t) reset_frames * FRAME : n;
  if (interval_bytes == 0) interval_bytes = FRAME;
  m.prev = (int32_t *) malloc(sizeof(int32_t) * ((interval_bytes < n ? interval_bytes : n) + rl) + 64);
  ls = (lens_state_t *) calloc(1, sizeof(*ls));
  toks = (tok_t *) malloc(sizeof(tok_t) * ((size_t) o.block_size + 8));

  for (istart = 0; istart < n || (n == 0 && istart == 0); istart += interval_bytes) {
    size_t iend = istart + interval_bytes < n ? istart + interval_bytes : n;
    size_t p = istart;
    int pending_pad = 0;
    /* interval start: encoder + decoder state restart (lzxd.c:257-270); always word-aligned here */
    memset(m.head, 0, sizeof(int32_t) << HBITS);
    m.istart = istart; m.iend = iend; m.R[0] = m.R[1] = m.R[2] = 1;
    if (rl) {                    /* DELTA: positions are taken from the start of the reference data */
      size_t q;
      m.src = srcbuf; m.istart = 0; m.iend = rl + iend; m.bias = rl;
      for (q = 0; q < rl; q++) mf_insert(&m, q);
    }
    memset(ls, 0, sizeof(*ls));
    if (frame_off) frame_off[istart / FRAME] = w.n;
    if (o.delta && n) put_chunk_size(&w, istart, n);
    if (o.intel_filesize) {
      bw_put(&w, 1, 1);
      bw_put(&w, (uint32_t) o.intel_filesize >> 16, 16);
      bw_put(&w, (uint32_t) o.intel_filesize & 0xFFFF, 16);
    }
    else bw_put(&w, 0, 1);
    if (n == 0) break;
    while (p < iend) {
      size_t bend = p + (size_t) o.block_size < iend ? p + (size_t) o.block_size : iend;
      uint32_t bbytes = (uint32_t)(bend - p);
      int mode = o.block_mode == 4 ? (int)(blk_no % 3) + 1 : o.block_mode;
      blk_no++;
      if (pending_pad) { bw_byte(&w, 0); pending_pad = 0; }          /* lzxd.c:469-474 */
      if (mode == 3) {
        size_t q;
        bw_put(&w, 3, 3); bw_put(&w, bbytes >> 8, 16); bw_put(&w, bbytes & 0xFF, 8);
        if (w.nbits == 0) bw_put(&w, 0, 16); else bw_align(&w);      /* lzxd.c:506-507 */
        for (q = 0; q < 3; q++) {
          uint32_t r = m.R[q];
          bw_byte(&w, r & 0xFF); bw_byte(&w, (r >> 8) & 0xFF); bw_byte(&w, (r >> 16) & 0xFF); bw_byte(&w, r >> 24);
        }
        for (q = p; q < bend; q++) {
          bw_byte(&w, src[q]); mf_insert(&m, q + rl);
          /* a frame end inside a stored block needs no padding: the bit buffer is empty */
          if (((q + 1) % FRAME) == 0 && frame_off) frame_off[(q + 1) / FRAME] = w.n;
          if (((q + 1) % FRAME) == 0 && o.delta && q + 1 < n) put_chunk_size(&w, q + 1, n);
        }
        pending_pad = (int)(bbytes & 1);
      }
      else {
        size_t nt = parse_block(&m, p + rl, bend + rl, toks), ti;
        for (ti = 0; ti < nt; ti++) {
          if (toks[ti].len) {
            unsigned o_ = toks[ti].off, lg_ = 0, l_ = toks[ti].len, lb_;
            while (o_ > 1) { o_ >>= 1; lg_++; }
            lb_ = l_ <= 4 ? l_ - 2 : l_ <= 8 ? 3 : l_ <= 16 ? 4 : l_ <= 32 ? 5 : l_ <= 64 ? 6 : l_ <= 128 ? 7 : 8;
            if (toks[ti].rslot == 3) __sync_fetch_and_add(&mspk_lzx_stat_offhist[lg_ < 23 ? lg_ : 23], 1);
            else __sync_fetch_and_add(&mspk_lzx_stat_offhist[23], 1);
            __sync_fetch_and_add(&mspk_lzx_stat_lenhist[lb_], 1);
            __sync_fetch_and_add(&mspk_lzx_stat_match_bytes, toks[ti].len);
          }
          else __sync_fetch_and_add(&mspk_lzx_stat_literals, 1);
        }
        __sync_fetch_and_add(&mspk_lzx_stat_tokens, nt);
        emit_compressed_block(&w, ls, toks, nt, bbytes, num_main, mode, p, frame_off, o.delta, n);
      }
      p = bend;
    }
    bw_align(&w);   /* short final frame: the decoder re-aligns after it too */
  }
  if (frame_off) frame_off[nframes] = w.n;
  free(srcbuf); free(m.head); free(m.prev); free(ls); free(toks);
  return w.overflow ? 0 : w.n;
}
