/* oracle/qtm_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * CPU restatement of the reference Quantum folder decoder (adaptive arithmetic-coded LZ77).
 * Follows (by behaviour):
 *   bit reader ............ libmspack/mspack/readbits.h:133-214 with qtmd.c:27-36 (BE16 words, MSB first)
 *   static tables ......... qtmd.c:66-82 (generated here from the recipe in qtmd.c:52-64)
 *   arithmetic decoder .... qtmd.c:92-123 (integer widths reproduced exactly)
 *   model maintenance ..... qtmd.c:125-182
 *   frame / window loop ... qtmd.c:257-479
 * The window starts zero-filled where the reference's is uninitialised malloc memory.
 */
#include <stdlib.h>
#include <string.h>
#include "oracle.h"
extern __thread int oracle_hard_eof_;   /* lzx_oracle.c: oracle_set_hard_eof() */

#define FRAME 32768u

typedef struct { uint16_t sym, cumfreq; } msym_t;
typedef struct { int shiftsleft, entries; msym_t *syms; } model_t;

typedef struct {
  const uint8_t *in; size_t in_len, pos;
  uint32_t bb; int bl; int err;
} qbits_t;

static int q_byte(qbits_t *b, unsigned *v) {
  if (b->pos < b->in_len) { *v = b->in[b->pos++]; return 0; }
  if (!oracle_hard_eof_ && b->pos < b->in_len + 2) { b->pos++; *v = 0; return 0; }   /* (a FAILED read -- sys->read < 0 -- fabricates nothing: readbits.h:196-198) */
  b->err = ORC_READ; return 1;
}
static int q_fill(qbits_t *b) {          /* one READ_BYTES: 16 bits, big-endian byte pair */
  unsigned b0, b1;
  if (q_byte(b, &b0) || q_byte(b, &b1)) return 1;
  b->bb |= ((b0 << 8) | b1) << (32 - 16 - b->bl);
  b->bl += 16;
  return 0;
}
static int q_ensure(qbits_t *b, int n) { while (b->bl < n) if (q_fill(b)) return 1; return 0; }
#define QPEEK(b, n) ((b)->bb >> (32 - (n)))
#define QDROP(b, n) do { (b)->bb <<= (n); (b)->bl -= (n); } while (0)
static int q_bits(qbits_t *b, int n, unsigned *v) {
  if (q_ensure(b, n)) return 1;
  *v = QPEEK(b, n); QDROP(b, n); return 0;
}
static int q_many(qbits_t *b, int n, unsigned *v) {     /* readbits.h:143-153 */
  unsigned val = 0;
  while (n > 0) {
    int run;
    if (b->bl <= 16) { if (q_fill(b)) return 1; }
    run = (b->bl < n) ? b->bl : n;
    val = (val << run) | QPEEK(b, run);
    QDROP(b, run);
    n -= run;
  }
  *v = val; return 0;
}

typedef struct {
  qbits_t b;
  uint8_t *win; uint32_t wsize, wpos, frame_todo;
  uint16_t H, L, C; int header_read;
  model_t m0, m1, m2, m3, m4, m5, m6, m6len, m7;
  msym_t s0[65], s1[65], s2[65], s3[65], s4[25], s5[37], s6[43], s6l[28], s7[8];
} qtm_t;

static uint32_t pos_base[42]; static uint8_t pos_extra[42], len_base[27], len_extra[27];
static void init_tables(void) {
  unsigned i, off;
  if (pos_base[1]) return;
  for (i = 0, off = 0; i < 42; i++) {
    pos_base[i] = off; pos_extra[i] = (uint8_t)(((i < 2) ? 0 : (i - 2)) >> 1); off += 1u << pos_extra[i];
  }
  for (i = 0, off = 0; i < 26; i++) {
    len_base[i] = (uint8_t) off; len_extra[i] = (uint8_t)(((i < 2) ? 0 : (i - 2)) >> 2); off += 1u << len_extra[i];
  }
  len_base[26] = 254; len_extra[26] = 0;
}

static void model_init(model_t *m, msym_t *syms, int start, int len) {
  int i;
  m->shiftsleft = 4; m->entries = len; m->syms = syms;
  for (i = 0; i <= len; i++) { syms[i].sym = (uint16_t)(start + i); syms[i].cumfreq = (uint16_t)(len - i); }
}

static void model_update(model_t *m) {                  /* qtmd.c:125-166 */
  int i, j;
  if (--m->shiftsleft) {
    for (i = m->entries - 1; i >= 0; i--) {
      m->syms[i].cumfreq >>= 1;
      if (m->syms[i].cumfreq <= m->syms[i + 1].cumfreq)
        m->syms[i].cumfreq = (uint16_t)(m->syms[i + 1].cumfreq + 1);
    }
    return;
  }
  m->shiftsleft = 50;
  for (i = 0; i < m->entries; i++) {
    m->syms[i].cumfreq = (uint16_t)(m->syms[i].cumfreq - m->syms[i + 1].cumfreq);
    m->syms[i].cumfreq++;
    m->syms[i].cumfreq >>= 1;
  }
  /* the exact (unstable) exchange pattern is part of the format */
  for (i = 0; i < m->entries - 1; i++)
    for (j = i + 1; j < m->entries; j++)
      if (m->syms[i].cumfreq < m->syms[j].cumfreq) { msym_t t = m->syms[i]; m->syms[i] = m->syms[j]; m->syms[j] = t; }
  for (i = m->entries - 1; i >= 0; i--)
    m->syms[i].cumfreq = (uint16_t)(m->syms[i].cumfreq + m->syms[i + 1].cumfreq);
}

/* returns 0 ok / 1 input exhausted; *out = decoded symbol */
static int get_symbol(qtm_t *q, model_t *m, int *out) {
  uint16_t H = q->H, L = q->L, C = q->C, symf;
  unsigned int range;
  int i;
  range = (unsigned int)((((int) H - (int) L) & 0xFFFF) + 1);
  symf = (uint16_t)((((unsigned int)((((int) C - (int) L + 1) * (int) m->syms[0].cumfreq) - 1)) / range) & 0xFFFF);
  for (i = 1; i < m->entries; i++) if (m->syms[i].cumfreq <= symf) break;
  *out = m->syms[i - 1].sym;
  range = (unsigned int)(((int) H - (int) L) + 1);
  symf = m->syms[0].cumfreq;
  H = (uint16_t)((unsigned int) L + (((unsigned int) m->syms[i - 1].cumfreq * range) / (unsigned int) symf) - 1u);
  L = (uint16_t)((unsigned int) L + (((unsigned int) m->syms[i].cumfreq * range) / (unsigned int) symf));
  do { m->syms[--i].cumfreq += 8; } while (i > 0);
  if (m->syms[0].cumfreq > 3800) model_update(m);
  for (;;) {
    if ((L & 0x8000) != (H & 0x8000)) {
      if ((L & 0x4000) && !(H & 0x4000)) { C ^= 0x4000; L &= 0x3FFF; H |= 0x4000; }
      else break;
    }
    L = (uint16_t)(L << 1); H = (uint16_t)((H << 1) | 1);
    if (q_ensure(&q->b, 1)) return 1;
    C = (uint16_t)((C << 1) | QPEEK(&q->b, 1));
    QDROP(&q->b, 1);
  }
  q->H = H; q->L = L; q->C = C;
  return 0;
}

/* oracle_qtm_set_marks(): for the NEXT oracle_qtm_decode() of this thread -- positions (ascending) at which later requests of the
 * same stream may end; log[i] = how far the token that covers the byte in front of marks[i] runs past it, i.e. what a request
 * ending there leaves in the window for the next call (qtmd.c:268-276; the token sequence does not depend on where requests end).
 * 0 where a token ends exactly there, and for marks the decoding never reached; 0xFFFFFFFF where a request ending there FAILS in
 * the reference: the mark lies inside a match that crosses the window's end, in front of that end (qtmd.c:358-374: "can't flush
 * up to the end of the window, but can't break out either" -> MSPACK_ERR_DECRUNCH).  The ground truth of MSPACK_HIP_UF_QTM_MARKS. */
static __thread const uint32_t *qtm_marks_;
static __thread uint32_t qtm_n_marks_;
static __thread uint32_t *qtm_mark_log_;
void oracle_qtm_set_marks(const uint32_t *marks, uint32_t n, uint32_t *log) { qtm_marks_ = marks; qtm_n_marks_ = n; qtm_mark_log_ = log; }

int oracle_qtm_decode(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap,
                      uint64_t out_bytes, int window_bits, oracle_result *res)
{
  const uint32_t *marks = qtm_marks_; uint32_t n_marks = qtm_n_marks_, *mark_log = qtm_mark_log_, mk = 0;
  uint64_t lin = 0;                                /* bytes decoded so far (the linear position of wpos) */
  qtm_t *q;
  uint64_t written = 0;
  int64_t need = (int64_t) out_bytes;
  uint32_t o_ptr = 0, o_end = 0;
  int err = ORC_OK, wb2;

  qtm_marks_ = NULL; qtm_n_marks_ = 0; qtm_mark_log_ = NULL;
  { uint32_t i; for (i = 0; i < n_marks; i++) mark_log[i] = 0; }
#define MARKS() do { while (mk < n_marks && lin >= marks[mk]) { mark_log[mk] = (uint32_t)(lin - marks[mk]); mk++; } } while (0)
  memset(res, 0, sizeof(*res));
  if (window_bits < 10 || window_bits > 21) { res->err = ORC_ARGS; return ORC_ARGS; }
  init_tables();
  q = (qtm_t *) calloc(1, sizeof(*q));
  q->wsize = 1u << window_bits;
  q->win = (uint8_t *) calloc(1, q->wsize);
  q->b.in = in; q->b.in_len = in_len;
  q->frame_todo = FRAME;
  wb2 = window_bits * 2;
  model_init(&q->m0, q->s0, 0, 64);   model_init(&q->m1, q->s1, 64, 64);
  model_init(&q->m2, q->s2, 128, 64); model_init(&q->m3, q->s3, 192, 64);
  model_init(&q->m4, q->s4, 0, wb2 > 24 ? 24 : wb2);
  model_init(&q->m5, q->s5, 0, wb2 > 36 ? 36 : wb2);
  model_init(&q->m6, q->s6, 0, wb2);
  model_init(&q->m6len, q->s6l, 0, 27);
  model_init(&q->m7, q->s7, 0, 7);

#define EMIT(from, n) do { uint32_t n_ = (n); \
    if (out && written < out_cap) { size_t room = out_cap - (size_t) written; \
      memcpy(out + written, q->win + (from), n_ < room ? n_ : room); } \
    written += n_; } while (0)

  while ((int64_t)(o_end - o_ptr) < need) {
    uint32_t frame_end;
    unsigned v;
    if (!q->header_read) {
      q->H = 0xFFFF; q->L = 0;
      if (q_bits(&q->b, 16, &v)) goto rderr;
      q->C = (uint16_t) v; q->header_read = 1;
    }
    frame_end = (uint32_t)((int64_t) q->wpos + (need - (int64_t)(o_end - o_ptr)));
    if (q->wpos + q->frame_todo < frame_end) frame_end = q->wpos + q->frame_todo;
    if (frame_end > q->wsize) frame_end = q->wsize;

    while (q->wpos < frame_end) {
      int sel, sym;
      uint32_t moff; int mlen;
      /* (the marks the tokens so far have passed: said once the checks behind a token -- the frame's end, its trailer -- are
       * through too; a request that ends inside a token which fails those fails with it) */
      MARKS();
      if (get_symbol(q, &q->m7, &sel)) goto rderr;
      if (sel < 4) {
        model_t *m = sel == 0 ? &q->m0 : sel == 1 ? &q->m1 : sel == 2 ? &q->m2 : &q->m3;
        if (get_symbol(q, m, &sym)) goto rderr;
        q->win[q->wpos++] = (uint8_t) sym; q->frame_todo--;
        lin++;
        continue;
      }
      if (sel == 4) {
        if (get_symbol(q, &q->m4, &sym) || q_many(&q->b, pos_extra[sym], &v)) goto rderr;
        moff = pos_base[sym] + v + 1; mlen = 3;
      }
      else if (sel == 5) {
        if (get_symbol(q, &q->m5, &sym) || q_many(&q->b, pos_extra[sym], &v)) goto rderr;
        moff = pos_base[sym] + v + 1; mlen = 4;
      }
      else if (sel == 6) {
        if (get_symbol(q, &q->m6len, &sym) || q_many(&q->b, len_extra[sym], &v)) goto rderr;
        mlen = (int) len_base[sym] + (int) v + 5;
        if (get_symbol(q, &q->m6, &sym) || q_many(&q->b, pos_extra[sym], &v)) goto rderr;
        moff = pos_base[sym] + v + 1;
      }
      else { err = ORC_DECRUNCH; goto done; }

      q->frame_todo -= (uint32_t) mlen;
      if (q->wpos + (uint32_t) mlen > q->wsize) {                   /* qtmd.c:358-390 */
        uint32_t i = q->wsize - q->wpos, d = q->wpos;
        int32_t j = (int32_t) q->wpos - (int32_t) moff;
        while (i--) q->win[d++] = q->win[(uint32_t)(j++) & (q->wsize - 1)];
        /* (marks inside the part of the match in front of the window's end: a request that ends there cannot be served -- the
         * reference gives up below whenever it is asked for less than the window still holds, qtmd.c:366-374) */
        while (mk < n_marks && marks[mk] < lin + (q->wsize - q->wpos)) mark_log[mk++] = 0xFFFFFFFFu;
        i = q->wsize - o_ptr;
        if ((int64_t) i > need) { err = ORC_DECRUNCH; goto done; }
        EMIT(o_ptr, i); need -= i; o_ptr = o_end = 0;
        d = 0; i = (uint32_t) mlen - (q->wsize - q->wpos);
        while (i--) q->win[d++] = q->win[(uint32_t)(j++) & (q->wsize - 1)];
        q->wpos = q->wpos + (uint32_t) mlen - q->wsize;
        lin += (uint32_t) mlen;
        break;
      }
      else {
        uint32_t i = (uint32_t) mlen, d = q->wpos, s;
        if (moff > q->wpos) {
          uint32_t j = moff - q->wpos;
          if ((int32_t) j > (int32_t) q->wsize) { err = ORC_DECRUNCH; goto done; }
          s = q->wsize - j;
          if (j < i) { i -= j; while (j-- > 0) q->win[d++] = q->win[s++]; s = 0; }
          while (i-- > 0) q->win[d++] = q->win[s++];
        }
        else { s = d - moff; while (i-- > 0) q->win[d++] = q->win[s++]; }
        q->wpos += (uint32_t) mlen;
        lin += (uint32_t) mlen;
      }
    }
    o_end = q->wpos;
    if (q->frame_todo > FRAME) { err = ORC_DECRUNCH; goto done; }
    if (q->frame_todo == 0) {
      if (q->b.bl & 7) QDROP(&q->b, q->b.bl & 7);
      do { if (q_bits(&q->b, 8, &v)) goto rderr; } while (v != 0xFF);
      q->header_read = 0; q->frame_todo = FRAME;
    }
    if (q->wpos == q->wsize) {
      uint32_t i = o_end - o_ptr;
      if ((int64_t) i >= need) break;
      EMIT(o_ptr, i); need -= i; o_ptr = o_end = 0; q->wpos = 0;
    }
  }
  if (need) { EMIT(o_ptr, (uint32_t) need); o_ptr += (uint32_t) need; }
  /* what the call decoded beyond the request -- the rest of the match that covers its last byte -- stays in the window and is
   * the first thing the NEXT call hands to its output (qtmd.c:268-276: o_end - o_ptr) */
  res->in_next = o_end - o_ptr;
  MARKS();
  goto done;
rderr:
  err = ORC_READ;
done:
  res->err = err; res->out_len = written; res->in_used = q->b.pos;
  free(q->win); free(q);
  return err;
}
