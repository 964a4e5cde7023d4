// qtm_kernel.hpp -- Quantum unit decoder: one wavefront per CAB folder stream.
//
// Replaces, for one unit, qtmd_init + qtmd_decompress(out_len) of the reference
// (libmspack/mspack/qtmd.c:187-479).
//   models ........... the nine adaptive models (qtm.h:49-77, qtmd.c:169-182,242-251) live in NINE
//                      VGPRs: lane i of a model register holds entry i as sym<<16 | cumfreq.
//   GET_SYMBOL ....... qtmd.c:92-123.  The linear scan "first i with cumfreq[i] <= symf" becomes one
//                      per-lane multiply/compare + ballot (cumfreq[i]*range <= X  <=>  cumfreq[i] <=
//                      X/range, no division); the "+8 to every entry before i" is one masked VALU add;
//                      the two divisions by the model total share one per-lane division (numerators
//                      in lanes 0 and 1).  Integer widths follow the C promotions of the reference.
//   qtmd_update_model  qtmd.c:125-166.  Halving keeps strict monotonicity with a suffix-max scan;
//                      the every-50th re-sort reproduces the reference's exchange pattern exactly:
//                      per outer step the strict prefix maxima ("records") rotate one place
//                      (prefix-max scan + ballot + bpermute) -- no serial O(n^2) loop.
//   window ........... output buffer is the window (offsets never exceed window_size); bytes the
//                      reference would read from never-written window memory read as zero.
//   bit reader ....... readbits.h:133-166,143-153 + qtmd.c:27-36 (BE16 words, MSB first); the
//                      reference's bits_left is tracked exactly at all times (it is cheap next to the
//                      arithmetic decoder), so ERR_READ fires exactly where the reference's does.
#pragma once
#include "wave_common.hpp"
#include "spec_queue.hpp"

#define QTM_FRAME 32768u
#ifdef LZX_MARKS        /* analysis builds: static instruction counts between marks (tools/count_isa.py) */
#define QTM_MARK(name) asm volatile("; MARK " name)
#else
#define QTM_MARK(name) do { } while (0)
#endif

struct QtmShared { SpecQueueLds spq; };

struct QtmDec {
  InWindow w;
  u64 bb; int bl;     // my bit buffer (MSB aligned)
  int rbl;            // the reference's bits_left
  u32 lane;
  u32 H, L, C;        // 16-bit registers kept in 32-bit scalars

  __device__ __forceinline__ u32 cons_bits() const { return w.wi * 32u - (u32) bl; }
  __device__ __forceinline__ void refill() {
    u32 d = w.next_dword(lane);
    bb |= (u64) __builtin_bswap32(d) << (32 - bl);      // two BE16 words in stream order (qtmd.c:30-35)
    bl += 32;
  }
  __device__ __forceinline__ void need(int n) { if (bl < n) refill(); }
  // one READ_BYTES of the reference (16 bits); false = ERR_READ
  __device__ __forceinline__ bool ref_fill() {
    u32 i = w.origin + ((cons_bits() + (u32) rbl) >> 3);
    if (i + 2u > w.in_len + w.eofs) return false;
    rbl += 16;
    return true;
  }
  __device__ __forceinline__ bool read_bits(int n, u32 &v) {      // READ_BITS, n <= 16
    need(n);
    while (rbl < n) { if (!ref_fill()) return false; }
    v = (u32)(bb >> (64 - n));
    bb <<= n; bl -= n; rbl -= n;
    return true;
  }
  __device__ __forceinline__ bool read_many(int n, u32 &v) {      // READ_MANY_BITS (readbits.h:143-153)
    u32 val = 0;
    while (n > 0) {
      if (rbl <= 16) { if (!ref_fill()) return false; }
      int run = rbl < n ? rbl : n;
      need(run);
      val = (val << run) | (u32)(bb >> (64 - run));
      bb <<= run; bl -= run; rbl -= run;
      n -= run;
    }
    v = val;
    return true;
  }
};

// ---- model maintenance (qtmd.c:125-166); `m` is the per-lane model register ---------------------
__device__ __forceinline__ u32 qtm_shfl_down(u32 v, u32 delta, u32 lane) {
  // value of lane+delta (0 beyond the wave)
  u32 r = (u32) __builtin_amdgcn_ds_bpermute((int)((lane + delta) << 2), (int) v);
  return (lane + delta < WAVE) ? r : 0u;
}
__device__ __forceinline__ u32 qtm_shfl_up(u32 v, u32 delta, u32 lane, u32 fill) {
  u32 r = (u32) __builtin_amdgcn_ds_bpermute((int)((lane - delta) << 2), (int) v);
  return (lane >= delta) ? r : fill;
}

__device__ void qtm_update_model(u32 &m, u32 entries, int &shiftsleft, u32 lane)
{
  u32 sym = m >> 16, cf = m & 0xFFFFu;
  const bool act = lane < entries;
  if (--shiftsleft) {
    // c'[i] = max(h[i], c'[i+1] + 1), c'[entries] = 0  ==>  c'[i] = max_{j>=i}(h[j] + j) - i
    u32 g = act ? ((cf >> 1) + lane) : 0u;
#pragma unroll
    for (u32 dlt = 1; dlt < WAVE; dlt <<= 1) { u32 o = qtm_shfl_down(g, dlt, lane); g = o > g ? o : g; }
    if (g < entries) g = entries;                       // the sentinel term (0 + entries)
    cf = (g - lane) & 0xFFFFu;
  }
  else {
    shiftsleft = 50;
    // cumulative -> frequency, +1, halve (unsigned short arithmetic)
    u32 nextcf = qtm_shfl_down(cf, 1, lane);
    if (lane + 1u >= entries) nextcf = 0;
    u32 f = act ? ((((cf - nextcf) & 0xFFFFu) + 1u) & 0xFFFFu) >> 1 : 0u;
    u32 pair = act ? ((sym << 16) | f) : 0u;            // the element that moves: (sym, freq)
    // for i in 0..n-2: for j in i+1..n-1: if (a[i] < a[j]) swap.  Within one i the strict prefix
    // maxima over positions i..n-1 ("records") each take over the previous record's element and
    // position i receives the last record's.
    for (u32 i = 0; i + 1u < entries; i++) {
      u32 key = (lane >= i && lane < entries) ? (pair & 0xFFFFu) : 0u;
      // exclusive prefix max over lanes >= i  (lanes < i contribute nothing)
      u32 pm = key;
#pragma unroll
      for (u32 dlt = 1; dlt < WAVE; dlt <<= 1) { u32 o = qtm_shfl_up(pm, dlt, lane, 0u); pm = o > pm ? o : pm; }
      u32 excl = qtm_shfl_up(pm, 1, lane, 0u);          // max over lanes < this one
      bool rec = (lane == i) || (lane > i && lane < entries && key > excl);
      u64 rm = ballot(rec);
      // previous record below this lane / the last record of all
      u64 below = rm & ((1ull << lane) - 1ull);
      u32 prev = below ? (63u - (u32) __clzll((long long) below)) : 0u;
      u32 lastrec = 63u - (u32) __clzll((long long) rm);
      u32 src = (lane == i) ? lastrec : prev;
      u32 moved = (u32) __builtin_amdgcn_ds_bpermute((int)(src << 2), (int) pair);
      if (rec) pair = moved;
    }
    // frequency -> cumulative: suffix sum
    u32 c = act ? (pair & 0xFFFFu) : 0u;
#pragma unroll
    for (u32 dlt = 1; dlt < WAVE; dlt <<= 1) c += qtm_shfl_down(c, dlt, lane);
    sym = pair >> 16; cf = c & 0xFFFFu;
  }
  if (act) m = (sym << 16) | cf;
}

// GET_SYMBOL (qtmd.c:92-123).  Returns the symbol, or -1 on ERR_READ.
__device__ __forceinline__ int qtm_get_symbol(QtmDec &d, u32 &m, u32 entries, int &shiftsleft)
{
  const u32 lane = d.lane;
  QTM_MARK("qtm_sym_begin");
  u32 H = d.H, L = d.L, C = d.C;
  const u32 cf = m & 0xFFFFu;
  const u32 tot = rdl(cf, 0);
  u32 range = ((H - L) & 0xFFFFu) + 1u;
  u32 X = (u32)((int)(C - L + 1u) * (int) tot - 1);      // int arithmetic, then unsigned (qtmd.c:94)
  bool hit;
  if (range < 65536u && X >= (range << 16)) {            // quotient would not fit 16 bits: do as C does
    u32 symf = (X / range) & 0xFFFFu;
    hit = cf <= symf;
  }
  else hit = cf * range <= X;                            // cf <= floor(X / range), division-free
  u64 hm = ballot(hit && lane >= 1u && lane < entries);
  u32 i = hm ? ((u32) __ffsll((long long) hm) - 1u) : entries;
  u32 e_im1 = rdl(m, i - 1u);
  u32 cf_i = (i < entries) ? (rdl(m, i & 63u) & 0xFFFFu) : 0u;
  int sym = (int)(e_im1 >> 16);
  u32 range2 = (u32)((int) H - (int) L + 1);
  // H = L + (cf[i-1]*range)/tot - 1 ; L = L + (cf[i]*range)/tot : both quotients from ONE division
  u32 num = (lane == 0u) ? (e_im1 & 0xFFFFu) * range2 : cf_i * range2;
  // exact 32-bit / 16-bit division through one single-precision reciprocal (a generic integer division is ~25
  // dependent instructions; round 2 used a double-precision reciprocal + a Newton step: four slow instructions on the
  // chain).  num < 2^32 and the quotient is at most 65536: float(num) is off by 2^-24, v_rcp_f32 by one ulp, the product
  // by 2^-24 -- relative 2^-22, i.e. less than 0.02 on the quotient: the truncated estimate is off by at most one.
  u32 quo;
  {
    const float rc = __builtin_amdgcn_rcpf((float) tot);
    quo = (u32)((float) num * rc);
    u32 rem = num - quo * tot;
    if ((int) rem < 0) quo--;
    else if (rem >= tot) quo++;
  }
  H = (L + rdl(quo, 0) - 1u) & 0xFFFFu;
  L = (L + rdl(quo, 1)) & 0xFFFFu;
  // cumfreq[0..i-1] += 8; rescale when the total passes 3800
  if (lane < i) m += 8u;
  QTM_MARK("qtm_sym_interval_done");
  if (tot + 8u > 3800u) qtm_update_model(m, entries, shiftsleft, lane);
  QTM_MARK("qtm_sym_renorm_begin");
  // Renormalisation (qtmd.c:107-122) in closed form.  The reference's bit-at-a-time loop is always
  // n shifts while the top bits of L and H agree, then m "underflow" steps while L = 01.., H = 10..
  // (each drops bit 14 and keeps the top bits 0 / 1), then it stops: n = leading equal bits,
  // m = leading positions below the top where L has 1 and H has 0.
  {
    u32 n = (u32) __builtin_clz(((L ^ H) << 16) | 0x8000u);            // 0..16
    u32 L1 = (L << n) & 0xFFFFu, H1 = ((H << n) | ((1u << n) - 1u)) & 0xFFFFu;
    u32 t = ((L1 & ~H1) & 0x7FFFu) << 17;
    u32 mu = (u32) __builtin_clz(~t);                                   // 0..15
    if (n == 16u) mu = 0;                                               // L == H: 16 shifts, then 0000 / FFFF
    const u32 k = n + mu;
    if (k) {
      // k one-bit reads: the reference refills 16 bits whenever bits_left is 0 at a read
      u32 have = (u32) d.rbl;
      while (have < k) {
        if (!d.ref_fill()) {                                            // ERR_READ at the bit that needed it
          u32 used = (u32) d.rbl;                                       // bits consumed before the failing read
          d.need((int) used); if (used) { d.bb <<= used; d.bl -= (int) used; }
          d.rbl = 0; d.H = H; d.L = L; d.C = C;
          return -1;
        }
        have = (u32) d.rbl;
      }
      d.need((int) k);
      const u32 nb = (u32)(d.bb >> (64u - k));
      d.bb <<= k; d.bl -= (int) k; d.rbl -= (int) k;
      u32 C1 = ((C << n) | (nb >> mu)) & 0xFFFFu;
      if (mu) {
        u32 top = (~(C1 >> (15u - mu)) & 1u) << 15;
        C1 = (((C1 << mu) | (nb & ((1u << mu) - 1u))) & 0x7FFFu) | top;
        L1 = (L1 << mu) & 0x7FFFu;
        H1 = 0x8000u | (((H1 << mu) | ((1u << mu) - 1u)) & 0x7FFFu);
      }
      L = L1; H = H1; C = C1;
    }
  }
  d.H = H; d.L = L; d.C = C;
  QTM_MARK("qtm_sym_end");
  return sym;
}

__device__ __forceinline__ u32 qtm_model_init(u32 lane, u32 start, u32 len) {   // qtmd.c:169-182
  return (lane <= len) ? (((start + lane) << 16) | (len - lane)) : 0u;
}

// static tables in closed form (qtmd.c:52-64)
__device__ __forceinline__ void qtm_pos_slot(u32 s, u32 &base, u32 &extra) {
  extra = (s < 2u) ? 0u : ((s - 2u) >> 1);
  base = (s < 2u) ? s : ((2u + (s & 1u)) << extra);
}
__device__ __forceinline__ void qtm_len_slot(u32 s, u32 &base, u32 &extra) {
  if (s < 6u) { base = s; extra = 0; }
  else if (s == 26u) { base = 254u; extra = 0; }
  else { extra = (s - 2u) >> 2; base = ((4u + ((s - 2u) & 3u)) << extra) - 2u; }
}

// byte-serial window copy semantics (qtmd.c:393-414) on the linear buffer; bytes before the start
// of the stream read as zero; stores are clipped to the unit's output
__device__ __forceinline__ void qtm_copy(u8 *out, u32 P, u32 off, u32 len, u32 out_len, u32 lane)
{
  if (off > P) {
    // part of the source lies before the first decoded byte: zeros, then (byte-serial) the rest
    for (u32 k = lane; k < len; k += WAVE) {
      // source byte index relative to stream start: P - off + k (negative => 0, or a byte this very
      // match produced earlier, which then is itself defined by the same rule)
      u32 kk = k;
      u32 b = 0;
      // walk back through the period until the source is outside this match
      while (kk >= off) kk -= off;                    // now source = P - off + kk, kk < off
      if (P + kk >= off) b = out[P + kk - off];
      if (P + k < out_len) out[P + k] = (u8) b;
    }
    return;
  }
  const u8 *src = out + P - off;
  if (off >= len || off >= WAVE) {
    for (u32 k = lane; k < len; k += WAVE) { u32 b = src[k]; if (P + k < out_len) out[P + k] = (u8) b; }
  }
  else {
    u32 r = lane, s = off << 5, step = 64u, ss = off << 5;
#pragma unroll
    for (int q = 0; q < 6; q++) { u32 t = r - s; r = t < r ? t : r; s >>= 1; }
#pragma unroll
    for (int q = 0; q < 6; q++) { u32 t = step - ss; step = t < step ? t : step; ss >>= 1; }
    for (u32 k = lane; k < len; k += WAVE) {
      u32 b = src[r];
      if (P + k < out_len) out[P + k] = (u8) b;
      r += step; if (r >= off) r -= off;
    }
  }
}

__device__ void qtm_decode_unit(const mspack_hip_unit &u, const u8 *in_arena, u8 *out_arena,
                                mspack_hip_result *res, QtmShared *sh)
{
  const u32 lane = threadIdx.x;
  const u32 wb = u.window_bits;
  if (wb < 10u || wb > 21u) {
    if (lane == 0) { res->err = ERR_ARGS; res->flags = 0; res->out_len = 0; res->in_used = 0; res->good_len = 0; res->in_next = 0; }
    return;
  }
  QtmDec d;
  d.lane = lane;
  d.w.unit = in_arena + u.in_off; d.w.in_len = u.in_len;
  d.w.eofs = (u.flags & MSPACK_HIP_UF_HARD_EOF) ? 0u : 2u;
  d.w.seek(0, lane);
  d.bb = 0; d.bl = 0; d.rbl = 0; d.H = 0; d.L = 0; d.C = 0;
  u8 *out = out_arena + u.out_off;
  const u32 wsize = 1u << wb, out_len = u.out_len;

  const u32 n4 = (2u * wb > 24u) ? 24u : 2u * wb, n5 = (2u * wb > 36u) ? 36u : 2u * wb, n6 = 2u * wb;
  u32 m0 = qtm_model_init(lane, 0, 64), m1 = qtm_model_init(lane, 64, 64), m2 = qtm_model_init(lane, 128, 64),
      m3 = qtm_model_init(lane, 192, 64), m4 = qtm_model_init(lane, 0, n4), m5 = qtm_model_init(lane, 0, n5),
      m6 = qtm_model_init(lane, 0, n6), m6l = qtm_model_init(lane, 0, 27), m7 = qtm_model_init(lane, 0, 7);
  int s0 = 4, s1 = 4, s2 = 4, s3 = 4, s4 = 4, s5 = 4, s6 = 4, s6l = 4, s7 = 4;

  // the reference's loop variables (qtmd.c:257-479): window_posn, frame_todo, o_ptr/o_end as window
  // indices, out_bytes still wanted.  P is the linear position of window_posn.
  u32 P = 0, wpos = 0, frame_todo = QTM_FRAME, o_ptr = 0, o_end = 0, written = 0;
  long long need = (long long) out_len;
  bool header_read = false;
  int err = ERR_OK;
  u32 good = 0;                 // position up to which a request would have succeeded
  // matches are queued and resolved 64 output bytes at a time (spec_queue.hpp) instead of paying a
  // dependent load -> store round trip per match; literals are stored directly
  SpecQueue Q;
  spq_init(sh->spq, Q, 0u, lane);
  // literals wait in two registers, lane i holding the i-th pending one and its position, and leave 64 at a time in one
  // store instruction (round 2 stored every literal on its own from lane 0: 12 M store instructions per launch of
  // config 4); they go out before anything reads the output (a resolve, a direct copy)
  u32 lit_buf = 0, lit_pos = 0, lit_n = 0;
#define QTM_FLUSH()                                                                           \
  do {                                                                                        \
    if (lit_n) { if (lane < lit_n && lit_pos < out_len) out[lit_pos] = (u8) lit_buf; lit_n = 0; } \
  } while (0)
#define QTM_COPY(P_, off_, len_)                                                              \
  do {                                                                                        \
    if ((off_) <= (P_) && (P_) + (len_) - (Q.Pf & ~63u) <= SPQ_RING && Q.mcount < SPQ_CAP) {   \
      spq_push(sh->spq, Q, lane == 0u, 0u, 1u, (P_), (off_), (len_));                         \
    }                                                                                         \
    else {                                                                                    \
      QTM_FLUSH();                                                                            \
      spq_resolve(sh->spq, Q, out, (P_), true, lane, out_len);                                \
      qtm_copy(out, (P_), (off_), (len_), out_len, lane);                                     \
      Q.Pf = (P_) + (len_);                                                                   \
    }                                                                                         \
  } while (0)

  while ((long long)(o_end - o_ptr) < need) {
    u32 v;
    if (!header_read) {
      d.H = 0xFFFFu; d.L = 0;
      if (!d.read_bits(16, v)) { err = ERR_READ; break; }
      d.C = v; header_read = true;
    }
    u32 frame_end = (u32)((long long) wpos + (need - (long long)(o_end - o_ptr)));
    if (wpos + frame_todo < frame_end) frame_end = wpos + frame_todo;
    if (frame_end > wsize) frame_end = wsize;
    bool stop = false;

    while (wpos < frame_end) {
      good = P;
      int sel = qtm_get_symbol(d, m7, 7, s7);
      if (sel < 0) { err = ERR_READ; stop = true; break; }
      if (sel < 4) {
        // ONE copy of the symbol decoder for the four literal models: the model moves through a
        // temporary (a few selects) instead of four inlined copies of GET_SYMBOL
        u32 mm = sel == 0 ? m0 : (sel == 1 ? m1 : (sel == 2 ? m2 : m3));
        int ss = sel == 0 ? s0 : (sel == 1 ? s1 : (sel == 2 ? s2 : s3));
        int sym = qtm_get_symbol(d, mm, 64, ss);
        if (sel == 0) { m0 = mm; s0 = ss; } else if (sel == 1) { m1 = mm; s1 = ss; }
        else if (sel == 2) { m2 = mm; s2 = ss; } else { m3 = mm; s3 = ss; }
        if (sym < 0) { err = ERR_READ; stop = true; break; }
        lit_buf = wrl(lit_buf, (u32) sym, lit_n); lit_pos = wrl(lit_pos, P, lit_n);
        if (++lit_n == WAVE) QTM_FLUSH();
        P++; wpos++; frame_todo--;
        continue;
      }
      u32 moff, mlen, base, extra;
      int sym;
      if (sel == 4 || sel == 5) {
        u32 mm = sel == 4 ? m4 : m5;
        int ss = sel == 4 ? s4 : s5;
        sym = qtm_get_symbol(d, mm, sel == 4 ? n4 : n5, ss);
        if (sel == 4) { m4 = mm; s4 = ss; } else { m5 = mm; s5 = ss; }
        if (sym < 0) { err = ERR_READ; stop = true; break; }
        qtm_pos_slot((u32) sym, base, extra);
        if (!d.read_many((int) extra, v)) { err = ERR_READ; stop = true; break; }
        moff = base + v + 1u; mlen = sel == 4 ? 3u : 4u;
      }
      else if (sel == 6) {
        sym = qtm_get_symbol(d, m6l, 27, s6l);
        if (sym < 0) { err = ERR_READ; stop = true; break; }
        qtm_len_slot((u32) sym, base, extra);
        if (!d.read_many((int) extra, v)) { err = ERR_READ; stop = true; break; }
        mlen = base + v + 5u;
        sym = qtm_get_symbol(d, m6, n6, s6);
        if (sym < 0) { err = ERR_READ; stop = true; break; }
        qtm_pos_slot((u32) sym, base, extra);
        if (!d.read_many((int) extra, v)) { err = ERR_READ; stop = true; break; }
        moff = base + v + 1u;
      }
      else { err = ERR_DECRUNCH; stop = true; break; }

      frame_todo -= mlen;
      if (wpos + mlen > wsize) {                                      // qtmd.c:358-390
        u32 i = wsize - o_ptr;
        // (the copy itself is the same on the linear buffer; only the flush bookkeeping differs)
        if ((long long) i > need) {
          // first part was already copied by the reference before it bails out
          QTM_COPY(P, moff, wsize - wpos);
          err = ERR_DECRUNCH; stop = true; break;
        }
        QTM_COPY(P, moff, mlen);
        written += i; need -= i; o_ptr = 0; o_end = 0;
        P += mlen; wpos = wpos + mlen - wsize;
        break;
      }
      if (moff > wpos && (moff - wpos) > wsize) { err = ERR_DECRUNCH; stop = true; break; }   // qtmd.c:399
      QTM_COPY(P, moff, mlen);
      P += mlen; wpos += mlen;
      if (spq_due(Q, P)) { QTM_FLUSH(); spq_resolve(sh->spq, Q, out, P, false, lane, out_len); }
    }
    if (stop) break;
    o_end = wpos;
    // a match that overshot the frame fails every request that needed that match: `good` still
    // holds the position where it started
    if (frame_todo > QTM_FRAME) { err = ERR_DECRUNCH; break; }         // qtmd.c:424
    good = P ? P - 1u : 0u;       // errors in the frame-end handling hit the request ending here
    if (frame_todo == 0u) {
      int n = d.rbl & 7;                                               // qtmd.c:432
      if (n) { d.need(n); d.bb <<= n; d.bl -= n; d.rbl -= n; }
      bool ok = true;
      do { if (!d.read_bits(8, v)) { ok = false; break; } } while (v != 0xFFu);
      if (!ok) { err = ERR_READ; break; }
      header_read = false; frame_todo = QTM_FRAME;
    }
    good = P;
    if (wpos == wsize) {
      u32 i = o_end - o_ptr;
      if ((long long) i >= need) break;
      written += i; need -= i; o_ptr = 0; o_end = 0; wpos = 0;
    }
  }
  QTM_FLUSH();
  // whatever is still queued lies below the highest position any token reached
  {
    u32 top = Q.Pf;
    if (Q.mcount) { uint2 lr = sh->spq.mlist[Q.mcount - 1u]; top = rfl(lr.x) + (rfl(lr.y) & 511u); }
    if (P > top) top = P;
    spq_resolve(sh->spq, Q, out, top, true, lane, out_len);
  }
#undef QTM_COPY
#undef QTM_FLUSH
  if (err == ERR_OK && need) { written += (u32) need; }
  if (err == ERR_OK) good = written;
  if (lane == 0) {
    res->err = err; res->flags = 0; res->out_len = written; res->good_len = good; res->in_next = 0;
    res->in_used = d.w.origin + ((d.cons_bits() + (u32) d.rbl) >> 3);
  }
}
