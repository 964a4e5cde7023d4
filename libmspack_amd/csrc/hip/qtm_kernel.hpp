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
#include <type_traits>

#define QTM_FRAME 32768u
#ifdef LZX_MARKS        /* analysis builds: static instruction counts between marks (tools/count_isa.py) */
#define QTM_MARK(name) asm volatile("; MARK " name)
#else
#define QTM_MARK(name) do { } while (0)
#endif

struct QtmShared {
  SpecQueueLds spq;
  /* MSPACK_HIP_UF_QTM_MARKS: where the unit is in its table of marks -- in LDS, not in registers: the decode loop carries ONE extra
   * scalar (the next mark) and the rest is looked at when a mark is passed (qtm_marks_passed) */
  u32 mark_i, n_marks;
  const u32 *marks; u32 *mark_log;
};
// the marks a token has passed (P: the position behind it): their entries of the log; -> the next mark.  fail_below: marks below
// this position get 0xFFFFFFFF instead (they lie inside the part of a window-crossing match in front of the window's end)
__device__ __attribute__((noinline)) u32 qtm_marks_passed(QtmShared *sh, const u32 P, const u32 fail_below, const u32 lane)
{
  u32 i = rfl(sh->mark_i);
  const u32 n = rfl(sh->n_marks);
  const u32 *const marks = sh->marks;
  u32 *const log = sh->mark_log;
  u32 next = 0xFFFFFFFFu;
  for (;;) {
    next = i < n ? rfl(gld(marks + i)) : 0xFFFFFFFFu;
    if (fail_below ? next >= fail_below : P < next) break;
    if (lane == 0) log[i] = fail_below ? 0xFFFFFFFFu : P - next;
    i++;
  }
  if (lane == 0) sh->mark_i = i;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  return next;
}
#ifdef QTM_TIMERS       /* analysis builds: cycles per part of the token loop, block 0's wave (mspack_hip_debug_qtm_timers) */
__device__ unsigned long long g_qtm_tm[8];
#define QT0() unsigned long long qt_ = __builtin_amdgcn_s_memtime()
#define QT(k) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); d.tm[k] += (u32)(n_ - qt_); qt_ = n_; } while (0)
#else
#define QT0() do { } while (0)
#define QT(k) do { } while (0)
#endif

struct QtmDec {
  InWindow w;
  u64 bb; int bl;     // my bit buffer (MSB aligned)
  int rbl;            // the reference's bits_left
  u32 lane;
  u32 H, L, C;        // 16-bit registers kept in 32-bit scalars
#ifdef QTM_TIMERS
  u32 tm[8];
#endif

  __device__ __forceinline__ u32 cons_bits() const { return w.wi * 32u - (u32) bl; }
  // The window's dwords are byte-swapped ONCE PER LANE when a chunk is loaded (two BE16 words in stream order,
  // qtmd.c:30-35), so that a refill is a v_readlane into an SGPR and nothing else: swapped behind the readlane (there is
  // no scalar byte swap) the dword came back in a VGPR, and the whole coder chain -- bit buffer, C, L, H, the
  // renormalisation -- ran as vector instructions on wave-uniform values at twice the scalar latency.
  __device__ __forceinline__ void seek_swapped(u32 byte_pos) {
    w.origin = byte_pos; w.wi = 0;
    w.cur = __builtin_bswap32(w.load_chunk(0, lane));
    w.nxt = __builtin_bswap32(w.load_chunk(1, lane));
  }
  __device__ __forceinline__ void refill() {
    const u32 d = rdl(w.cur, w.wi & 63u);
    w.wi++;
    if ((w.wi & 63u) == 0u) { w.cur = w.nxt; w.nxt = __builtin_bswap32(w.load_chunk((w.wi >> 6) + 1u, lane)); }
    bb |= (u64) d << (32 - bl);
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
  // the same far from the end of the input (`FAST`: the caller has checked that >= 96 bytes lie ahead, a READ_BYTES cannot
  // fail): the reference's bits_left is still tracked -- two adds -- so that the exact reader can take over at any token
  template <bool FAST> __device__ __forceinline__ bool fill_t() { if (FAST) { rbl += 16; return true; } return ref_fill(); }
  template <bool FAST> __device__ __forceinline__ bool read_many_t(int n, u32 &v) {
    if (!FAST) return read_many(n, v);
    u32 val = 0;
    while (n > 0) {
      if (rbl <= 16) rbl += 16;
      int run = rbl < n ? rbl : n;
      need(run);
      val = (val << run) | (u32)(bb >> (64 - run));
      bb <<= run; bl -= run; rbl -= run;
      n -= run;
    }
    v = val;
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

// (a real call -- it runs once per ~50..400 symbols of a model -- and BY VALUE: with reference parameters the models'
// registers had their address taken and lived in scratch memory, a load and a store per symbol)
struct QtmUpd { u32 m; int shiftsleft; };
__device__ __attribute__((noinline)) QtmUpd qtm_update_model(u32 m, const u32 entries, int shiftsleft, const u32 lane)
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
  QtmUpd r; r.m = m; r.shiftsleft = shiftsleft;
  return r;
}

// GET_SYMBOL (qtmd.c:92-123).  Returns the symbol, or -1 on ERR_READ.
// Round 4: (1) every lane divides ITS OWN entry's cumfreq * range by the total while the search for the symbol runs --
// H and L are then two readlanes of finished quotients (the division used to start only when the symbol was known: two
// readlanes, a multiply and the reciprocal chain behind the search); (2) FAST = far from the end of the input no read can
// fail: no bounds arithmetic, no failure paths (the reference's bits_left is still tracked, see QtmDec::fill_t).
// `shifts`: the nine models' shiftsleft counters (qtm.h:52), one per LANE of a vector register -- they are touched once
// per ~50..400 symbols, and nine more scalar registers were what made the coder's hot state spill.
// Round 5: `tots` = the model's TOTAL (cumfreq[0]) kept on the scalar side, 16 bits of an SGPR per model (TSH = this model's
// shift): it grows by 8 per symbol and is re-read from the register only after a rescale.  The total is the first thing a symbol
// needs (X = (C - L + 1) * total - 1, qtmd.c:94) and used to arrive through a v_readlane hop (~36 cycles, micro-benchmarks of
// round 2) behind the model-select branch, where nothing can be scheduled in front of it.
template <bool FAST, u32 TSH>
__device__ __forceinline__ int qtm_get_symbol(QtmDec &d, u32 &m, u32 entries, u32 &shifts, const u32 model_ix, u32 &tots)
{
  const u32 lane = d.lane;
  QTM_MARK("qtm_sym_begin");
  QT0();
  u32 H = d.H, L = d.L, C = d.C;
  const u32 cf = m & 0xFFFFu;
  const u32 tot = (tots >> TSH) & 0xFFFFu;
  const u32 range = ((H - L) & 0xFFFFu) + 1u;
  const u32 range2 = (u32)((int) H - (int) L + 1);
  // ---- per entry: floor(cumfreq * range / total), exact through one single-precision reciprocal (num < 2^32, quotient
  // <= 65536: float(num) is off by 2^-24, v_rcp_f32 by one ulp, the product by 2^-24 -- the truncated estimate is off by
  // at most one) ----
  u32 quo;
  {
    const u32 num = cf * range2;
    const float rc = __builtin_amdgcn_rcpf((float) tot);
    quo = (u32)((float) num * rc);
    const u32 rem = num - quo * tot;
    if ((int) rem < 0) quo--;
    else if (rem >= tot) quo++;
  }
  // ---- the symbol: first entry i >= 1 with cumfreq[i] <= floor(X / range) ----
  const u32 X = (u32)((int)(C - L + 1u) * (int) tot - 1);      // int arithmetic, then unsigned (qtmd.c:94)
  bool hit;
  if (range < 65536u && X >= (range << 16)) {                  // quotient would not fit 16 bits: do as C does
    const u32 symf = (X / range) & 0xFFFFu;
    hit = cf <= symf;
  }
  else hit = cf * range <= X;                                  // cf <= floor(X / range), division-free
  // (entries 1 .. entries - 1 as a mask of the ballot: a constant for most models -- as a lane predicate it was a
  // loop-invariant exec mask per model that spilled and came back through two v_readlane per symbol)
  const u64 hm = ballot(hit) & (((entries < 64u ? (1ull << entries) : 0ull) - 1ull) & ~1ull);
  const u32 i = hm ? ((u32) __ffsll((long long) hm) - 1u) : entries;
  const int sym = (int)(rdl(m, i - 1u) >> 16);
  // H = L + (cf[i-1]*range)/tot - 1 ; L = L + (cf[i]*range)/tot
  const u32 q_hi = rdl(quo, i - 1u), q_lo = (i < entries) ? rdl(quo, i & 63u) : 0u;
  H = (L + q_hi - 1u) & 0xFFFFu;
  L = (L + q_lo) & 0xFFFFu;
  // cumfreq[0..i-1] += 8; rescale when the total passes 3800
  if (lane < i) m += 8u;
  tots += 8u << TSH;                                             // (entry 0 is below every i: the total always grows)
  QTM_MARK("qtm_sym_interval_done");
  QT(0);
  if (tot + 8u > 3800u) {
    const QtmUpd up = qtm_update_model(m, entries, (int) rdl(shifts, model_ix), lane);
    m = up.m; shifts = wrl(shifts, (u32) up.shiftsleft, model_ix);
    tots = (tots & ~(0xFFFFu << TSH)) | (rdl(m & 0xFFFFu, 0u) << TSH);
  }
  QT(1);
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
      if (FAST) { if ((u32) d.rbl < k) { d.rbl += 16; if ((u32) d.rbl < k) d.rbl += 16; } }
      else {
        u32 have = (u32) d.rbl;
        while (have < k) {
          if (!d.ref_fill()) {                                          // ERR_READ at the bit that needed it
            u32 used = (u32) d.rbl;                                     // bits consumed before the failing read
            d.need((int) used); if (used) { d.bb <<= used; d.bl -= (int) used; }
            d.rbl = 0; d.H = H; d.L = L; d.C = C;
            return -1;
          }
          have = (u32) d.rbl;
        }
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
  QT(2);
#ifdef QTM_TIMERS
  d.tm[4]++;
#endif
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

// the nine models (qtm.h:49-77) and their shift counters
struct QtmModels {
  u32 m0, m1, m2, m3, m4, m5, m6, m6l, m7;
  u32 shifts;                      /* lane k: shiftsleft of model k (0-3 literals, 4, 5, 6, 7 = 6len, 8 = selector) */
  u32 n4, n5, n6;
  u32 t01, t23, t45, t66, t7;      /* the models' totals (cumfreq[0]) on the scalar side, two per register: qtm_get_symbol */
};
enum { QTM_T_LIT = 0, QTM_T_MATCH = 1, QTM_T_READ = -1, QTM_T_DECRUNCH = -2 };
// one token (qtmd.c:292-350): the selector, then a literal from one of four models or a match's length / offset.
// Every model has its own inlined copy of GET_SYMBOL (round 3 moved the literal models through a temporary: a chain of
// selects in front of and behind every literal); FAST as in qtm_get_symbol.
template <bool FAST>
__device__ __forceinline__ int qtm_token(QtmDec &d, QtmModels &M, u32 &val, u32 &mlen)
{
  const int sel = qtm_get_symbol<FAST, 0>(d, M.m7, 7, M.shifts, 8u, M.t7);
  int sym;
  u32 base, extra, v;
  switch (sel) {
  case 0: sym = qtm_get_symbol<FAST, 0>(d, M.m0, 64, M.shifts, 0u, M.t01); if (sym < 0) return QTM_T_READ; val = (u32) sym; return QTM_T_LIT;
  case 1: sym = qtm_get_symbol<FAST, 16>(d, M.m1, 64, M.shifts, 1u, M.t01); if (sym < 0) return QTM_T_READ; val = (u32) sym; return QTM_T_LIT;
  case 2: sym = qtm_get_symbol<FAST, 0>(d, M.m2, 64, M.shifts, 2u, M.t23); if (sym < 0) return QTM_T_READ; val = (u32) sym; return QTM_T_LIT;
  case 3: sym = qtm_get_symbol<FAST, 16>(d, M.m3, 64, M.shifts, 3u, M.t23); if (sym < 0) return QTM_T_READ; val = (u32) sym; return QTM_T_LIT;
  case 4:
    sym = qtm_get_symbol<FAST, 0>(d, M.m4, M.n4, M.shifts, 4u, M.t45); if (sym < 0) return QTM_T_READ;
    qtm_pos_slot((u32) sym, base, extra);
    if (!d.template read_many_t<FAST>((int) extra, v)) return QTM_T_READ;
    val = base + v + 1u; mlen = 3u; return QTM_T_MATCH;
  case 5:
    sym = qtm_get_symbol<FAST, 16>(d, M.m5, M.n5, M.shifts, 5u, M.t45); if (sym < 0) return QTM_T_READ;
    qtm_pos_slot((u32) sym, base, extra);
    if (!d.template read_many_t<FAST>((int) extra, v)) return QTM_T_READ;
    val = base + v + 1u; mlen = 4u; return QTM_T_MATCH;
  case 6:
    sym = qtm_get_symbol<FAST, 16>(d, M.m6l, 27, M.shifts, 7u, M.t66); if (sym < 0) return QTM_T_READ;
    qtm_len_slot((u32) sym, base, extra);
    if (!d.template read_many_t<FAST>((int) extra, v)) return QTM_T_READ;
    mlen = base + v + 5u;
    sym = qtm_get_symbol<FAST, 0>(d, M.m6, M.n6, M.shifts, 6u, M.t66); if (sym < 0) return QTM_T_READ;
    qtm_pos_slot((u32) sym, base, extra);
    if (!d.template read_many_t<FAST>((int) extra, v)) return QTM_T_READ;
    val = base + v + 1u; return QTM_T_MATCH;
  default:
    return sel < 0 ? QTM_T_READ : QTM_T_DECRUNCH;
  }
}

// (forced inline: as a real function its arguments -- and with them the whole coder state -- arrive in VECTOR registers and
// count as divergent; the chain then runs on the vector unit at twice the latency)
// MARKS: the instantiation for units that carry marks (MSPACK_HIP_UF_QTM_MARKS) -- a kernel of its own (mspack_decode_qtm_marks):
// inside the one loop the compare per token cost the folders WITHOUT marks 5.7 % (config 4: 342 -> 362 ms; the coder's chain is one
// wave's in-order instruction stream, every instruction in the loop is on it)
template <bool MARKS>
__device__ __forceinline__ void qtm_decode_unit(const mspack_hip_unit &u, const u8 *in_arena, u8 *out_arena,
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
  d.seek_swapped(0);
  d.bb = 0; d.bl = 0; d.rbl = 0; d.H = 0; d.L = 0; d.C = 0;
#ifdef QTM_TIMERS
  for (int k_ = 0; k_ < 8; k_++) d.tm[k_] = 0;
  const unsigned long long qt_unit_ = __builtin_amdgcn_s_memtime();
#endif
  u8 *out = out_arena + u.out_off;
  const u32 wsize = 1u << wb, out_len = u.out_len;

  QtmModels M;
  M.n4 = (2u * wb > 24u) ? 24u : 2u * wb; M.n5 = (2u * wb > 36u) ? 36u : 2u * wb; M.n6 = 2u * wb;
  M.m0 = qtm_model_init(lane, 0, 64); M.m1 = qtm_model_init(lane, 64, 64); M.m2 = qtm_model_init(lane, 128, 64);
  M.m3 = qtm_model_init(lane, 192, 64); M.m4 = qtm_model_init(lane, 0, M.n4); M.m5 = qtm_model_init(lane, 0, M.n5);
  M.m6 = qtm_model_init(lane, 0, M.n6); M.m6l = qtm_model_init(lane, 0, 27); M.m7 = qtm_model_init(lane, 0, 7);
  M.shifts = 4u;
  // (qtm_model_init: cumfreq[0] = len)
  M.t01 = 64u | (64u << 16); M.t23 = 64u | (64u << 16); M.t45 = M.n4 | (M.n5 << 16); M.t66 = M.n6 | (27u << 16); M.t7 = 7u;

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
  // MSPACK_HIP_UF_QTM_MARKS (mspack_hip.h): how far the token that covers the byte in front of a marked position runs past it --
  // what qtmd keeps in its window when a request ends there and hands to the NEXT request's output before it decodes anything
  // (qtmd.c:268-276).  One scalar compare per token; the positions come out of their table one at a time as they are passed.
  const u32 n_marks = (MARKS && (u.flags & MSPACK_HIP_UF_QTM_MARKS)) ? u.ref_len : 0u;
  u32 next_mark = 0xFFFFFFFFu;
  if (MARKS) {
    const u32 *const marks = (const u32 *)(in_arena + (size_t) u.in_chunk * 4u);
    u32 *const mark_log = (u32 *)(out + (((size_t) out_len + 15u) & ~(size_t) 15u));
    for (u32 i = lane; i < n_marks; i += WAVE) mark_log[i] = 0u;
    if (lane == 0) { sh->mark_i = 0u; sh->n_marks = n_marks; sh->marks = marks; sh->mark_log = mark_log; }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    if (n_marks) next_mark = rfl(gld(marks));
  }
#define QTM_MARKS() do { if (MARKS && P >= next_mark) next_mark = qtm_marks_passed(sh, P, 0u, lane); } while (0)
#define QTM_MARKS_FAIL(below_) do { if (MARKS && next_mark < (below_)) next_mark = qtm_marks_passed(sh, P, (below_), lane); } while (0)
#ifdef QTM_NO_OUTPUT     /* analysis builds, never shipped: the arithmetic decoder ALONE -- no literal buffer, no
                            match queue, no copies, nothing written.  What a launch of this build takes is the floor of any split of a
                            folder into a decoding wave and a writing wave (tools/bench_qtm_config4.py; profiles/round6_qtm.txt) */
#define QTM_FLUSH() do { lit_n = 0; } while (0)
#define QTM_COPY(P_, off_, len_) do { } while (0)
#define QTM_LIT(v_, P_) do { } while (0)
#define QTM_RESOLVE_DUE(P_) do { } while (0)
#else
#define QTM_LIT(v_, P_) do { lit_buf = wrl(lit_buf, (v_), lit_n); lit_pos = wrl(lit_pos, (P_), lit_n); if (++lit_n == WAVE) QTM_FLUSH(); } while (0)
#define QTM_RESOLVE_DUE(P_) do { if (spq_due(Q, (P_))) { QTM_FLUSH(); spq_resolve(sh->spq, Q, out, (P_), false, lane, out_len); } } while (0)
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
#endif

  // The decode loop (qtmd.c:283-470).  Far from the end of the input -- a token reads fewer than 96 bytes -- no read can fail:
  // the lean decoder (FAST).  Round 5: the lean and the exact decoder are TWO copies of the whole loop, one run after the
  // other, not two arms per token inside one loop: the hot loop holds nine GET_SYMBOL bodies instead of eighteen, and what
  // only the exact reader needs stays out of its registers.  The lean copy leaves BETWEEN two tokens when the input's last
  // 96 bytes begin (switch_; o_end = wpos as at the end of every pass: the head of the loop then recomputes the same frame_end
  // from the state), so the exact copy just goes on.
  // (Two loops in a row inside the frame loop -- the first attempt -- made the compiler treat the coder's wave-uniform
  // state as divergent: the whole chain moved to the vector unit.  Checked in the ISA: build/isa/qtm_r5.txt.)
#ifndef QTM_SPLIT_LOOPS
#define QTM_SPLIT_LOOPS 1          /* 0: one loop, lean or exact decided per token (rounds 3-4) */
#endif
#define QTM_FAR_FROM_END() (d.w.origin + d.w.wi * 4u + 96u <= d.w.in_len)
#define QTM_DECODE_LOOP(LEAVE_WHEN_NEAR_END_, TOKEN_, switch_)                                        \
  while ((long long)(o_end - o_ptr) < need) {                                                         \
    u32 v;                                                                                            \
    if (!header_read) {                                                                               \
      d.H = 0xFFFFu; d.L = 0;                                                                         \
      if (!d.read_bits(16, v)) { err = ERR_READ; break; }                                             \
      d.C = v; header_read = true;                                                                    \
    }                                                                                                 \
    u32 frame_end = (u32)((long long) wpos + (need - (long long)(o_end - o_ptr)));                    \
    if (wpos + frame_todo < frame_end) frame_end = wpos + frame_todo;                                 \
    if (frame_end > wsize) frame_end = wsize;                                                         \
    bool stop = false;                                                                                \
    while (wpos < frame_end) {                                                                        \
      if (LEAVE_WHEN_NEAR_END_ && !QTM_FAR_FROM_END()) { o_end = wpos; switch_ = true; stop = true; break; } \
      QTM_MARKS();                                                                                    \
      good = P;                                                                                       \
      u32 moff = 0, mlen = 0;                                                                         \
      const int tk = TOKEN_;                                                                          \
      if (tk < 0) { err = tk == QTM_T_READ ? ERR_READ : ERR_DECRUNCH; stop = true; break; }           \
      if (tk == QTM_T_LIT) {                                                                          \
        QTM_LIT(moff, P);                                                                             \
        P++; wpos++; frame_todo--;                                                                    \
        continue;                                                                                     \
      }                                                                                               \
      frame_todo -= mlen;                                                                             \
      if (wpos + mlen > wsize) {                                      /* qtmd.c:358-390 */            \
        u32 i = wsize - o_ptr;                                                                        \
        /* (marks inside the part of the match in front of the window's end: a request ending there fails, qtmd.c:366-374) */ \
        QTM_MARKS_FAIL(P + (wsize - wpos));                                                           \
        /* (the copy itself is the same on the linear buffer; only the flush bookkeeping differs) */  \
        if ((long long) i > need) {                                                                   \
          /* first part was already copied by the reference before it bails out */                    \
          QTM_COPY(P, moff, wsize - wpos);                                                            \
          err = ERR_DECRUNCH; stop = true; break;                                                     \
        }                                                                                             \
        QTM_COPY(P, moff, mlen);                                                                      \
        written += i; need -= i; o_ptr = 0; o_end = 0;                                                \
        P += mlen; wpos = wpos + mlen - wsize;                                                        \
        break;                                                                                        \
      }                                                                                               \
      if (moff > wpos && (moff - wpos) > wsize) { err = ERR_DECRUNCH; stop = true; break; }   /* qtmd.c:399 */ \
      QTM_COPY(P, moff, mlen);                                                                        \
      P += mlen; wpos += mlen;                                                                        \
      QTM_RESOLVE_DUE(P);                                                                             \
    }                                                                                                 \
    if (stop) break;                                                                                  \
    o_end = wpos;                                                                                     \
    /* a match that overshot the frame fails every request that needed that match: `good` still       \
     * holds the position where it started */                                                         \
    if (frame_todo > QTM_FRAME) { err = ERR_DECRUNCH; break; }         /* qtmd.c:424 */               \
    good = P ? P - 1u : 0u;       /* errors in the frame-end handling hit the request ending here */  \
    if (frame_todo == 0u) {                                                                           \
      int n = d.rbl & 7;                                               /* qtmd.c:432 */               \
      if (n) { d.need(n); d.bb <<= n; d.bl -= n; d.rbl -= n; }                                        \
      bool ok = true;                                                                                 \
      do { if (!d.read_bits(8, v)) { ok = false; break; } } while (v != 0xFFu);                       \
      if (!ok) { err = ERR_READ; break; }                                                             \
      header_read = false; frame_todo = QTM_FRAME;                                                    \
    }                                                                                                 \
    good = P;                                                                                         \
    if (wpos == wsize) {                                                                              \
      u32 i = o_end - o_ptr;                                                                          \
      if ((long long) i >= need) break;                                                               \
      written += i; need -= i; o_ptr = 0; o_end = 0; wpos = 0;                                        \
    }                                                                                                 \
  }
  {
    bool to_exact = false;
#if QTM_SPLIT_LOOPS
    QTM_DECODE_LOOP(true, qtm_token<true>(d, M, moff, mlen), to_exact)
    if (to_exact) { bool never = false; QTM_DECODE_LOOP(false, qtm_token<false>(d, M, moff, mlen), never) (void) never; }
#else
    QTM_DECODE_LOOP(false, (QTM_FAR_FROM_END() ? qtm_token<true>(d, M, moff, mlen) : qtm_token<false>(d, M, moff, mlen)), to_exact)
    (void) to_exact;
#endif
  }
#undef QTM_DECODE_LOOP
#undef QTM_FAR_FROM_END
  if (err == ERR_OK) QTM_MARKS();      // (the marks the last tokens passed -- unless what follows those tokens failed: a request that
                                       // ends inside them then fails too: the frame's end, its trailer, qtmd.c:424-441)
#undef QTM_MARKS
#undef QTM_MARKS_FAIL
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
#undef QTM_LIT
#undef QTM_RESOLVE_DUE
#ifdef QTM_TIMERS
  if (blockIdx.x == 0 && lane == 0) {
    for (int k_ = 0; k_ < 5; k_++) g_qtm_tm[k_] = d.tm[k_];
    g_qtm_tm[5] = __builtin_amdgcn_s_memtime() - qt_unit_; g_qtm_tm[6] = P;
  }
#endif
  if (err == ERR_OK && need) { written += (u32) need; }
  if (err == ERR_OK) good = written;
  if (lane == 0) {
    res->err = err; res->flags = 0; res->out_len = written; res->good_len = good; res->in_next = 0;
    res->in_used = d.w.origin + ((d.cons_bits() + (u32) d.rbl) >> 3);
  }
}
