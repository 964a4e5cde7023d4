// mszip_kernel.hpp -- MSZIP unit decoder: one wavefront per CAB folder stream.
//
// Replaces, for one unit, mszipd_init + mszipd_decompress(out_len) of the reference
// (libmspack/mspack/mszipd.c:335-460) including inflate (mszipd.c:154-316), the dynamic-header
// reader (mszipd.c:91-151), the "CK" scan (mszipd.c:406-414) and the 32 KiB history that is NOT
// cleared between blocks (mszipd.c:267-268).
//   bit reader ...... readbits.h:133-180 + mszipd.c:19-27 (LSB first, bytewise) -> 64-bit SGPR
//                     buffer refilled 32 bits at a time from the lane-resident input chunk
//   READ_HUFFSYM .... one LDS lookup (10 / 7 direct bits, index = bit-reversed code); longer codes
//                     by wave-wide limit compare + ballot
//   window .......... the reference's window[32768] restarts at index 0 for every block and keeps
//                     older blocks' bytes above the current position.  Here block b decodes straight
//                     into out[B_b + idx]; an index the current block has not written yet is
//                     served from the most recent earlier block that was long enough (a short
//                     monotonic stack of (B, length) pairs) -- bit-identical to the ring.
// Output regions of MSZIP units need 32768 bytes of slack after out_len (a block is decoded in
// full even when only part of it is requested, mszipd.c:442-446).
#pragma once
#include "wave_common.hpp"
#include "spec_queue.hpp"

#define ZIP_FRAME 32768u
#define ZIP_LIT_P 10
#define ZIP_DIST_P 9
#define ZIP_BL_P 7
#define ZIP_HIST 8

#define ZIP_STAGE_WORDS 768u       /* zip_parse_lanes: 3 KiB of a CFDATA block's input per pass */
#define ZIP_SEG 8u                 /* zip_parse_lanes: tokens per segment of the balanced last walk (lzx_parse_emit's scheme) */
#define ZIP_LIT_RING 1024u         /* zip_parse_lanes: bytes of the literal ring (the last walk's literals leave as 16-byte rows) */
#define ZIP_LANE_TAIL 384u         /* bits a lane walks in front of its stretch's end to find its exit */
#define ZIP_LANE_ROUNDS 5u         /* walks before the consistent prefix is taken as it is */

#define ZIP_TOK_CAP (REC_CHUNK * REC_CHUNKS)   /* match records a CFDATA block can have at most in the launch's pool (a block has <= 10923) */

// what a parse wave leaves for the unit's wave (same slots as LzxFrameRec; only the head is used)
struct ZipBlockRec {
  u32 status;                      /* 0 = not yet, 1 = the whole CFDATA block was parsed, 2 = the parse wave gave up */
  u32 n_tokens;
  u32 start_bit;                   /* first bit of the deflate data (behind 'C','K'), from the unit's first byte */
  u32 end_bit;                     /* first bit behind the last end-of-block symbol */
  u32 eob_rbl;                     /* the reference's bits_left there */
  u32 total_out;                   /* bytes the block produces (<= 32768) */
  u32 chunk[REC_CHUNKS];           /* where the block's match records are (wave_common.hpp: RecPool) */
  u32 fold;                        /* mspack_mszip_fold (zip_fold_block): 0 = not (yet), 1 = this block's matches are copied and so are those
                                      of every block of the folder below it, 2 = the chain of folded blocks ended at or below this one */
  u32 pad[329];
};
static_assert(sizeof(ZipBlockRec) == 1408, "ZipBlockRec slot size");

struct __align__(16) MszipShared {
  u16 lit_tab[1 << ZIP_LIT_P];
  u16 lit_sorted[288];
  u16 dist_tab[1 << ZIP_DIST_P];
  u16 dist_sorted[32];
  u16 bl_tab[1 << ZIP_BL_P];
  u16 bl_sorted[20];
  u32 cnt[20];
  u32 hist_B[ZIP_HIST], hist_len[ZIP_HIST];
  u32 ctab[REC_CHUNKS];            /* parse waves: the block's chunk list (RecWriter) */
  u8  lit_len[288];
  u8  dist_len[32];
  u8  bl_len[20];
  u8  lens[324];
  union {
    struct {
      u32 inbuf[128 + 4];        /* speculative path: two 256-byte input chunks */
      SpecQueueLds spq;          /* speculative path: queued matches + start flags (spec_queue.hpp) */
      u32 tq0[128], tq1[128];    /* speculative path: parsed tokens waiting for their commit (zip_run_spec) */
    };
    struct {
      u32 stage[ZIP_STAGE_WORDS + 64u + 4u];   /* parse waves (zip_parse_lanes): the input of one pass, stream order */
      alignas(16) u32 litring[ZIP_LIT_RING / 4u];   /* ... the literals of the last walk's rounds on their way out */
      u8 owner[512];                            /* ... which lane's stretch a segment of the last walk belongs to */
    };
  };
};

// inflate() failure classes: <0 = format error (-> MSPACK_ERR_DECRUNCH), >0 = MSPACK_ERR_READ
#define ZIP_E_FORMAT (-1)

struct ZipDec {
  InWindow w;
  u64 bb; int bl;
  bool near_end, careful; int rbl;
  u32 lane;
  MszipShared *sh;
  HuffRegs hr_lit, hr_dist, hr_bl;
  // output / window
  u8 *out;
  u32 B, wpos; bool flushed;
  u32 ref_bo;              // the reference's zip->bytes_output for the current block: every flush adds, also the one that overflows (mszipd.c:320-331)
  u32 hist_n;                    // entries in the (B, length) stack, most recent first
  u32 lit_buf, lit_n;
  // what the reference's stream struct holds (STORE_BITS, readbits.h:119-124): repair mode restarts from it
  u32 snap_iptr; int snap_rbl;

  __device__ __forceinline__ u32 cons_bits() const { return w.wi * 32u - (u32) bl; }
  // the reference's i_ptr: every byte up to here has been moved into its bit buffer
  __device__ __forceinline__ u32 abs_iptr() const { return w.origin + ((cons_bits() + (u32) rbl) >> 3); }
  __device__ __forceinline__ void store_bits() { snap_iptr = abs_iptr(); snap_rbl = rbl; }
  __device__ __forceinline__ void refill() {
    u32 d = w.next_dword(lane);
    bb |= (u64) d << bl;
    bl += 32;
    u32 fetched = w.origin + w.wi * 4u;
    if (fetched >= w.in_len || w.in_len - fetched <= 64u) near_end = true;
  }
  __device__ __forceinline__ void need(int n) { if (bl < n) refill(); }
  __device__ bool ref_ensure(int n) {             // bytewise ENSURE_BITS, EOF-exact
    while (rbl < n) {
      u32 i = w.origin + ((cons_bits() + (u32) rbl) >> 3);
      if (i >= w.in_len + w.eofs) return false;     // fabricated zero bytes, then ERR_READ
      rbl += 8;
    }
    return true;
  }
  __device__ __forceinline__ bool sym_ensure() {  // ENSURE_BITS(16)
    if (careful) return ref_ensure(16);
    if (near_end) { careful = true; rbl = 16 + (int)((0u - cons_bits()) & 7u); }
    return true;
  }
  __device__ __forceinline__ void drop(int n) { bb >>= n; bl -= n; if (careful) rbl -= n; }
  // READ_BITS(n), 0 <= n <= 16; returns false on ERR_READ
  __device__ __forceinline__ bool read_bits(int n, u32 &v) {
    need(n);
    if (careful && !ref_ensure(n)) return false;
    v = (u32) bb & ((1u << n) - 1u);
    drop(n);
    return true;
  }
  // returns symbol, -1 = format error, -2 = ERR_READ
  template <int TP>
  __device__ __forceinline__ int decode_sym(const u16 *tab, const u16 *sorted, const HuffRegs &hr, bool ens16) {
    if (ens16) { if (!sym_ensure()) return -2; }
    u32 e = rfl((u32) tab[(u32) bb & ((1u << TP) - 1u)]);
    if (e == 0) {
      e = huff_long(hr, sorted, __brev((u32) bb) >> 16, lane);
      if (e == 0) return -1;
    }
    drop((int)(e >> 10));
    return (int)(e & 1023u);
  }
  // byte-align the stream (drop bits_left & 7, mszipd.c:173,407)
  __device__ __forceinline__ void byte_align() { int n = (int)((0u - cons_bits()) & 7u); if (n) drop(n); }
  // stream byte position of the next unread bit (must be byte aligned)
  __device__ __forceinline__ u32 byte_pos() const { return w.origin + (cons_bits() >> 3); }
  // restart bit reading at an absolute byte position with an empty buffer (bits_left = 0)
  __device__ __forceinline__ void restart(u32 pos) {
    w.seek(pos, lane); bb = 0; bl = 0; rbl = 0;
    near_end = (pos >= w.in_len || w.in_len - pos <= 64u);
    if (near_end) careful = true;
  }

  // ---- window --------------------------------------------------------------------------------
  // byte at window index x (< 32768) given that the current block has `hw` valid bytes from 0
  __device__ __forceinline__ u32 win_byte(u32 x, u32 hw) const {
    if (x < hw) return out[B + x];
    for (u32 k = 0; k < hist_n; k++) {
      u32 hl = sh->hist_len[k];
      if (x < hl) return out[sh->hist_B[k] + x];
    }
    return 0u;                                     // never written: the reference reads junk here
  }
  __device__ __forceinline__ void flush_lits() {
    if (lit_n) {
      if (lane < lit_n) out[B + wpos - lit_n + lane] = (u8) lit_buf;
      lit_n = 0;
    }
  }
  // FLUSH_IF_NEEDED (mszipd.c:37-44): returns false on overflow
  __device__ __forceinline__ bool wrap_if_needed() {
    if (wpos == ZIP_FRAME) {
      flush_lits();
      ref_bo += ZIP_FRAME;
      if (flushed) return false;
      flushed = true; wpos = 0;
    }
    return true;
  }
  // copy `n` bytes (no destination wrap inside) from window index mpos, distance `dist`
  __device__ __forceinline__ void copy_part(u32 mpos, u32 dist, u32 n) {
    const u32 hw0 = flushed ? ZIP_FRAME : wpos;
    if (dist >= n || dist >= WAVE) {
      for (u32 c = 0; c < n; c += WAVE) {
        u32 k = c + lane;
        u32 hw = flushed ? ZIP_FRAME : (wpos + c);
        if (k < n) out[B + wpos + k] = (u8) win_byte((mpos + k) & (ZIP_FRAME - 1u), hw);
      }
    }
    else {
      u32 r = lane, s = dist << 5, step = 64u, ss = dist << 5;
#pragma unroll
      for (int q = 0; q < 6; q++) { u32 t = r - s; r = t < r ? t : r; s >>= 1; }
#pragma unroll
      for (int q = 0; q < 6; q++) { u32 t = step - ss; step = t < step ? t : step; ss >>= 1; }
      for (u32 k = lane; k < n; k += WAVE) {
        out[B + wpos + k] = (u8) win_byte((mpos + r) & (ZIP_FRAME - 1u), hw0);
        r += step; if (r >= dist) r -= dist;
      }
    }
  }
};

// RFC 1951 3.2.5 tables in closed form (mszipd.c:46-68)
__device__ __forceinline__ void zip_len_code(u32 code, u32 &base, u32 &extra) {
  if (code < 8u) { base = 3u + code; extra = 0; }
  else if (code == 28u) { base = 258u; extra = 0; }
  else { extra = (code - 4u) >> 2; base = 3u + ((4u + (code & 3u)) << extra); }
}
__device__ __forceinline__ void zip_dist_code(u32 code, u32 &base, u32 &extra) {
  if (code < 4u) { base = 1u + code; extra = 0; }
  else { extra = (code - 2u) >> 1; base = 1u + ((2u + (code & 1u)) << extra); }
}

// zip_read_lens (mszipd.c:91-151): 0 ok, <0 format error, >0 ERR_READ
__device__ __forceinline__ int zip_read_dynamic(ZipDec &d)
{
  MszipShared *sh = d.sh;
  static const u8 order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
  u32 nlit, ndist, nbl, v;
  if (!d.read_bits(5, nlit) || !d.read_bits(5, ndist) || !d.read_bits(4, nbl)) return ERR_READ;
  nlit += 257u; ndist += 1u; nbl += 4u;
  if (nlit > 288u || ndist > 32u) return ZIP_E_FORMAT;
  if (d.lane < 20u) sh->bl_len[d.lane] = 0;
  for (u32 i = 0; i < nbl; i++) { if (!d.read_bits(3, v)) return ERR_READ; sh->bl_len[order[i]] = (u8) v; }
  if (huff_build<ZIP_BL_P>(sh->bl_len, 19, 7, sh->bl_tab, sh->bl_sorted, sh->cnt, d.hr_bl, d.lane, true)) return ZIP_E_FORMAT;
  u32 last = 0, total = nlit + ndist;
  for (u32 i = 0; i < total; i++) {
    d.need(16);
    if (d.careful && !d.ref_ensure(7)) return ERR_READ;          // ENSURE_BITS(7), mszipd.c:122
    int code = d.decode_sym<ZIP_BL_P>(sh->bl_tab, sh->bl_sorted, d.hr_bl, false);
    if (code < 0) return code == -2 ? ERR_READ : ZIP_E_FORMAT;
    if (code < 16) { sh->lens[i] = (u8) code; last = (u32) code; continue; }
    u32 run, val;
    if (code == 16) { if (!d.read_bits(2, run)) return ERR_READ; run += 3u; val = last; }
    else if (code == 17) { if (!d.read_bits(3, run)) return ERR_READ; run += 3u; val = 0; }
    else if (code == 18) { if (!d.read_bits(7, run)) return ERR_READ; run += 11u; val = 0; }
    else return ZIP_E_FORMAT;
    if (i + run > total) return ZIP_E_FORMAT;
    for (u32 k = d.lane; k < run; k += WAVE) sh->lens[i + k] = (u8) val;
    i += run - 1u;
  }
  for (u32 k = d.lane; k < 288u; k += WAVE) sh->lit_len[k] = (k < nlit) ? sh->lens[k] : (u8) 0;
  if (d.lane < 32u) sh->dist_len[d.lane] = (d.lane < ndist) ? sh->lens[nlit + d.lane] : (u8) 0;
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// Speculative decode of the literal/length + distance token loop (mszipd.c:228-303), the same two-phase
// scheme as lzx_run_spec.  PARSE, per round: all 64 lanes decode a COMPLETE token (literal/length symbol,
// length extra bits, distance symbol, distance extra bits) starting at bit (bitpos + lane) of the
// LSB-first stream; the chain of real tokens is followed with v_readlane; the on-chain tokens go into a
// token queue in LDS.  COMMIT, per 64 queued tokens (one per lane): a DPP prefix sum gives every token its
// output position, literals go out in one store and matches are queued for spec_queue.hpp.
// The parse stops -- always at a token boundary -- at the end-of-block symbol, at a token this path does
// not take (an invalid code, a long distance code) and near the end of the input; the commit stops at a
// match that reaches into an earlier block's bytes and where the output reaches the 32 KiB mark, and
// hands the bit position of that token back; the EOF-exact scalar loop of zip_inflate takes over from there.
// Returns 1 when the end-of-block symbol was consumed, 0 otherwise.
// ---------------------------------------------------------------------------------------------------
struct ZipTok { u32 tot, sym, kind, olen, dist; bool unk; };      // kind 0 literal, 1 match, 2 end of block

__device__ __forceinline__ ZipTok zip_spec_token(const MszipShared *sh, const u32 lit_fov, const u32 *llim, u64 r,
                                                 const u32 *dlim = nullptr, const u32 dist_fov = 0u)
{
  ZipTok t;
  const u32 lo = (u32) r;
  u32 e = sh->lit_tab[lo & ((1u << ZIP_LIT_P) - 1u)];
  {
    // codes longer than the direct table: canonical length = number of per-length limits the
    // (MSB-first) 16-bit peek is not below; symbol via the sorted list (readhuff.h:144-172)
    u32 peek16 = __brev(lo) >> 16, ln = ZIP_LIT_P + 1u;
#pragma unroll
    for (int l = ZIP_LIT_P + 1; l <= 16; l++) ln += (peek16 >= llim[l - ZIP_LIT_P - 1]) ? 1u : 0u;
    u32 lq = ln <= 16u ? ln : 0u;
    u32 fo = (u32) __builtin_amdgcn_ds_bpermute((int)(lq << 2), (int) lit_fov);
    u32 idx = (fo >> 16) + ((peek16 >> (16u - lq)) - (fo & 0xFFFFu));
    if (idx >= 288u) idx = 0;
    u32 ls = sh->lit_sorted[idx];
    if (e == 0u && lq != 0u) e = ls | (lq << 10);
  }
  bool unk = (e == 0u);
  u32 tot = e >> 10;
  const u32 sym = e & 1023u;
  r >>= tot;
  const bool is_match = sym > 256u;
  const u32 code = sym - 257u;
  if (is_match && code >= 29u) unk = true;                        // mszipd.c:246
  u32 lbase, lextra, dbase, dextra;
  zip_len_code(code < 29u ? code : 0u, lbase, lextra);
  const u32 lev = (u32) r & ((1u << lextra) - 1u);
  r >>= lextra;
  u32 e2 = sh->dist_tab[(u32) r & ((1u << ZIP_DIST_P) - 1u)];
  if (dlim && ballot(is_match && e2 == 0u)) {                     // a distance code beyond the direct table (rare)
    const u32 pk = __brev((u32) r) >> 16;
    u32 ln = ZIP_DIST_P + 1u;
#pragma unroll
    for (int l = ZIP_DIST_P + 1; l <= 16; l++) ln += (pk >= dlim[l - ZIP_DIST_P - 1]) ? 1u : 0u;
    const u32 lq = ln <= 16u ? ln : 0u;
    const u32 fo = (u32) __builtin_amdgcn_ds_bpermute((int)(lq << 2), (int) dist_fov);
    u32 idx = (fo >> 16) + ((pk >> (16u - lq)) - (fo & 0xFFFFu));
    if (idx >= 32u) idx = 0;
    const u32 dsr = sh->dist_sorted[idx];
    if (e2 == 0u && lq != 0u) e2 = dsr | (lq << 10);
  }
  const u32 ds = e2 & 1023u;
  zip_dist_code(ds < 30u ? ds : 0u, dbase, dextra);
  r >>= (e2 >> 10);
  const u32 dev = (u32) r & ((1u << dextra) - 1u);
  if (is_match) {
    unk = unk || e2 == 0u || ds >= 30u;                           // long distance codes: scalar loop
    tot += lextra + (e2 >> 10) + dextra;
  }
  t.tot = tot; t.sym = sym; t.unk = unk;
  t.kind = is_match ? 1u : (sym == 256u ? 2u : 0u);
  t.olen = is_match ? lbase + lev : (sym == 256u ? 0u : 1u);
  t.dist = dbase + dev;
  return t;
}

__device__ __forceinline__ int zip_run_spec(ZipDec &d)
{
  MszipShared *sh = d.sh;
  const u32 lane = d.lane;
  // Positions are linear positions in the unit's output: block base + window index.  When the
  // previous block filled the whole 32 KiB window, the bytes a match finds "above" the current
  // position in the reference's ring (mszipd.c:267-268) are exactly the previous block's, which lie
  // right below this block in the output: such a source is an ordinary linear copy.
  u8 *const out = d.out;
  const u32 B = rfl(d.B);
  const bool lin_hist = d.hist_n > 0u && rfl(sh->hist_len[0]) == ZIP_FRAME && rfl(sh->hist_B[0]) + ZIP_FRAME == B;
  u32 P = B + rfl(d.wpos);
  // same margin reasoning as lzx_run_spec: a round consumes at most 64 + 48 bits
  const u32 room_bytes = (d.w.in_len > d.w.origin + 56u) ? (d.w.in_len - d.w.origin - 56u) : 0u;
  const u32 bit_limit = rfl(room_bytes * 8u);
  u32 bitpos = rfl(d.cons_bits());
  if (bitpos >= bit_limit || P - B >= ZIP_FRAME - 1u) return 0;
  d.flush_lits();
  u32 cb = bitpos >> 11;
  {
    u32 lo = d.w.load_chunk(cb, lane), hi = d.w.load_chunk(cb + 1u, lane);
    sh->inbuf[lane] = lo; sh->inbuf[64u + lane] = hi;
    if (lane < 4u) sh->inbuf[128u + lane] = 0;
  }
  u32 pf = d.w.load_chunk(cb + 2u, lane);
  u32 llim[16 - ZIP_LIT_P];
#pragma unroll
  for (int l = ZIP_LIT_P + 1; l <= 16; l++) llim[l - ZIP_LIT_P - 1] = rdl(d.hr_lit.limv, (u32) l);
  SpecQueue Q;
  spq_init(sh->spq, Q, P, lane);
  u32 *const tq0 = sh->tq0, *const tq1 = sh->tq1;
  u32 th = 0, tt = 0;                                  // token queue: committed / parsed (counters)
  bool stop = false;                                   // the parser is done
  int rc = 0, eob_rbl = 0;
  u32 eob_end = 0;                                     // bit position behind the end-of-block symbol

  for (;;) {
    // =================================== PARSE ===================================
    if (!stop && tt - th < 64u) {
      if ((bitpos >> 11) != cb) {                       // slide the LDS window by one chunk
        u32 up = sh->inbuf[64u + lane];
        sh->inbuf[lane] = up; sh->inbuf[64u + lane] = pf;
        cb++;
        pf = d.w.load_chunk(cb + 2u, lane);
      }
      // ---- every lane decodes the token that would start at bit (bitpos + lane) ----
      const u32 rel = bitpos - (cb << 11) + lane;
      const u32 k = rel >> 5, sft = rel & 31u;
      const u32 i0 = sh->inbuf[k], i1 = sh->inbuf[k + 1u], i2 = sh->inbuf[k + 2u];
      const u64 q01 = ((u64) i1 << 32) | i0, q12 = ((u64) i2 << 32) | i1;
      const u64 r = (u64)(u32)(q01 >> sft) | ((u64)(u32)(q12 >> sft) << 32);
      const ZipTok t = zip_spec_token(sh, d.hr_lit.fov, llim, r);
      // next token start; >= 128 ends the walk: 128 + lane = not for this path, 192 + lane = end of block
      const u32 vnext = t.unk ? (128u + lane) : (t.kind == 2u ? (192u + lane) : (lane + t.tot));
      // ---- follow the real token boundaries ----
      u64 chain = 0;
      u32 q = 0;
      do { chain |= 1ull << q; q = rdl(vnext, q); } while (q < WAVE);
      if (q >= 192u) {
        // the end-of-block symbol stays on the chain and ends the parse.  The reference's bits_left
        // behind it: ENSURE_BITS(16) at its first bit, minus its length
        q -= 192u;
        const u32 st = bitpos + q, tl = rdl(t.tot, q);
        eob_rbl = (int)(16u + ((0u - st) & 7u) - tl);
        eob_end = st + tl;
        stop = true;
      }
      else if (q >= 128u) { q -= 128u; chain &= ~(1ull << q); stop = true; }   // the scalar loop takes this token
      // ---- queue the tokens on the chain ----
      {
        const u32 rank = __builtin_amdgcn_mbcnt_hi((u32)(chain >> 32), __builtin_amdgcn_mbcnt_lo((u32) chain, 0u));
        if ((chain >> lane) & 1ull) {
          const u32 ti = (tt + rank) & 127u;
          tq0[ti] = t.kind | (t.olen << 3) | (((bitpos + lane) & 0xFFFFu) << 12);
          tq1[ti] = t.kind == 0u ? t.sym : t.dist;
        }
        tt += (u32) __popcll(chain);
      }
      bitpos += q;
      if (bitpos >= bit_limit) stop = true;
      if (!stop && tt - th < 64u) continue;
    }
    // =================================== COMMIT ===================================
    u32 n = tt - th;
    if (n > 64u) n = 64u;
    if (n == 0u) break;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    const u32 ci = (th + lane) & 127u;
    const u32 c0 = tq0[ci], c1 = tq1[ci];
    const u32 kind = c0 & 7u;
    const u32 olen = lane < n ? ((c0 >> 3) & 511u) : 0u;
    const u32 incl = wave_incl_scan(olen);
    const u32 opos = P + incl - olen;
    u32 newP = P + rdl(incl, 63u);
    // tokens this path leaves to the scalar loop: output reaching the 32 KiB mark (FLUSH_IF_NEEDED,
    // mszipd.c:37-44) and matches whose source lies before the start of this block (mszipd.c:267-268)
    bool cut = false;
    {
      const u64 cm = ballot(lane < n && (opos - B + olen >= ZIP_FRAME ||
                                         (kind == 1u && c1 > opos - B && !lin_hist)));
      if (cm) { const u32 j = (u32) __ffsll((long long) cm) - 1u; n = j; newP = rdl(opos, j); cut = true; }
    }
    const bool valid = lane < n;
    if (valid && kind == 0u) out[opos] = (u8) c1;
    const bool eob = ballot(valid && kind == 2u) != 0ull;
    u64 mm = ballot(valid && kind == 1u);
    if (mm) {
      bool ism = (mm >> lane) & 1ull;
      if (Q.mcount + (u32) __popcll(mm) > SPQ_CAP) spq_resolve(sh->spq, Q, out, P, true, lane);
      for (;;) {
        // a push must keep every start flag inside the ring (spec_queue.hpp): take the matches that end
        // inside it, resolve up to the first one that does not, go on
        const u32 limit = (Q.Pf & ~63u) + SPQ_RING;
        const u64 fit = newP <= limit ? mm : ballot(ism && opos + olen <= limit);
        if (fit) {
          const u32 rank = __builtin_amdgcn_mbcnt_hi((u32)(fit >> 32), __builtin_amdgcn_mbcnt_lo((u32) fit, 0u));
          spq_push(sh->spq, Q, (fit >> lane) & 1ull, rank, (u32) __popcll(fit), opos, c1, olen);
          mm &= ~fit;
          ism = (mm >> lane) & 1ull;
        }
        if (!mm) break;
        spq_resolve(sh->spq, Q, out, rdl(opos, (u32) __ffsll((long long) mm) - 1u), true, lane);
      }
    }
    th += n;
    P = newP;
    if (spq_due(Q, P)) spq_resolve(sh->spq, Q, out, P, false, lane);
    if (eob) { rc = 1; bitpos = eob_end; break; }       // (the end-of-block symbol is the last token parsed)
    if (cut) break;
  }
  spq_resolve(sh->spq, Q, out, P, true, lane);
  // parsed but not committed: the bit position goes back to the first such token
  if (!rc && tt != th) {
    const u32 lo = rfl(tq0[th & 127u]) >> 12;
    bitpos -= (bitpos - lo) & 0xFFFFu;
  }
  // hand the exact bit position back to the scalar reader; its bits_left restarts from the byte the
  // position lies in (every later ENSURE_BITS re-derives the reference's value from there)
  d.wpos = P - B;
  {
    u32 wi = bitpos >> 5, ch = wi >> 6;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    u32 lo = sh->inbuf[lane], hi = sh->inbuf[64u + lane];
    if (ch == cb) { d.w.cur = lo; d.w.nxt = hi; }
    else if (ch == cb + 1u) { d.w.cur = hi; d.w.nxt = pf; }
    else { d.w.cur = d.w.load_chunk(ch, lane); d.w.nxt = d.w.load_chunk(ch + 1u, lane); }   // went back
    d.w.wi = wi; d.bb = 0; d.bl = 0;
    d.refill(); d.refill();
    u32 sk = bitpos & 31u;
    if (sk) { d.bb >>= sk; d.bl -= (int) sk; }
    d.rbl = rc ? eob_rbl : (int)((0u - bitpos) & 7u);
  }
  return rc;
}

// ---------------------------------------------------------------------------------------------------
// Block-level parse parallelism (units that carry a frame table: the CFDATA blocks' offsets in the folder stream).
// Every CFDATA block of an MSZIP folder is a deflate stream of its own ('C','K' + blocks with their own Huffman
// tables, mszipd.c:406-418); only its MATCHES may reach into the previous block's bytes (mszipd.c:267-268).  So one
// PARSE wave per CFDATA block (mspack_mszip_parse) decodes the block's tokens into global memory, and the folder's own
// wave (mspack_decode_mszip) commits them block after block (zip_run_tokens: positions, literals, match queue) --
// what stays serial per folder is a quarter of the work.  A parse wave gives up silently on anything unusual (a
// stored deflate block, an error, the last bytes of the input, more than 32 KiB of output); the folder's wave
// adopts a record only if it was parsed from exactly its bit position, and decodes everything else as before.
// ---------------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------------
// zip_parse_lanes -- the lane-stretch parser of the LZX path (lzx_kernel.hpp, lzx_parse_emit) for deflate: same
// tokens of the current deflate block from the reader's bit position on; returns 1 = the end-of-block symbol was
// consumed (bit position and the reference's bits_left behind it are in the decoder), 0 = stopped in front of a token
// this path does not take (scalar reader positioned there), -1 = the block cannot be parsed here.  A pass covers up to
// 4 KiB of input (round 2's parse: 64 bit positions per round, the chain of real tokens followed):
//   * the pass's input is staged in LDS; its bits are cut into 64 stretches, one per lane (lane 0's is short: it
//     starts at a real token and walks its stretch while the others look for their exits);
//   * round 0: every other lane walks the last ZIP_LANE_TAIL bits in front of its stretch's end from an arbitrary
//     bit -- Huffman-coded token streams fall into step after a few tokens -- and hands the bit at which it leaves
//     its stretch to the next lane as that lane's entry; next rounds: lanes whose entry moved walk their stretch from
//     it, until no entry moves (a lane's tokens are then exactly the tokens of the serial decoder: its entry is the
//     exit of a lane for which that holds, down to lane 0);
//   * the end-of-block symbol and a token this path does not take (an invalid code, a distance code longer than the
//     direct table) end a lane's walk; the first such lane in chain order ends the pass;
//   * prefix sums of the lanes' output bytes and match counts give every lane its place in the block's output and in
//     the record list; a last walk stores the LITERALS straight into the output (`fout` = where the block's bytes go if
//     every earlier block of the folder is a full one -- the folder's wave checks that before it adopts the record)
//     and one record per MATCH: (position in the block, distance << 9 | length), what zip_run_tokens queues.
// mszipd.c:228-303 (the token loop), readhuff.h:144-172 (the codes).
// ---------------------------------------------------------------------------------------------------
#define ZIP_STAGE_R(p_, r_)                                                                      \
  u64 r_;                                                                                        \
  {                                                                                              \
    const u32 k_ = (p_) >> 5, s_ = (p_) & 31u;                                                   \
    const u32 x0_ = sh->stage[k_], x1_ = sh->stage[k_ + 1u], x2_ = sh->stage[k_ + 2u];          \
    r_ = (u64)(u32) __builtin_amdgcn_alignbit(x1_, x0_, s_) | ((u64)(u32) __builtin_amdgcn_alignbit(x2_, x1_, s_) << 32); \
  }

__device__ __forceinline__ int zip_parse_lanes(ZipDec &d, u8 *const fout, RecWriter &W, u32 &tt, u32 &outc)
{
  MszipShared *sh = d.sh;
  const u32 lane = d.lane;
  // Tokens may start anywhere below the end of the input (the bytes behind it are staged as zeros, which is what the
  // reference's reader fabricates there, readbits.h:194); whether the reference got as far without MSPACK_ERR_READ is
  // decided once, at the end-of-block symbol (below).
  const u32 bit_limit = rfl(d.w.in_len > d.w.origin ? (d.w.in_len - d.w.origin) * 8u : 0u);
  u32 bitpos = rfl(d.cons_bits());                       // relative to d.w.origin, like everything below
  if (bitpos >= bit_limit) return -1;
  u32 llim[16 - ZIP_LIT_P];
#pragma unroll
  for (int l = ZIP_LIT_P + 1; l <= 16; l++) llim[l - ZIP_LIT_P - 1] = rdl(d.hr_lit.limv, (u32) l);
  u32 dlim[16 - ZIP_DIST_P];
#pragma unroll
  for (int l = ZIP_DIST_P + 1; l <= 16; l++) dlim[l - ZIP_DIST_P - 1] = rdl(d.hr_dist.limv, (u32) l);
  const u32 lit_fov = d.hr_lit.fov, dist_fov = d.hr_dist.fov;
  int rc = 0, eob_rbl = 0;
  u32 lit_flushed = (outc + 15u) & ~15u;                  // literals below this position have left the ring (a multiple of 16)
  for (;;) {
    // ---- stage the input from the dword that holds bit `bitpos` ----
    const u32 sw = bitpos >> 5, sb_bit = sw << 5;
    const u32 b0 = bitpos - sb_bit;
    u32 e0 = ZIP_STAGE_WORDS * 32u; if (e0 > bit_limit - sb_bit) e0 = bit_limit - sb_bit;
    {
      InWindow ws = d.w; ws.origin = d.w.origin + sw * 4u;
      const u32 nck = (e0 + 128u + 2047u) >> 11;         // a token that starts below e0 ends below e0 + 48
      constexpr int NCH = (int)(ZIP_STAGE_WORDS / 64u) + 1;
      u32 sv[NCH];
#pragma unroll
      for (int c = 0; c < NCH; c++) sv[c] = (u32) c < nck ? ws.load_chunk((u32) c, lane) : 0u;
#pragma unroll
      for (int c = 0; c < NCH; c++) if ((u32) c < nck) sh->stage[(u32) c * 64u + lane] = sv[c];
      if (lane < 4u) sh->stage[nck * 64u + lane] = 0u;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    // ---- stretches ----
    const u32 Lb = e0 - b0;
    u32 S = (Lb + 63u) >> 6; if (S < 64u) S = 64u;
    u32 S0 = S;
    if (S > ZIP_LANE_TAIL + 64u) { S0 = ZIP_LANE_TAIL; S = (Lb - S0 + 62u) / 63u; }
    const u32 nl = Lb <= S0 ? 1u : 1u + (Lb - S0 + S - 1u) / S;
    const u32 rstart = lane == 0u ? b0 : b0 + S0 + (lane - 1u) * S;
    u32 rend = rstart + (lane == 0u ? S0 : S); if (rend > e0) rend = e0;
    u32 entry = lane == 0u ? b0 : (rend > rstart + ZIP_LANE_TAIL ? rend - ZIP_LANE_TAIL : rstart);
    u32 n = 0, nb = 0, nmr = 0, exitp = entry, stop_at = 0, stop_tot = 0, ended = 0;     // ended: 1 end of block, 2 a token not taken
    bool changed = lane < nl;
    // checkpoints of the lane's walk, one per ZIP_SEG tokens: bit position | output bytes so far << 16, and matches so far
    // (a byte each).  All walking lanes take a token per step, so the capture is a wave-uniform branch every ZIP_SEG steps.
    u32 ckA1 = 0, ckA2 = 0, ckA3 = 0, ckA4 = 0, ckA5 = 0, ckA6 = 0, ckA7 = 0, ckM0 = 0, ckM1 = 0;
    for (u32 round = 0; ; ) {
      u32 p = entry, cnt = 0, cb = 0, cm = 0, sa = 0, st = 0, en = 0;
      for (u32 it = 0; ; it++) {
        const bool act = changed && p < rend;
        if (!ballot(act)) break;
        if ((it & (ZIP_SEG - 1u)) == 0u && it != 0u && it < 8u * ZIP_SEG) {
          // (a lane that has stopped keeps cnt < it: its checkpoints beyond its last token are never used)
          const u32 a = p | (cb << 16), k = it / ZIP_SEG;
          if (changed) {
            if (k == 1u) ckA1 = a; else if (k == 2u) ckA2 = a; else if (k == 3u) ckA3 = a; else if (k == 4u) ckA4 = a;
            else if (k == 5u) ckA5 = a; else if (k == 6u) ckA6 = a; else ckA7 = a;
            if (k <= 4u) ckM0 = (ckM0 & ~(0xFFu << (8u * (k - 1u)))) | (cm << (8u * (k - 1u)));
            else ckM1 = (ckM1 & ~(0xFFu << (8u * (k - 5u)))) | (cm << (8u * (k - 5u)));
          }
        }
        ZIP_STAGE_R(act ? p : 0u, r)
        const ZipTok t = zip_spec_token(sh, lit_fov, llim, r, dlim, dist_fov);
        const bool die = act && t.unk, eob = act && !t.unk && t.kind == 2u, ok = act && !die && !eob;
        if (die || eob) { en = die ? 2u : 1u; sa = p; st = t.tot; }
        cnt += ok ? 1u : 0u; cb += ok ? t.olen : 0u; cm += (ok && t.kind == 1u) ? 1u : 0u;
        p = (die || eob) ? rend : p + (ok ? t.tot : 0u);
      }
      if (changed) { n = cnt; nb = cb; nmr = cm; exitp = p; ended = en; stop_at = sa; stop_tot = st; }
      round++;
      const u32 pe = (u32) __builtin_amdgcn_ds_bpermute((int)(((lane - 1u) & 63u) << 2), (int) exitp);
      const u32 ne = lane == 0u ? b0 : pe;
      changed = lane < nl && ne != entry;
      entry = ne;
      if (!ballot(changed) || round >= ZIP_LANE_ROUNDS) break;
    }
    // ---- the consistent prefix: lanes < mm ----
    u32 m = nl;
    { const u64 chm = ballot(changed); if (chm) m = (u32) __ffsll((long long) chm) - 1u; }
    u32 mm = m, dl = 0;
    bool hit = false;
    { const u64 dm = ballot(ended != 0u && lane < m); if (dm) { dl = (u32) __ffsll((long long) dm) - 1u; mm = dl + 1u; hit = true; } }
    if (mm == 0u) return -1;
    const u32 cvn = lane < mm ? n : 0u, cvb = lane < mm ? nb : 0u, cvm = lane < mm ? nmr : 0u;
    const u32 inclb = wave_incl_scan(cvb), inclm = wave_incl_scan(cvm);
    const u32 tot_m = rdl(inclm, 63u), tot_b = rdl(inclb, 63u);
    if (outc + tot_b > ZIP_FRAME || !W.ensure(tt + tot_m, lane)) return -1;
    // ---- last walk, BALANCED (lzx_parse_emit's scheme): the pass's tokens are cut into segments of ZIP_SEG tokens (the
    // lanes' checkpoints) and segment r * 64 + l goes to lane l in round r -- every lane decodes the same number of tokens
    // per round, and a round's 64 segments are NEIGHBOURS in the output and in the record list.  Literals go into an LDS
    // ring and leave as whole 16-byte rows behind the round (the bytes of the matches in between are whatever the ring
    // held: they are not final before the folder's wave has copied the block's matches, zip_run_tokens).
    {
      u32 segc = lane < mm ? (cvn + ZIP_SEG - 1u) / ZIP_SEG : 0u;
      if (segc > 8u) segc = 8u;                                     // (a stretch of more than 8 segments: the last one is long)
      const u32 seginc = wave_incl_scan(segc);
      const u32 T = rdl(seginc, 63u);
      u8 *const owner = sh->owner;
      for (u32 q = 0; q < segc; q++) owner[seginc - segc + q] = (u8) lane;
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      const u32 info0 = entry | (cvn << 16), info1 = outc + inclb - cvb, info2 = tt + inclm - cvm, info3 = (seginc - segc) | (segc << 16);
      for (u32 r = 0; r * 64u < T; r++) {
        const u32 sg = r * 64u + lane;
        const bool sact = sg < T;
        const u32 o = sact ? (u32) owner[sg] : 0u, oa = o << 2;
        const u32 i0_ = (u32) __builtin_amdgcn_ds_bpermute((int) oa, (int) info0), i1_ = (u32) __builtin_amdgcn_ds_bpermute((int) oa, (int) info1);
        const u32 i2_ = (u32) __builtin_amdgcn_ds_bpermute((int) oa, (int) info2), i3_ = (u32) __builtin_amdgcn_ds_bpermute((int) oa, (int) info3);
        const u32 k = sg - (i3_ & 0xFFFFu), osegc = i3_ >> 16, on_ = i0_ >> 16;
        const u32 a1 = (u32) __builtin_amdgcn_ds_bpermute((int) oa, (int) ckA1), a2 = (u32) __builtin_amdgcn_ds_bpermute((int) oa, (int) ckA2);
        const u32 a3 = (u32) __builtin_amdgcn_ds_bpermute((int) oa, (int) ckA3), a4 = (u32) __builtin_amdgcn_ds_bpermute((int) oa, (int) ckA4);
        const u32 a5 = (u32) __builtin_amdgcn_ds_bpermute((int) oa, (int) ckA5), a6 = (u32) __builtin_amdgcn_ds_bpermute((int) oa, (int) ckA6);
        const u32 a7 = (u32) __builtin_amdgcn_ds_bpermute((int) oa, (int) ckA7);
        const u32 m0_ = (u32) __builtin_amdgcn_ds_bpermute((int) oa, (int) ckM0), m1_ = (u32) __builtin_amdgcn_ds_bpermute((int) oa, (int) ckM1);
        const u32 ca = k == 0u ? (i0_ & 0xFFFFu) : (k == 1u ? a1 : (k == 2u ? a2 : (k == 3u ? a3 : (k == 4u ? a4 : (k == 5u ? a5 : (k == 6u ? a6 : a7))))));
        const u32 cmk = k == 0u ? 0u : (k <= 4u ? (m0_ >> (8u * (k - 1u))) & 0xFFu : (m1_ >> (8u * (k - 5u))) & 0xFFu);
        const u32 ntok = sact ? (k + 1u == osegc ? on_ - k * ZIP_SEG : ZIP_SEG) : 0u;
        u32 p = ca & 0xFFFFu, i = 0, pos = i1_ + (k == 0u ? 0u : ca >> 16), j = i2_ + cmk;
        for (;;) {
          const bool on = i < ntok;
          if (!ballot(on)) break;
          ZIP_STAGE_R(on ? p : 0u, rr)
          const ZipTok t = zip_spec_token(sh, lit_fov, llim, rr, dlim, dist_fov);
          if (on && t.kind == 0u) {
            // (inside the ring's window: into LDS; a literal below the first whole row or beyond the window -- long matches
            // between the segments -- goes out on its own)
            if (pos >= lit_flushed && pos - lit_flushed < ZIP_LIT_RING) ((u8 *) sh->litring)[pos & (ZIP_LIT_RING - 1u)] = (u8) t.sym;
            else gst_stream(fout + pos, (u8) t.sym);
          }
          if (on && t.kind == 1u) gst_stream(W.at(j), make_uint2(pos, (t.dist << 9) | t.olen));
          p += on ? t.tot : 0u; i += on ? 1u : 0u;
          pos += on ? t.olen : 0u; j += (on && t.kind == 1u) ? 1u : 0u;
        }
        {
          // rows up to the last complete one; the rest waits for the next round.  A round that outran the ring stored its
          // far literals itself: the rows behind the window are skipped for good.
          u32 rmax = rdl(wave_incl_max(sact ? pos : 0u), 63u);
          if (rmax > ZIP_FRAME) rmax = ZIP_FRAME;
          if (rmax > lit_flushed) {
            const bool outran = rmax - lit_flushed > ZIP_LIT_RING;
            const u32 upto = outran ? (rmax + 15u) & ~15u : rmax & ~15u;
            u32 lim = upto; if (outran) lim = lit_flushed + ZIP_LIT_RING;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            for (u32 row = lit_flushed + 16u * lane; row < lim; row += 16u * WAVE)
              gst((uint4 *)(fout + row), *(const uint4 *)((const u8 *) sh->litring + (row & (ZIP_LIT_RING - 1u))));
            if (upto > lit_flushed) lit_flushed = upto;
          }
        }
      }
    }
    tt += tot_m; outc += tot_b;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");        // the stage is rewritten by the next pass
    if (hit) {
      const u32 sa = sb_bit + rdl(stop_at, dl);
      if (rdl(ended, dl) == 1u) {                                 // the end-of-block symbol: not a token, ends the parse
        // ENSURE_BITS(16) in front of it (mszipd.c:229 via readhuff.h:44) is the farthest the reference reads in this
        // block: byte (sa + 15) >> 3 of the stream must exist or be one of the bytes fabricated at the end of the input
        if (((sa + 15u) >> 3) >= d.w.in_len - d.w.origin + d.w.eofs) return -1;
        const u32 tl = rdl(stop_tot, dl);
        eob_rbl = (int)(16u + ((0u - sa) & 7u) - tl);
        bitpos = sa + tl; rc = 1;
      }
      else bitpos = sa;
      break;
    }
    bitpos = sb_bit + rdl(exitp, mm - 1u);
    if (bitpos >= bit_limit) return -1;
  }
  if (outc > lit_flushed && outc - lit_flushed <= ZIP_LIT_RING) {
    // the ring's last rows (where this call's tokens end): byte by byte behind the last complete row
    const u32 full = outc & ~15u;
    for (u32 row = lit_flushed + 16u * lane; row < full; row += 16u * WAVE)
      gst((uint4 *)(fout + row), *(const uint4 *)((const u8 *) sh->litring + (row & (ZIP_LIT_RING - 1u))));
    const u32 t0 = full > lit_flushed ? full : lit_flushed;
    if (t0 + lane < outc) gst(fout + t0 + lane, ((const u8 *) sh->litring)[(t0 + lane) & (ZIP_LIT_RING - 1u)]);
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  // hand the exact bit position to the scalar reader
  {
    const u32 wi = bitpos >> 5, ch = wi >> 6;
    d.w.cur = d.w.load_chunk(ch, lane); d.w.nxt = d.w.load_chunk(ch + 1u, lane);
    d.w.wi = wi; d.bb = 0; d.bl = 0;
    d.refill(); d.refill();
    const u32 sk = bitpos & 31u;
    if (sk) { d.bb >>= sk; d.bl -= (int) sk; }
    d.rbl = rc ? eob_rbl : (int)((0u - bitpos) & 7u);
  }
  return rc;
}

// Hand-off of a block record between waves of ONE launch (mspack_mszip_pipe): what the parse wave stored (literals in the
// output, match records, record fields) is published with an agent-scope release, a drained store queue and a relaxed
// status store; the folder's wave polls the status relaxed and then takes one agent-scope acquire (the recipe of the
// LZX path, lzx_kernel.hpp).  Status: 0 = not yet, 1 = the whole CFDATA block was parsed, 2 = the parse wave gave up.
__device__ __forceinline__ u32 zip_status_load(const u32 *p) {
  return rfl(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void zip_status_publish(u32 *p, const u32 v, const u32 lane) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#ifndef MSPACK_WAVE_EMU
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  if (lane == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// one parse wave: CFDATA block `b` of unit u; true = the record is complete
__device__ __forceinline__ bool zip_parse_block_body(const mspack_hip_unit &u, const u32 b, const u8 *in_arena, u8 *out_arena, ZipBlockRec *rec,
                                const RecPool &pool, MszipShared *sh)
{
  RecWriter W;
  W.begin(pool, sh->ctab, rec->chunk);
  u8 *const fout = out_arena + u.out_off + (size_t) b * ZIP_FRAME;      // (inside the unit's region: b < ceil(out_len / 32768), 32 KiB of slack)
  const u32 lane = threadIdx.x;
  ZipDec d;
  d.lane = lane; d.sh = sh;
  d.w.unit = in_arena + u.in_off; d.w.in_len = u.in_len;
  d.w.eofs = (u.flags & MSPACK_HIP_UF_HARD_EOF) ? 0u : 2u;
  d.out = nullptr; d.B = 0; d.wpos = 0; d.flushed = false; d.hist_n = 0; d.lit_buf = 0; d.lit_n = 0;
  d.snap_iptr = 0; d.snap_rbl = 0;
  const u32 *ftab = (const u32 *)(in_arena + (size_t) u.in_chunk * 4u);
  const u32 fo = rfl(ftab[b]);
  if (fo >= u.in_len || u.in_len - fo <= 64u) return false;       // the last bytes of the input belong to the EOF-exact reader
  d.restart(fo);
  u32 v, last_block, type;
  if (!d.read_bits(8, v) || v != 'C' || !d.read_bits(8, v) || v != 'K') return false;
  const u32 start_bit = d.w.origin * 8u + d.cons_bits();
  u32 tt = 0, outc = 0;
  do {
    if (!d.read_bits(1, last_block) || !d.read_bits(2, type)) return false;
    if (type == 1u) {
      for (u32 k = lane; k < 288u; k += WAVE) sh->lit_len[k] = (u8)(k < 144u ? 8 : (k < 256u ? 9 : (k < 280u ? 7 : 8)));
      if (lane < 32u) sh->dist_len[lane] = 5;
    }
    else if (type == 2u) { if (zip_read_dynamic(d)) return false; }
    else return false;                                             // stored (or invalid): the folder's wave does it
    if (huff_build<ZIP_LIT_P>(sh->lit_len, 288, 9, sh->lit_tab, sh->lit_sorted, sh->cnt, d.hr_lit, lane, true)) return false;
    if (huff_build<ZIP_DIST_P>(sh->dist_len, 32, 6, sh->dist_tab, sh->dist_sorted, sh->cnt, d.hr_dist, lane, true)) return false;
    for (;;) {
      const int rc = zip_parse_lanes(d, fout, W, tt, outc);
      if (rc < 0) return false;
      if (rc == 1) break;
      // a token the lane-parallel decoder does not take (a long distance code, ...): one scalar token (mszipd.c:228-303)
      if (d.bl <= 32) d.refill();
      const u32 st = d.cons_bits();
      int sym = d.decode_sym<ZIP_LIT_P>(sh->lit_tab, sh->lit_sorted, d.hr_lit, true);
      if (sym < 0) return false;
      if (outc >= ZIP_FRAME || !W.ensure(tt + 1u, lane)) return false;
      if (sym < 256) { if (lane == 0) fout[outc] = (u8) sym; outc++; continue; }
      if (sym == 256) {                                      // bits_left behind it: ENSURE_BITS(16) at its first bit, minus its length
        d.rbl = (int)(16u + ((0u - st) & 7u) - (d.cons_bits() - st));
        break;
      }
      u32 code = (u32) sym - 257u, lbase, lextra, dbase, dextra, ev;
      if (code >= 29u) return false;
      zip_len_code(code, lbase, lextra);
      if (!d.read_bits((int) lextra, ev)) return false;
      const u32 length = lbase + ev;
      if (d.bl <= 32) d.refill();
      int ds = d.decode_sym<ZIP_DIST_P>(sh->dist_tab, sh->dist_sorted, d.hr_dist, true);
      if (ds < 0 || ds >= 30) return false;
      zip_dist_code((u32) ds, dbase, dextra);
      if (!d.read_bits((int) dextra, ev)) return false;
      if (outc + length > ZIP_FRAME) return false;
      if (lane == 0) *W.at(tt) = make_uint2(outc, ((dbase + ev) << 9) | length);
      tt++; outc += length;
    }
  } while (!last_block);
  if (lane == 0) {
    rec->n_tokens = tt; rec->start_bit = start_bit; rec->end_bit = d.w.origin * 8u + d.cons_bits();
    rec->eob_rbl = (u32) d.rbl; rec->total_out = outc;
  }
  return true;
}
__device__ void zip_parse_block(const mspack_hip_unit &u, const u32 b, const u8 *in_arena, u8 *out_arena, ZipBlockRec *rec,
                                const RecPool &pool, MszipShared *sh)
{
  const bool ok = zip_parse_block_body(u, b, in_arena, out_arena, rec, pool, sh);
  zip_status_publish(&rec->status, ok ? 1u : 2u, threadIdx.x);
}

// commit a CFDATA block a parse wave has taken apart: its literals are in the output already, its matches are records
// (position in the block, distance << 9 | length) in position order; they are queued and resolved in position space
// (spec_queue.hpp).  false: a match needs bytes this path cannot serve (history that is not a full block right
// below) or a record is not what a parse wave writes: the caller decodes the block the serial way.
__device__ __forceinline__ bool zip_run_tokens(ZipDec &d, const uint2 *pool_base, const u32 *chunk, const u32 n_tok_, const u32 total_)
{
  MszipShared *sh = d.sh;
  const u32 lane = d.lane;
  u8 *const out = d.out;
  const u32 B = rfl(d.B), n_tok = rfl(n_tok_), total = rfl(total_);
  const bool lin_hist = d.hist_n > 0u && rfl(sh->hist_len[0]) == ZIP_FRAME && rfl(sh->hist_B[0]) + ZIP_FRAME == B;
  if (n_tok > ZIP_TOK_CAP || total > ZIP_FRAME) return false;
  u32 P = B;
  SpecQueue Q;
  spq_init(sh->spq, Q, P, lane);
  bool ok = true;
  // Records come from memory four batches (256 records) at a time: the next four loads are issued before the current
  // four batches are queued, and the registers change hands once per four batches -- a load is only waited for
  // long after it was issued
  uint2 cur0 = make_uint2(0u, 0u), cur1 = cur0, cur2 = cur0, cur3 = cur0;
  if (n_tok) {
    const uint2 *g0 = rec_group(pool_base, chunk, 0u);            // (groups of four batches: one chunk lookup per 256 records)
    if (lane < n_tok) cur0 = gld(g0 + lane);
    if (64u + lane < n_tok) cur1 = gld(g0 + 64u + lane);
    if (128u + lane < n_tok) cur2 = gld(g0 + 128u + lane);
    if (192u + lane < n_tok) cur3 = gld(g0 + 192u + lane);
  }
  bool done = false;
  u32 th = 0;
  while (!done && th < n_tok) {
    uint2 nx0 = make_uint2(0u, 0u), nx1 = nx0, nx2 = nx0, nx3 = nx0;
    const u32 tb = th + 256u + lane;
    if (th + 256u < n_tok) {
      const uint2 *g1 = rec_group(pool_base, chunk, th + 256u);
      if (tb < n_tok) nx0 = gld(g1 + lane);
      if (tb + 64u < n_tok) nx1 = gld(g1 + 64u + lane);
      if (tb + 128u < n_tok) nx2 = gld(g1 + 128u + lane);
      if (tb + 192u < n_tok) nx3 = gld(g1 + 192u + lane);
    }
#pragma unroll 1
    for (u32 k = 0; k < 4u; k++, th += 64u) {
      if (th >= n_tok) { done = true; break; }
      const u32 n = n_tok - th < 64u ? n_tok - th : 64u;
      const uint2 cur = k == 0u ? cur0 : (k == 1u ? cur1 : (k == 2u ? cur2 : cur3));
      const bool valid = lane < n;
      const u32 rpos = cur.x, olen = valid ? (cur.y & 511u) : 0u, dist = cur.y >> 9;
      const u32 opos = B + rpos;
      // in order, inside the block, at or above everything queued so far; a source below the block only with the
      // previous block right below it
      const u32 prev_end = (u32) __builtin_amdgcn_ds_bpermute((int)(((lane - 1u) & 63u) << 2), (int)(opos + olen));
      const bool bad = valid && (olen < 3u || dist == 0u || rpos + olen > total || opos < (lane == 0u ? P : prev_end) ||
                                 (dist > rpos && (!lin_hist || dist > rpos + ZIP_FRAME)));
      if (ballot(bad)) { ok = false; done = true; break; }
      const u32 newP = rdl(opos + olen, n - 1u);
      // (runs -- matches in a row at one distance -- are written as periodic fills, the rest goes through the queue: spec_queue.hpp)
      spq_push_runs(sh->spq, Q, out, valid, n, opos, olen, dist, lane);
      P = newP;
      if (spq_due(Q, P)) spq_resolve(sh->spq, Q, out, P, false, lane);
    }
    cur0 = nx0; cur1 = nx1; cur2 = nx2; cur3 = nx3;
  }
  spq_resolve(sh->spq, Q, out, P, true, lane);
  return ok;
}

// ---------------------------------------------------------------------------------------------------
// zip_fold_block -- the copies of CFDATA block b of a folder as a FOLD task (fold_common.hpp; mspack_mszip_fold, shim.hip): what
// zip_run_tokens does on the folder's own wave, block after block (0.25 ms per block: one folder of ordinary data at 130 MB/s),
// done by one wave per block with only the gather of bytes that come from the block BELOW left on the folder's chain.
// Same records, same checks as zip_run_tokens; a block is only folded when every block below it is a full one that was folded
// too (its history is then the 32 KiB right below it).  The folder's wave (mszip_decode_unit) stays the judge: it takes a
// folded block only if the record was parsed from exactly its bit position and every earlier block was taken the same way;
// else it copies the block's matches itself, or decodes the block serially, as before.
// ---------------------------------------------------------------------------------------------------
#define ZIP_FOLD_DONE 1u
#define ZIP_FOLD_ENDED 2u
__device__ void zip_fold_block(const mspack_hip_unit &u, const u32 b, u8 *out_arena, ZipBlockRec *urecs, const uint2 *pool_base, FoldLds *L)
{
  // (FOLD_WAVES waves: wave 0 reads the records and talks to the other tasks, all waves do what is per byte -- fold_common.hpp)
  const u32 lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
  u8 *const out = out_arena + u.out_off;
  ZipBlockRec *rec = &urecs[b];
  const u32 B = b * ZIP_FRAME;
  if (wid == 0u) {
    const u32 st = rfl(gld(&rec->status));
    const u32 n_tok = rfl(gld(&rec->n_tokens)), total = rfl(gld(&rec->total_out));
    bool ok = st == 1u && n_tok <= ZIP_TOK_CAP && total <= ZIP_FRAME;
    // (the block below: parsed, and a full one -- checked again through the chain word, which says the same of every block below IT)
    if (ok && b != 0u) ok = rfl(gld(&urecs[b - 1u].status)) == 1u && rfl(gld(&urecs[b - 1u].total_out)) == ZIP_FRAME;
    if (ok) {
      fold_init(L, B, total, lane);
      u32 P = 0;                                                    // everything below this position of the block is covered
      uint2 cur0 = make_uint2(0u, 0u), cur1 = cur0, cur2 = cur0, cur3 = cur0;
      if (n_tok) {
        const uint2 *g0 = rec_group(pool_base, rec->chunk, 0u);
        if (lane < n_tok) cur0 = gld(g0 + lane);
        if (64u + lane < n_tok) cur1 = gld(g0 + 64u + lane);
        if (128u + lane < n_tok) cur2 = gld(g0 + 128u + lane);
        if (192u + lane < n_tok) cur3 = gld(g0 + 192u + lane);
      }
      u32 th = 0;
      while (th < n_tok && ok) {
        uint2 nx0 = make_uint2(0u, 0u), nx1 = nx0, nx2 = nx0, nx3 = nx0;
        if (th + 256u < n_tok) {
          const uint2 *g1 = rec_group(pool_base, rec->chunk, th + 256u);
          const u32 tb = th + 256u + lane;
          if (tb < n_tok) nx0 = gld(g1 + lane);
          if (tb + 64u < n_tok) nx1 = gld(g1 + 64u + lane);
          if (tb + 128u < n_tok) nx2 = gld(g1 + 128u + lane);
          if (tb + 192u < n_tok) nx3 = gld(g1 + 192u + lane);
        }
#pragma unroll 1
        for (u32 k = 0; k < 4u && th < n_tok && ok; k++) {
          const u32 n = n_tok - th < 64u ? n_tok - th : 64u;
          const uint2 cur = k == 0u ? cur0 : (k == 1u ? cur1 : (k == 2u ? cur2 : cur3));
          const bool valid = lane < n;
          const u32 rpos = cur.x, olen = valid ? (cur.y & 511u) : 0u, dist = cur.y >> 9;
          // zip_run_tokens' checks: in order, inside the block; a source below the block only within the 32 KiB right below it
          const u32 prev_end = (u32) __builtin_amdgcn_ds_bpermute((int)(((lane - 1u) & 63u) << 2), (int)(rpos + olen));
          const bool bad = valid && (olen < 3u || dist == 0u || rpos + olen > total || rpos < (lane == 0u ? P : prev_end) ||
                                     (dist > rpos && (b == 0u || dist > rpos + ZIP_FRAME)));
          if (ballot(bad)) { ok = false; break; }
          fold_fill_batch(L, B, valid, n, rpos, olen, dist, lane);
          P = rdl(rpos + olen, n - 1u);
          th += n;
        }
        cur0 = nx0; cur1 = nx1; cur2 = nx2; cur3 = nx3;
      }
    }
#if defined(MSPACK_WAVE_EMU)
    if (lane == 0 && getenv("MSPACK_EMU_FOLD_TRACE")) fprintf(stderr, "zip_fold_block: block %u: %u records, %u bytes, ok %d\n", b, n_tok, total, (int) ok);
#endif
    if (lane == 0) { L->ctl[1] = ok ? 1u : 0u; L->ctl[2] = total; }
  }
  fold_barrier();
  const bool ok = L->ctl[1] != 0u;
  const u32 total = L->ctl[2];
  u32 nl = 0;
  if (ok) {
    fold_jump_all(L, B, total, wid, lane);
    fold_write_own(L, out, B, total, wid, lane);
    // (a deflate distance reaches 32 KiB: whatever comes from below the block comes from the block right below it -- nothing to
    // gather a link early; the list of those bytes is made now, off the chain)
    nl = fold_write_early(L, out, B, total, wid, lane);
  }
  // the folder's chain: every block below folded?  (a task only waits for lower blocks of its folder: earlier tickets)
  if (wid == 0u) {
    u32 pch = ZIP_FOLD_DONE;
    if (b != 0u) {
      pch = zip_status_load(&urecs[b - 1u].fold);
      for (u32 tries = 0; pch == 0u && tries < (1u << 24); tries++) { __builtin_amdgcn_s_sleep(4); pch = zip_status_load(&urecs[b - 1u].fold); }
    }
    if (lane == 0) L->ctl[3] = pch;
  }
  fold_barrier();
  if (L->ctl[3] != ZIP_FOLD_DONE || !ok) {
    if (wid == 0u) zip_status_publish(&rec->fold, ZIP_FOLD_ENDED, lane);
    return;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  if (nl) fold_write_late(L, out, B, total, nl, 1u, wid, lane);        // (all of them come from the block right below)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");           // (every wave's stores out of the door before wave 0 says so)
#ifndef MSPACK_WAVE_EMU
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  fold_barrier();
  if (wid == 0u) zip_status_publish(&rec->fold, ZIP_FOLD_DONE, lane);
}

// inflate (mszipd.c:154-316): 0 ok, <0 format error, >0 ERR_READ.  *bytes_output as the reference.
__device__ __forceinline__ int zip_inflate(ZipDec &d, u32 &bytes_output)
{
  MszipShared *sh = d.sh;
  const u32 lane = d.lane;
  u32 last_block, type;
  do {
    if (!d.read_bits(1, last_block) || !d.read_bits(2, type)) return ERR_READ;
    if (type == 0u) {
      // stored: byte-align, LEN/NLEN, raw bytes (mszipd.c:168-207)
      d.byte_align();
      u32 pos = d.byte_pos();
      if (pos + 4u > d.w.in_len + d.w.eofs) return ERR_READ;
      u32 hb = (lane < 4u) ? d.w.byte_at(pos + lane) : 0u;
      u32 length = rdl(hb, 0) | (rdl(hb, 1) << 8), ncomp = rdl(hb, 2) | (rdl(hb, 3) << 8);
      pos += 4u;
      if (length != (~ncomp & 0xFFFFu)) { d.restart(pos); return ZIP_E_FORMAT; }
      d.flush_lits();
      while (length > 0u) {
        u32 n = length, room = ZIP_FRAME - d.wpos;
        if (n > room) n = room;
        // the reference copies what the input buffer holds, including the two fabricated bytes
        u32 avail = (pos < d.w.in_len + d.w.eofs) ? (d.w.in_len + d.w.eofs - pos) : 0u;
        if (avail == 0u) return ERR_READ;
        if (n > avail) n = avail;
        for (u32 k = lane; k < n; k += WAVE) d.out[d.B + d.wpos + k] = (u8) d.w.byte_at(pos + k);
        pos += n; d.wpos += n; length -= n;
        if (!d.wrap_if_needed()) { d.restart(pos); return ZIP_E_FORMAT; }
      }
      d.restart(pos);
    }
    else if (type == 1u || type == 2u) {
      if (type == 1u) {
        for (u32 k = lane; k < 288u; k += WAVE) sh->lit_len[k] = (u8)(k < 144u ? 8 : (k < 256u ? 9 : (k < 280u ? 7 : 8)));
        if (lane < 32u) sh->dist_len[lane] = 5;
      }
      else { d.store_bits(); int r = zip_read_dynamic(d); if (r) return r; d.store_bits(); }   // mszipd.c:223,149
      if (huff_build<ZIP_LIT_P>(sh->lit_len, 288, 9, sh->lit_tab, sh->lit_sorted, sh->cnt, d.hr_lit, lane, true)) return ZIP_E_FORMAT;
      if (huff_build<ZIP_DIST_P>(sh->dist_len, 32, 6, sh->dist_tab, sh->dist_sorted, sh->cnt, d.hr_dist, lane, true)) return ZIP_E_FORMAT;
      u32 cooldown = 0;                                // scalar tokens to take before the next speculative run
      for (;;) {
#ifndef ZIP_NO_SPEC
        if (cooldown == 0u && !d.flushed) {
          if (zip_run_spec(d)) break;                    // consumed the end-of-block symbol
          cooldown = 8u;
        }
        else if (cooldown) cooldown--;
#endif
        if (d.bl <= 32) d.refill();
        int sym = d.decode_sym<ZIP_LIT_P>(sh->lit_tab, sh->lit_sorted, d.hr_lit, true);
        if (sym < 0) return sym == -2 ? ERR_READ : ZIP_E_FORMAT;
        if (sym < 256) {
          d.lit_buf = wrl(d.lit_buf, (u32) sym, d.lit_n);
          d.lit_n++; d.wpos++;
          if (d.lit_n == WAVE) d.flush_lits();
          if (!d.wrap_if_needed()) return ZIP_E_FORMAT;
          continue;
        }
        if (sym == 256) break;
        u32 code = (u32) sym - 257u, lbase, lextra, dbase, dextra, ev;
        if (code >= 29u) return ZIP_E_FORMAT;
        zip_len_code(code, lbase, lextra);
        if (!d.read_bits((int) lextra, ev)) return ERR_READ;
        u32 length = lbase + ev;
        if (d.bl <= 32) d.refill();
        int ds = d.decode_sym<ZIP_DIST_P>(sh->dist_tab, sh->dist_sorted, d.hr_dist, true);
        if (ds < 0) return ds == -2 ? ERR_READ : ZIP_E_FORMAT;
        if (ds >= 30) return ZIP_E_FORMAT;
        zip_dist_code((u32) ds, dbase, dextra);
        if (!d.read_bits((int) dextra, ev)) return ERR_READ;
        u32 dist = dbase + ev;
        u32 mpos = ((dist > d.wpos) ? ZIP_FRAME : 0u) + d.wpos - dist;          // mszipd.c:267-268
        d.flush_lits();
        // the destination may cross the 32 KiB mark once: split there (FLUSH_IF_NEEDED semantics)
        while (length > 0u) {
          u32 n = length, room = ZIP_FRAME - d.wpos;
          if (n > room) n = room;
          d.copy_part(mpos, dist, n);
          d.wpos += n; length -= n; mpos = (mpos + n) & (ZIP_FRAME - 1u);
          if (!d.wrap_if_needed()) return ZIP_E_FORMAT;
        }
      }
    }
    else return ZIP_E_FORMAT;
  } while (!last_block);
  d.flush_lits();
  bytes_output = (d.flushed ? ZIP_FRAME : 0u) + d.wpos;
  d.ref_bo += d.wpos;
  if (d.flushed && d.wpos) return ZIP_E_FORMAT;                                   // mszipd.c:309-311,326-331
  return 0;
}

// recs / toks: the parse waves' records and tokens for this launch (NULL: none), indexed by frame slot
__device__ __forceinline__ void mszip_decode_unit(const mspack_hip_unit &u, const u8 *in_arena, u8 *out_arena,
                                  mspack_hip_result *res, MszipShared *sh, const ZipBlockRec *recs, const uint2 *toks)
{
  const u32 lane = threadIdx.x;
  ZipDec d;
  d.lane = lane; d.sh = sh;
  d.w.unit = in_arena + u.in_off; d.w.in_len = u.in_len;
  d.w.eofs = (u.flags & MSPACK_HIP_UF_HARD_EOF) ? 0u : 2u;
  d.w.seek(0, lane);
  d.bb = 0; d.bl = 0; d.rbl = 0;
  // MSZIP has long stretches without an ENSURE_BITS(16) (dynamic headers, the CK scan), so the
  // reference's bits_left is simply tracked from the first bit on (a few scalar ops per event)
  d.near_end = true; d.careful = true;
  d.out = out_arena + u.out_off; d.B = 0; d.wpos = 0; d.flushed = false; d.hist_n = 0;
  d.lit_buf = 0; d.lit_n = 0;
  const bool repair = (u.flags & MSPACK_HIP_UF_MSZIP_REPAIR) != 0u;
  const bool kwaj = (u.flags & MSPACK_HIP_UF_MSZIP_KWAJ) != 0u;
  u32 remaining = u.out_len, written = 0, rflags = 0;
  u32 leftover = 0;               // what the last block inflated to beyond the request (mszipd keeps it for its next call)
  int err = ERR_OK;
  u32 state0 = 0;                 // 'C','K' scanner state carried over a repair restart
  d.snap_iptr = 0; d.snap_rbl = 0;
  const bool use_recs = recs != nullptr && (u.flags & MSPACK_HIP_UF_FRAME_TABLE) != 0u && !repair && !kwaj;
  bool fold_chain = true;         // every block so far was taken as mspack_mszip_fold left it (zip_fold_block)
  u32 blk = 0;                    // CFDATA blocks started so far (the frame slot of the next one)
  const u32 nblk = (u.out_len + ZIP_FRAME - 1u) / ZIP_FRAME;
  // repair mode with MSPACK_HIP_UF_MSZIP_LOG: which blocks were repaired and how many bytes each lost -- what the reference
  // tells sys->message (mszipd.c:427) -- for the driver: behind the unit's slack, u32 count, then (output offset, bytes lost)
  const bool logging = repair && (u.flags & MSPACK_HIP_UF_MSZIP_LOG) != 0u;
  u32 *const rlog = (u32 *)(out_arena + u.out_off + (((size_t) u.out_len + ZIP_FRAME + 15u) & ~(size_t) 15u));
  const u32 rlog_cap = logging ? (u32) u.e8_base : 0u;
  u32 n_repaired = 0;

  while (remaining > 0u || kwaj) {
    d.byte_align();
    u32 state = state0, v;
    bool rd_ok = true;
    state0 = 0;
    if (kwaj) {
      // mszipd_decompress_kwaj (mszipd.c:462-495): block length (only 0 matters: end of stream), then
      // exactly 'C','K'
      u32 lo, hi;
      if (!d.read_bits(8, lo) || !d.read_bits(8, hi)) { err = ERR_READ; break; }
      if ((lo | (hi << 8)) == 0u) break;
      if (!d.read_bits(8, v)) { err = ERR_READ; break; }
      if (v != 'C') { err = ERR_DATAFORMAT; break; }
      if (!d.read_bits(8, v)) { err = ERR_READ; break; }
      if (v != 'K') { err = ERR_DATAFORMAT; break; }
      if (remaining < ZIP_FRAME) { rflags |= MSPACK_HIP_F_OUT_FULL; break; }   // a block needs 32 KiB of room
    }
    else {
      // skip to the next 'C','K' (mszipd.c:406-414)
      while (state != 2u) {
        if (!d.read_bits(8, v)) { rd_ok = false; break; }
        if (v == 'C') state = 1;
        else if (state == 1u && v == 'K') state = 2;
        else state = 0;
      }
      if (!rd_ok) { err = ERR_READ; break; }
    }

    d.wpos = 0; d.flushed = false; d.ref_bo = 0;
    d.store_bits();                                                              // mszipd.c:419
    u32 bytes_output = 0;
    int r = 0;
    bool adopted = false;
    if (use_recs && blk < nblk) {
      // a parse wave's record for this block?  adopt it if it was parsed from exactly this bit position
      const ZipBlockRec *rc_ = &recs[u.frame_base + blk];
      const u32 st_ = rfl(rc_->status);       // (the parse kernel ran before this one)
      // (and only where the parse wave put the literals: every earlier block of the folder a full one)
      // (its matches may have been copied already, as a fold task: only usable when every block before it was taken that way)
      const bool at = st_ == 1u && rfl(rc_->start_bit) == d.w.origin * 8u + d.cons_bits() && d.B == blk * ZIP_FRAME;
      const bool folded = at && fold_chain && rfl(rc_->fold) == ZIP_FOLD_DONE;
      if (!folded) fold_chain = false;
      if (at && (folded || zip_run_tokens(d, toks, rc_->chunk, rc_->n_tokens, rc_->total_out))) {
        const u32 eb = rfl(rc_->end_bit), total = rfl(rc_->total_out);
        d.restart(eb >> 3);
        { const u32 sk = eb & 7u; if (sk) { d.need((int) sk); d.bb >>= sk; d.bl -= (int) sk; } }
        d.careful = true; d.rbl = (int) rfl(rc_->eob_rbl);
        d.flushed = (total == ZIP_FRAME); d.wpos = d.flushed ? 0u : total;
        bytes_output = total;
        adopted = true;
        rflags |= MSPACK_HIP_F_FRAMES_ADOPTED;
      }
    }
    else fold_chain = false;
    blk++;
    if (!adopted) r = zip_inflate(d, bytes_output);
    if (r) {
      d.flush_lits();
      if (repair) {                                                              // mszipd.c:422-433
        u32 bo = d.flushed ? ZIP_FRAME : 0u;
        if (bo == 0u && d.wpos > 0u) bo = d.wpos;
        for (u32 k = bo + lane; k < ZIP_FRAME; k += WAVE) d.out[d.B + k] = 0;
        bytes_output = ZIP_FRAME;
        if (logging) {
          // (the reference's count: its bytes_output includes a flush that overflowed the frame -- the number can wrap)
          const u32 rbo = (d.ref_bo == 0u && d.wpos > 0u) ? d.wpos : d.ref_bo;
          if (lane == 0 && n_repaired < rlog_cap) { rlog[1u + 2u * n_repaired] = d.B; rlog[2u + 2u * n_repaired] = ZIP_FRAME - rbo; }
          n_repaired++;
        }
        if (r < 0) {
          // The next block starts from the reference's STRUCT state (RESTORE_BITS, mszipd.c:404): bit
          // buffer and bits_left as of the last STORE_BITS (mszipd.c:419,223,149), but i_ptr rewound to
          // the start of the input chunk when the feeder refilled since then (read_input stores
          // i_ptr/i_end in the struct, readbits.h:184-214).  Chunks are `bufsz` bytes of the folder stream
          // and the last one is whatever is left, so refills happen when byte k*bufsz or byte in_len is
          // first asked for.
          const u32 bufsz = u.in_chunk ? u.in_chunk : 4096u;
          const u32 e = d.abs_iptr(), s = d.snap_iptr, in_len = d.w.in_len;
          u32 R = s;
          if (e > s) {
            const u32 m = (e - 1u < in_len) ? e - 1u : in_len;
            const u32 cand = (in_len <= e - 1u) ? in_len : (m / bufsz) * bufsz;
            if (cand > s) R = cand;
          }
          // whole bytes of the stale bit buffer go through the 'C','K' scanner first, then the stream from R
          const u32 nb = (u32) d.snap_rbl >> 3;
          for (u32 k = 0; k < nb; k++) {
            const u32 c = rfl(d.w.byte_at(s - nb + k));
            if (c == 'C') state0 = 1;
            else if (state0 == 1u && c == 'K') state0 = 2;
            else state0 = 0;
          }
          d.restart(R);
        }
      }
      else { err = (r > 0) ? r : ERR_DECRUNCH; break; }
    }
    u32 n = remaining < bytes_output ? remaining : bytes_output;
    written += n;
    if (r > 0 && repair) { err = r; break; }                                     // mszipd.c:449
    remaining -= n;
    leftover = bytes_output - n;
    // remember this block for later blocks' history reads; entries it shadows (not longer than
    // it) are dropped, so the stack stays strictly increasing in length towards older blocks
    if (bytes_output) {
      u32 nh = 0, tl[ZIP_HIST - 1], tb[ZIP_HIST - 1];
#pragma unroll
      for (u32 k = 0; k < ZIP_HIST - 1u; k++) { tl[k] = 0; tb[k] = 0; }
#pragma unroll
      for (u32 k = 0; k < ZIP_HIST; k++) {
        if (k < d.hist_n) {
          u32 hl = rfl(sh->hist_len[k]), hb = rfl(sh->hist_B[k]);
          if (hl > bytes_output && nh < ZIP_HIST - 1u) {
#pragma unroll
            for (u32 q = 0; q < ZIP_HIST - 1u; q++) if (q == nh) { tl[q] = hl; tb[q] = hb; }
            nh++;
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      if (lane == 0) {
        sh->hist_len[0] = bytes_output; sh->hist_B[0] = d.B;
#pragma unroll
        for (u32 k = 0; k < ZIP_HIST - 1u; k++) { sh->hist_len[k + 1u] = tl[k]; sh->hist_B[k + 1u] = tb[k]; }
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      d.hist_n = 1u + nh;
    }
    d.B += n;
  }
  if (logging && lane == 0) rlog[0] = n_repaired;
  if (lane == 0) {
    // in_next: a block is as long as its deflate stream, whatever the CFDATA header said (mszipd.c:386-392, 440-452): the bytes the
    // last block produced beyond out_len lie in the unit's slack, and this says how many
    res->err = err; res->flags = rflags; res->out_len = written; res->good_len = written; res->in_next = (err == ERR_OK && !kwaj) ? leftover : 0u;
    res->in_used = d.w.origin + ((d.cons_bits() + (d.careful ? (u32) d.rbl : 0u)) >> 3);
  }
}
