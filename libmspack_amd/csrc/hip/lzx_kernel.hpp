// lzx_kernel.hpp -- LZX unit decoder: one wavefront per CAB folder / CHM reset interval.
//
// Replaces, for one unit, lzxd_init + lzxd_decompress(out_len) of the reference
// (libmspack/mspack/lzxd.c:274-346, 388-771) with bit-exact output and error code:
//   bit reader ........ readbits.h:133-166 + lzxd.c:85-91 -> 64-bit SGPR bit buffer, refilled 32 bits
//                       at a time by v_readlane from the lane-resident input chunk
//   READ_HUFFSYM ...... readhuff.h:39-66 -> one LDS lookup (10/8/7/6 direct bits); long codes by a
//                       wave-wide limit compare + ballot (wave_common.hpp)
//   make_decode_table . readhuff.h:83-176 -> lane-parallel build, same accept/reject set
//   lzxd_read_lens .... lzxd.c:138-183 -> lzx_read_lens_spec: pretree tokens decoded 64 bit positions at a
//                       time, run lengths prefix-summed into indices (each length is a delta on the
//                       previous block's value, so the lengths array itself stays in LDS)
//   main decode loop .. lzxd.c:538-651 -> lzx_run_spec: speculative parse of 64 bit positions per round +
//                       commit of 64 tokens at a time (positions, literals, R0-R2 by prefix scan, checks),
//                       match copies deferred to spec_queue.hpp.  The EOF-exact scalar loop (last bytes of
//                       the input, DELTA's extended lengths) gathers literals 64 at a time into one
//                       coalesced store and copies a match with one coalesced 64-lane load/store per 64
//                       bytes, the overlapping case (offset < length) served from the periodic source
//   frame logic ....... lzxd.c:419-466, 677-697, 749-754
//   E8 translation .... lzxd.c:706-736 -> NOT done here: the window must keep untranslated bytes
//                       and our window IS the output, so the decode kernel only records, per frame,
//                       the intel_filesize to apply; a second, frame-parallel kernel translates.
// The output buffer doubles as the LZ77 window ("linear window"): src = pos - offset, valid because
// frames never straddle the window wrap (lzxd.c:655-656); positions modulo window_size are kept
// only for the reference's error checks (lzxd.c:613-634).
//
// The header is compiled twice by shim.hip, each time inside its own namespace: once as is (LZX of
// CAB folders and CHM sections) and once with LZX_DELTA defined -- LZX DELTA of OAB files
// (lzxd.c:288-293, 348-382, 440-444, 588-611): windows 2^17..2^25 (main alphabets of up to 2576
// symbols: the main-tree table entries are 32 bits wide with a 12-bit symbol field), a 16-bit
// chunk size in front of every frame, match lengths extended beyond 257, and reference data that
// sits in the output arena right below the unit's output (positions are then biased by its size,
// so a source inside the reference data is an ordinary linear copy).
#include "wave_common.hpp"
#include "spec_queue.hpp"

#define LZX_FRAME 32768u
#undef LZX_MAIN_P
#ifdef LZX_PARSE_ONLY
/* third compilation (namespace lzxp, shim.hip): the parse waves of the frame-parallel path.  They need no match
 * queue and no token queue, and a main-tree table of 8 direct bits (codes beyond it are resolved lane-parallel
 * anyway): 5.5 KiB of LDS instead of 9.75, i.e. 7 waves per SIMD instead of 4 -- parse throughput is a matter of
 * how many serial chains a SIMD can interleave */
#ifndef LZX_PARSE_MAIN_P
#define LZX_PARSE_MAIN_P 8
#endif
#define LZX_MAIN_P LZX_PARSE_MAIN_P
#ifndef LZX_STAGE_WORDS
#define LZX_STAGE_WORDS 2048u   /* the parse wave's LDS stage: 8 KiB of the frame's input per pass */
#endif
#else
#define LZX_MAIN_P 10
#endif
#define LZX_LEN_P 9
#define LZX_ALI_P 7
#define LZX_PRE_P 6
#undef LZX_MAIN_SYMS
#undef LZX_MSH
#undef LZX_MTAB_T
#ifdef LZX_DELTA
#define LZX_MAIN_SYMS 2640     /* 256 + 290*8 + 64 (w<=25) */
#define LZX_MSH 12             /* main-tree table entries: symbol | length << 12, in 32 bits */
#define LZX_MTAB_T u32
#else
#define LZX_MAIN_SYMS 720      /* 256 + 50*8 + 64: every index that can ever be non-zero (w<=21) */
#define LZX_MSH 10             /* symbol | length << 10, in 16 bits */
#define LZX_MTAB_T u16
#endif
#define LZX_MMASK ((1u << LZX_MSH) - 1u)
#define LZX_LEN_SYMS 250
#ifdef LZX_EXP_STATS
#define HT0() u64 ht_ = __builtin_amdgcn_s_memtime()
#define HT(k) do { u64 n_ = __builtin_amdgcn_s_memtime(); d.st_h[k] += (u32)(n_ - ht_); ht_ = n_; } while (0)
#else
#define HT0() do { } while (0)
#define HT(k) do { } while (0)
#endif
#ifdef LZX_MARKS      /* analysis builds: region markers in the assembly (tools/count_isa.py) */
#define LZX_MARK(name) asm volatile("; MARK " name)
#else
#define LZX_MARK(name) do { } while (0)
#endif

struct __align__(16) LzxShared {
  LZX_MTAB_T main_tab[1 << LZX_MAIN_P];
  u16 main_sorted[LZX_MAIN_SYMS];
  u16 len_tab[1 << LZX_LEN_P];
  u16 len_sorted[256];
  u16 ali_tab[1 << LZX_ALI_P];
  u16 ali_sorted[8];
#if defined(LZX_DELTA) || defined(LZX_PARSE_ONLY)
  u16 pre_tab[1 << LZX_PRE_P];
  u16 pre_sorted[24];
  u32 cnt[20];
  u8  pre_len[24];
#ifndef LZX_PARSE_ONLY
  u32 inbuf[128 + 4];            /* speculative path: two 256-byte input chunks, words pre-swapped */
#else
  u32 stage[LZX_STAGE_WORDS + 64]; /* lzx_parse_emit: a stretch of a frame's input, words pre-swapped */
#ifdef LZX_LIT_RING
  alignas(16) u32 litring[LZX_LIT_RING / 4u];  /* lzx_parse_emit: the literals of the last walk's rounds on their way out (whole 16-byte rows) */
#endif
  /* what only the block header needs -- its input window and the code lengths -- shares its room with the main tree's
   * second-level table (lzx_build_sub), which is built when the header is done and the lengths are in the frame's record */
  union {
    struct {
      u32 inbuf[128 + 4];
      u8  main_len[LZX_MAIN_SYMS + 16];
      u8  len_len[LZX_LEN_SYMS + 70];
      u8  ali_len[8];
    };
    u16 sub_tab[(528 + LZX_MAIN_SYMS + 16 + LZX_LEN_SYMS + 70 + 8) / 2];
  };
#endif
#else
  /* lzx_run_spec2's token queue (start bits of parsed tokens) shares its room with what only block headers
   * use (pretree tables, the table builder's counters): a header is never decoded while tokens are queued */
  union {
    u32 tq0[256];
    struct { u16 pre_tab[1 << LZX_PRE_P]; u16 pre_sorted[24]; u32 cnt[20]; u8 pre_len[24]; };
  };
  u32 inbuf[192 + 4];            /* lzx_run_spec2: three chunks (the one behind the parse position too: queued
                                    tokens are decoded from their start bit at commit time) */
#endif
#ifndef LZX_PARSE_ONLY
  u8  main_len[LZX_MAIN_SYMS + 16];
  u8  len_len[LZX_LEN_SYMS + 70];
  u8  ali_len[8];
  SpecQueueLds spq;              /* speculative path: queued matches + start flags (spec_queue.hpp) */
#ifdef LZX_DELTA
  u32 tq0[128], tq1[128];        /* lzx_run_spec keeps whole tokens (kind/length, value) */
#else
  u32 side0[16], side1[16];      /* lzx_run_spec2 keeps start bits; the few tokens the scalar decoder took are here */
#endif
#endif
};

struct LzxDec {
  // ---- input / bit buffer (wave-uniform unless noted) ----
  InWindow w;
  u64 bb; int bl;
  bool near_end, careful; int rbl;   // reference bits_left is simulated only near end of input
  int err;
  u32 lane;
  // ---- output ----
  u8 *out; u32 P;                    // linear position == bytes decoded since unit start
  u32 lit_buf; u32 lit_n;            // lit_buf is per-lane
  u32 st_rounds, st_unknown;         // statistics (LZX_EXP_STATS builds only)
  u32 st_t[10];
  u32 st_h[3];                       // header timers: pretree, length symbols, table builds
  LzxShared *sh;
  HuffRegs hr_main, hr_len, hr_ali, hr_pre;

  __device__ __forceinline__ u32 cons_bits() const { return w.wi * 32u - (u32) bl; }
  __device__ __forceinline__ void refill() {
    u32 d = w.next_dword(lane);
    u32 x = (d << 16) | (d >> 16);               // two LE16 words, first word on top (lzxd.c:85-91)
    bb |= (u64) x << (32 - bl);
    bl += 32;
    u32 fetched = w.origin + w.wi * 4u;
    if (fetched >= w.in_len || w.in_len - fetched <= 64u) near_end = true;
  }
  __device__ __forceinline__ void need(int n) { if (bl < n) refill(); }   // n <= 32
  __device__ __forceinline__ bool ref_ensure(int n) {             // ENSURE_BITS(n) of the reference, EOF-exact
    while (rbl < n) {
      u32 i = w.origin + ((cons_bits() + (u32) rbl) >> 3);   // the reference's i_ptr
      if (i + 2u > w.in_len + w.eofs) { err = ERR_READ; return false; }   // fake bytes, then ERR_READ
      rbl += 16;
    }
    return true;
  }
  __device__ __forceinline__ bool sym_ensure() {  // ENSURE_BITS(16): bits_left becomes a pure
    if (careful) return ref_ensure(16);           // function of the bit position
    if (near_end) { careful = true; rbl = 16 + (int)((0u - cons_bits()) & 15u); }
    return true;
  }
  __device__ __forceinline__ void drop(int n) { bb <<= n; bl -= n; if (careful) rbl -= n; }
  __device__ __forceinline__ bool read_bits(int n, u32 &v) {    // READ_BITS, 1 <= n <= 17
    need(n);
    if (careful && !ref_ensure(n)) return false;
    v = (u32)(bb >> (64 - n));
    drop(n);
    return true;
  }
  template <int TP, int SH = 10, typename TabT = u16>
  __device__ __forceinline__ int decode_sym(const TabT *tab, const u16 *sorted, const HuffRegs &hr) {
    if (!sym_ensure()) return -1;
    u32 e = rfl((u32) tab[(u32)(bb >> (64 - TP))]);
    if (e == 0) {
      e = huff_long<SH>(hr, sorted, (u32)(bb >> 48), lane);
      if (e == 0) { err = ERR_DECRUNCH; return -1; }
    }
    drop((int)(e >> SH));
    return (int)(e & ((1u << SH) - 1u));
  }
  // the reference's i_ptr (bytes) -- exact in careful mode, a lower bound otherwise
  __device__ __forceinline__ u32 iptr() const {
    u32 c = cons_bits();
    return w.origin + (careful ? ((c + (u32) rbl) >> 3) : (((c + 15u) & ~15u) >> 3));
  }

  __device__ __forceinline__ void flush_lits() {
    if (lit_n) {
      if (lane < lit_n) gst(out + P - lit_n + lane, (u8) lit_buf);
      lit_n = 0;
    }
  }
};

__device__ __forceinline__ u32 lzx_read_lens_spec(LzxDec &d, u8 *lens, u32 first, u32 last);

// lzxd_read_lens (lzxd.c:138-183).  Every length is a delta against the previous block's lens[x], but
// the tokens of one call never depend on each other: far from the end of the input they are decoded
// 64 bit positions at a time (lzx_read_lens_spec); the scalar loop below is the EOF-exact version.
__device__ __forceinline__ bool lzx_read_lens(LzxDec &d, u8 *lens, u32 first, u32 last)
{
  LzxShared *sh = d.sh;
  u32 v;
  HT0();
  for (u32 x = 0; x < 20; x++) {
    if (!d.read_bits(4, v)) return false;
    sh->pre_len[x] = (u8) v;
  }
  if (d.lane < 4u) sh->pre_len[20 + d.lane] = 0;
  if (huff_build<LZX_PRE_P>(sh->pre_len, 20, 6, sh->pre_tab, sh->pre_sorted, sh->cnt, d.hr_pre, d.lane, false)) {
    d.err = ERR_DECRUNCH; return false;                       // incl. the all-zero pretree
  }
  HT(0);
  if (!d.careful) first = lzx_read_lens_spec(d, lens, first, last);
  for (u32 x = first; x < last; ) {
    d.need(32);
    int z = d.decode_sym<LZX_PRE_P>(sh->pre_tab, sh->pre_sorted, d.hr_pre);
    if (z < 0) return false;
    if (z == 17 || z == 18) {
      u32 y;
      if (!d.read_bits(z == 17 ? 4 : 5, y)) return false;
      y += (z == 17) ? 4u : 20u;
      for (u32 i = d.lane; i < y; i += WAVE) lens[x + i] = 0;  // runs are NOT clipped (lzxd.c:159)
      x += y;
    }
    else if (z == 19) {
      u32 y;
      if (!d.read_bits(1, y)) return false;
      y += 4u;
      d.need(16);
      int z2 = d.decode_sym<LZX_PRE_P>(sh->pre_tab, sh->pre_sorted, d.hr_pre);
      if (z2 < 0) return false;
      int nv = (int) rfl((u32) lens[x]) - z2; if (nv < 0) nv += 17;
      if (d.lane < y) lens[x + d.lane] = (u8) nv;
      x += y;
    }
    else {
      int nv = (int) rfl((u32) lens[x]) - z; if (nv < 0) nv += 17;
      lens[x] = (u8) nv;
      x++;
    }
  }
  HT(1);
#ifdef LZX_TRACE
  { u32 h_ = 0; for (u32 x = first; x < last; x++) h_ = h_ * 31u + lens[x];
    if (d.lane == 0) printf("read_lens [%u,%u) hash %08x cons %u\n", first, last, h_, d.cons_bits()); }
#endif
  return true;
}

struct LzxState {
  u32 R0, R1, R2;
  u32 block_type, block_length, block_remaining;
  u32 wsize, wpos, frame_posn, frame, reset_frames, num_offsets;
  u32 offset;            // bytes written (lzx->offset)
  u32 length;            // lzx->length
  int32_t intel_filesize;
  bool header_read, intel_started, length_empty;
  bool raw_mode; u32 raw_pos;   // inside / right after an uncompressed block: input byte position
  u32 ref_size;          // LZX DELTA: bytes of reference data below position 0 (0 otherwise)
};

// a match source before the window position is legal when the stream has produced that much, or
// (DELTA) when it stays inside the reference data; never beyond the window (lzxd.c:622-634)
#define LZX_BAD_SOURCE(off_, wp_, written_, ref_, wsize_)                                       \
  ((off_) > (wp_) && ((((off_) > (written_)) && (((off_) - (wp_)) > (ref_))) || (((off_) - (wp_)) > (wsize_))))

__device__ __forceinline__ void lzx_reset_state(LzxDec &d, LzxState &s) {      // lzxd.c:257-270
  s.R0 = s.R1 = s.R2 = 1;
  s.header_read = false; s.block_remaining = 0; s.block_type = 0;
  for (u32 i = d.lane; i < LZX_MAIN_SYMS + 16; i += WAVE) d.sh->main_len[i] = 0;
  // NB the reference clears exactly MAXSYMBOLS entries; the safety area beyond is never cleared
  // but also never read back for the length tree (index 249 is inside MAXSYMBOLS = 250)
  for (u32 i = d.lane; i < LZX_LEN_SYMS; i += WAVE) d.sh->len_len[i] = 0;
}

// leave raw (uncompressed-block) input mode: bit reading restarts at raw_pos with an empty buffer
__device__ __forceinline__ void lzx_leave_raw(LzxDec &d, LzxState &s) {
  if (s.raw_mode) {
    d.w.seek(s.raw_pos, d.lane);
    d.bb = 0; d.bl = 0; d.rbl = 0;
    u32 fetched = s.raw_pos;
    d.near_end = (fetched >= d.w.in_len || d.w.in_len - fetched <= 64u);
    if (d.near_end) d.careful = true;      // bits_left == 0 here: a determined point
    s.raw_mode = false;
  }
}

// block header (lzxd.c:467-523); returns false on error (d.err set)
__device__ __forceinline__ bool lzx_block_header(LzxDec &d, LzxState &s, const bool tables = true)
{
  LzxShared *sh = d.sh;
  u32 v, hi, lo;
  if (s.block_type == 3u && (s.block_length & 1u)) {            // odd-sized stored block: pad byte
    // bit buffer is empty here; the byte is skipped at i_ptr (lzxd.c:469-474)
    if (s.raw_mode) {
      if (s.raw_pos >= d.w.in_len + d.w.eofs) { d.err = ERR_READ; return false; }
      s.raw_pos++;
    }
    else {
      // stored block ended earlier in raw mode and we already re-seeked: cannot happen, raw_mode
      // is only left here or at a reset (where block_type is cleared)
    }
  }
  lzx_leave_raw(d, s);
  if (!d.read_bits(3, v) || !d.read_bits(16, hi) || !d.read_bits(8, lo)) return false;
  s.block_type = v;
  s.block_remaining = s.block_length = (hi << 8) | lo;
  if (v == 2u) {
    for (u32 i = 0; i < 8; i++) { u32 t; if (!d.read_bits(3, t)) return false; sh->ali_len[i] = (u8) t; }
    if (huff_build<LZX_ALI_P>(sh->ali_len, 8, 7, sh->ali_tab, sh->ali_sorted, sh->cnt, d.hr_ali, d.lane, false)) {
      d.err = ERR_DECRUNCH; return false;
    }
  }
  if (v == 1u || v == 2u) {
    // three pretree-coded runs (lzxd.c:491-497); one inlined call site keeps the code small
    int r = 0;
    for (int part = 0; part < 3; part++) {
      u8 *lens = (part == 2) ? sh->len_len : sh->main_len;
      u32 first = (part == 1) ? 256u : 0u;
      u32 last = (part == 0) ? 256u : (part == 1 ? 256u + s.num_offsets : 249u);
      if (!lzx_read_lens(d, lens, first, last)) return false;
      HT0();
      if (part == 1 && tables) {
        if (huff_build<LZX_MAIN_P, LZX_MSH, LZX_MTAB_T>(sh->main_len, 256 + (int) s.num_offsets + 64, 12, sh->main_tab, sh->main_sorted,
                                   sh->cnt, d.hr_main, d.lane, false)) {
          d.err = ERR_DECRUNCH; return false;
        }
        if (rfl((u32) sh->main_len[0xE8]) != 0u) s.intel_started = true;
        HT(2);
      }
    }
    if (!tables) return true;              // a parse wave walking the headers in front of its own frame: lengths only
    HT0();
    r = huff_build<LZX_LEN_P>(sh->len_len, LZX_LEN_SYMS, 12, sh->len_tab, sh->len_sorted, sh->cnt,
                              d.hr_len, d.lane, false);
    HT(2);
    if (r == 1) { d.err = ERR_DECRUNCH; return false; }
    s.length_empty = (r == 2);                                   // lzxd.c:111-125
    return true;
  }
  if (v == 3u) {
    s.intel_started = true;
    // discard 1..16 bits up to the next word boundary (a whole word if already aligned),
    // lzxd.c:506-507.  After the 27 header bits the reference holds < 16 bits, so the data starts
    // at the end of the current word, or one word further when exactly aligned.
    u32 c = d.cons_bits();
    u32 data = d.w.origin + (((c + 15u) & ~15u) >> 3) + (((c & 15u) == 0u) ? 2u : 0u);
    if (d.careful) {
      if (d.rbl == 0 && !d.ref_ensure(16)) return false;
    }
    // 12 bytes R0,R1,R2 (LE32), then the raw bytes; bytes up to in_len+1 exist (two fake zeros)
    if (data + 12u > d.w.in_len + d.w.eofs) { d.err = ERR_READ; return false; }
    u32 b = (d.lane < 12u) ? d.w.byte_at(data + d.lane) : 0u;
    u32 r[3];
#pragma unroll
    for (int k = 0; k < 3; k++)
      r[k] = rdl(b, 4 * k) | (rdl(b, 4 * k + 1) << 8) | (rdl(b, 4 * k + 2) << 16) | (rdl(b, 4 * k + 3) << 24);
    s.R0 = r[0]; s.R1 = r[1]; s.R2 = r[2];
    s.raw_mode = true; s.raw_pos = data + 12u;
    d.bb = 0; d.bl = 0; d.rbl = 0;
    return true;
  }
  d.err = ERR_DECRUNCH;
  return false;
}

// copy a match of `len` bytes at distance `off` (1 <= off <= window) to the linear position P
__device__ __forceinline__ void lzx_copy_match(u8 *out, u32 P, u32 off, u32 len, u32 lane)
{
  u8 *dst = out + P;
  const u8 *src = dst - off;
  if (off >= len || off >= WAVE) {
    // every 64-byte step only reads bytes that are already complete (earlier steps/tokens)
    for (u32 i = lane; i < len; i += WAVE) gst(dst + i, gld(src + i));
  }
  else {
    // overlapping copy, period `off` < 64: lane i takes pattern byte (i mod off)
    u32 r = lane, s = off << 5;
#pragma unroll
    for (int k = 0; k < 6; k++) { u32 t = r - s; r = t < r ? t : r; s >>= 1; }     // r = lane mod off
    u32 step = 64u, ss = off << 5;
#pragma unroll
    for (int k = 0; k < 6; k++) { u32 t = step - ss; step = t < step ? t : step; ss >>= 1; }  // 64 mod off
    for (u32 i = lane; i < len; i += WAVE) {
      gst(dst + i, gld(src + r));
      r += step; if (r >= off) r -= off;
    }
  }
}

// reference-exact slow path for offsets no encoder produces (0, or beyond the window: only reachable
// through R0-R2 loaded from a stored-block header).  Byte-serial ring semantics of lzxd.c:618-646.
__device__ void lzx_copy_match_odd(u8 *out, u32 P, u32 wpos, u32 wsize, u32 off, u32 len)
{
  u32 base = P - wpos;                       // linear position of window index 0 in this pass
  for (u32 k = 0; k < len; k++) {
    u32 sidx = (wpos - off + k) & (wsize - 1u);
    if (off > wpos) { u32 j = off - wpos; sidx = (k < j) ? (wsize - j + k) : (k - j); }
    u8 b = 0;
    if (sidx < wpos + k) b = out[base + sidx];
    else if (base + sidx >= wsize) b = out[base + sidx - wsize];
    out[P + k] = b;
  }
}


// what a steady-state run (lzx_run_spec / lzx_run_spec2 below) ends with: the run is through, the generic (EOF-exact) loop of
// lzx_decode_unit takes over -- always at a token boundary, where the reference's bits_left is a pure function of the bit
// position (LzxDec::sym_ensure) --, or the stream is bad
enum { LZX_RUN_DONE = 0, LZX_RUN_SWITCH = 1, LZX_RUN_FAIL = 2 };



// ---------------------------------------------------------------------------------------------------
// Speculative window decode: the steady-state path.
//
// A Huffman/LZ bitstream is serial only because a symbol's start is known once the previous symbol's
// length is.  So all 64 lanes decode a COMPLETE token (main symbol, length footer, offset bits,
// aligned symbol) each starting at a different bit -- lane l at bit (pos + l) -- with gathered LDS
// table lookups, and publish "bits consumed / kind / length / offset" in two registers.  The true
// symbol boundaries are then followed through those registers with v_readlane: a token costs two
// readlanes and a handful of scalar ops instead of a chain of dependent LDS round trips, and the
// lookups of ~6 consecutive tokens (64 bits / ~10 bits per token) overlap.  Tokens with a code
// longer than the direct table are decoded by the scalar routine from the same 64 bits.
// Input: the two current 256-byte chunks live in LDS with each dword's 16-bit halves swapped, so the
// stream is a plain MSB-first bit string; the next chunk is prefetched in a register.
// ---------------------------------------------------------------------------------------------------
template <bool ALIGNED>
__device__ __forceinline__ u32 lzx_scalar_token(const LzxDec &d, bool length_empty, u64 r,
                                                u32 &kind, u32 &val, u32 &off)
{
  const LzxShared *sh = d.sh;
  u32 tot = 0;
  u32 e = rfl((u32) sh->main_tab[(u32)(r >> (64 - LZX_MAIN_P))]);
  if (e == 0) { e = huff_long<LZX_MSH>(d.hr_main, sh->main_sorted, (u32)(r >> 48), d.lane); if (e == 0) return 0; }
  { u32 l = e >> LZX_MSH; r <<= l; tot += l; }
  u32 sym = e & LZX_MMASK;
  if (sym < 256u) { kind = 0; val = sym; off = 0; return tot; }
  u32 m = sym - 256u, slot = m >> 3, len = (m & 7u) + 2u;
  if ((m & 7u) == 7u) {
    if (length_empty) return 0;
    u32 f = rfl((u32) sh->len_tab[(u32)(r >> (64 - LZX_LEN_P))]);
    if (f == 0) { f = huff_long(d.hr_len, sh->len_sorted, (u32)(r >> 48), d.lane); if (f == 0) return 0; }
    { u32 l = f >> 10; r <<= l; tot += l; }
    len += f & 1023u;
  }
  val = len;
  if (slot < 3u) { kind = 2u + slot; off = 0; return tot; }
  u32 extra = slot < 4u ? 0u : (slot < 36u ? (slot >> 1) - 1u : 17u);
  u32 base = slot < 4u ? slot : (slot < 36u ? ((2u + (slot & 1u)) << extra) : ((slot - 34u) << 17));
  off = base - 2u;
  if (ALIGNED && extra >= 3u) {
    u32 nb = extra - 3u;
    if (nb) { off += (u32)(r >> (64 - nb)) << 3; r <<= nb; tot += nb; }
    u32 a = rfl((u32) sh->ali_tab[(u32)(r >> (64 - LZX_ALI_P))]);
    if (a == 0) return 0;
    tot += a >> 10; off += a & 1023u;
  }
  else if (extra) { off += (u32)(r >> (64 - extra)); tot += extra; }
  kind = 1;
  return tot;
}


// one speculative token: everything lane-local, decoded from 64 bits of the stream
struct SpecTok { u32 tot, sym, kind, olen, off; bool unk;
};

template <bool ALIGNED>
__device__ __forceinline__ SpecTok lzx_spec_token(const LzxShared *sh, const u32 main_fov, const u32 *mlim,
                                                  const bool length_empty, const u32 w0, const u32 w1)
{
  SpecTok t;
  u64 r = ((u64) w0 << 32) | w1;
  u32 e = sh->main_tab[w0 >> (32 - LZX_MAIN_P)];
  {
    // codes longer than the direct table, for all lanes at once: canonical length = number of
    // per-length limits the 16-bit peek is not below; symbol via the sorted list (readhuff.h:144-172)
    u32 peek16 = w0 >> 16, ln = LZX_MAIN_P + 1u;
#pragma unroll
    for (int l = LZX_MAIN_P + 1; l <= 16; l++) ln += (peek16 >= mlim[l - LZX_MAIN_P - 1]) ? 1u : 0u;
    u32 lq = ln <= 16u ? ln : 0u;
    u32 fo = (u32) __builtin_amdgcn_ds_bpermute((int)(lq << 2), (int) main_fov);
    u32 idx = (fo >> 16) + ((peek16 >> (16u - lq)) - (fo & 0xFFFFu));
    if (idx >= LZX_MAIN_SYMS) idx = 0;
    u32 ls = sh->main_sorted[idx];
    if (e == 0u && lq != 0u) e = ls | (lq << LZX_MSH);
  }
  bool unk = (e == 0u);
  u32 tot = e >> LZX_MSH, sym = e & LZX_MMASK;
  r <<= tot;
  bool is_match = sym >= 256u;
  u32 m = sym - 256u, slot = m >> 3, lh = m & 7u;
  u32 e2 = sh->len_tab[(u32)(r >> (64 - LZX_LEN_P))];
  bool need_len = is_match && lh == 7u;
  if (need_len) { unk = unk || e2 == 0u || length_empty; u32 l2 = e2 >> 10; r <<= l2; tot += l2; }
  u32 mlen = lh + 2u + (need_len ? (e2 & 1023u) : 0u);
#ifdef LZX_DELTA
  if (is_match && mlen == 257u) unk = true;             // announces an extended length (lzxd.c:588-611)
#endif
  // position_base / extra_bits in closed form (lzxd.c:202-207), branch-free; only slots >= 3 matter here
  // (0..2 are the repeats): extra = clamp(slot/2 - 1, 0, 17), base = (slot < 36 ? 2 + (slot & 1) : slot - 34) << extra
  const int ex_ = (int)(slot >> 1) - 1;
  const u32 extra = (u32)(ex_ < 0 ? 0 : (ex_ > 17 ? 17 : ex_));
  u32 off = (((slot < 36u) ? 2u + (slot & 1u) : slot - 34u) << extra) - 2u;
  bool expl = is_match && slot >= 3u;
  if (ALIGNED) {
    bool ali = extra >= 3u;
    u32 nb = ali ? extra - 3u : extra;
    u32 vb = nb ? (u32)(r >> (64u - nb)) : 0u;
    u64 r2 = r << nb;
    u32 e3 = sh->ali_tab[(u32)(r2 >> (64 - LZX_ALI_P))];
    if (expl) {
      tot += nb;
      if (ali) { off += (vb << 3) + (e3 & 1023u); tot += e3 >> 10; unk = unk || e3 == 0u; }
      else off += vb;
    }
  }
  else {
    u32 vb = extra ? (u32)(r >> (64u - extra)) : 0u;
    if (expl) { off += vb; tot += extra; }
  }
  t.tot = tot; t.sym = sym; t.unk = unk; t.off = off;
  t.kind = !is_match ? 0u : (expl ? 1u : 2u + slot);
  t.olen = is_match ? mlen : 1u;
  return t;
}


// how many bits does the token take whose main-tree entry is e (symbol | code length << LZX_MSH, not 0)?
template <bool ALIGNED>
__device__ __forceinline__ u32 lzx_adv_from_entry(const LzxShared *sh, const bool length_empty, const u32 e,
                                                  const u32 w0, const u32 w1, bool &unk)
{
  const u32 mlen = e >> LZX_MSH, sym = e & LZX_MMASK;
  const bool is_match = sym >= 256u;
  const u32 m = sym - 256u, slot = m >> 3;
  const bool need_len = is_match && (m & 7u) == 7u;
  const u32 e2 = sh->len_tab[(w0 << mlen) >> (32 - LZX_LEN_P)];        // the footer's code starts right behind the main code
  u32 tot = mlen;
  unk = false;
  if (need_len) { unk = (e2 == 0u) || length_empty; tot += e2 >> 10; }
  const int ex_ = (int)(slot >> 1) - 1;
  const u32 extra = (u32)(ex_ < 0 ? 0 : (ex_ > 17 ? 17 : ex_));
  const bool expl = is_match && slot >= 3u;
  if (ALIGNED) {
    const bool ali = extra >= 3u;
    const u32 nb = ali ? extra - 3u : extra;
    const u64 r = ((u64) w0 << 32) | w1;
    const u32 e3 = sh->ali_tab[(u32)((r << (tot + nb)) >> (64 - LZX_ALI_P))];   // behind the verbatim bits (bit <= 46)
    if (expl) { tot += nb; if (ali) { tot += e3 >> 10; unk = unk || e3 == 0u; } }
  }
  else if (expl) tot += extra;
  return tot;
}


// ---- input staging shared by the speculative paths ----------------------------------------------
// The two current 256-byte input chunks live in LDS with each dword's 16-bit halves swapped (the
// stream becomes a plain MSB-first bit string); the next chunk is prefetched in a register.
#define SWAP16(x) (((x) << 16) | ((x) >> 16))
__device__ __forceinline__ void spec_stage(LzxDec &d, u32 &bitpos, u32 &cb, u32 &pf)
{
  LzxShared *sh = d.sh;
  const u32 lane = d.lane;
  bitpos = rfl(d.cons_bits());
  cb = bitpos >> 11;                                    // chunk cb = dwords [64cb, 64cb+64)
  u32 lo = d.w.load_chunk(cb, lane), hi = d.w.load_chunk(cb + 1u, lane);
  sh->inbuf[lane] = SWAP16(lo); sh->inbuf[64u + lane] = SWAP16(hi);
  if (lane < 4u) sh->inbuf[128u + lane] = 0;
  pf = d.w.load_chunk(cb + 2u, lane);
}
__device__ __forceinline__ void spec_slide(LzxDec &d, const u32 bitpos, u32 &cb, u32 &pf)
{
  if ((bitpos >> 11) != cb) {                           // slide the LDS window by one chunk
    LzxShared *sh = d.sh;
    const u32 lane = d.lane;
    u32 up = sh->inbuf[64u + lane];
    sh->inbuf[lane] = up; sh->inbuf[64u + lane] = SWAP16(pf);
    cb++;
    pf = d.w.load_chunk(cb + 2u, lane);
  }
}
// hand the exact bit position back to the scalar reader
__device__ __forceinline__ void spec_resync(LzxDec &d, const u32 bitpos, const u32 cb, const u32 pf)
{
  LzxShared *sh = d.sh;
  const u32 lane = d.lane;
  u32 wi = bitpos >> 5, ch = wi >> 6;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  u32 lo = sh->inbuf[lane], hi = sh->inbuf[64u + lane];
  lo = SWAP16(lo); hi = SWAP16(hi);
  if (ch == cb) { d.w.cur = lo; d.w.nxt = hi; }
  else if (ch == cb + 1u) { d.w.cur = hi; d.w.nxt = pf; }
  else { d.w.cur = d.w.load_chunk(ch, lane); d.w.nxt = d.w.load_chunk(ch + 1u, lane); }   // went back (end of a run)
  d.w.wi = wi; d.bb = 0; d.bl = 0;
  d.refill(); d.refill();
  u32 sk = bitpos & 31u;
  if (sk) { d.bb <<= sk; d.bl -= (int) sk; }
}
// last bit position (relative to the window origin) from which a speculative round may start; see
// the margin discussion in lzx_run_spec
__device__ __forceinline__ u32 spec_bit_limit(const LzxDec &d, const u32 margin)
{
  const u32 room_bytes = (d.w.in_len > d.w.origin + margin) ? (d.w.in_len - d.w.origin - margin) : 0u;
  return rfl(room_bytes * 8u);
}

// lzxd_read_lens, vector form.  Decodes pretree tokens from `first` on and returns the index the
// scalar loop has to continue from (== last when everything was done here).  A token = pretree
// symbol z [+ 4 / 5 / 1 extra bits for z = 17 / 18 / 19] [+ a second symbol for z = 19]; it sets
// y = 1 or a run of y lengths (lzxd.c:150-179).  All 64 lanes decode the token starting at bit
// (bitpos + lane); the chain of real tokens is followed with v_readlane; a prefix sum of the run
// lengths gives every token its index x; then every on-chain lane rewrites its own lens[x .. x+y).
__device__ __forceinline__ u32 lzx_read_lens_spec(LzxDec &d, u8 *lens, u32 first, u32 last_)
{
  LzxShared *sh = d.sh;
  const u32 lane = d.lane;
  const u32 last = rfl(last_);
  u32 X = rfl(first);
  const u32 bit_limit = spec_bit_limit(d, 56u);
  if (rfl(d.cons_bits()) >= bit_limit) return X;
  u32 bitpos, cb, pf;
  spec_stage(d, bitpos, cb, pf);
  u32 plim[16 - LZX_PRE_P];                             // limits of the code lengths beyond the table
#pragma unroll
  for (int l = LZX_PRE_P + 1; l <= 16; l++) plim[l - LZX_PRE_P - 1] = rdl(d.hr_pre.limv, (u32) l);

  while (X < last && bitpos < bit_limit) {
    spec_slide(d, bitpos, cb, pf);
    const u32 rel = bitpos - (cb << 11) + lane;
    const u32 k = rel >> 5, sft = rel & 31u;
    const u32 i0 = sh->inbuf[k], i1 = sh->inbuf[k + 1u], i2 = sh->inbuf[k + 2u];
    const u32 w0 = (u32)(((((u64) i0 << 32) | i1) << sft) >> 32);
    const u32 w1 = (u32)(((((u64) i1 << 32) | i2) << sft) >> 32);
    u64 r = ((u64) w0 << 32) | w1;
    u32 e = sh->pre_tab[w0 >> (32 - LZX_PRE_P)];
    {
      u32 peek16 = w0 >> 16, ln = LZX_PRE_P + 1u;
#pragma unroll
      for (int l = LZX_PRE_P + 1; l <= 16; l++) ln += (peek16 >= plim[l - LZX_PRE_P - 1]) ? 1u : 0u;
      u32 lq = ln <= 16u ? ln : 0u;
      u32 fo = (u32) __builtin_amdgcn_ds_bpermute((int)(lq << 2), (int) d.hr_pre.fov);
      u32 idx = (fo >> 16) + ((peek16 >> (16u - lq)) - (fo & 0xFFFFu));
      if (idx >= 20u) idx = 0;
      u32 ls = sh->pre_sorted[idx];
      if (e == 0u && lq != 0u) e = ls | (lq << 10);
    }
    bool unk = (e == 0u);
    const u32 z = e & 1023u;
    u32 tot = e >> 10;
    r <<= tot;
    const u32 nb = z == 17u ? 4u : (z == 18u ? 5u : (z == 19u ? 1u : 0u));
    const u32 xb = nb ? (u32)(r >> (64u - nb)) : 0u;
    r <<= nb; tot += nb;
    const u32 y = z == 17u ? 4u + xb : (z == 18u ? 20u + xb : (z == 19u ? 4u + xb : 1u));
    const u32 e2 = sh->pre_tab[(u32)(r >> (64 - LZX_PRE_P))];       // second symbol of a "same" run
    if (z == 19u) { unk = unk || e2 == 0u; tot += e2 >> 10; }       // (a long second code: scalar loop)
    const u32 zz = z == 19u ? (e2 & 1023u) : z;
    const bool zero = z == 17u || z == 18u;
    const u32 vnext = unk ? (128u + lane) : (lane + tot);

    u64 chain = 0;
    u32 q = 0;
    do { chain |= 1ull << q; q = rdl(vnext, q); } while (q < WAVE);
    bool hit_unknown = false;
    if (q >= 128u) { q -= 128u; chain &= ~(1ull << q); hit_unknown = true; }
    bool on = (chain >> lane) & 1ull;
    const u32 yy = on ? y : 0u;
    const u32 incl = wave_incl_scan(yy);
    const u32 x = X + incl - yy;
    u32 newX = X + rdl(incl, 63u);
    // lengths are read only while x < last (lzxd.c:148); a run may overshoot (it is not clipped)
    const u64 late = ballot(on && x >= last);
    if (late) {
      u32 j = (u32) __ffsll((long long) late) - 1u;
      chain &= (1ull << j) - 1ull;
      on = (chain >> lane) & 1ull;
      q = j; hit_unknown = false; newX = rdl(x, j);
    }
    if (on) {
      int nv = 0;
      if (!zero) { nv = (int)(u32) lens[x] - (int) zz; if (nv < 0) nv += 17; }
      for (u32 i = 0; i < y; i++) lens[x + i] = (u8) nv;
    }
    X = newX;
    bitpos += q;
    if (hit_unknown) break;                             // the scalar loop takes (and judges) this token
  }
  spec_resync(d, bitpos, cb, pf);
  return X;
}

// ---- R0-R2 as a prefix scan ----------------------------------------------------------------------
// What a token does to the three recent offsets (lzxd.c:565-586) is a map "new slot i <- old slot j or
// this token's own offset": an explicit offset is (own, R0, R1), a repeat of R0 changes nothing, a
// repeat of R1 / R2 swaps it with R0.  Such maps compose associatively, so the state after every token
// of a batch is an inclusive scan.  A map is three bytes, one per new slot: 0..2 = old slot, 0x80 | lane =
// the offset of the token in that lane.  v_perm_b32 composes two maps in one instruction.
#define LRU_ID 0x020100u
__device__ __forceinline__ u32 lru_compose(u32 first, u32 then)
{
  const u32 L = then & 0x808080u;                        // slots `then` fills with a token's own offset
  const u32 full = (L << 1) - (L >> 7);                  // 0xFF in those bytes
  const u32 sel = (then & ~full) | (0x060504u & full) | 0x0C000000u;
  return __builtin_amdgcn_perm(then, first, sel);        // byte i: first[then[i]] or then[i] itself
}
__device__ __forceinline__ u32 lru_scan(u32 x)
{
  u32 v = x, t;
  t = (u32) __builtin_amdgcn_update_dpp((int) LRU_ID, (int) v, 0x111, 0xf, 0xf, false); v = lru_compose(t, v);
  t = (u32) __builtin_amdgcn_update_dpp((int) LRU_ID, (int) v, 0x112, 0xf, 0xf, false); v = lru_compose(t, v);
  t = (u32) __builtin_amdgcn_update_dpp((int) LRU_ID, (int) v, 0x114, 0xf, 0xf, false); v = lru_compose(t, v);
  t = (u32) __builtin_amdgcn_update_dpp((int) LRU_ID, (int) v, 0x118, 0xf, 0xf, false); v = lru_compose(t, v);
  t = (u32) __builtin_amdgcn_update_dpp((int) LRU_ID, (int) v, 0x142, 0xa, 0xf, false); v = lru_compose(t, v);
  t = (u32) __builtin_amdgcn_update_dpp((int) LRU_ID, (int) v, 0x143, 0xc, 0xf, false); v = lru_compose(t, v);
  return v;
}

#ifndef LZX_PARSE_ONLY
// ---- COMMIT: one batch of parsed tokens, one token per lane ---------------------------------------------
// Used by the speculative runs of the serial path (tokens from the LDS queue).
struct LzxCommit {                  // wave-uniform commit-side state of a run
  u32 P, R0, R1, R2;
  u32 run_end, wbase, wsize, offset_written, ref_size;
  SpecQueue Q;
};
#define LZX_TK_BAIL 6u             /* DELTA: a match length that announces an extension (lzxd.c:588-611) */
#define LZX_TK_FAIL 7u

// c0 = kind | output length << 3 | ..., c1 = literal or explicit offset; lanes >= n are idle.  Returns the number
// of tokens taken: fewer than n at a marker (its kind in `marker`) or where the run ends (lzxd.c:538).
__device__ __forceinline__ u32 lzx_commit_batch(LzxDec &d, LzxCommit &C, const u32 c0, const u32 c1, u32 n,
                                                u32 &marker, bool &fail_after)
{
  LzxShared *sh = d.sh;
  const u32 lane = d.lane;
  u8 *const out = d.out;
  const u32 run_end = C.run_end, wbase = C.wbase, wsize = C.wsize;
  const u32 P = C.P;
  const u32 kind = c0 & 7u;
  marker = 0; fail_after = false;
  {
    const u64 mk = ballot(lane < n && kind >= LZX_TK_BAIL);
    if (mk) { const u32 jm = (u32) __ffsll((long long) mk) - 1u; marker = rdl(kind, jm); n = jm; }
  }
  const u32 olen = lane < n ? ((c0 >> 3) & 511u) : 0u;
  const u32 incl = wave_incl_scan(olen);
  const u32 opos = P + incl - olen;                   // output position of this lane's token
  u32 newP = P + rdl(incl, 63u);
  // tokens are decoded only while the run lasts (lzxd.c:538): the first one that would start at or
  // after run_end, and everything parsed behind it, is not part of this run
  if (newP >= run_end) {
    // (a literal run that would cross the end of the run is not taken either: the serial path goes on there)
    const u64 late = ballot(lane < n && (opos >= run_end || (kind == 0u && opos + olen > run_end)));
    if (late) { const u32 j = (u32) __ffsll((long long) late) - 1u; n = j; newP = rdl(opos, j); marker = 0; }
  }
  const bool valid = lane < n;
  if (valid && kind == 0u) {
    gst(out + opos, (u8) c1);
    if (olen > 1u) {                                        // a literal run: 2..4 bytes, first literal in the low byte
      gst(out + opos + 1u, (u8)(c1 >> 8));
      if (olen > 2u) gst(out + opos + 2u, (u8)(c1 >> 16));
      if (olen > 3u) gst(out + opos + 3u, (u8)(c1 >> 24));
    }
  }
  const bool ism0 = valid && kind != 0u;
  u64 mm = ballot(ism0);
  if (mm) {
    // (1) every match's offset through the R0-R2 LRU (lzxd.c:565-586)
    const u32 sR0 = C.R0, sR1 = C.R1, sR2 = C.R2;
    u32 vmoff = c1;
    const u64 k1 = ballot(ism0 && kind == 1u);
    if (!ballot(ism0 && kind >= 3u)) {
      // only explicit offsets and repeats of R0: a repeat takes the nearest explicit offset before
      // it, and the last three explicit offsets are the new R0-R2
      const u64 below = k1 & ((1ull << lane) - 1ull);
      const u32 src = below ? 63u - (u32) __clzll((long long) below) : 0u;
      const u32 pv = (u32) __builtin_amdgcn_ds_bpermute((int)(src << 2), (int) c1);
      if (kind == 2u) vmoff = below ? pv : sR0;
      if (k1) {
        u64 m = k1;
        const u32 j0 = 63u - (u32) __clzll((long long) m);
        u32 nb = sR0, nc = sR1;
        m &= ~(1ull << j0);
        if (m) {
          const u32 j1 = 63u - (u32) __clzll((long long) m);
          nb = rdl(c1, j1); nc = sR0;
          m &= ~(1ull << j1);
          if (m) nc = rdl(c1, 63u - (u32) __clzll((long long) m));
        }
        C.R0 = rdl(c1, j0); C.R1 = nb; C.R2 = nc;
      }
    }
    else {
      u32 x = LRU_ID;
      if (ism0) x = kind == 1u ? (0x010080u | lane) : (kind == 3u ? 0x020001u : (kind == 4u ? 0x000102u : LRU_ID));
      const u32 Cm = lru_scan(x);
      const u32 e0 = Cm & 0xFFu;
      const u32 pv = (u32) __builtin_amdgcn_ds_bpermute((int)((e0 & 63u) << 2), (int) c1);
      vmoff = (e0 & 0x80u) ? pv : (e0 == 0u ? sR0 : (e0 == 1u ? sR1 : sR2));
      const u32 Cl = rdl(Cm, 63u);
      const u32 f0 = Cl & 0xFFu, f1 = (Cl >> 8) & 0xFFu, f2 = (Cl >> 16) & 0xFFu;
      C.R0 = (f0 & 0x80u) ? rdl(c1, f0 & 63u) : (f0 == 0u ? sR0 : (f0 == 1u ? sR1 : sR2));
      C.R1 = (f1 & 0x80u) ? rdl(c1, f1 & 63u) : (f1 == 0u ? sR0 : (f1 == 1u ? sR1 : sR2));
      C.R2 = (f2 & 0x80u) ? rdl(c1, f2 & 63u) : (f2 == 0u ? sR0 : (f2 == 1u ? sR1 : sR2));
    }
    // (2) the reference's checks (lzxd.c:613-634, 678-693) for all matches at once
    {
      const u32 wp = opos - wbase;
      const bool bad = ism0 && (opos + olen > run_end || wp + olen > wsize ||
                                LZX_BAD_SOURCE(vmoff, wp, C.offset_written, C.ref_size, wsize));
      const u64 badm = ballot(bad);
      if (badm) { mm &= (1ull << ((u32) __ffsll((long long) badm) - 1u)) - 1ull; fail_after = true; }
    }
    // (3) queue the matches
    if (mm) {
      bool ism = lane_in(mm);
      // Offsets no linear copy can serve (0, or beyond the window: only from a stored block's R0-R2;
      // DELTA: beyond the 23 bits the queue holds) take the slow way: resolve the queue, copy this
      // batch's matches one at a time with the reference's ring semantics.
      if (ballot(ism && (vmoff == 0u || vmoff > wsize || (vmoff >> 23) != 0u))) {
        spq_resolve(sh->spq, C.Q, out, P, true, lane);
        for (u64 dm = mm; dm; dm &= dm - 1ull) {
          const u32 l = (u32) __ffsll((long long) dm) - 1u;
          const u32 pos_l = rdl(opos, l), len_l = rdl(olen, l), off_l = rdl(vmoff, l);
          if (off_l != 0u && off_l <= wsize) lzx_copy_match(out, pos_l, off_l, len_l, lane);
          else { if (lane == 0) lzx_copy_match_odd(out, pos_l, pos_l - wbase, wsize, off_l, len_l); }
        }
        C.Q.Pf = newP;
      }
      else {
        if (C.Q.mcount + (u32) __popcll(mm) > SPQ_CAP) spq_resolve(sh->spq, C.Q, out, P, true, lane);
        for (;;) {
          // a push must keep every start flag inside the ring (spec_queue.hpp): take the matches that
          // end inside it, resolve up to the first one that does not, go on
          const u32 limit = (C.Q.Pf & ~63u) + SPQ_RING;
          const u64 fit = newP <= limit ? mm : ballot(ism && opos + olen <= limit);
          if (fit) {
            const u32 rank = __builtin_amdgcn_mbcnt_hi((u32)(fit >> 32), __builtin_amdgcn_mbcnt_lo((u32) fit, 0u));
            spq_push(sh->spq, C.Q, lane_in(fit), rank, (u32) __popcll(fit), opos, vmoff, olen);
            mm &= ~fit;
            ism = lane_in(mm);
          }
          if (!mm) break;
          spq_resolve(sh->spq, C.Q, out, rdl(opos, (u32) __ffsll((long long) mm) - 1u), true, lane);
        }
      }
    }
  }
  C.P = newP;
  return n;
}

// ---------------------------------------------------------------------------------------------------
// The speculative decode of a run of tokens (lzxd.c:538-651), in two alternating phases.
//
// PARSE (per round of 64 bit positions): every lane decodes the complete token that would start at bit
// (bitpos + lane); the chain of real tokens is followed with v_readlane; the tokens on the chain are
// appended to a token queue in LDS (kind, output length, offset or literal, start bit).  Nothing else
// happens in a round: no output position is needed to find the next token, so the serial chain of the
// whole decoder is just "decode 64 positions, walk".  A token the lane-parallel decoder does not take
// (a code beyond the direct length / aligned tables, an invalid one) is decoded on the scalar side from
// the same 64 bits and queued like the others.
//
// COMMIT (per 64 queued tokens, one token per lane, all lanes busy): prefix sum of the output lengths
// -> positions; literals are stored; R0-R2 are resolved for all matches at once (lru_scan); the
// reference's checks (lzxd.c:613-634, 678-693) run for all matches at once; the matches go to the
// deferred-copy queue of spec_queue.hpp, which is resolved 64 output bytes per pass.
//
// The parser runs ahead of the committed position, so at the end of a run (lzxd.c:538: tokens are read
// only while this_run > 0) it has parsed tokens that do not belong to the run -- bits of the next block
// header read as tokens.  They are dropped and the bit position goes back to the first of them (every
// record carries the low 16 bits of its start).  For the same reason the parser never fails: what it
// cannot decode becomes a FAIL marker that only counts when the commit reaches it.
// ---------------------------------------------------------------------------------------------------
#define LZX_TQ 128u                /* token queue entries (two commits' worth) */
#ifdef LZX_DELTA

template <bool ALIGNED>
__device__ __forceinline__ int lzx_run_spec(LzxDec &d, LzxState &s, const u32 run_end_, const u32 wbase_)
{
  LzxShared *sh = d.sh;
  const u32 lane = d.lane;
  u8 *const out = d.out;
  // everything below is wave-uniform; readfirstlane tells the compiler so (SGPRs, scalar branches)
  LzxCommit C;
  C.run_end = rfl(run_end_); C.wbase = rfl(wbase_);
  C.P = rfl(d.P);
  C.R0 = rfl(s.R0); C.R1 = rfl(s.R1); C.R2 = rfl(s.R2);
  C.wsize = rfl(s.wsize); C.offset_written = rfl(s.offset); C.ref_size = rfl(s.ref_size);
  const bool length_empty = rfl((u32) s.length_empty) != 0u;
  int rc = LZX_RUN_DONE;

  // The parser stops `margin` bytes before the end of the input.  A round (64 starts + a 53-bit
  // token) plus one scalar token consumes at most 22 bytes, a block header read without any symbol
  // decode 17 more and the first symbol after it 7: with 56 the EOF-exact reader
  // (LzxDec::sym_ensure) still takes over at a symbol boundary at least 6 bytes before the
  // reference's read pointer can reach the end of the input.
  const u32 bit_limit = spec_bit_limit(d, 56u);
  if (rfl(d.cons_bits()) >= bit_limit) return LZX_RUN_SWITCH;
  // pending literals of the scalar path go out first: this path stores literals directly
  d.flush_lits();
  u32 bitpos, cb, pf;                                   // next unparsed bit (relative to d.w.origin)
  spec_stage(d, bitpos, cb, pf);
  u32 mlim[16 - LZX_MAIN_P];                            // limits of the code lengths beyond the table
#pragma unroll
  for (int l = LZX_MAIN_P + 1; l <= 16; l++) mlim[l - LZX_MAIN_P - 1] = rdl(d.hr_main.limv, (u32) l);

  spq_init(sh->spq, C.Q, C.P, lane);
  u32 *const tq0 = sh->tq0, *const tq1 = sh->tq1;
  u32 th = 0, tt = 0;                                   // token queue: committed / parsed (counters)
  bool stop = false;                                    // the parser is done (input margin, marker)
  bool bail = false;

#ifdef LZX_EXP_STATS
#define TICK(k) do { u64 n_ = __builtin_amdgcn_s_memtime(); d.st_t[k] += (u32)(n_ - tk_); tk_ = n_; } while (0)
#else
#define TICK(k) do { } while (0)
#endif
  while (rc == LZX_RUN_DONE && C.P < C.run_end && !bail) {
#ifdef LZX_EXP_STATS
    u64 tk_ = __builtin_amdgcn_s_memtime();
#endif
    // =================================== PARSE ===================================
    if (!stop && tt - th < 64u) {
      spec_slide(d, bitpos, cb, pf);
      const u32 rel = bitpos - (cb << 11) + lane;
      const u32 k = rel >> 5, sft = rel & 31u;
      const u32 i0 = sh->inbuf[k], i1 = sh->inbuf[k + 1u], i2 = sh->inbuf[k + 2u];
      const u32 w0 = (u32)(((((u64) i0 << 32) | i1) << sft) >> 32);
      const u32 w1 = (u32)(((((u64) i1 << 32) | i2) << sft) >> 32);
      const SpecTok t = lzx_spec_token<ALIGNED>(sh, d.hr_main.fov, mlim, length_empty, w0, w1);
      // next token start (in bits from bitpos); >= 256 marks "needs the scalar decoder" and ends the walk
      const u32 vn = t.unk ? (256u + lane) : (lane + t.tot);
      TICK(0);
      // ---- follow the real token boundaries: which positions start a token? ----
      u64 chain = 0, chain2 = 0;
      u32 q = 0;
      do { chain |= 1ull << q; q = rdl(vn, q); } while (q < 64u);
      bool hit_unknown = false;
      if (q >= 256u) {
        q -= 256u; hit_unknown = true;
        if (q < 64u) chain &= ~(1ull << q); else chain2 &= ~(1ull << (q - 64u));
      }
      u32 nA = (u32) __popcll(chain), nB = (u32) __popcll(chain2);
      TICK(1);
      // ---- queue the tokens on the chain ----
      {
        const u32 rank = __builtin_amdgcn_mbcnt_hi((u32)(chain >> 32), __builtin_amdgcn_mbcnt_lo((u32) chain, 0u));
        if ((chain >> lane) & 1ull) {
          const u32 ti = (tt + rank) & (LZX_TQ - 1u);
          tq0[ti] = t.kind | (t.olen << 3) | (((bitpos + lane) & 0xFFFFu) << 12);
          tq1[ti] = t.kind == 0u ? t.sym : t.off;
        }
        tt += nA + nB;
      }
      bitpos += q;
      d.st_rounds++;
      if (hit_unknown) {
        u32 tk_kind = 0, tk_val = 0, tk_off = 0;
        const u64 rq = ((u64) rdl(w0, q) << 32) | rdl(w1, q);
        const u32 tk_tot = lzx_scalar_token<ALIGNED>(d, length_empty, rq, tk_kind, tk_val, tk_off);
        u32 r0, r1 = tk_kind == 0u ? tk_val : tk_off;
        if (tk_tot == 0u) { r0 = LZX_TK_FAIL; stop = true; }
#ifdef LZX_DELTA
        else if (tk_kind != 0u && tk_val == 257u) { r0 = LZX_TK_BAIL; stop = true; }
#endif
        else r0 = tk_kind | ((tk_kind == 0u ? 1u : tk_val) << 3);
        if (lane == 0u) {
          const u32 ti = tt & (LZX_TQ - 1u);
          tq0[ti] = r0 | ((bitpos & 0xFFFFu) << 12);
          tq1[ti] = r1;
        }
        tt++;
        if (!stop) bitpos += tk_tot;
      }
      if (bitpos >= bit_limit) stop = true;
      TICK(2);
      if (!stop && tt - th < 64u) continue;
    }

    // =================================== COMMIT ===================================
    u32 n = tt - th;
    if (n > 64u) n = 64u;
    if (n == 0u) { rc = LZX_RUN_SWITCH; break; }         // the input margin was reached and all is committed
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    const u32 ci = (th + lane) & (LZX_TQ - 1u);
    const u32 c0 = tq0[ci], c1 = tq1[ci];
    u32 marker; bool fail_after;
    th += lzx_commit_batch(d, C, c0, c1, n, marker, fail_after);
    TICK(5);
    if (spq_due(C.Q, C.P)) spq_resolve(sh->spq, C.Q, out, C.P, false, lane);
    TICK(6);
    if (fail_after || marker == LZX_TK_FAIL) { d.err = ERR_DECRUNCH; rc = LZX_RUN_FAIL; }
    else if (marker == LZX_TK_BAIL) bail = true;
  }
  spq_resolve(sh->spq, C.Q, out, C.P, true, lane);
  // parsed but not committed: the bit position goes back to the first such token
  if (tt != th) {
    const u32 lo = rfl(tq0[th & (LZX_TQ - 1u)]) >> 12;
    bitpos -= (bitpos - lo) & 0xFFFFu;
  }
  d.P = C.P;
  s.R0 = C.R0; s.R1 = C.R1; s.R2 = C.R2;
  spec_resync(d, bitpos, cb, pf);
  return rc;
}

#endif  /* LZX_DELTA: lzx_run_spec */

#ifndef LZX_DELTA
// ---------------------------------------------------------------------------------------------------
// lzx_run_spec2 -- the speculative run of plain LZX: parse token LENGTHS, decode token VALUES at commit time.
//
// Measured on the box (profiles/round2_*): a unit's time is its wave's instruction count times the latency of its
// dependent steps -- and three quarters of the vector instructions were the 64-position token decode, executed
// for 64 lanes of which ~7 hold a real token.  What the chain needs from a position is only HOW LONG the token that
// would start there is.  So a round computes just that (main-tree entry -> code length, length footer's code
// length, number of offset bits, aligned symbol's length) and queues the START BITS of the tokens on the chain;
// the values (literal, match length, offset) are decoded when 64 queued tokens are committed -- one real token
// per lane, every lane busy.  A main code longer than the direct table stops the walk; only then are the long
// codes of the round resolved (lane-parallel, once) and the walk resumes.  Tokens the scalar decoder had to take
// wait in 16 side slots.  The LDS input window holds three 256-byte chunks (the one behind the parse position
// too), and the parser never runs more than a chunk ahead of the oldest queued token.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void spec3_stage(LzxDec &d, u32 &bitpos, u32 &cb, u32 &pf)
{
  LzxShared *sh = d.sh;
  const u32 lane = d.lane;
  bitpos = rfl(d.cons_bits());
  cb = bitpos >> 11;                                    // the window is chunks cb-1, cb, cb+1
  const u32 lo = d.w.load_chunk(cb, lane), hi = d.w.load_chunk(cb + 1u, lane);
  sh->inbuf[lane] = 0; sh->inbuf[64u + lane] = SWAP16(lo); sh->inbuf[128u + lane] = SWAP16(hi);
  if (lane < 4u) sh->inbuf[192u + lane] = 0;
  pf = d.w.load_chunk(cb + 2u, lane);
}
__device__ __forceinline__ void spec3_slide(LzxDec &d, const u32 bitpos, u32 &cb, u32 &pf)
{
  if ((bitpos >> 11) != cb) {
    LzxShared *sh = d.sh;
    const u32 lane = d.lane;
    const u32 mid = sh->inbuf[64u + lane], up = sh->inbuf[128u + lane];
    sh->inbuf[lane] = mid; sh->inbuf[64u + lane] = up; sh->inbuf[128u + lane] = SWAP16(pf);
    cb++;
    pf = d.w.load_chunk(cb + 2u, lane);
  }
}
__device__ __forceinline__ void spec3_resync(LzxDec &d, const u32 bitpos, const u32 cb, const u32 pf)
{
  LzxShared *sh = d.sh;
  const u32 lane = d.lane;
  const u32 wi = bitpos >> 5, ch = wi >> 6;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  u32 lo = sh->inbuf[64u + lane], hi = sh->inbuf[128u + lane], lw = sh->inbuf[lane];
  lo = SWAP16(lo); hi = SWAP16(hi); lw = SWAP16(lw);
  if (ch == cb) { d.w.cur = lo; d.w.nxt = hi; }
  else if (ch == cb + 1u) { d.w.cur = hi; d.w.nxt = pf; }
  else if (ch + 1u == cb) { d.w.cur = lw; d.w.nxt = lo; }                                   // went back (end of a run)
  else { d.w.cur = d.w.load_chunk(ch, lane); d.w.nxt = d.w.load_chunk(ch + 1u, lane); }
  d.w.wi = wi; d.bb = 0; d.bl = 0;
  d.refill(); d.refill();
  const u32 sk = bitpos & 31u;
  if (sk) { d.bb <<= sk; d.bl -= (int) sk; }
}

#define LZX_TQ2 256u                /* lzx_run_spec2: token queue entries */
#ifndef LZX_SETS
#define LZX_SETS 6                  /* position sets per lane: a round covers 64 * LZX_SETS bit positions (measured:
                                       2 sets 4.63 ms, 4 4.57, 6 4.46, 8 4.74 on the headline batch) */
#endif

template <bool ALIGNED>
__device__ __forceinline__ int lzx_run_spec2(LzxDec &d, LzxState &s, const u32 run_end_, const u32 wbase_)
{
  LzxShared *sh = d.sh;
  const u32 lane = d.lane;
  u8 *const out = d.out;
  LzxCommit C;
  C.run_end = rfl(run_end_); C.wbase = rfl(wbase_);
  C.P = rfl(d.P);
  C.R0 = rfl(s.R0); C.R1 = rfl(s.R1); C.R2 = rfl(s.R2);
  C.wsize = rfl(s.wsize); C.offset_written = rfl(s.offset); C.ref_size = rfl(s.ref_size);
  const bool length_empty = rfl((u32) s.length_empty) != 0u;
  int rc = LZX_RUN_DONE;
  // The parser stops `margin` bytes before the end of the input: a round (256 starts + a 53-bit token) plus one
  // scalar token is at most 46 bytes, a block header read without any symbol decode 17 more and the first symbol
  // after it 7 -- with 88 the EOF-exact reader still takes over at a symbol boundary well before the reference's
  // read pointer can reach the end of the input (cf. lzx_run_spec)
  const u32 bit_limit = spec_bit_limit(d, 56u + 8u * LZX_SETS);
  if (rfl(d.cons_bits()) >= bit_limit) return LZX_RUN_SWITCH;
  d.flush_lits();
  u32 bitpos, cb, pf;
  spec3_stage(d, bitpos, cb, pf);
  u32 mlim[16 - LZX_MAIN_P];
#pragma unroll
  for (int l = LZX_MAIN_P + 1; l <= 16; l++) mlim[l - LZX_MAIN_P - 1] = rdl(d.hr_main.limv, (u32) l);
  spq_init(sh->spq, C.Q, C.P, lane);
  u32 *const tq0 = sh->tq0;
  u32 th = 0, tt = 0;                                   // token queue: committed / parsed (counters)
  u32 qbase = 0;                                        // start bit of the oldest queued token (valid while tt != th)
  u32 sw = 0, nside = 0;                                // side slots: written (counter) / pending
  bool stop = false;

  while (rc == LZX_RUN_DONE && C.P < C.run_end) {
    // =================================== PARSE ===================================
    // (not while the oldest queued token would fall out of the LDS window, nor with the side slots nearly full)
    if (!stop && tt - th < 64u && nside < 12u && (tt == th || ((bitpos + 64u * LZX_SETS + 128u) >> 11) <= (qbase >> 11) + 1u)) {
      LZX_MARK("parse_begin");
      spec3_slide(d, bitpos, cb, pf);
      const u32 rel = bitpos - ((cb - 1u) << 11) + lane;
      const u32 k = rel >> 5, sft = rel & 31u;
      // ---- token lengths at 64 * LZX_SETS positions: lane l looks at bits bitpos + l + 64 j ----
      u32 vn[LZX_SETS];
#pragma unroll
      for (int j = 0; j < LZX_SETS; j++) {
        const u32 i0 = sh->inbuf[k + 2u * j], i1 = sh->inbuf[k + 2u * j + 1u];
        const u32 w0 = (u32)(((((u64) i0 << 32) | i1) << sft) >> 32);
        u32 w1 = 0;
        if (ALIGNED) { const u32 i2 = sh->inbuf[k + 2u * j + 2u]; w1 = (u32)(((((u64) i1 << 32) | i2) << sft) >> 32); }
        const u32 e = sh->main_tab[w0 >> (32 - LZX_MAIN_P)];
        bool unk;
        const u32 tot = lzx_adv_from_entry<ALIGNED>(sh, length_empty, e, w0, w1, unk);
        // next token start (in bits from bitpos); >= 1024: the scalar decoder must look, >= 2048: a main code longer
        // than the direct table (resolved below, lane-parallel, if the walk gets there)
        const u32 pos = lane + 64u * j;
        vn[j] = e == 0u ? (2048u + pos) : (unk ? (1024u + pos) : (pos + tot));
      }
      LZX_MARK("walk_begin");
      // ---- follow the real token boundaries through the sets ----
      u64 chain[LZX_SETS];
      u32 q = 0, ntok = 0;
#pragma unroll
      for (int j = 0; j < LZX_SETS; j++) {
        chain[j] = 0;
        if (q < 64u * (j + 1) && ntok <= 64u) {               // (a round queues at most 128 tokens)
          for (;;) {
            while (q < 64u * (j + 1)) { chain[j] |= 1ull << (q & 63u); q = rdl(vn[j], q & 63u); }
            if (q < 2048u) break;
            // the walk ran into a main code longer than the direct table: resolve this set's long codes (canonical
            // length = number of per-length limits the 16-bit peek is not below) and go on from there
            q -= 2048u;
            const u32 i0 = sh->inbuf[k + 2u * j], i1 = sh->inbuf[k + 2u * j + 1u];
            const u32 w0 = (u32)(((((u64) i0 << 32) | i1) << sft) >> 32);
            u32 w1 = 0;
            if (ALIGNED) { const u32 i2 = sh->inbuf[k + 2u * j + 2u]; w1 = (u32)(((((u64) i1 << 32) | i2) << sft) >> 32); }
            const u32 peek16 = w0 >> 16;
            u32 ln = LZX_MAIN_P + 1u;
#pragma unroll
            for (int l = LZX_MAIN_P + 1; l <= 16; l++) ln += (peek16 >= mlim[l - LZX_MAIN_P - 1]) ? 1u : 0u;
            const u32 lq = ln <= 16u ? ln : 0u;
            const u32 fo = (u32) __builtin_amdgcn_ds_bpermute((int)(lq << 2), (int) d.hr_main.fov);
            u32 idx = (fo >> 16) + ((peek16 >> (16u - lq)) - (fo & 0xFFFFu));
            if (idx >= LZX_MAIN_SYMS) idx = 0;
            const u32 e = lq ? ((u32) sh->main_sorted[idx] | (lq << LZX_MSH)) : 0u;
            bool unk2;
            const u32 tot2 = lzx_adv_from_entry<ALIGNED>(sh, length_empty, e, w0, w1, unk2);
            const u32 pos = lane + 64u * j;
            if (vn[j] >= 2048u) vn[j] = (unk2 || e == 0u) ? (1024u + pos) : (pos + tot2);
            chain[j] &= ~(1ull << (q & 63u));
          }
          ntok += (u32) __popcll(chain[j]);
        }
      }
      LZX_MARK("walk_end");
      bool hit_unknown = false;
      if (q >= 1024u) {
        q -= 1024u; hit_unknown = true;
#pragma unroll
        for (int j = 0; j < LZX_SETS; j++) if ((q >> 6) == (u32) j) { chain[j] &= ~(1ull << (q & 63u)); ntok--; }
      }
      // ---- queue the start bits of the tokens on the chain ----
      {
        if (tt == th && ntok) qbase = bitpos;
        u32 base = tt;
#pragma unroll
        for (int j = 0; j < LZX_SETS; j++) {
          const u32 rank = __builtin_amdgcn_mbcnt_hi((u32)(chain[j] >> 32), __builtin_amdgcn_mbcnt_lo((u32) chain[j], 0u));
          if (lane_in(chain[j])) tq0[(base + rank) & (LZX_TQ2 - 1u)] = (bitpos + 64u * j + lane) & 0xFFFFu;
          base += (u32) __popcll(chain[j]);
        }
        tt = base;
      }
      d.st_rounds++;
      LZX_MARK("queue_end");
      if (hit_unknown) {
        // a token the lane-parallel decoder does not take: the scalar decoder reads it from its 64 bits
        const u32 tb = bitpos + q;
        const u32 r2 = tb - ((cb - 1u) << 11);
        const u32 k2 = r2 >> 5, s2 = r2 & 31u;
        const u32 a0 = rfl(sh->inbuf[k2]), a1 = rfl(sh->inbuf[k2 + 1u]), a2 = rfl(sh->inbuf[k2 + 2u]);
        const u64 hi64 = ((u64) a0 << 32) | a1, lo64 = (u64) a2 << 32;
        const u64 rq = s2 ? ((hi64 << s2) | (lo64 >> (64u - s2))) : hi64;
        u32 tk_kind = 0, tk_val = 0, tk_off = 0;
        const u32 tk_tot = lzx_scalar_token<ALIGNED>(d, length_empty, rq, tk_kind, tk_val, tk_off);
        u32 r0;
        const u32 r1 = tk_kind == 0u ? tk_val : tk_off;
        if (tk_tot == 0u) { r0 = LZX_TK_FAIL; stop = true; }
        else r0 = tk_kind | ((tk_kind == 0u ? 1u : tk_val) << 3);
        if (lane == 0u) {
          sh->side0[sw & 15u] = r0; sh->side1[sw & 15u] = r1;
          tq0[tt & (LZX_TQ2 - 1u)] = (tb & 0xFFFFu) | (((sw & 15u) + 1u) << 16);
        }
        if (tt == th) qbase = tb;
        sw++; nside++; tt++;
        bitpos = tb + (stop ? 0u : tk_tot);
      }
      else bitpos += q;
      if (bitpos >= bit_limit) stop = true;
      LZX_MARK("parse_end");
      if (!stop && tt - th < 64u) continue;
    }

    // =================================== COMMIT ===================================
    LZX_MARK("commit_begin");
    u32 n = tt - th;
    if (n > 64u) n = 64u;
    if (n == 0u) { rc = LZX_RUN_SWITCH; break; }         // the input margin was reached and all is committed
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    const u32 qe = tq0[(th + lane) & (LZX_TQ2 - 1u)];
    u32 c0, c1;
    {
      // decode the queued tokens' values, one token per lane, from their start bits
      const u32 sb = qbase + ((qe - qbase) & 0xFFFFu);                 // full start bit
      const u32 rel = sb - ((cb - 1u) << 11);
      const u32 k = (lane < n) ? (rel >> 5) : 0u, sft = rel & 31u;
      const u32 i0 = sh->inbuf[k], i1 = sh->inbuf[k + 1u], i2 = sh->inbuf[k + 2u];
      const u32 w0 = (u32)(((((u64) i0 << 32) | i1) << sft) >> 32);
      const u32 w1 = (u32)(((((u64) i1 << 32) | i2) << sft) >> 32);
      const SpecTok t = lzx_spec_token<ALIGNED>(sh, d.hr_main.fov, mlim, length_empty, w0, w1);
      c0 = t.unk ? LZX_TK_FAIL : (t.kind | (t.olen << 3));
      c1 = t.kind == 0u ? t.sym : t.off;
      const u32 si = qe >> 16;
      if (si) { c0 = sh->side0[si - 1u]; c1 = sh->side1[si - 1u]; }
      c0 |= (qe & 0xFFFFu) << 12;
    }
    LZX_MARK("values_end");
    u32 marker; bool fail_after;
    const u32 took = lzx_commit_batch(d, C, c0, c1, n, marker, fail_after);
    LZX_MARK("commit_batch_end");
    nside -= (u32) __popcll(ballot(lane < took && (qe >> 16) != 0u));
    th += took;
    if (tt != th) {                                                     // start bit of the token that is the oldest now
      const u32 nx = rfl(tq0[th & (LZX_TQ2 - 1u)]) & 0xFFFFu;
      qbase += (nx - qbase) & 0xFFFFu;
    }
    LZX_MARK("resolve_begin");
    if (spq_due(C.Q, C.P)) spq_resolve(sh->spq, C.Q, out, C.P, false, lane);
    LZX_MARK("resolve_end");
    if (fail_after || marker == LZX_TK_FAIL) { d.err = ERR_DECRUNCH; rc = LZX_RUN_FAIL; }
  }
  spq_resolve(sh->spq, C.Q, out, C.P, true, lane);
  // parsed but not committed: the bit position goes back to the first such token
  if (tt != th) bitpos = qbase;
  d.P = C.P;
  s.R0 = C.R0; s.R1 = C.R1; s.R2 = C.R2;
  spec3_resync(d, bitpos, cb, pf);
  return rc;
}
#endif  /* !LZX_DELTA: lzx_run_spec2 */

#endif  /* !LZX_PARSE_ONLY */

#ifndef LZX_DELTA
// ---------------------------------------------------------------------------------------------------
// Frame-level parse parallelism (plain LZX; units that carry a frame table, MSPACK_HIP_UF_FRAME_TABLE).
//
// Every 32 KiB frame starts on a 16-bit boundary of the compressed stream (lzxd.c:695-697) at an offset the
// container states up front -- one CFDATA block per frame in a cabinet (cabd.c:1362-1479), one reset-table
// entry per frame in a CHM (chmd.c:1146-1149).  The serial chain of a unit is "where does the next token
// start"; it needs the Huffman tables, not the window and not R0-R2.  So mspack_lzx_pipe gives every FRAME a parse
// task (lzx_pipe_parse): it waits for the code lengths of the frame before it (code lengths are deltas on the previous
// block's, lzxd.c:138-183: a chain, but a short one -- one header per link), reads its own block header, publishes
// its code lengths, builds the tables and parses the frame's tokens with every lane walking its own stretch of the bits
// (lzx_parse_emit): literals go straight to the output, matches become 8-byte records in the launch's record pool.
// Rounds 2-4 parsed on the guess that every frame holds exactly ONE verbatim / aligned block that begins where the frame
// begins -- what this build's own encoder writes.  Microsoft's encoder does not: the reference's large-files cabinets hold
// blocks of megabytes (8 384 624 bytes, 7 379 562 ...), so their frames lie INSIDE a block, and the guess failed for every
// frame of every real cabinet tried (they all took the serial path: 180 MB/s).  Round 5: the chain from frame to frame is
// "code lengths + what is left of the open block"; a frame inside a block inherits both and has no header to read, a frame
// that holds a block's end parses up to it, reads the next header there and goes on with the new tables (lzx_pipe_parse).
// Stored blocks, a block that ends where nothing can be parsed, damage: the task gives up silently.  A resolve task per frame (lzx_pipe_resolve) then
// turns the records into copies in stream order: R0-R2, the reference's checks, the match queue.  Whatever the tasks
// do not cover -- the last bytes of the input, a frame with several blocks, stored blocks, a damaged stream, a wrong
// table -- ends the unit's chain there (rs_* in the unit's first record) and is decoded by the serial path
// (mspack_decode_lzx, resume) from that very bit, so error codes and byte counts cannot differ.
// ---------------------------------------------------------------------------------------------------

struct __align__(16) LzxFrameRec {
  u32 status;                       /* LZX_ST_*: 2 = header known (code lengths published), 7 = literals stored, records written */
  u32 n_tokens;
  u32 hdr_start_bit;                /* bit positions count from the unit's first compressed byte */
  u32 end_bit;                      /* first bit that was not parsed */
  u32 block_type, block_length;
  u32 flags;                        /* 1: the length tree is empty, 2: literal 0xE8 has a code */
  u32 prog;                         /* mspack_lzx_pipe, while status is 2: match records | output bytes << 15 that are in memory
                                       already (published after every pass of lzx_parse_emit but the last) */
  u8 ali_len[8];
  u32 rem_out;                      /* lzx_pipe_parse: bytes of the block that is open BEHIND this frame (0: the next frame starts with a
                                       block header).  Published with the code lengths (status 2): the next frame's task inherits both */
  u32 run_rem;                      /* what a decoder that goes on INSIDE this frame (a record that ends early) has as block_remaining
                                       at the frame's first byte, counting the block the record ends in as if it had begun there */
  u8 main_len[LZX_MAIN_SYMS + 16];
  u8 len_len[LZX_LEN_SYMS + 70];
  /* ---- mspack_lzx_pipe (lzx_pipe_parse / lzx_pipe_commit) ---- */
  u32 frame_start_bit;              /* where the frame begins: in front of a reset interval's 1 + 32 header bits */
  u32 intel_filesize;               /* the interval header's value when this frame carries it (else 0) */
  u32 bytes_done;                   /* output bytes the record covers (== the frame's size: a complete frame) */
  u32 n_edge;                       /* literals kept in edge_lit (the frame's first bytes share a cache line with the
                                       bytes below them, which another wave may be writing: the commit wave stores them) */
  u32 edge_mask[4];
  /* what the unit's commit task leaves for mspack_decode_lzx: where serial decoding resumes (in the unit's FIRST record) */
  u32 rs_valid, rs_frame, rs_partial, rs_P, rs_next_bit, rs_R0, rs_R1, rs_R2;
  u8 edge_lit[128];
  /* the unit's chain of frames (lzx_pipe_resolve): 0 = open, 1 = this frame and every frame before it are complete in the
   * output (cR0-cR2: R0-R2 behind its last match), 2 = the chain ended at or before this frame */
  u32 chain, cR0, cR1, cR2;
  /* lzx_fold.hpp: R0-R2 behind the frame's last match, published as soon as they are known -- long before its bytes are
   * final (rst: 0 open, 1 valid, 2 the chain ends at or before this frame) */
  u32 rst, rR0, rR1, rR2;
  u8 pad2[16];
  u32 chunk[REC_CHUNKS];            /* where the frame's match records are: wave_common.hpp, RecPool */
};
static_assert(sizeof(LzxFrameRec) == 1408, "LzxFrameRec layout");
// LzxFrameRec::status.  The separate header / parse launches only use 0, 2, 1.  In the dependency-driven launch
// (mspack_lzx_pipe, shim.hip) the word is also the hand-off flag between the frame's parse task and the unit's wave:
//   0 untouched | 5 a parse wave claimed the frame | 2 its code lengths are in the record, tokens still being parsed |
//   1 tokens parsed (final) | 3 code lengths valid, no tokens (final) | 4 nothing usable (final; the chain of code
//   lengths is broken for the rest of the reset interval) | 6 the unit's own wave took the frame (decodes it serially)
#ifdef LZX_PIPE_TRACE      /* analysis builds: time a unit task spends waiting for parse tasks (shim.hip: g_pipe_wait) */
__device__ unsigned long long g_pipe_wait[1 << 16];
__device__ unsigned long long g_pipe_phase[16];     /* summed over all waves: s_memrealtime ticks per phase (PH below) */
/* (accumulated in registers, added to the global sums once per task: an atomic per stamp would serialise the waves) */
#define PHDECL() u32 pha_[16] = { 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u }
#define PH0() unsigned long long ph_ = __builtin_amdgcn_s_memrealtime()
#define PH(k) do { const unsigned long long n_ = __builtin_amdgcn_s_memrealtime(); pha_[k] += (u32)(n_ - ph_); ph_ = n_; } while (0)
#define PHE0() unsigned long long phe_ = __builtin_amdgcn_s_memrealtime()
#define PHE(k) do { const unsigned long long n_ = __builtin_amdgcn_s_memrealtime(); d.st_t[k] += (u32)(n_ - phe_); phe_ = n_; } while (0)
#define PHFLUSH() do { if (threadIdx.x == 0) for (int k_ = 0; k_ < 16; k_++) if (pha_[k_]) atomicAdd(&g_pipe_phase[k_], (unsigned long long) pha_[k_]); } while (0)
#define PHCNT(k, n) do { d.st_t[k] += (n); } while (0)       /* (12..15: counts, not times -- steps of the count walks, rounds, steps of the last walk, passes) */
#else
#define PHDECL() do { } while (0)
#define PHE0() do { } while (0)
#define PHE(k) do { } while (0)
#define PHCNT(k, n) do { } while (0)
#define PH0() do { } while (0)
#define PH(k) do { } while (0)
#define PHFLUSH() do { } while (0)
#endif
#ifdef LZX_PIPE_TRACE
#define LZX_PIPE_WAIT_BEGIN() const unsigned long long pw_ = __builtin_amdgcn_s_memrealtime()
#define LZX_PIPE_WAIT_END() do { if (threadIdx.x == 0) g_pipe_wait[blockIdx.x & 0xFFFFu] += __builtin_amdgcn_s_memrealtime() - pw_; } while (0)
#else
#define LZX_PIPE_WAIT_BEGIN() do { } while (0)
#define LZX_PIPE_WAIT_END() do { } while (0)
#endif
#define LZX_ST_NONE 0u
#define LZX_ST_PARSED 1u
#define LZX_ST_HEADER 2u
#define LZX_ST_HDRONLY 3u
#define LZX_ST_FAILED 4u
#define LZX_ST_CLAIMED 5u
#define LZX_ST_TAKEN 6u
#define LZX_ST_EMITTED 7u         /* lzx_pipe_parse: literals stored, match records written (final) */
/* a match record of lzx_pipe_parse (uint2): x = position in the unit's output, y = offset << 11 | length << 2 | which:
 * 0 explicit offset, 1..3 repeat of R0 / R1 / R2 (lzxd.c:565-586) */
__device__ __forceinline__ u32 lzx_status_load(const u32 *p) {
  return rfl(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
// publish what this wave stored (tokens, record fields), then the status word: agent-scope release, a drained
// store queue (the compiler may drop the wait behind the write-back: MI355X guide, hand-off recipe), relaxed flag
__device__ __forceinline__ void lzx_status_publish(u32 *p, const u32 v, const u32 lane) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#ifndef MSPACK_WAVE_EMU
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  if (lane == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// continue reading at an absolute bit position (from the unit's first byte)
__device__ __forceinline__ void lzx_seek_bit(LzxDec &d, const u32 abs_bit)
{
  const u32 par = d.w.origin & 1u;                       // 16-bit words start at bytes of this parity
  const u32 wbyte = ((((abs_bit >> 3) - par) >> 1) << 1) + par;
  const u32 sk = abs_bit - wbyte * 8u;
  d.w.seek(wbyte, d.lane);
  d.bb = 0; d.bl = 0; d.rbl = 0;
  d.near_end = (wbyte >= d.w.in_len || d.w.in_len - wbyte <= 64u);
  d.refill(); d.refill();
  if (sk) { d.bb <<= sk; d.bl -= (int) sk; }
}

#ifdef LZX_PARSE_ONLY
// ---------------------------------------------------------------------------------------------------
// The lane parser -- a frame's tokens, every lane walking its own stretch of the bit stream.
//
// The 64-positions-per-round parser above spends its vector instructions on 64 lanes of which the ~5 on the chain
// matter.  Here the frame's bits [B, E) -- E is what the frame table says, a hint -- are cut into 64 stretches and
// lane l walks the tokens of stretch l one after the other: length of the token at p, p += length, until p leaves
// the stretch.  Lane 0 starts at B, a real token start; the others start at their stretch's first bit, which is
// almost never one.  But a walk that starts in the middle of a token falls into step with the real chain after a
// few tokens (each landing is a real token start with probability ~1/mean token length), so its EXIT -- the first
// position beyond the stretch -- is almost always the real chain's.  Round 2: every lane starts again from its
// left neighbour's exit.  Lane l's walk is the real chain if lane l-1's was and its entry is lane l-1's exit: an
// induction from lane 0, checked after every round (entry == left exit for all lanes: done, usually after round
// 2; otherwise only the lanes whose entry moved walk again).  A last walk decodes the token VALUES and stores
// them: lane l's i-th token at (tokens of lanes < l) + i.  Nothing here depends on E being right: a wrong table
// only makes the stretches unequal.  What a lane cannot decode (a code the tables do not hold) ends the record
// there; the unit's own wave judges that token.  The input sits in LDS (8 KiB per pass, halves of every dword
// swapped: a plain MSB-first bit string); all walks are LDS lookups, one token per lane per step.
// ---------------------------------------------------------------------------------------------------
#ifndef LZX_LANE_ROUNDS
#define LZX_LANE_ROUNDS 5u          /* walks before the consistent prefix is taken as it is */
#endif
#ifndef LZX_LANE_TAIL
#define LZX_LANE_TAIL 384u
#endif
#ifndef LZX_SEG
#define LZX_SEG 8u                  /* lzx_parse_emit: tokens per segment of the balanced last walk (a power of two): a round's 64 segments
                                       cover ~1.1 KiB of output -- what the literal ring holds */
#endif
#endif  /* LZX_PARSE_ONLY */


#ifdef LZX_PARSE_ONLY
// ---------------------------------------------------------------------------------------------------
// lzx_parse_emit -- the lane parser taken one step further (mspack_lzx_pipe): the parse wave does not leave TOKENS
// for the unit's wave, it leaves the frame's LITERALS IN PLACE and a list of MATCH RECORDS.
//
// A frame starts at a known output position (f * 32 KiB), so once the lanes' stretches are consistent every lane
// knows, by a prefix sum over the stretches' output lengths, where its first token's bytes go: in its last walk it
// stores its literals straight into the output and writes one record per match (position, length, explicit offset or
// which of R0-R2 it repeats).  What is left for the unit's wave -- the part LZ77 makes serial -- is resolving R0-R2
// along the record list and copying the matches (lzx_pipe_commit): no token ever travels through memory, and the
// positions / literal stores of all frames of a unit run in parallel.
// The frame's first bytes may share a cache line with bytes another wave is writing at that moment (the end of the
// previous frame, of the previous unit): literals there (`edge_n` positions) are kept in the record and stored by the
// commit wave.  The walk stops where the frame is full (frame_size bytes), at a token that would cross its end, at a
// token the tables do not hold and 56 bytes before the end of the input (the EOF-exact reader's): bytes_done / end_bit
// say how far it got; the rest is decoded serially (mspack_decode_lzx resumes there).
// ---------------------------------------------------------------------------------------------------
template <bool ALIGNED>
__device__ __forceinline__ u32 lzx_adv_olen(const LzxShared *sh, const bool length_empty, const u32 e, const u32 e2,
                                            const u32 w0, const u32 w1, bool &unk, u32 &olen)
{
  const u32 mlen = e >> LZX_MSH, sym = e & LZX_MMASK;
  const bool is_match = sym >= 256u;
  const u32 m = sym - 256u, slot = m >> 3;
  const bool need_len = is_match && (m & 7u) == 7u;
  u32 tot = mlen;
  unk = false;
  olen = is_match ? (m & 7u) + 2u : 1u;
  if (need_len) { unk = (e2 == 0u) || length_empty; tot += e2 >> 10; olen += e2 & 1023u; }
  const int ex_ = (int)(slot >> 1) - 1;
  const u32 extra = (u32)(ex_ < 0 ? 0 : (ex_ > 17 ? 17 : ex_));
  const bool expl = is_match && slot >= 3u;
  if (ALIGNED) {
    const bool ali = extra >= 3u;
    const u32 nb = ali ? extra - 3u : extra;
    const u64 r = ((u64) w0 << 32) | w1;
    const u32 e3 = sh->ali_tab[(u32)((r << (tot + nb)) >> (64 - LZX_ALI_P))];
    if (expl) { tot += nb; if (ali) { tot += e3 >> 10; unk = unk || e3 == 0u; } }
  }
  else if (expl) tot += extra;
  return tot;
}


// ---------------------------------------------------------------------------------------------------
// lzx_build_sub -- second level of the parse waves' main-tree table.
// The direct table has 2^8 entries (LDS), and a main tree of 656 symbols has many codes of 9..16 bits: in nearly every
// step of a walk SOME lane meets one, and the lane-parallel resolve of codes beyond the table (eight limit compares, a
// ds_bpermute, a sorted-symbol lookup: ~35 instructions) ran for the whole wave.  With a second level -- per 8-bit
// prefix that starts longer codes, a sub-table indexed by the next Lmax(prefix) - 8 bits -- a long code costs one more
// LDS read and no branch.  Canonical codes: symbol i of the sorted list (length L, i-th of its length) has the code
// first(L) + (i - offs(L)); hr.fov holds first | offs << 16 per length.  Returns false (tables untouched) when the
// sub-tables do not fit LZX_SUB_CAP entries: the walks then resolve long codes the old way.
// Level-1 entry of such a prefix: 0x8000 | (sub-table bits - 1) << 11 | sub-table base.
// ---------------------------------------------------------------------------------------------------
#define LZX_SUB_CAP ((528u + LZX_MAIN_SYMS + 16u + LZX_LEN_SYMS + 70u + 8u) / 2u)
__device__ __forceinline__ bool lzx_build_sub(LzxShared *sh, const HuffRegs &hr, const u32 nsorted, const u32 lane)
{
  static_assert(LZX_MAIN_P == 8, "lzx_build_sub: 8 direct bits");
  u32 *const lmax = sh->stage;                                  // 256 words of scratch (the stage is filled later)
  for (u32 x = lane; x < 256u; x += WAVE) lmax[x] = 0u;
  u32 first[8], offs[9];                                        // lengths 9..16
#pragma unroll
  for (int l = 9; l <= 16; l++) { const u32 fo = rdl(hr.fov, (u32) l); first[l - 9] = fo & 0xFFFFu; offs[l - 9] = fo >> 16; }
  offs[8] = nsorted;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  const u32 lo = offs[0];
  if (lo >= nsorted) return true;                               // no code is longer than the direct table
  // ---- the longest code under every prefix ----
  for (u32 i = lo + lane; i < nsorted; i += WAVE) {
    u32 L = 9u;
#pragma unroll
    for (int l = 10; l <= 16; l++) L += (i >= offs[l - 9]) ? 1u : 0u;
    u32 fc = first[0], of = offs[0];
#pragma unroll
    for (int l = 10; l <= 16; l++) if (L == (u32) l) { fc = first[l - 9]; of = offs[l - 9]; }
    const u32 code16 = (fc + (i - of)) << (16u - L);
    atomicMax(&lmax[(code16 >> 8) & 255u], L);
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  // ---- sub-table sizes -> bases; level-1 entries ----
  u32 total = 0;
  u32 bases[4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const u32 x = (u32) r * 64u + lane;
    const u32 lm = lmax[x];
    const u32 sz = lm ? 1u << (lm - 8u) : 0u;
    const u32 inc = wave_incl_scan(sz);
    bases[r] = total + inc - sz;
    total += rdl(inc, 63u);
  }
  if (total > LZX_SUB_CAP) return false;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const u32 x = (u32) r * 64u + lane;
    const u32 lm = lmax[x];
    if (lm) { sh->main_tab[x] = (LZX_MTAB_T)(0x8000u | ((lm - 9u) << 11) | bases[r]); lmax[x] = lm | (bases[r] << 8); }
  }
  for (u32 q = lane; q < total; q += WAVE) sh->sub_tab[q] = 0;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  // ---- every long symbol fills its share of its prefix's sub-table ----
  for (u32 i = lo + lane; i < nsorted; i += WAVE) {
    u32 L = 9u;
#pragma unroll
    for (int l = 10; l <= 16; l++) L += (i >= offs[l - 9]) ? 1u : 0u;
    u32 fc = first[0], of = offs[0];
#pragma unroll
    for (int l = 10; l <= 16; l++) if (L == (u32) l) { fc = first[l - 9]; of = offs[l - 9]; }
    const u32 code16 = (fc + (i - of)) << (16u - L);
    const u32 lb = lmax[(code16 >> 8) & 255u];
    const u32 lm = lb & 255u, base = lb >> 8, sb = lm - 8u;
    const u32 start = (code16 & 255u) >> (8u - sb), cnt = 1u << (lm - L);
    const u32 ent = (u32) sh->main_sorted[i] | (L << 10);
    for (u32 r = 0; r < cnt; r++) sh->sub_tab[base + start + r] = (u16) ent;
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  return true;
}

// 32 bits of the staged input (sh->stage: dwords of an MSB-first bit string) from bit p on, and the 32 behind them.
// The window is taken one bit early -- dwords ((p + 31) >> 5) - 1 and the next, shifted right by 31 - ((p + 31) & 31)
// -- so that the shift is always 0..31: one v_alignbit_b32 per word, no 64-bit shift and no special case for p % 32 == 0
// (for p == 0 the dword in front of the stage is read and shifted out entirely).
#define STAGE_BITS(p_, w0_, w1_, WANT1)                                                        \
  u32 w0_, w1_ = 0u;                                                                           \
  {                                                                                            \
    const u32 t_ = (p_) + 31u, a_ = ~t_ & 31u;                                                 \
    const u32 *q_ = sh->stage + (t_ >> 5);                                                     \
    const u32 x0_ = q_[-1], x1_ = q_[0];                                                       \
    w0_ = (u32) __builtin_amdgcn_alignbit(x0_, x1_, a_);                                       \
    if (WANT1) { const u32 x2_ = q_[1]; w1_ = (u32) __builtin_amdgcn_alignbit(x1_, x2_, a_); } \
  }


// One token at the bits (w0, w1), every lane its own: main-tree entry (codes beyond the direct table resolved for all
// lanes at once when any lane has one), length footer, offset bits, aligned-offset symbol.  Everything is computed for
// every lane and selected -- no divergent branches in the walks' loop bodies.  unk: the tables do not hold this token.
struct EmitTok { u32 tot, olen, sym, slot, off; bool is_match, expl, unk; };
template <bool ALIGNED, bool VALUES>
__device__ __forceinline__ EmitTok lzx_emit_token(const LzxShared *sh, const bool act, const bool length_empty,
                                                  const u32 *mlim, const u32 *llim, const u32 main_fov, const u32 len_fov,
                                                  const u32 w0, const u32 w1, const bool two_level)
{
  EmitTok t;
  u32 e = sh->main_tab[w0 >> (32 - LZX_MAIN_P)];
  if (two_level) {
    // (second level: always read, selected -- no branch; a direct entry's fields index some harmless slot)
    const u32 sb = ((e >> 11) & 7u) + 1u;
    u32 ix = (e & 0x7FFu) + (((w0 >> 16) & 255u) >> (8u - sb));
    ix = ix < LZX_SUB_CAP ? ix : 0u;
    const u32 e2_ = sh->sub_tab[ix];
    e = (e & 0x8000u) ? e2_ : e;
  }
  else if (ballot(act && e == 0u)) {
    const u32 pk = w0 >> 16;
    u32 ln = LZX_MAIN_P + 1u;
#pragma unroll
    for (int l = LZX_MAIN_P + 1; l <= 16; l++) ln += (pk >= mlim[l - LZX_MAIN_P - 1]) ? 1u : 0u;
    const u32 lq = ln <= 16u ? ln : 0u;
    const u32 fo = (u32) __builtin_amdgcn_ds_bpermute((int)(lq << 2), (int) main_fov);
    u32 ix = (fo >> 16) + ((pk >> (16u - lq)) - (fo & 0xFFFFu));
    ix = ix < LZX_MAIN_SYMS ? ix : 0u;
    const u32 el = (u32) sh->main_sorted[ix] | (lq << LZX_MSH);
    e = (e == 0u && lq != 0u) ? el : e;
  }
  const u32 ml = e >> LZX_MSH, sy = e & LZX_MMASK;
  const bool is_match = sy >= 256u;
  const u32 mq = sy - 256u, slot = mq >> 3, lh = mq & 7u;
  const bool foot = is_match && lh == 7u;
  const u32 wl = w0 << ml;
  u32 e2 = sh->len_tab[wl >> (32 - LZX_LEN_P)];
  if (ballot(act && foot && e2 == 0u)) {
    const u32 pk = wl >> 16;
    u32 ln = LZX_LEN_P + 1u;
#pragma unroll
    for (int l = LZX_LEN_P + 1; l <= 16; l++) ln += (pk >= llim[l - LZX_LEN_P - 1]) ? 1u : 0u;
    const u32 lq = ln <= 16u ? ln : 0u;
    const u32 fo = (u32) __builtin_amdgcn_ds_bpermute((int)(lq << 2), (int) len_fov);
    u32 ix = (fo >> 16) + ((pk >> (16u - lq)) - (fo & 0xFFFFu));
    ix = ix < 256u ? ix : 0u;
    const u32 el = (u32) sh->len_sorted[ix] | (lq << 10);
    e2 = (e2 == 0u && lq != 0u) ? el : e2;
  }
  u32 tot = ml + (foot ? e2 >> 10 : 0u);
  t.olen = is_match ? lh + 2u + (foot ? e2 & 1023u : 0u) : 1u;
  bool unk = e == 0u || (foot && (e2 == 0u || length_empty));
  const int ex_ = (int)(slot >> 1) - 1;
  const u32 extra = (u32)(ex_ < 0 ? 0 : (ex_ > 17 ? 17 : ex_));
  const bool expl = is_match && slot >= 3u;
  u32 off = 0;
  if (VALUES) off = (((slot < 36u) ? 2u + (slot & 1u) : slot - 34u) << extra) - 2u;
  // the 32 bits behind the codes read so far (tot <= 32; a shift of 32 - tot == 0 hands back w1: right for tot == 32)
  const u32 v = (u32) __builtin_amdgcn_alignbit(w0, w1, 32u - tot);
  if (ALIGNED) {
    const bool ali = extra >= 3u;
    const u32 nb = ali ? extra - 3u : extra;
    const u32 vb = nb ? v >> (32u - nb) : 0u;
    const u32 e3 = sh->ali_tab[(v << nb) >> (32 - LZX_ALI_P)];
    tot += expl ? nb + (ali ? e3 >> 10 : 0u) : 0u;
    unk = unk || (expl && ali && e3 == 0u);
    if (VALUES) off += ali ? (vb << 3) + (e3 & 1023u) : vb;
  }
  else {
    if (VALUES) off += extra ? v >> (32u - extra) : 0u;
    tot += expl ? extra : 0u;
  }
  t.tot = tot; t.sym = sy; t.slot = slot; t.off = off; t.is_match = is_match; t.expl = expl; t.unk = unk;
  return t;
}

template <bool ALIGNED>
__device__ __forceinline__ void lzx_parse_emit(LzxDec &d, const bool length_empty, const u32 start_bit, const u32 frame_end_bit,
                                               u8 *const fout, const u32 frame_pos, const u32 frame_size, const u32 edge_n,
                                               LzxFrameRec *rec, RecWriter &W, u32 &n_rec, u32 &end_bit, u32 &bytes_done,
                                               const bool two_level, const bool stream, const u32 plimit, const bool first_seg)
{
  // (between two calls for one frame the edge literals' position mask -- its LDS words are the table builder's counters -- and
  // the record writer's chunk list -- the pretree's table -- wait in the stage's last 64 words, which only a pass's look-ahead
  // uses: nothing between the calls touches them)
  // n_rec / bytes_done: in and out -- a frame that holds the end of one block and the beginning of the next is parsed in two
  // calls (lzx_pipe_parse), each with its own tables, the second one going on where the first one stopped; plimit: the frame
  // position the call may not pass (the end of its block or of the frame: a match that crosses either is the serial path's to
  // report, lzxd.c:678-693)
  LzxShared *sh = d.sh;
  const u32 lane = d.lane;
  const u32 in_limit = d.w.in_len > 56u ? (d.w.in_len - 56u) * 8u : 0u;
  const u32 Eall = frame_end_bit < in_limit ? frame_end_bit : in_limit;
  u32 mlim[16 - LZX_MAIN_P], llim[16 - LZX_LEN_P];
#pragma unroll
  for (int l = LZX_MAIN_P + 1; l <= 16; l++) mlim[l - LZX_MAIN_P - 1] = rdl(d.hr_main.limv, (u32) l);
#pragma unroll
  for (int l = LZX_LEN_P + 1; l <= 16; l++) llim[l - LZX_LEN_P - 1] = rdl(d.hr_len.limv, (u32) l);
  const u32 main_fov = d.hr_main.fov, len_fov = d.hr_len.fov;
  u32 tt = rfl(n_rec), B = rfl(start_bit), P = rfl(bytes_done);   // records written, next bit, bytes of the frame done
  bool stop = false;
  if (first_seg) { if (lane < 4u) sh->cnt[lane] = 0u; }        // the edge literals' positions (128 bits)
  else {
    if (lane < 4u) sh->cnt[lane] = sh->stage[LZX_STAGE_WORDS + REC_CHUNKS + lane];
    W.restore(sh->stage + LZX_STAGE_WORDS, lane);
  }
#ifdef LZX_LIT_RING
  // literals below this position have left the ring (a multiple of 16).  (A second call starts with the first whole row at or
  // above P: the literals in front of it are stored on their own -- the row they lie in holds the first call's bytes)
  u32 lit_flushed = first_seg ? edge_n : (((P + 15u) & ~15u) > edge_n ? ((P + 15u) & ~15u) : edge_n);
#endif

  while (!stop && B < Eall && P < plimit) {
    PHE0();
    PHCNT(3, 1u);
    // ---- stage the input from the dword that holds bit B ----
    const u32 sb_byte = (B >> 5) << 2, sb_bit = sb_byte * 8u;
    u32 E = sb_bit + LZX_STAGE_WORDS * 32u; if (E > Eall) E = Eall;
    const u32 b0 = B - sb_bit, e0 = E - sb_bit;
    d.w.origin = sb_byte;
    {
      // every chunk of the pass is requested before the first one is waited for: one memory round trip per pass
      const u32 nck = (e0 + 128u + 2047u) >> 11;               // a token that starts below e0 ends below e0 + 53
      constexpr int NCH = (int)(LZX_STAGE_WORDS / 64u) + 1;
      u32 sv[NCH];
      if ((((size_t) d.w.unit) & 3u) == 0u) {
        // dword-aligned input (sb_byte is a multiple of 4): plain loads from clamped addresses, nothing between them
        // that waits -- the chunks' loads are all in flight before the first LDS store
#pragma unroll
        for (int c = 0; c < NCH; c++) {
          const u32 o = sb_byte + (u32) c * 256u + lane * 4u;
          sv[c] = gld((const u32 *)(d.w.unit + (((u32) c < nck && o < d.w.in_len) ? o : 0u)));
        }
#pragma unroll
        for (int c = 0; c < NCH; c++) {
          const u32 o = sb_byte + (u32) c * 256u + lane * 4u;
          u32 v = o < d.w.in_len ? sv[c] : 0u;
          const u32 rem = d.w.in_len - o;
          v = (o < d.w.in_len && rem < 4u) ? v & ((1u << (8u * rem)) - 1u) : v;
          if ((u32) c < nck) sh->stage[(u32) c * 64u + lane] = SWAP16(v);
        }
      }
      else {
#pragma unroll
        for (int c = 0; c < NCH; c++) sv[c] = (u32) c < nck ? d.w.load_chunk((u32) c, lane) : 0u;
#pragma unroll
        for (int c = 0; c < NCH; c++) if ((u32) c < nck) sh->stage[(u32) c * 64u + lane] = SWAP16(sv[c]);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    PHE(6);
    // Stretches: equal in bits, but lane 0's is only as long as the others' FIRST walk (their last LZX_LANE_TAIL bits):
    // lane 0 starts at a real token and walks its whole stretch in the first round, while the others find their exits --
    // with equal stretches that round lasted as long as a full walk.
    const u32 Lb = e0 - b0;
    u32 S = (Lb + 63u) >> 6; if (S < 64u) S = 64u;
    u32 S0 = S;
    if (S > LZX_LANE_TAIL + 64u) { S0 = LZX_LANE_TAIL; S = (Lb - S0 + 62u) / 63u; }
    const u32 nl = Lb <= S0 ? 1u : 1u + (Lb - S0 + S - 1u) / S;
    const u32 rstart = lane == 0u ? b0 : b0 + S0 + (lane - 1u) * S;
    u32 rend = rstart + (lane == 0u ? S0 : S); if (rend > e0) rend = e0;
    u32 entry = lane == 0u ? b0 : (rend > rstart + LZX_LANE_TAIL ? rend - LZX_LANE_TAIL : rstart);
    u32 n = 0, nb = 0, nmr = 0, exitp = entry, stop_at = 0;      // tokens / output bytes / matches of the stretch
    bool dead = false, changed = lane < nl;
    // checkpoints of the lane's walk, one per LZX_SEG tokens: bit position | output bytes so far << 16, and matches so far
    // (a byte each).  All walking lanes take a token per step, so the capture is a wave-uniform branch every LZX_SEG steps.
    u32 ckA1 = 0, ckA2 = 0, ckA3 = 0, ckA4 = 0, ckA5 = 0, ckA6 = 0, ckA7 = 0, ckM0 = 0, ckM1 = 0;
    for (u32 round = 0; ; ) {
      // ---- the lanes whose entry moved walk their stretch: token lengths, output lengths ----
      u32 p = entry, cnt = 0, cb = 0, cm = 0, sa = 0;
      bool dd = false;
      for (u32 it = 0; ; it++) {
        const bool act = changed && p < rend;
        if (!ballot(act)) break;
        if ((it & (LZX_SEG - 1u)) == 0u && it != 0u && it < 8u * LZX_SEG) {
          // (a lane that has stopped keeps cnt < it: its checkpoints beyond its last token are never used)
          const u32 a = p | (cb << 16), k = it / LZX_SEG;
          if (changed) {
            if (k == 1u) ckA1 = a; else if (k == 2u) ckA2 = a; else if (k == 3u) ckA3 = a; else if (k == 4u) ckA4 = a;
            else if (k == 5u) ckA5 = a; else if (k == 6u) ckA6 = a; else ckA7 = a;
            if (k <= 4u) ckM0 = (ckM0 & ~(0xFFu << (8u * (k - 1u)))) | (cm << (8u * (k - 1u)));
            else ckM1 = (ckM1 & ~(0xFFu << (8u * (k - 5u)))) | (cm << (8u * (k - 5u)));
          }
        }
        PHCNT(0, 1u);
        PHCNT(4, round >= 2u ? 1u : 0u);                   /* (steps of the walks behind the second) */
        LZX_MARK("emit_count_step_begin");
        STAGE_BITS(act ? p : 0u, w0, w1, ALIGNED)
        const EmitTok t = lzx_emit_token<ALIGNED, false>(sh, act, length_empty, mlim, llim, main_fov, len_fov, w0, w1, two_level);
        const bool ok = act && !t.unk, die = act && t.unk;
        dd = dd || die; sa = die ? p : sa;
        cnt += ok ? 1u : 0u; cb += ok ? t.olen : 0u; cm += (ok && t.is_match) ? 1u : 0u;
        p = die ? rend : p + (ok ? t.tot : 0u);
        LZX_MARK("emit_count_step_end");
      }
      if (changed) { n = cnt; nb = cb; nmr = cm; exitp = p; dead = dd; stop_at = sa; }
      round++;
      PHCNT(1, 1u);
      const u32 pe = (u32) __builtin_amdgcn_ds_bpermute((int)(((lane - 1u) & 63u) << 2), (int) exitp);
      const u32 ne = lane == 0u ? b0 : pe;
      changed = lane < nl && ne != entry;
      entry = ne;
      PHCNT(5, round >= 2u ? (u32) __popcll(ballot(changed)) : 0u);      /* (lanes that walk again behind the second walk) */
      if (!ballot(changed) || round >= LZX_LANE_ROUNDS) break;
    }
    // ---- the consistent prefix: lanes < mm ----
    u32 m = nl;
    { const u64 chm = ballot(changed); if (chm) m = (u32) __ffsll((long long) chm) - 1u; }
    u32 mm = m, dl = 0;
    bool hit = false;
    { const u64 dm = ballot(dead && lane < m); if (dm) { dl = (u32) __ffsll((long long) dm) - 1u; mm = dl + 1u; hit = true; } }
    const u32 cvb = lane < mm ? nb : 0u, cvm = lane < mm ? nmr : 0u;
    const u32 inclb = wave_incl_scan(cvb), inclm = wave_incl_scan(cvm);
    PHE(7);
    // room for this pass's match records (taken from the launch's pool, a chunk at a time): without it the frame ends here
    if (!W.ensure(tt + (mm ? rdl(inclm, mm - 1u) : 0u), lane)) { stop = true; break; }
    // ---- last walk, BALANCED: the pass's tokens are cut into segments of LZX_SEG tokens (the lanes' checkpoints) and
    // segment r * 64 + l goes to lane l in round r.  Every lane then decodes the same number of tokens per round (the
    // stretches are equal in bits, not in tokens: the longest one used to set the pace), and the 64 segments of a round
    // are NEIGHBOURS in the output and in the record list: a round writes ~2 KiB of adjacent literals and ~3 KiB of
    // adjacent records whose cache lines are complete when the round ends, instead of 64 lines per store that the
    // XCD's L2 has dropped again before the lane's next store to them arrives (traffic).
    u32 segc = lane < mm ? (n + LZX_SEG - 1u) / LZX_SEG : 0u;
    if (segc > 8u) segc = 8u;                                     // (a stretch of more than 8 segments: the last one is long)
    const u32 seginc = wave_incl_scan(segc);
    const u32 T = rdl(seginc, 63u);
    // (512 bytes of scratch: the sorted symbols are not needed once the second-level table stands
    // -- or, without one, the block header's input window: NOT the code lengths, a later header of this frame works on them)
    u8 *const owner = two_level ? (u8 *) sh->main_sorted : (u8 *) sh->inbuf;
    for (u32 q = 0; q < segc; q++) owner[seginc - segc + q] = (u8) lane;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    const u32 info0 = entry | (n << 16), info1 = P + inclb - cvb, info2 = tt + inclm - cvm, info3 = (seginc - segc) | (segc << 16);
    bool have_bad = false;
    u32 bad_s = 0xFFFFu, bad_pos = 0, bad_j = 0, bad_p = 0;
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
      const u32 ntok = sact ? (k + 1u == osegc ? on_ - k * LZX_SEG : LZX_SEG) : 0u;
      u32 p = ca & 0xFFFFu, i = 0, pos = i1_ + (k == 0u ? 0u : ca >> 16), j = i2_ + cmk;
      bool cross = false;
      for (;;) {
        const bool on = i < ntok && pos < plimit && !cross;
        if (!ballot(on)) break;
        PHCNT(2, 1u);
        LZX_MARK("emit_last_step_begin");
        STAGE_BITS(on ? p : 0u, w0, w1, true)
        const EmitTok t = lzx_emit_token<ALIGNED, true>(sh, on, length_empty, mlim, llim, main_fov, len_fov, w0, w1, two_level);
        const bool lit = on && !t.is_match;
        const bool crs = on && t.is_match && pos + t.olen > plimit;   // lzxd.c:678-693: the serial path reports it
        const bool mt = on && t.is_match && !crs;
        if (lit) {
#ifdef LZX_LIT_RING
          // (inside the ring's window: into LDS, written out row by row behind the round; a literal beyond it -- long matches
          // between the segments -- goes out on its own)
          if (pos >= edge_n) { if (pos - lit_flushed < LZX_LIT_RING) ((u8 *) sh->litring)[pos & (LZX_LIT_RING - 1u)] = (u8) t.sym; else gst_stream(fout + pos, (u8) t.sym); }
#else
          if (pos >= edge_n) gst_stream(fout + pos, (u8) t.sym);
#endif
          else { gst(&rec->edge_lit[pos], (u8) t.sym); atomicOr(&sh->cnt[pos >> 5], 1u << (pos & 31u)); }
        }
        // (an offset beyond the field -- only garbage decodes to one -- is recorded as 0: never valid, lzx_pipe_commit stops there)
        if (mt) gst_record(W.at(j), make_uint2(frame_pos + pos, (t.expl ? ((t.off < (1u << 21) ? t.off : 0u) << 11) : 0u) | (t.olen << 2) |
                                                         (t.expl ? 0u : t.slot + 1u)));
        cross = cross || crs;
        const bool adv = lit || mt;
        pos += lit ? 1u : (mt ? t.olen : 0u); j += mt ? 1u : 0u;
        p += adv ? t.tot : 0u; i += adv ? 1u : 0u;
        LZX_MARK("emit_last_step_end");
      }
      if (sact && !have_bad && (i < ntok || cross)) { have_bad = true; bad_s = sg; bad_pos = pos; bad_j = j; bad_p = p; }
#ifdef LZX_LIT_RING
      {
        // The round's 64 segments are neighbours in the output: what they left in the ring goes out as whole 16-byte rows (the
        // bytes of the matches in between are whatever the ring held -- they are not final before the frame's matches are
        // copied, lzx_pipe_resolve).  Rows up to the last complete one; the rest waits for the next round.  A round that
        // outran the ring stored its far literals itself: the rows behind the window are skipped for good.
        u32 rmax = rdl(wave_incl_max(sact ? pos : 0u), 63u);
        if (rmax > frame_size) rmax = frame_size;
        if (rmax > lit_flushed) {
          const bool outran = rmax - lit_flushed > LZX_LIT_RING;
          const u32 upto = outran ? (rmax + 15u) & ~15u : rmax & ~15u;
          u32 lim = upto; if (outran) lim = lit_flushed + LZX_LIT_RING;
          if (lim > (frame_size & ~15u)) lim = frame_size & ~15u;
          __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
          for (u32 row = lit_flushed + 16u * lane; row < lim; row += 16u * WAVE)
            gst_row((uint4 *)(fout + row), *(const uint4 *)((const u8 *) sh->litring + (row & (LZX_LIT_RING - 1u))));
          if (upto > lit_flushed) lit_flushed = upto;
        }
      }
#endif
    }
    PHE(8);
    // ---- where did this pass get to?  the first segment that was not emitted completely ends the frame ----
    u32 smin = have_bad ? bad_s : 0xFFFFu;
#pragma unroll
    for (u32 dlt = 1; dlt < WAVE; dlt <<= 1) {
      const u32 ot = (u32) __builtin_amdgcn_ds_bpermute((int)((lane ^ dlt) << 2), (int) smin);
      smin = ot < smin ? ot : smin;
    }
    smin = rfl(smin);
    if (smin != 0xFFFFu) {
      const u32 kq = smin & 63u;
      P = rdl(bad_pos, kq); tt = rdl(bad_j, kq); B = sb_bit + rdl(bad_p, kq); stop = true;
    }
    else {
      if (mm) { P += rdl(inclb, mm - 1u); tt += rdl(inclm, mm - 1u); }
      if (hit) { B = sb_bit + rdl(stop_at, dl); stop = true; }
      else if (mm == 0u) stop = true;
      else B = sb_bit + rdl(exitp, mm - 1u);
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");        // the stage is rewritten by the next pass
    // Another pass follows (and the launch has wave slots to spare, `stream`): what this one stored -- literals below P,
    // match records below tt -- is published now, so that the unit's commit task works on this frame while its later passes
    // are still being parsed (lzx_pipe_commit).  The edge literals all lie in the first 128 bytes: their mask is complete
    // once P has passed them.
    if (stream && !stop && B < Eall && P < plimit && P >= 128u && tt <= 0x7FFFu) {
      if (lane < 4u) rec->edge_mask[lane] = sh->cnt[lane];
      lzx_status_publish(&rec->prog, tt | (P << 15), lane);
#ifdef MSPACK_WAVE_EMU
      if (lane == 0) emu_test_delay();                             // (emulator test hook: lets the commit task see partial progress)
#endif
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
#if defined(LZX_LIT_RING)
  if (P > lit_flushed && P - lit_flushed <= LZX_LIT_RING) {
    // the last rows (the frame's end, or where the parse stopped): byte by byte behind the last complete row
    const u32 full = P & ~15u;
    for (u32 row = lit_flushed + 16u * lane; row < full; row += 16u * WAVE)
      gst_row((uint4 *)(fout + row), *(const uint4 *)((const u8 *) sh->litring + (row & (LZX_LIT_RING - 1u))));
    const u32 b0 = full > lit_flushed ? full : lit_flushed;
    if (b0 + lane < P) gst(fout + b0 + lane, ((const u8 *) sh->litring)[(b0 + lane) & (LZX_LIT_RING - 1u)]);
  }
#endif
  if (lane < 4u) { const u32 em = sh->cnt[lane]; rec->edge_mask[lane] = em; sh->stage[LZX_STAGE_WORDS + REC_CHUNKS + lane] = em; }
  W.save(sh->stage + LZX_STAGE_WORDS, lane);
  n_rec = tt; end_bit = B; bytes_done = P;
}
#endif  /* LZX_PARSE_ONLY */

// common set-up of the header wave and the parse waves: a decoder on the unit's input, nothing read yet
__device__ __forceinline__ bool lzx_side_setup(LzxDec &d, LzxState &s, const mspack_hip_unit &u, const u8 *in_arena, LzxShared *sh)
{
  d.lane = threadIdx.x; d.sh = sh; d.err = 0;
  d.w.unit = in_arena + u.in_off; d.w.in_len = u.in_len;
  d.w.eofs = (u.flags & MSPACK_HIP_UF_HARD_EOF) ? 0u : 2u;
  d.out = nullptr; d.P = 0; d.lit_buf = 0; d.lit_n = 0;
  d.st_rounds = 0; d.st_unknown = 0;
  for (int k_ = 0; k_ < 10; k_++) d.st_t[k_] = 0;
  d.st_h[0] = d.st_h[1] = d.st_h[2] = 0;
  d.bb = 0; d.bl = 0; d.rbl = 0; d.near_end = false; d.careful = false;
  d.w.origin = 0; d.w.wi = 0; d.w.cur = 0; d.w.nxt = 0;
  s.wsize = 1u << u.window_bits;
  s.wpos = 0; s.frame_posn = 0; s.frame = 0; s.reset_frames = u.reset_frames;
  s.offset = 0; s.length = u.out_len;
  s.intel_filesize = 0; s.intel_started = false; s.length_empty = false;
  s.raw_mode = false; s.raw_pos = 0; s.ref_size = 0;
  s.R0 = s.R1 = s.R2 = 1; s.header_read = false; s.block_remaining = 0; s.block_type = 0; s.block_length = 0;
  static const u16 slots[11] = { 30, 32, 34, 36, 38, 42, 50, 66, 98, 162, 290 };
  const u32 wb = u.window_bits;
  s.num_offsets = (wb >= 15u && wb <= 21u) ? ((u32) slots[wb - 15u] << 3) : 0u;
  return s.num_offsets != 0u;
}

#ifdef LZX_PARSE_ONLY
// ---------------------------------------------------------------------------------------------------
// mspack_lzx_pipe's PARSE task: header + tokens of frame f of unit u, by one wave.
// The block headers of a reset interval are a chain (code lengths are deltas on the previous block's,
// lzxd.c:138-183): the wave takes the previous frame's lengths from that frame's record as soon as its parse wave
// has published them (status HEADER or later), reads its own header at the position the frame table states, publishes
// its lengths, and only then parses its tokens (lzx_parse_emit) -- so the chain costs one header per link, not one
// frame.  It works on guesses (one block per frame, at the table's position) and gives up silently; the serial path stays the judge.
// Waiting is safe: the task it waits for has an earlier ticket (shim.hip), i.e. a live wave is working on it.
// ---------------------------------------------------------------------------------------------------
// the rest of a frame whose first block ended inside it: header, tables, tokens -- block by block to the frame's end.  A real
// call: frames like this are one in a few hundred, and inlined the general case's registers counted against every frame's
// parse (scratch accesses of the task 26 -> 104).  The code lengths of the block that ended are still in LDS (no second-level
// table was built over them), the record's first fields are written, `bytes_done` bytes / `n_rec` records are out.
__device__ __attribute__((noinline)) void lzx_pipe_parse_tail(const mspack_hip_unit *up, const u32 f, const u8 *in_arena, u8 *out_arena,
                                                              LzxFrameRec *urecs, const RecPool pool, LzxShared *sh)
{
  // (where lzx_pipe_parse stopped: LZX_TAIL_ARGS)
  u32 bytes_done = rfl(sh->stage[LZX_STAGE_WORDS + 32u]), n_rec = rfl(sh->stage[LZX_STAGE_WORDS + 33u]), cur_bit = rfl(sh->stage[LZX_STAGE_WORDS + 34u]);
  const u32 n_chunks = rfl(sh->stage[LZX_STAGE_WORDS + 35u]);
  const mspack_hip_unit u = *up;
  const u32 lane = threadIdx.x;
  LzxFrameRec *rec = &urecs[f];
  LzxDec d;
  LzxState s;
  lzx_side_setup(d, s, u, in_arena, sh);
  const u32 *ftab = (const u32 *)(in_arena + (size_t) u.in_chunk * 4u);
  const u32 nreal = (u.out_len + LZX_FRAME - 1u) / LZX_FRAME;
  const u32 fo = rfl(ftab[f]);
  u32 fsz = u.out_len - f * LZX_FRAME; if (fsz > LZX_FRAME) fsz = LZX_FRAME;
  u32 fe = (f + 1u < nreal) ? rfl(ftab[f + 1u]) : u.in_len;
  if (fe > u.in_len || fe <= fo) fe = u.in_len;
  u8 *const fout = out_arena + u.out_off + (size_t) f * LZX_FRAME;
  const u32 edge_n = (128u - (u32)((size_t) fout & 127u)) & 127u;
  RecWriter W;
  W.begin(pool, (u32 *) sh->pre_tab, rec->chunk);
  W.n_chunks = n_chunks;
  bool published = false, failed = false;
  u32 rem = 0, btype = 0, end_bit = cur_bit, pub_p0 = 0, e8flag = 0;
  while (bytes_done < fsz) {
    const u32 seg_p0 = bytes_done;
    if (rem == 0u) {
      lzx_seek_bit(d, cur_bit);
      d.err = 0; s.block_type = 0;
      const bool hok = lzx_block_header(d, s, false) && !d.careful && !d.near_end;
      if (!hok || (s.block_type != 1u && s.block_type != 2u) || s.block_length == 0u) { failed = true; break; }
      rem = s.block_length; btype = s.block_type;
      if (rfl((u32) sh->main_len[0xE8]) != 0u) e8flag = 2u;       // lzxd.c:497
      cur_bit = rfl(d.w.origin) * 8u + rfl(d.cons_bits());
    }
    const u32 need = fsz - seg_p0;
    if (!published && rem >= need) {
      for (u32 i = lane; i < (LZX_MAIN_SYMS + 16) / 4u; i += WAVE) gst((u32 *) rec->main_len + i, ((const u32 *) sh->main_len)[i]);
      for (u32 i = lane; i < (LZX_LEN_SYMS + 70) / 4u; i += WAVE) gst((u32 *) rec->len_len + i, ((const u32 *) sh->len_len)[i]);
      if (lane < 8u) rec->ali_len[lane] = sh->ali_len[lane];
      pub_p0 = seg_p0;
      if (lane == 0) {
        rec->end_bit = cur_bit; rec->block_type = btype; rec->block_length = rem; rec->rem_out = rem - need;
        rec->run_rem = seg_p0 + rem;      // (the block the record may end in, counted from the frame's first byte: lzx_decode_unit)
      }
      lzx_status_publish(&rec->status, LZX_ST_HEADER, lane);
      published = true;
    }
    bool tables = true, two_level = false;
    {
      const int r = huff_build<LZX_LEN_P>(sh->len_len, LZX_LEN_SYMS, 12, sh->len_tab, sh->len_sorted, sh->cnt, d.hr_len, lane, false);
      tables = r != 1;
      s.length_empty = (r == 2);
    }
    if (tables && btype == 2u) tables = !huff_build<LZX_ALI_P>(sh->ali_len, 8, 7, sh->ali_tab, sh->ali_sorted, sh->cnt, d.hr_ali, lane, false);
    if (tables) {
      u32 nsorted = 0;
      tables = !huff_build<LZX_MAIN_P, LZX_MSH, LZX_MTAB_T>(sh->main_len, 256 + (int) s.num_offsets + 64, 12, sh->main_tab, sh->main_sorted,
                                                            sh->cnt, d.hr_main, lane, false, &nsorted);
      if (tables && published) two_level = rfl(lzx_build_sub(sh, d.hr_main, nsorted, lane) ? 1u : 0u) != 0u;
    }
    if (!tables) { failed = !published; break; }
    const u32 plimit = seg_p0 + (rem < need ? rem : need);
    if (btype == 2u) lzx_parse_emit<true>(d, s.length_empty, cur_bit, fe * 8u, fout, f * LZX_FRAME, fsz, edge_n, rec, W, n_rec, end_bit, bytes_done, two_level, false, plimit, false);
    else lzx_parse_emit<false>(d, s.length_empty, cur_bit, fe * 8u, fout, f * LZX_FRAME, fsz, edge_n, rec, W, n_rec, end_bit, bytes_done, two_level, false, plimit, false);
    if (bytes_done < plimit) break;                               // the record ends early: the serial path goes on behind it
    rem -= plimit - seg_p0;
    cur_bit = end_bit;
  }
  // nothing to hand on (a header that is no verbatim / aligned block, tables that do not build, a record that ends in front of
  // the frame's last header): nothing of this frame is used, the chain of code lengths ends here
  if (failed || !published) { lzx_status_publish(&rec->status, LZX_ST_FAILED, lane); return; }
  // a record that ends early must end INSIDE the frame's last block, behind at least one of its tokens
  if (bytes_done < fsz && bytes_done <= pub_p0) { lzx_status_publish(&rec->status, LZX_ST_HDRONLY, lane); return; }
  if (lane == 0) {
    rec->n_tokens = n_rec; rec->end_bit = end_bit; rec->bytes_done = bytes_done;
    rec->flags = rec->flags | e8flag | (s.length_empty ? 1u : 0u);
  }
  lzx_status_publish(&rec->status, LZX_ST_EMITTED, lane);
}

// ---------------------------------------------------------------------------------------------------
// The header chain without the headers on it (round 6).  A block header's code lengths are DELTAS on the previous block's
// (lzxd.c:138-183), so a folder written one block per frame -- this build's encoder, and others' -- chains its frames' parse
// tasks: wait for the frame below, read the own header (~60 us), publish; 512 frames: 32 ms, however many waves there are
// (measured: the whole fold path behind it takes 12).  But WHERE a header's bits end and WHAT it does to the lengths do not
// depend on the lengths it is applied to: every entry comes out as (old[x] + a) mod 17, as (old[x - i] + a) mod 17 for the
// i-th follower (i <= 4) of a run of equal lengths (pretree symbol 19: the run takes its value from ITS FIRST entry's old
// length), or as a value that depends on nothing old (zero runs, and whatever is written over an earlier run's overshoot --
// lzxd.c:159: runs are not clipped).  So a task whose predecessor is not ready reads its header at once, TWICE, against two
// probe vectors -- all zeros, and 1 + (x mod 5): five neighbours all different, none zero -- and keeps, per entry, a and what
// it is relative to (the difference of the two results names it: 0 = nothing, else the probe value of the entry it came
// from); when the frame below publishes, its lengths go through that program (~2 us) instead of through a header decode.
// Whether the frame STARTS with a header is the frame below's to say (rem_out): a frame inside a block throws the
// speculation away, as does a header that does not read the same way twice.  lzx_read_lens itself is untouched -- this is
// its own function applied to two inputs.  The program lives in the input stage's room (nothing is staged before the
// frame's first parse pass): LZX_SPEC_LENS bytes a | rel << 5 (rel 7: absolute), the aligned tree's 8 lengths, then the
// block's type, its length and the bit position behind the header.
// ---------------------------------------------------------------------------------------------------
#define LZX_SPEC_LENS (LZX_MAIN_SYMS + 16u + LZX_LEN_SYMS + 70u)     /* main_len and len_len lie back to back in LDS */
static_assert(LZX_SPEC_LENS + 8u + 16u <= LZX_STAGE_WORDS * 4u, "the header program fits the input stage");
static_assert(offsetof(LzxShared, len_len) == offsetof(LzxShared, main_len) + LZX_MAIN_SYMS + 16u, "main_len and len_len are contiguous");
__device__ __attribute__((noinline)) bool lzx_pipe_spec_header(const mspack_hip_unit *up, const u32 fo, const u8 *in_arena, LzxShared *sh)
{
  const mspack_hip_unit u = *up;
  const u32 lane = threadIdx.x;
  LzxDec d;
  LzxState s;
  if (!lzx_side_setup(d, s, u, in_arena, sh)) return false;
  u8 *const lens = sh->main_len;
  u8 *const prog = (u8 *) sh->stage;
  u32 bt = 0, bl = 0, cb = 0;
  for (u32 run = 0; run < 2u; run++) {
    for (u32 x = lane; x < LZX_SPEC_LENS; x += WAVE) lens[x] = run ? (u8)(1u + x % 5u) : (u8) 0u;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    d.w.seek(fo, lane);
    d.bb = 0; d.bl = 0; d.rbl = 0; d.near_end = false; d.careful = false; d.err = 0;
    s.block_type = 0; s.raw_mode = false;
    const bool hok = lzx_block_header(d, s, false) && !d.careful && !d.near_end;
    if (!hok || (s.block_type != 1u && s.block_type != 2u) || s.block_length == 0u) return false;
    const u32 c = rfl(d.w.origin) * 8u + rfl(d.cons_bits());
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    if (run == 0u) {
      bt = s.block_type; bl = s.block_length; cb = c;
      for (u32 x = lane; x < LZX_SPEC_LENS; x += WAVE) prog[x] = lens[x];
      if (lane < 8u) prog[LZX_SPEC_LENS + lane] = sh->ali_len[lane];
    }
    else {
      bool bad = s.block_type != bt || s.block_length != bl || c != cb;
      for (u32 x = lane; x < LZX_SPEC_LENS; x += WAVE) {
        const u32 a = prog[x], b = lens[x];
        const u32 diff = (b + 17u - a) % 17u;                  // 0: nothing old went into it; else the probe value of the entry that did
        const u32 rel = diff == 0u ? 7u : (x % 5u + 5u - (diff - 1u)) % 5u;
        bad = bad || a > 16u || b > 16u || diff > 5u || (diff != 0u && rel > x);
        prog[x] = (u8)(a | (rel << 5));
      }
      if (ballot(bad)) return false;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  }
  if (lane == 0) {
    u32 *w = (u32 *)(prog + ((LZX_SPEC_LENS + 8u + 3u) & ~3u));
    w[0] = bt; w[1] = bl; w[2] = cb;
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  return true;
}
// the previous block's code lengths (in LDS) through the program
__device__ __forceinline__ void lzx_pipe_apply_header(LzxShared *sh, const u32 lane)
{
  u8 *const lens = sh->main_len;
  const u8 *const prog = (const u8 *) sh->stage;
  u32 nv[(LZX_SPEC_LENS + 63u) / 64u];
#pragma unroll
  for (u32 k = 0; k < (LZX_SPEC_LENS + 63u) / 64u; k++) {
    const u32 x = k * 64u + lane;
    u32 v = 0;
    if (x < LZX_SPEC_LENS) {
      const u32 p = prog[x], a = p & 31u, rel = p >> 5;
      v = rel == 7u ? a : ((u32) lens[x - rel] + a) % 17u;
    }
    nv[k] = v;
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
#pragma unroll
  for (u32 k = 0; k < (LZX_SPEC_LENS + 63u) / 64u; k++) {
    const u32 x = k * 64u + lane;
    if (x < LZX_SPEC_LENS) lens[x] = (u8) nv[k];
  }
  if (lane < 8u) sh->ali_len[lane] = prog[LZX_SPEC_LENS + lane];
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
}

// Returns 1 when the frame's first block ended inside it and lzx_pipe_parse_tail has to go on (its arguments wait in the stage's
// spare words); 0 otherwise.  `spec`: the caller has read the frame's header ahead of the chain (lzx_pipe_spec_header: the
// program is in the stage).  Both are calls of the ticket loop (shim.hip), not of this function: nested, their frames -- and the
// registers this function had to save around them -- added up in every wave's scratch allocation (324 B per lane in round 5).
#define LZX_TAIL_ARGS (LZX_STAGE_WORDS + 32u)                  /* stage words: bytes done, records, bit position, record chunks */
__device__ u32 lzx_pipe_parse(const mspack_hip_unit &u, const mspack_hip_unit *up, const u32 f, const u8 *in_arena, u8 *out_arena,
                              LzxFrameRec *urecs, const RecPool &pool, LzxShared *sh, const bool stream, const bool spec)
{
  const u32 lane = threadIdx.x;
  LzxFrameRec *rec = &urecs[f];
  {
    u32 st = 0;
    if (lane == 0) st = atomicCAS(&rec->status, LZX_ST_NONE, LZX_ST_CLAIMED);
    if (rfl(st) != LZX_ST_NONE) return 0u;                        // the unit's wave was faster: it decodes this frame itself
  }
  LzxDec d;
  LzxState s;
  if (!lzx_side_setup(d, s, u, in_arena, sh) || u.in_len >= (1u << 28)) { lzx_status_publish(&rec->status, LZX_ST_FAILED, lane); return 0u; }
  const u32 *ftab = (const u32 *)(in_arena + (size_t) u.in_chunk * 4u);
  const u32 rf = u.reset_frames;
  const u32 nreal = (u.out_len + LZX_FRAME - 1u) / LZX_FRAME;
  const bool first = rf ? (f % rf) == 0u : f == 0u;
  PHDECL();
  PH0();
  // ---- the state in front of the frame: the code lengths of the last block header and what is left of that block ----
  u32 rem = 0, btype = 0;
  if (first) lzx_reset_state(d, s);
  else {
    const LzxFrameRec *pr = rec - 1;
    u32 ps;
    // (the previous frame's task has an earlier ticket: a live wave holds it.  The bound is a safety net -- giving up means
    // this frame and the ones behind it go to the serial path, never a hang)
    for (u32 tries = 0; ; tries++) {
      ps = lzx_status_load(&pr->status);
      if (ps != LZX_ST_NONE && ps != LZX_ST_CLAIMED) break;
      if (tries >= (1u << 24)) { ps = LZX_ST_FAILED; break; }
      __builtin_amdgcn_s_sleep(8);
    }
    if (ps == LZX_ST_FAILED || ps == LZX_ST_TAKEN) { lzx_status_publish(&rec->status, LZX_ST_FAILED, lane); return 0u; }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    // (1056 bytes, a dword per lane and step)
    for (u32 i = lane; i < (LZX_MAIN_SYMS + 16) / 4u; i += WAVE) ((u32 *) sh->main_len)[i] = gld((const u32 *) pr->main_len + i);
    for (u32 i = lane; i < (LZX_LEN_SYMS + 70) / 4u; i += WAVE) ((u32 *) sh->len_len)[i] = gld((const u32 *) pr->len_len + i);
    if (lane < 8u) sh->ali_len[lane] = gld(&pr->ali_len[lane]);
    rem = rfl(gld(&pr->rem_out)); btype = rfl(gld(&pr->block_type));
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  }
  PH(0);
  // ---- where the frame table says the frame begins ----
  const u32 fo = rfl(ftab[f]);
  bool ok = !(fo >= u.in_len || u.in_len - fo <= 64u);         // the last bytes of the input belong to the EOF-exact reader
  u32 intel = 0;
  if (ok) {
    d.w.seek(fo, lane);
    d.bb = 0; d.bl = 0; d.rbl = 0; d.near_end = false; d.careful = false; d.err = 0;
    if (first) {                                                // the interval's (stream's) 1 + 32 header bits, lzxd.c:447-453
      u32 v, hi = 0, lo = 0;
      ok = d.read_bits(1, v);
      if (ok && v) ok = d.read_bits(16, hi) && d.read_bits(16, lo);
      intel = (hi << 16) | lo;
    }
  }
  if (!ok || (rem != 0u && btype != 1u && btype != 2u)) { lzx_status_publish(&rec->status, LZX_ST_FAILED, lane); return 0u; }
  u32 fsz = u.out_len - f * LZX_FRAME; if (fsz > LZX_FRAME) fsz = LZX_FRAME;
  u32 fe = (f + 1u < nreal) ? rfl(ftab[f + 1u]) : u.in_len;       // where the table says the frame ends (a hint)
  if (fe > u.in_len || fe <= fo) fe = u.in_len;
  u8 *const fout = out_arena + u.out_off + (size_t) f * LZX_FRAME;
  // the frame's first bytes up to the next 128-byte line: another wave may be writing that line (see lzx_parse_emit)
  const u32 edge_n = (128u - (u32)((size_t) fout & 127u)) & 127u;
  // ---- the frame's first block (or what is left of the block it lies in).  Round 5: a frame need not be ONE block that begins
  // where it begins (what this build's own encoder writes, and all rounds 2-4 handled here): Microsoft's encoder writes blocks of
  // megabytes (the reference's large-files cabinets: one aligned block of 8 384 624 bytes, then the next), so a frame usually
  // lies INSIDE a block -- it inherits the previous frame's code lengths and has no header at all -- and now and then holds the
  // end of one block and the header and first tokens of the next (lzx_pipe_parse_tail).  The chain from frame to frame is "code
  // lengths + bytes left of the open block" (rem_out); it is published as soon as the LAST header of the frame has been read,
  // i.e. at once for a frame without one. ----
  u32 cur_bit = rfl(d.w.origin) * 8u + rfl(d.cons_bits());     // the frame's first block header, or its first token
  if (lane == 0) {
    // (what does not change any more goes into the record now: fewer values to carry through the parse)
    rec->hdr_start_bit = cur_bit; rec->frame_start_bit = fo * 8u; rec->intel_filesize = intel;
    rec->n_edge = edge_n < fsz ? edge_n : fsz; rec->n_tokens = 0; rec->bytes_done = 0; rec->prog = 0; rec->flags = 0;
  }
  if (rem == 0u && spec) {
    // the header was read ahead of the chain: the previous block's lengths go through its program
    lzx_pipe_apply_header(sh, lane);
    const u32 *w = (const u32 *)((const u8 *) sh->stage + ((LZX_SPEC_LENS + 8u + 3u) & ~3u));
    btype = rfl(w[0]); rem = rfl(w[1]); cur_bit = rfl(w[2]);
    s.block_type = btype; s.block_length = rem;
#if defined(MSPACK_WAVE_EMU)                                   /* emulator analysis runs: which frames took their header this way */
    if (lane == 0 && getenv("MSPACK_EMU_SPEC_TRACE")) fprintf(stderr, "lzx_pipe_parse: frame %u: header read ahead of the chain (block type %u, %u bytes)\n", f, btype, rem);
#endif
    if (lane == 0 && sh->main_len[0xE8] != 0) rec->flags = 2u;   // lzxd.c:497
  }
  else if (rem == 0u) {
    s.block_type = 0;
    const bool hok = lzx_block_header(d, s, false) && !d.careful && !d.near_end;
    if (!hok || (s.block_type != 1u && s.block_type != 2u) || s.block_length == 0u) { lzx_status_publish(&rec->status, LZX_ST_FAILED, lane); return 0u; }
    rem = s.block_length; btype = s.block_type;
    if (lane == 0 && sh->main_len[0xE8] != 0) rec->flags = 2u;   // lzxd.c:497: a block header with a code for 0xE8
    cur_bit = rfl(d.w.origin) * 8u + rfl(d.cons_bits());         // the block's first token
  }
  else s.block_type = btype;
  const bool published = rem >= fsz;
  if (published) {
    // the state behind this frame is known: the next frame's task may go on
    PH(1);
    for (u32 i = lane; i < (LZX_MAIN_SYMS + 16) / 4u; i += WAVE) gst((u32 *) rec->main_len + i, ((const u32 *) sh->main_len)[i]);
    for (u32 i = lane; i < (LZX_LEN_SYMS + 70) / 4u; i += WAVE) gst((u32 *) rec->len_len + i, ((const u32 *) sh->len_len)[i]);
    if (lane < 8u) rec->ali_len[lane] = sh->ali_len[lane];
    if (lane == 0) { rec->end_bit = cur_bit; rec->block_type = btype; rec->block_length = rem; rec->rem_out = rem - fsz; rec->run_rem = rem; }
    lzx_status_publish(&rec->status, LZX_ST_HEADER, lane);      // the next frame's wave may go on
    PH(2);
  }
  // ---- tables (cf. lzx_parse_frame): length and aligned trees first, the main tree last -- its second level takes the room of
  // the code lengths (which are in the record by then; not while a later header of this frame still works on them) ----
  bool tables = true, two_level = false;
  {
    const int r = huff_build<LZX_LEN_P>(sh->len_len, LZX_LEN_SYMS, 12, sh->len_tab, sh->len_sorted, sh->cnt, d.hr_len, lane, false);
    tables = r != 1;
    s.length_empty = (r == 2);
  }
  if (tables && btype == 2u) tables = !huff_build<LZX_ALI_P>(sh->ali_len, 8, 7, sh->ali_tab, sh->ali_sorted, sh->cnt, d.hr_ali, lane, false);
  if (tables) {
    u32 nsorted = 0;
    tables = !huff_build<LZX_MAIN_P, LZX_MSH, LZX_MTAB_T>(sh->main_len, 256 + (int) s.num_offsets + 64, 12, sh->main_tab, sh->main_sorted,
                                                          sh->cnt, d.hr_main, lane, false, &nsorted);
    if (tables && published) two_level = rfl(lzx_build_sub(sh, d.hr_main, nsorted, lane) ? 1u : 0u) != 0u;
  }
  if (!tables) { lzx_status_publish(&rec->status, published ? LZX_ST_HDRONLY : LZX_ST_FAILED, lane); return 0u; }
  PH(3);
  u32 n_rec = 0, end_bit = 0, bytes_done = 0;
  u32 n_chunks = 0;
  {
    // (the frame's chunk list in LDS: the room of the pretree's table -- only a block header uses that; a later header of this
    // frame finds the list put aside, lzx_parse_emit)
    RecWriter W;
    W.begin(pool, (u32 *) sh->pre_tab, rec->chunk);
    const u32 plimit = rem < fsz ? rem : fsz;
    if (btype == 2u) lzx_parse_emit<true>(d, s.length_empty, cur_bit, fe * 8u, fout, f * LZX_FRAME, fsz, edge_n, rec, W, n_rec, end_bit, bytes_done, two_level, stream, plimit, true);
    else lzx_parse_emit<false>(d, s.length_empty, cur_bit, fe * 8u, fout, f * LZX_FRAME, fsz, edge_n, rec, W, n_rec, end_bit, bytes_done, two_level, stream, plimit, true);
    n_chunks = W.n_chunks;
  }
  if (!published) {
    // the block ends inside the frame.  Parsed up to its end: the next header is read THERE (a real call: the hot path above
    // does not carry the general case's registers).  Not that far: nothing to hand on -- the chain of code lengths ends here
    if (bytes_done == rem) {
      if (lane == 0) { sh->stage[LZX_TAIL_ARGS] = bytes_done; sh->stage[LZX_TAIL_ARGS + 1u] = n_rec; sh->stage[LZX_TAIL_ARGS + 2u] = end_bit; sh->stage[LZX_TAIL_ARGS + 3u] = n_chunks; }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      PHFLUSH();
      return 1u;
    }
    lzx_status_publish(&rec->status, LZX_ST_FAILED, lane);
    PHFLUSH();
    return 0u;
  }
  // a record that ends early must end behind at least one token of the block: the serial path goes on from its last bit with
  // this block's tables.  Else: code lengths only
  if (bytes_done < fsz && bytes_done == 0u) { lzx_status_publish(&rec->status, LZX_ST_HDRONLY, lane); PHFLUSH(); return 0u; }
  if (lane == 0) {
    rec->n_tokens = n_rec; rec->end_bit = end_bit; rec->bytes_done = bytes_done;
    rec->flags = rec->flags | (s.length_empty ? 1u : 0u);
  }
  PH(4);
  lzx_status_publish(&rec->status, LZX_ST_EMITTED, lane);
  PH(5);
#ifdef LZX_PIPE_TRACE
  pha_[6] = d.st_t[6]; pha_[7] = d.st_t[7]; pha_[8] = d.st_t[8];
  pha_[12] = d.st_t[0]; pha_[13] = d.st_t[1]; pha_[14] = d.st_t[2]; pha_[15] = d.st_t[3];
  pha_[9] = d.st_t[4]; pha_[10] = d.st_t[5];
#endif
  PHFLUSH();
  return 0u;
}
#endif  /* LZX_PARSE_ONLY */

#ifndef LZX_PARSE_ONLY
// an adopted record's code lengths back into LDS (a later block header works on them, lzxd.c:138-183) ...
__device__ __forceinline__ void lzx_restore_lens(LzxDec &d, const LzxFrameRec *rec)
{
  LzxShared *sh = d.sh;
  for (u32 i = d.lane; i < LZX_MAIN_SYMS + 16; i += WAVE) sh->main_len[i] = rec->main_len[i];
  for (u32 i = d.lane; i < LZX_LEN_SYMS + 70; i += WAVE) sh->len_len[i] = rec->len_len[i];
  if (d.lane < 8u) sh->ali_len[d.lane] = rec->ali_len[d.lane];
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
}
// ... and its decode tables, when the serial path has to finish the block itself
__device__ __forceinline__ void lzx_restore_tables(LzxDec &d, LzxState &s)
{
  LzxShared *sh = d.sh;
  huff_build<LZX_MAIN_P, LZX_MSH, LZX_MTAB_T>(sh->main_len, 256 + (int) s.num_offsets + 64, 12, sh->main_tab, sh->main_sorted,
                                               sh->cnt, d.hr_main, d.lane, false);
  const int r = huff_build<LZX_LEN_P>(sh->len_len, LZX_LEN_SYMS, 12, sh->len_tab, sh->len_sorted, sh->cnt, d.hr_len, d.lane, false);
  s.length_empty = (r == 2);
  if (s.block_type == 2u) huff_build<LZX_ALI_P>(sh->ali_len, 8, 7, sh->ali_tab, sh->ali_sorted, sh->cnt, d.hr_ali, d.lane, false);
}

// ---------------------------------------------------------------------------------------------------
// lzx_pipe_resolve -- second half of a frame's task in mspack_lzx_pipe (round 4; round 3 had one COMMIT task per unit that
// walked the unit's frames one after the other: a serial chain per unit at the end of every launch).  The wave that parsed
// frame f (lzx_pipe_parse: literals stored, one record per match) waits until frame f - 1 is complete -- its task has an
// earlier ticket, so a live wave holds it --, checks that the frame continues the unit exactly where that frame ended
// (bit position, one block of the frame's size), stores the few literals the parse left in the record, and runs down
// the match records 64 at a time: R0-R2 resolved along the list (lzxd.c:565-586; the same prefix scan as
// lzx_commit_batch), the reference's source checks (lzxd.c:613-634), the copies through the position-space resolver
// (spec_queue.hpp).  Then it publishes the frame as complete (`chain` word; R0-R2 behind its last match for the next
// frame).  The first frame that is not a complete regular one ends the unit's chain: its task leaves, in the unit's first
// record, where serial decoding has to resume (frame, output position, bit position, R0-R2) and mspack_decode_lzx
// (launched behind the pipe) skips what is done, finishes the rest -- at least the last bytes of the input, which always
// belong to the EOF-exact reader -- and reports.  A failed check discards the frame: the serial path decodes it again
// from its first bit and reports the error with the reference's code and byte count.
// All frame tasks are alike (parse + resolve, ~1 ms): a launch's waves finish together instead of waiting for the last
// units' commit chains, and in a unit of many frames the parse of frame f + k runs beside the copies of frame f.
// ---------------------------------------------------------------------------------------------------
#define LZX_CH_OPEN 0u
#define LZX_CH_DONE 1u
#define LZX_CH_ENDED 2u

// one batch of match records: every match's offset through the R0-R2 LRU (lzxd.c:565-586; cf. lzx_commit_batch) and the
// reference's checks (lzxd.c:613-634) -- offsets no linear copy serves (0, beyond the window) end the fast path too.
// Returns false when a check fails.
// the LRU half on its own: the offsets only MOVE (no arithmetic on them), so symbolic values pass through it unchanged
// (lzx_fold.hpp runs it with "R0 / R1 / R2 as they are at the frame's first byte" as placeholders)
__device__ __forceinline__ u32 lzx_lru_batch(const bool ism, const u32 lane, const u32 which, const u32 c1, u32 &R0, u32 &R1, u32 &R2)
{
  const u32 sR0 = R0, sR1 = R1, sR2 = R2;
  u32 vmoff = c1;
  const u64 k1 = ballot(ism && which == 0u);
  if (!ballot(ism && which >= 2u)) {
    const u64 below = k1 & ((1ull << lane) - 1ull);
    const u32 src = below ? 63u - (u32) __clzll((long long) below) : 0u;
    const u32 pv = (u32) __builtin_amdgcn_ds_bpermute((int)(src << 2), (int) c1);
    if (which == 1u) vmoff = below ? pv : sR0;
    if (k1) {
      u64 m = k1;
      const u32 j0 = 63u - (u32) __clzll((long long) m);
      u32 nbv = sR0, ncv = sR1;
      m &= ~(1ull << j0);
      if (m) {
        const u32 j1 = 63u - (u32) __clzll((long long) m);
        nbv = rdl(c1, j1); ncv = sR0;
        m &= ~(1ull << j1);
        if (m) ncv = rdl(c1, 63u - (u32) __clzll((long long) m));
      }
      R0 = rdl(c1, j0); R1 = nbv; R2 = ncv;
    }
  }
  else {
    u32 x = LRU_ID;
    if (ism) x = which == 0u ? (0x010080u | lane) : (which == 2u ? 0x020001u : (which == 3u ? 0x000102u : LRU_ID));
    const u32 Cm = lru_scan(x);
    const u32 e0 = Cm & 0xFFu;
    const u32 pv = (u32) __builtin_amdgcn_ds_bpermute((int)((e0 & 63u) << 2), (int) c1);
    vmoff = (e0 & 0x80u) ? pv : (e0 == 0u ? sR0 : (e0 == 1u ? sR1 : sR2));
    const u32 Cl = rdl(Cm, 63u);
    const u32 f0 = Cl & 0xFFu, f1 = (Cl >> 8) & 0xFFu, f2 = (Cl >> 16) & 0xFFu;
    R0 = (f0 & 0x80u) ? rdl(c1, f0 & 63u) : (f0 == 0u ? sR0 : (f0 == 1u ? sR1 : sR2));
    R1 = (f1 & 0x80u) ? rdl(c1, f1 & 63u) : (f1 == 0u ? sR0 : (f1 == 1u ? sR1 : sR2));
    R2 = (f2 & 0x80u) ? rdl(c1, f2 & 63u) : (f2 == 0u ? sR0 : (f2 == 1u ? sR1 : sR2));
  }
  return vmoff;
}
__device__ __forceinline__ bool lzx_front_batch(const bool ism, const u32 lane, const u32 opos, const u32 olen, const u32 which, const u32 c1,
                                                u32 &R0, u32 &R1, u32 &R2, const u32 frame_pos, const u32 wbase, const u32 wsize, u32 &vmoff_out)
{
  const u32 vmoff = lzx_lru_batch(ism, lane, which, c1, R0, R1, R2);
  vmoff_out = vmoff;
  const u32 wp = opos - wbase;
  const bool b = ism && (wp + olen > wsize || LZX_BAD_SOURCE(vmoff, wp, frame_pos, 0u, wsize) ||
                         vmoff == 0u || vmoff > wsize || vmoff > opos);
  return !ballot(b);
}

// wait until the chain word of the frame below is one of the states a caller can act on
__device__ __forceinline__ u32 lzx_chain_wait(const u32 *p, const bool)
{
  u32 ch = lzx_status_load(p);
  LZX_PIPE_WAIT_BEGIN();
  for (u32 tries = 0; ch == LZX_CH_OPEN && tries < (1u << 24); tries++) {
    __builtin_amdgcn_s_sleep(4);
    ch = lzx_status_load(p);
  }
  LZX_PIPE_WAIT_END();
  return ch;
}

union LzxResolveLds { SpecQueueLds q; };
__device__ void lzx_pipe_resolve(const mspack_hip_unit &u, const u32 f, u8 *out_arena, LzxFrameRec *urecs, const uint2 *pool_base, LzxResolveLds *rl,
                                 const bool merged)
{
  // record j of this frame (wave_common.hpp: RecPool); a batch of 64 that starts at a multiple of 64 lies in one chunk
#define MREC(j_) rec_at(pool_base, rec->chunk, (j_))
  SpecQueueLds *const spq = &rl->q;
  const u32 lane = threadIdx.x;
  u8 *const out = out_arena + u.out_off;
  const u32 rf = u.reset_frames;
  const u32 nreal = (u.out_len + LZX_FRAME - 1u) / LZX_FRAME;
  const u32 wsize = 1u << u.window_bits;
  LzxFrameRec *rec = &urecs[f];
  LzxFrameRec *pr = rec - 1;
  const bool first = rf ? (f % rf) == 0u : f == 0u;
  PHDECL();
  // ---- the frame below: complete?  (Tried in round 4: R0-R2 published as soon as a first pass over the records has
  // resolved them, so that only the copies wait for the frame below -- no gain: a 512-frame folder's chain stayed at 242 us
  // per frame, which is the copies; the first pass is 10 % of a frame's resolve.) ----
  u32 R0 = 1, R1 = 1, R2 = 1, prev_end = 0;
  u32 pch = LZX_CH_DONE;
  if (f != 0u) {
    pch = lzx_chain_wait(&pr->chain, false);
    // (the chain ended below: whoever ended it has said where the serial path resumes.  Still open after the bound: nobody
    // says anything -- no rs_valid, the unit kernel decodes the unit from its first byte)
    if (pch != LZX_CH_DONE) { lzx_status_publish(&rec->chain, LZX_CH_ENDED, lane); return; }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    prev_end = (rfl(pr->end_bit) + 15u) & ~15u;
    if (!first) { R0 = rfl(pr->cR0); R1 = rfl(pr->cR1); R2 = rfl(pr->cR2); }       // (a reset frame: lzxd.c:257-270)
  }
  // ---- this frame's record: written by this wave (`merged`: parse and resolve are one task), else by the wave that holds
  // the frame's parse task -- an earlier ticket ----
  u32 st = lzx_status_load(&rec->status);
  if (!merged) {
    LZX_PIPE_WAIT_BEGIN();
    for (u32 tries = 0; (st == LZX_ST_NONE || st == LZX_ST_CLAIMED || st == LZX_ST_HEADER) && tries < (1u << 24); tries++) {
      __builtin_amdgcn_s_sleep(8);
      st = lzx_status_load(&rec->status);
    }
    LZX_PIPE_WAIT_END();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  PH0();
  u32 fsz = u.out_len - f * LZX_FRAME; if (fsz > LZX_FRAME) fsz = LZX_FRAME;
  const u32 frame_pos = f * LZX_FRAME;
  const u32 wbase = frame_pos & ~(wsize - 1u);                 // linear position of window index 0 in this pass
  const u32 eR0 = R0, eR1 = R1, eR2 = R2;
  u32 n_rec = 0, bytes = 0, end_bit = 0;
  bool bad = st != LZX_ST_EMITTED;
  if (!bad) {
    n_rec = rfl(rec->n_tokens); bytes = rfl(rec->bytes_done); end_bit = rfl(rec->end_bit);
    // (which blocks the frame lies in is the parse tasks' chain: code lengths and what is left of the open block travel from
    // frame to frame with the records, and a record only counts when every frame below it was complete)
    bad = rfl(rec->frame_start_bit) != prev_end || bytes > fsz || n_rec > REC_CHUNK * REC_CHUNKS;
  }
  if (!bad) {
    // ---- the literals of the frame's first cache line ----
    const u32 ne = rfl(rec->n_edge);
    for (u32 i = lane; i < ne; i += WAVE)
      if ((gld(&rec->edge_mask[i >> 5]) >> (i & 31u)) & 1u) gst(out + frame_pos + i, gld(&rec->edge_lit[i]));
    // ---- the match records ----
    SpecQueue Q;
    spq_init(*spq, Q, frame_pos, lane);
    u32 th = 0;
    uint2 cur0 = make_uint2(0u, 0u), cur1 = cur0, cur2 = cur0, cur3 = cur0;
    {
      const uint2 *g0 = rec_group(pool_base, rec->chunk, 0u);     // (groups of four batches: one chunk lookup per 256 records)
      if (th + lane < n_rec) cur0 = gld(g0 + lane);
      if (th + 64u + lane < n_rec) cur1 = gld(g0 + 64u + lane);
      if (th + 128u + lane < n_rec) cur2 = gld(g0 + 128u + lane);
      if (th + 192u + lane < n_rec) cur3 = gld(g0 + 192u + lane);
    }
    for (; th < n_rec && !bad; ) {
      uint2 nx0 = make_uint2(0u, 0u), nx1 = nx0, nx2 = nx0, nx3 = nx0;
      const u32 tb = th + 256u + lane;
      if (th + 256u < n_rec) {
        const uint2 *g1 = rec_group(pool_base, rec->chunk, th + 256u);
        if (tb < n_rec) nx0 = gld(g1 + lane);
        if (tb + 64u < n_rec) nx1 = gld(g1 + 64u + lane);
        if (tb + 128u < n_rec) nx2 = gld(g1 + 128u + lane);
        if (tb + 192u < n_rec) nx3 = gld(g1 + 192u + lane);
      }
#pragma unroll 1
      for (u32 k = 0; k < 4u && th < n_rec && !bad; k++) {
        u32 n = n_rec - th; if (n > 64u) n = 64u;
        const uint2 cur = k == 0u ? cur0 : (k == 1u ? cur1 : (k == 2u ? cur2 : cur3));
        const bool ism = lane < n;
        const u32 opos = cur.x, olen = (cur.y >> 2) & 511u, which = cur.y & 3u, c1 = cur.y >> 11;
        const u64 mm = ballot(ism);
        u32 vmoff = c1;
        // (1) offsets through the R0-R2 LRU, (2) the reference's checks
        if (!lzx_front_batch(ism, lane, opos, olen, which, c1, R0, R1, R2, frame_pos, wbase, wsize, vmoff)) { bad = true; break; }
        PH(9);
        // (3) queue the copies (cf. lzx_commit_batch)
        // (runs -- matches in a row at one offset -- are written as periodic fills, the rest goes through the queue: spec_queue.hpp)
        const u32 newP = rdl(opos + olen, n - 1u);
        spq_push_runs(*spq, Q, out, ism, n, opos, olen, vmoff, lane);
        PH(10);
        if (spq_due(Q, newP)) spq_resolve(*spq, Q, out, newP, false, lane);
        PH(11);
        th += n;
      }
      cur0 = nx0; cur1 = nx1; cur2 = nx2; cur3 = nx3;
    }
    if (!bad) spq_resolve(*spq, Q, out, frame_pos + bytes, true, lane);
    PH(11);
  }
  // ---- the frame is complete: the next frame's task may go on.  Anything else ends the unit's chain here: the serial path
  // (mspack_decode_lzx) resumes at this frame's first bit, or behind its last record when only its end is missing ----
  const bool whole = !bad && bytes == fsz;
  if (lane == 0) {
    if (whole) { rec->cR0 = R0; rec->cR1 = R1; rec->cR2 = R2; }
    if (!whole || f + 1u == nreal) {
      LzxFrameRec *r0 = &urecs[0];
      const bool partial = !bad && !whole;
      r0->rs_frame = whole ? f + 1u : f; r0->rs_partial = partial ? 1u : 0u;
      r0->rs_P = whole ? (f + 1u) * LZX_FRAME : (partial ? frame_pos + bytes : frame_pos);
      r0->rs_next_bit = whole ? ((end_bit + 15u) & ~15u) : (partial ? end_bit : prev_end);
      r0->rs_R0 = bad ? eR0 : R0; r0->rs_R1 = bad ? eR1 : R1; r0->rs_R2 = bad ? eR2 : R2;
      r0->rs_valid = 1u;
    }
  }
  lzx_status_publish(&rec->chain, whole ? LZX_CH_DONE : LZX_CH_ENDED, lane);
  PHFLUSH();
#undef MREC
}
// ---------------------------------------------------------------------------------------------------
// lzx_pipe_resolve_stream -- the resolve task of a launch that has wave slots to spare (round 6; round 3's commit task had this,
// round 4's restructure dropped it, and BASELINE config 3's launch shape -- 1024 intervals: 4096 tickets for 4096 waves -- got slower
// every round since: 1.44 -> 1.58 -> 1.60 ms).  In such a launch every ticket is pulled at once, and a unit's chain is
// P(f0) -> R(f0) -> R(f1): the resolve task of a frame sat idle until the frame's parse task had stored its last record.  Here it
// takes the records up WHILE the frame is parsed: lzx_parse_emit publishes, behind every pass but the last, how many match records
// and output bytes are in memory (`prog`, with the same release recipe as a status word), and this task works through what has
// arrived -- whole groups of 256 records -- one acquire per event.  The frame's chain is then the longer of its parse and its
// resolve, not their sum.  Only where waves are spare (shim.hip: control word 3): a resolve wave that has started on a frame
// holds its slot until the frame's parse task is through.  Same records, same checks, same hand-over as lzx_pipe_resolve.
// ---------------------------------------------------------------------------------------------------
__device__ void lzx_pipe_resolve_stream(const mspack_hip_unit &u, const u32 f, u8 *out_arena, LzxFrameRec *urecs, const uint2 *pool_base, LzxResolveLds *rl)
{
  SpecQueueLds *const spq = &rl->q;
  const u32 lane = threadIdx.x;
  u8 *const out = out_arena + u.out_off;
  const u32 rf = u.reset_frames;
  const u32 nreal = (u.out_len + LZX_FRAME - 1u) / LZX_FRAME;
  const u32 wsize = 1u << u.window_bits;
  LzxFrameRec *rec = &urecs[f];
  LzxFrameRec *pr = rec - 1;
  const bool first = rf ? (f % rf) == 0u : f == 0u;
  u32 R0 = 1, R1 = 1, R2 = 1, prev_end = 0;
  if (f != 0u) {
    const u32 pch = lzx_chain_wait(&pr->chain, false);
    if (pch != LZX_CH_DONE) { lzx_status_publish(&rec->chain, LZX_CH_ENDED, lane); return; }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    prev_end = (rfl(pr->end_bit) + 15u) & ~15u;
    if (!first) { R0 = rfl(pr->cR0); R1 = rfl(pr->cR1); R2 = rfl(pr->cR2); }
  }
  // the frame's parse task: an earlier ticket.  Its header (status HEADER: the record's first fields stand) or its end
  u32 st = lzx_status_load(&rec->status);
  for (u32 tries = 0; (st == LZX_ST_NONE || st == LZX_ST_CLAIMED) && tries < (1u << 24); tries++) {
    __builtin_amdgcn_s_sleep(8);
    st = lzx_status_load(&rec->status);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  u32 fsz = u.out_len - f * LZX_FRAME; if (fsz > LZX_FRAME) fsz = LZX_FRAME;
  const u32 frame_pos = f * LZX_FRAME;
  const u32 wbase = frame_pos & ~(wsize - 1u);
  const u32 eR0 = R0, eR1 = R1, eR2 = R2;
  u32 n_rec = 0, bytes = 0, end_bit = 0;
  bool bad = !(st == LZX_ST_EMITTED || st == LZX_ST_HEADER);
  if (!bad) bad = rfl(gld(&rec->frame_start_bit)) != prev_end;
  bool fin = false;                                          // the parse task has said its last word
  u32 avail = 0, th = 0;
  bool edge_done = false;
  SpecQueue Q;
  spq_init(*spq, Q, frame_pos, lane);
  while (!bad) {
    if (!fin) {
      st = lzx_status_load(&rec->status);
      if (st != LZX_ST_HEADER) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        fin = true;
        if (st != LZX_ST_EMITTED) { bad = true; break; }
        n_rec = rfl(gld(&rec->n_tokens)); bytes = rfl(gld(&rec->bytes_done)); end_bit = rfl(gld(&rec->end_bit));
        if (bytes > fsz || n_rec > REC_CHUNK * REC_CHUNKS || n_rec < avail) { bad = true; break; }
        avail = n_rec;
      }
      else {
        const u32 pg = lzx_status_load(&rec->prog) & 0x7FFFu;
        if (pg > avail) {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); avail = pg;
#if defined(MSPACK_WAVE_EMU)
          if (lane == 0 && getenv("MSPACK_EMU_STREAM_TRACE")) fprintf(stderr, "lzx_pipe_resolve_stream: frame %u: %u records in while the frame is parsed\n", f, avail);
#endif
        }
        else if (th + 256u > avail) { __builtin_amdgcn_s_sleep(8); continue; }
      }
    }
    if (!edge_done && (fin || avail != 0u)) {
      // the literals of the frame's first cache line (their mask is complete once a pass has been published: lzx_parse_emit)
      const u32 ne = rfl(gld(&rec->n_edge));
      for (u32 i = lane; i < ne; i += WAVE)
        if ((gld(&rec->edge_mask[i >> 5]) >> (i & 31u)) & 1u) gst(out + frame_pos + i, gld(&rec->edge_lit[i]));
      edge_done = true;
    }
    // whole groups of 256 records (all that is left once the parse is through)
    while (th < avail && (fin || th + 256u <= avail) && !bad) {
      const uint2 *g0 = rec_group(pool_base, rec->chunk, th);
      uint2 c4[4];
#pragma unroll
      for (u32 k = 0; k < 4u; k++) { c4[k] = make_uint2(0u, 0u); if (th + 64u * k + lane < avail) c4[k] = gld(g0 + 64u * k + lane); }
#pragma unroll 1
      for (u32 k = 0; k < 4u && th < avail && !bad; k++) {
        u32 n = avail - th; if (n > 64u) n = 64u;
        const uint2 cur = k == 0u ? c4[0] : (k == 1u ? c4[1] : (k == 2u ? c4[2] : c4[3]));
        const bool ism = lane < n;
        const u32 opos = cur.x, olen = (cur.y >> 2) & 511u, which = cur.y & 3u, c1 = cur.y >> 11;
        u32 vmoff = c1;
        if (!lzx_front_batch(ism, lane, opos, olen, which, c1, R0, R1, R2, frame_pos, wbase, wsize, vmoff)) { bad = true; break; }
        const u32 newP = rdl(opos + olen, n - 1u);
        spq_push_runs(*spq, Q, out, ism, n, opos, olen, vmoff, lane);
        if (spq_due(Q, newP)) spq_resolve(*spq, Q, out, newP, false, lane);
        th += n;
      }
    }
    if (fin && th >= avail) break;
  }
  if (!bad) {
    if (!edge_done) {
      const u32 ne = rfl(gld(&rec->n_edge));
      for (u32 i = lane; i < ne; i += WAVE)
        if ((gld(&rec->edge_mask[i >> 5]) >> (i & 31u)) & 1u) gst(out + frame_pos + i, gld(&rec->edge_lit[i]));
    }
    spq_resolve(*spq, Q, out, frame_pos + bytes, true, lane);
  }
  const bool whole = !bad && bytes == fsz;
  if (lane == 0) {
    if (whole) { rec->cR0 = R0; rec->cR1 = R1; rec->cR2 = R2; }
    if (!whole || f + 1u == nreal) {
      LzxFrameRec *r0 = &urecs[0];
      const bool partial = !bad && !whole;
      r0->rs_frame = whole ? f + 1u : f; r0->rs_partial = partial ? 1u : 0u;
      r0->rs_P = whole ? (f + 1u) * LZX_FRAME : (partial ? frame_pos + bytes : frame_pos);
      r0->rs_next_bit = whole ? ((end_bit + 15u) & ~15u) : (partial ? end_bit : prev_end);
      r0->rs_R0 = bad ? eR0 : R0; r0->rs_R1 = bad ? eR1 : R1; r0->rs_R2 = bad ? eR2 : R2;
      r0->rs_valid = 1u;
    }
  }
  lzx_status_publish(&rec->chain, whole ? LZX_CH_DONE : LZX_CH_ENDED, lane);
}

#include "lzx_fold.hpp"
#endif  /* !LZX_PARSE_ONLY */
#endif  /* !LZX_DELTA */

#ifndef LZX_PARSE_ONLY
// decode one LZX unit.  frame_meta[frame_base + f] receives the intel_filesize to apply to frame f
// (0 = none).  Returns via *res.
#ifdef LZX_DELTA
__device__ __forceinline__ void lzx_decode_unit(const mspack_hip_unit &u, const u8 *in_arena, u8 *out_arena,
                                int32_t *frame_meta, mspack_hip_result *res, LzxShared *sh)
#else
// recs: the pipe's frame records for this launch (NULL: none), indexed by frame slot; toks: its record pool (not read here)
// resume: the launch ran mspack_lzx_pipe first -- the unit's first record says how far its commit task got (rs_*: so many
// complete frames, possibly part of the next one).  Those frames are not decoded again: only their bookkeeping (interval
// header, E8 decision, offsets, in_next) is replayed from their records, and decoding goes on serially where the pipe stopped
__device__ __forceinline__ void lzx_decode_unit(const mspack_hip_unit &u, const u8 *in_arena, u8 *out_arena,
                                int32_t *frame_meta, mspack_hip_result *res, LzxShared *sh,
                                const LzxFrameRec *recs, const uint2 *toks, const bool resume = false)
#endif
{
  const u32 lane = threadIdx.x;
  LzxDec d;
  LzxState s;
  u32 flags = 0;
  const u32 out_bytes = u.out_len;
  u32 remaining = out_bytes;
  u32 in_next = 0;

  d.lane = lane; d.sh = sh; d.err = 0;
  d.w.unit = in_arena + u.in_off; d.w.in_len = u.in_len;
  d.w.eofs = (u.flags & MSPACK_HIP_UF_HARD_EOF) ? 0u : 2u;
  d.w.seek(0, lane);
  d.bb = 0; d.bl = 0; d.rbl = 0;
  d.near_end = (u.in_len <= 64u); d.careful = d.near_end;
#ifdef LZX_DELTA
  // reference data lies right below the unit's output; positions are biased by its size
  d.out = out_arena + u.out_off - u.ref_len; d.P = u.ref_len;
#else
  d.out = out_arena + u.out_off; d.P = 0;
#endif
  d.lit_buf = 0; d.lit_n = 0;
  d.st_rounds = 0; d.st_unknown = 0;
  for (int k_ = 0; k_ < 10; k_++) d.st_t[k_] = 0;
  d.st_h[0] = d.st_h[1] = d.st_h[2] = 0;

  s.wsize = 1u << u.window_bits;
  s.wpos = 0; s.frame_posn = 0; s.frame = 0; s.reset_frames = u.reset_frames;
  s.offset = 0; s.length = out_bytes;
  s.intel_filesize = 0; s.intel_started = false; s.length_empty = false;
  s.raw_mode = false; s.raw_pos = 0;
  {
    static const u16 slots[11] = { 30, 32, 34, 36, 38, 42, 50, 66, 98, 162, 290 };
    u32 wb = u.window_bits;
#ifdef LZX_DELTA
    s.ref_size = u.ref_len;
    s.num_offsets = (wb >= 17u && wb <= 25u && u.ref_len <= (1u << wb)) ? ((u32) slots[wb - 15u] << 3) : 0u;
#else
    s.ref_size = 0;
    s.num_offsets = (wb >= 15u && wb <= 21u) ? ((u32) slots[wb - 15u] << 3) : 0u;
#endif
  }
  if (s.num_offsets == 0u) {
    if (lane == 0) {
      res->err = ERR_ARGS; res->flags = 0; res->out_len = 0; res->in_used = 0; res->good_len = 0; res->in_next = 0;
#ifndef LZX_DELTA
      if (u.flags & MSPACK_HIP_UF_LZX_LOG) *(u32 *)(out_arena + u.out_off + (((size_t) u.out_len + LZX_FRAME + 15u) & ~(size_t) 15u)) = 0u;
#endif
    }
    return;
  }
  lzx_reset_state(d, s);
#ifdef LZX_EXP_STATS
  u64 tstart_ = __builtin_amdgcn_s_memtime();
#endif
#ifdef LZX_PHASE_TIMERS
  const u64 pt_unit_ = __builtin_amdgcn_s_memtime();
#endif

#ifndef LZX_DELTA
  const bool use_recs = recs != nullptr && (u.flags & MSPACK_HIP_UF_FRAME_TABLE) != 0u;
  const LzxFrameRec *stale = nullptr;       // adopted record whose code lengths / tables are not in LDS (yet)
  bool stale_tables = false;
  // where mspack_lzx_pipe's commit task stopped (resume): rs_frame complete frames, then possibly part of frame rs_frame
  bool rs_on = false, rs_partial = false, rs_inject = false, positioned = true;
  u32 rs_frame = 0, rs_P = 0, rs_next = 0, rs_R0 = 1, rs_R1 = 1, rs_R2 = 1, ff_end = 0;
  if (resume && use_recs) {
    const LzxFrameRec *r0 = &recs[u.frame_base];
    if (rfl(r0->rs_valid) == 1u) {
      rs_on = true; positioned = false;
      rs_frame = rfl(r0->rs_frame); rs_partial = rfl(r0->rs_partial) != 0u; rs_P = rfl(r0->rs_P); rs_next = rfl(r0->rs_next_bit);
      rs_R0 = rfl(r0->rs_R0); rs_R1 = rfl(r0->rs_R1); rs_R2 = rfl(r0->rs_R2);
    }
  }
#endif
#ifndef LZX_DELTA
  u32 *const olog = (u32 *)(out_arena + u.out_off + (((size_t) u.out_len + LZX_FRAME + 15u) & ~(size_t) 15u));
  u32 n_open_resets = 0;
#endif
  if (out_bytes != 0u) {
    const u32 end_frame = out_bytes / LZX_FRAME + 1u;                      // lzxd.c:419
    while (s.frame < end_frame) {
      if (s.reset_frames && (s.frame % s.reset_frames) == 0u) {
#ifndef LZX_DELTA
        // a block that is still open at a reset point: a format error the reference warns about and decodes through
        // (lzxd.c:423-431); MSPACK_HIP_UF_LZX_LOG: the frame goes into the unit's log for the driver's sys->message
        if (s.block_remaining != 0u && (u.flags & MSPACK_HIP_UF_LZX_LOG) != 0u) {
          if (d.lane == 0 && n_open_resets < u.ref_len) olog[1u + n_open_resets] = s.frame;
          n_open_resets++;
        }
#endif
        // a reset in raw mode keeps reading bits from raw_pos (no pad byte: block_type is cleared)
        lzx_reset_state(d, s);
#ifndef LZX_DELTA
        stale = nullptr; stale_tables = false;
#endif
      }
#ifndef LZX_DELTA
      const bool ff = rs_on && s.frame < rs_frame;                   // done by the pipe: bookkeeping only
      const bool pf = rs_on && s.frame == rs_frame && rs_partial;    // partly done: go on behind its last record
      const LzxFrameRec *frec = (ff || pf) ? &recs[u.frame_base + s.frame] : nullptr;
      if (rs_on && s.frame == rs_frame && !rs_partial) {
        // serial decoding starts with this frame: the state the pipe left at its first bit
        lzx_seek_bit(d, rs_next);
        d.P = rs_P; s.R0 = rs_R0; s.R1 = rs_R1; s.R2 = rs_R2;
        if (rs_frame != 0u && !(s.reset_frames && (s.frame % s.reset_frames) == 0u)) {
          stale = &recs[u.frame_base + rs_frame - 1u];
          stale_tables = s.block_remaining != 0u;                  // inside a block the pipe's frames left open: its tables too
        }
        rs_on = false; positioned = true;
      }
#endif
#ifdef LZX_DELTA
      {                                                               // chunk size (lzxd.c:440-444)
        u32 cs;
        if (s.raw_mode) {
          // inside a stored block the bit buffer is empty: ENSURE_BITS(16) reads two bytes, REMOVE drops them
          if (s.raw_pos + 2u > d.w.in_len + d.w.eofs) { d.err = ERR_READ; break; }
          s.raw_pos += 2u;
        }
        else if (!d.read_bits(16, cs)) break;
      }
#endif
      if (!s.header_read) {
        u32 v, hi = 0, lo = 0;
#ifndef LZX_DELTA
        if (frec) { const u32 iv = rfl(frec->intel_filesize); hi = iv >> 16; lo = iv & 0xFFFFu; }   // (its parse wave read the bits)
        else
#endif
        {
          lzx_leave_raw(d, s);
          if (!d.read_bits(1, v)) break;
          if (v) { if (!d.read_bits(16, hi) || !d.read_bits(16, lo)) break; }
        }
        s.intel_filesize = (int32_t)((hi << 16) | lo);
        if (s.intel_filesize) flags |= MSPACK_HIP_F_INTEL_HEADER;
        s.header_read = true;
      }
      u32 frame_size = LZX_FRAME;
      if (s.length && (s.length - s.offset) < frame_size) frame_size = s.length - s.offset;

      int todo = (int)(s.frame_posn + frame_size - s.wpos);
      bool fail = false;
#ifndef LZX_DELTA
      if (ff) {
        // a frame the pipe finished, decoded and in place: the block it ends in and what is left of that block
        s.block_type = rfl(frec->block_type); s.block_length = rfl(frec->block_length); s.block_remaining = rfl(frec->rem_out);
        const u32 rfl_ = rfl(frec->flags);
        s.length_empty = (rfl_ & 1u) != 0u;
        if (rfl_ & 2u) s.intel_started = true;
        flags |= MSPACK_HIP_F_FRAMES_ADOPTED;
        d.P += frame_size; s.wpos += frame_size;
        ff_end = (rfl(frec->end_bit) + 15u) & ~15u;                  // behind the 16-bit realignment (lzxd.c:695-697)
        todo = 0;
      }
      if (pf) {
        // (the block the record ends in, counted as if it had begun with the frame: the loop below takes the frame's bytes off it)
        s.block_type = rfl(frec->block_type);
        s.block_length = rfl(frec->block_length); s.block_remaining = rfl(frec->run_rem);
        const u32 rfl_ = rfl(frec->flags);
        s.length_empty = (rfl_ & 1u) != 0u;
        if (rfl_ & 2u) s.intel_started = true;
        stale = frec; stale_tables = true;
        flags |= MSPACK_HIP_F_FRAMES_ADOPTED;
        rs_inject = true; rs_on = false;
      }
#endif
      while (todo > 0) {
#ifdef LZX_EXP_STATS
        u64 t0_ = __builtin_amdgcn_s_memtime();
#endif
#ifndef LZX_DELTA
        if (s.block_remaining == 0u && stale) { lzx_restore_lens(d, stale); stale = nullptr; stale_tables = false; }
#endif
        if (s.block_remaining == 0u) { if (!lzx_block_header(d, s)) { fail = true; break; } }
#ifdef LZX_EXP_STATS
        d.st_unknown += (u32)((__builtin_amdgcn_s_memtime() - t0_) >> 6);
#endif
        int run = (int) s.block_remaining;
        if (run > todo) run = todo;
        todo -= run; s.block_remaining -= (u32) run;

        if (s.block_type == 1u || s.block_type == 2u) {
          // ---------------- the hot loop (lzxd.c:538-651) ----------------
          const bool aligned = (s.block_type == 2u);
          const u32 run_end = d.P + (u32) run;
          const u32 wbase = d.P - s.wpos;          // linear position of window index 0
          bool respec = true;                      // try the speculative path (again)
#ifndef LZX_DELTA
          if (rs_inject) {
            // the pipe committed this frame's records up to rs_P: go on from the bit behind the last of them
            rs_inject = false; positioned = true;
            d.flush_lits();
            d.P = rs_P; s.R0 = rs_R0; s.R1 = rs_R1; s.R2 = rs_R2;
            lzx_seek_bit(d, rs_next);
          }
          if (!fail && d.P < run_end && stale_tables) {      // the record did not reach the end of the run
            if (stale) { lzx_restore_lens(d, stale); stale = nullptr; }
            lzx_restore_tables(d, s); stale_tables = false;
          }
          if (fail) break;
#endif
          while (d.P < run_end) {
            if (respec && !d.careful && !d.near_end) {
#ifndef LZX_DELTA
              int rc = aligned ? lzx_run_spec2<true>(d, s, run_end, wbase) : lzx_run_spec2<false>(d, s, run_end, wbase);
#else
              int rc = aligned ? lzx_run_spec<true>(d, s, run_end, wbase) : lzx_run_spec<false>(d, s, run_end, wbase);
#endif
              if (rc == LZX_RUN_FAIL) { fail = true; break; }
              if (d.P >= run_end) break;
            }
#ifdef LZX_EXP_STATS
            u64 ts_ = __builtin_amdgcn_s_memtime();
#define TS9() d.st_t[9] += (u32)(__builtin_amdgcn_s_memtime() - ts_)
#else
#define TS9() do { } while (0)
#endif
#ifdef LZX_DELTA
            respec = true;                           // it hands single tokens over (extended match lengths)
#else
            respec = false;
#endif
            if (d.bl <= 32) d.refill();
            int sym = d.decode_sym<LZX_MAIN_P, LZX_MSH, LZX_MTAB_T>(sh->main_tab, sh->main_sorted, d.hr_main);
            if (sym < 0) { fail = true; break; }
            if (sym < 256) {
              d.lit_buf = wrl(d.lit_buf, (u32) sym, d.lit_n);
              d.lit_n++; d.P++;
              if (d.lit_n == WAVE) d.flush_lits();
              TS9();
              continue;
            }
            u32 m = (u32) sym - 256u, slot = m >> 3, len = m & 7u, off;
            if (len == 7u) {
              if (s.length_empty) { d.err = ERR_DECRUNCH; fail = true; break; }
              int foot = d.decode_sym<LZX_LEN_P>(sh->len_tab, sh->len_sorted, d.hr_len);
              if (foot < 0) { fail = true; break; }
              len += (u32) foot;
            }
            len += 2u;
            if (slot == 0u) off = s.R0;
            else if (slot == 1u) { off = s.R1; s.R1 = s.R0; s.R0 = off; }
            else if (slot == 2u) { off = s.R2; s.R2 = s.R0; s.R0 = off; }
            else {
              // position_base / extra_bits from their closed form (lzxd.c:202-207)
              u32 extra = slot < 4u ? 0u : (slot < 36u ? (slot >> 1) - 1u : 17u);
              u32 base = slot < 4u ? slot : (slot < 36u ? ((2u + (slot & 1u)) << extra) : ((slot - 34u) << 17));
              off = base - 2u;
              if (d.bl <= 32) d.refill();
              if (extra >= 3u && aligned) {
                if (extra > 3u) { u32 vb; if (!d.read_bits((int) extra - 3, vb)) { fail = true; break; } off += vb << 3; }
                int a = d.decode_sym<LZX_ALI_P>(sh->ali_tab, sh->ali_sorted, d.hr_ali);
                if (a < 0) { fail = true; break; }
                off += (u32) a;
              }
              else if (extra) { u32 vb; if (!d.read_bits((int) extra, vb)) { fail = true; break; } off += vb; }
              s.R2 = s.R1; s.R1 = s.R0; s.R0 = off;
            }
#ifdef LZX_DELTA
            if (len == 257u) {                                          // lzxd.c:588-611
              u32 p3, x;
              if (d.bl <= 32) d.refill();
              if (d.careful && !d.ref_ensure(3)) { fail = true; break; }
              p3 = (u32)(d.bb >> 61);
              if ((p3 & 4u) == 0u)      { d.drop(1); if (!d.read_bits(8, x)) { fail = true; break; } }
              else if ((p3 >> 1) == 2u) { d.drop(2); if (!d.read_bits(10, x)) { fail = true; break; } x += 0x100u; }
              else if (p3 == 6u)        { d.drop(3); if (!d.read_bits(12, x)) { fail = true; break; } x += 0x500u; }
              else                      { d.drop(3); if (!d.read_bits(15, x)) { fail = true; break; } }
              len += x;
            }
#endif
            u32 wp = d.P - wbase;
            // a match running past the run is an error in every case (lzxd.c:678-693); test it
            // before copying so that nothing is ever written past the unit's output
            if (d.P + len > run_end) { d.err = ERR_DECRUNCH; fail = true; break; }
            if (wp + len > s.wsize) { d.err = ERR_DECRUNCH; fail = true; break; }       // lzxd.c:613
            if (LZX_BAD_SOURCE(off, wp, s.offset, s.ref_size, s.wsize)) { d.err = ERR_DECRUNCH; fail = true; break; }
            d.flush_lits();
            if (off != 0u && off <= s.wsize) lzx_copy_match(d.out, d.P, off, len, lane);
            else { if (lane == 0) lzx_copy_match_odd(d.out, d.P, wp, s.wsize, off, len); }
            d.P += len;
            TS9();
          }
          d.flush_lits();
          if (fail) break;
          s.wpos = d.P - wbase;
          run = (int)(run_end - d.P);              // <= 0: overrun of the last match
        }
        else if (s.block_type == 3u) {
          // stored bytes: coalesced copy input -> output (lzxd.c:654-671)
          u32 n = (u32) run;
          if (s.raw_pos + n > d.w.in_len + d.w.eofs || s.raw_pos + n < s.raw_pos) { d.err = ERR_READ; fail = true; break; }
          for (u32 i = lane; i < n; i += WAVE) d.out[d.P + i] = (u8) d.w.byte_at(s.raw_pos + i);
          s.raw_pos += n; d.P += n; s.wpos += n;
          run = 0;
        }
        else { d.err = ERR_DECRUNCH; fail = true; break; }

        if (run < 0) {                                                          // lzxd.c:678-685
          if ((u32)(-run) > s.block_remaining) { d.err = ERR_DECRUNCH; fail = true; break; }
          s.block_remaining -= (u32)(-run);
        }
      }
      if (fail) break;
      if ((s.wpos - s.frame_posn) != frame_size) { d.err = ERR_DECRUNCH; break; }  // lzxd.c:689

      // re-align the bitstream to 16 bits (lzxd.c:695-697)
#ifndef LZX_DELTA
      if (ff) {                    // (a frame the pipe finished: its record says where the stream goes on)
        in_next = ff_end >> 3;
        flags = s.block_remaining ? (flags | MSPACK_HIP_F_BLOCK_OPEN) : (flags & ~MSPACK_HIP_F_BLOCK_OPEN);
      }
      else
#endif
      {
        if (!s.raw_mode) {
          if (d.careful) { if (d.rbl > 0 && !d.ref_ensure(16)) break; }
          int n = d.bl & 15;
          if (d.bl < n) d.refill();
          if (n) d.drop(n);
        }
        if (frame_size) {            // for callers that chain units (CHM reset intervals): where the next frame starts
          in_next = s.raw_mode ? s.raw_pos : d.w.origin + (d.cons_bits() >> 3);
          flags = s.block_remaining ? (flags | MSPACK_HIP_F_BLOCK_OPEN) : (flags & ~MSPACK_HIP_F_BLOCK_OPEN);
        }
      }

      // E8: record what the translation pass must do for this frame (lzxd.c:707-708)
      {
        int32_t fs = 0;
        if (s.intel_started && s.intel_filesize && s.frame < 32768u && frame_size > 10u) {
          fs = s.intel_filesize; flags |= MSPACK_HIP_F_E8_APPLIED;
        }
        if (lane == 0 && frame_meta) frame_meta[u.frame_base + s.frame] = fs;
      }
      {
        u32 n = remaining < frame_size ? remaining : frame_size;
        s.offset += n; remaining -= n;
      }
      s.frame_posn += frame_size; s.frame++;
      if (s.wpos == s.wsize) s.wpos = 0;
      if (s.frame_posn == s.wsize) s.frame_posn = 0;
    }
  }
#ifndef LZX_DELTA
  if (!positioned && d.err == 0) lzx_seek_bit(d, ff_end);          // every frame came from the pipe: the reader stands behind the last one
#endif
#if defined(LZX_PHASE_TIMERS) && !defined(LZX_DELTA)
  if (lane == 0 && (blockIdx.x & 1023u) == 0u)
    printf("lzx unit %u: total %llu clk; run_tokens %u clk (commit_batch %u, resolve %u) in %u batches, %u tokens\n", blockIdx.x,
           (unsigned long long)(__builtin_amdgcn_s_memtime() - pt_unit_), d.st_t[3], d.st_t[0], d.st_t[1], d.st_t[2], d.st_t[4]);
#endif
  int err = d.err;
  if (err == 0 && remaining) err = ERR_DECRUNCH;                                  // lzxd.c:758-761
  if (err == ERR_READ && remaining == 0u) flags |= MSPACK_HIP_F_LOOKAHEAD_READ;
  if (lane == 0) {
#ifndef LZX_DELTA
    if (u.flags & MSPACK_HIP_UF_LZX_LOG) olog[0] = n_open_resets;
#endif
    res->err = err; res->flags = flags; res->out_len = s.offset; res->good_len = s.offset; res->in_next = in_next;
    res->in_used = s.raw_mode ? s.raw_pos : d.iptr();
#ifdef LZX_EXP_STATS
    {   // scratch builds only: section timers overwrite the head of the unit's output
      u32 *so = (u32 *) d.out;
      for (int k_ = 0; k_ < 6; k_++) so[k_] = d.st_t[k_];
      so[6] = (u32)(__builtin_amdgcn_s_memtime() - tstart_); so[7] = d.st_rounds; so[8] = d.st_unknown;
      so[9] = d.st_h[0]; so[10] = d.st_h[1]; so[11] = d.st_h[2];
      so[12] = d.st_t[6]; so[13] = d.st_t[7]; so[14] = d.st_t[8]; so[15] = d.st_t[9];
    }
#endif
  }
}

// E8 translation of one 32 KiB frame (lzxd.c:706-736), in place; one wavefront per frame.
// The scan is sequential in the reference (an E8 consumes the 4 following bytes, which are then not
// examined); here 64 bytes are examined at a time, candidates are found with a ballot and the
// skip rule is resolved on the 64-bit mask.
__device__ void lzx_e8_frame(u8 *frame, u32 frame_size, int32_t curpos0, int32_t filesize, u32 lane)
{
  if (frame_size <= 10u) return;
  const u32 end = frame_size - 10u;
  u32 skip_until = 0;                      // bytes below this index belong to an earlier operand
  for (u32 base = 0; base < end; base += WAVE) {
    u32 i = base + lane;
    bool cand = (i < end) && (i >= skip_until) && (frame[i] == 0xE8);
    u64 m = ballot(cand);
    u64 keep = 0;
    while (m) {
      u32 l = (u32) __ffsll((long long) m) - 1u;
      keep |= 1ull << l;
      u64 clr = (l + 5u >= 64u) ? ~0ull << l : (((1ull << 5) - 1ull) << l);
      m &= ~clr;
      skip_until = base + l + 5u;
    }
    // curpos at an accepted E8 at index i equals curpos0 + i (every byte advances it by one:
    // a skipped operand advances it by 5 for 5 bytes, lzxd.c:721,731)
    if ((keep >> lane) & 1ull) {
      int32_t curpos = curpos0 + (int32_t) i;
      int32_t abs_off = (int32_t)((u32) frame[i + 1] | ((u32) frame[i + 2] << 8) | ((u32) frame[i + 3] << 16) |
                                  ((u32) frame[i + 4] << 24));
      if (abs_off >= -curpos && abs_off < filesize) {
        int32_t rel = (abs_off >= 0) ? abs_off - curpos : abs_off + filesize;
        frame[i + 1] = (u8) rel; frame[i + 2] = (u8)(rel >> 8);
        frame[i + 3] = (u8)(rel >> 16); frame[i + 4] = (u8)(rel >> 24);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  }
}
#endif  /* !LZX_PARSE_ONLY */
