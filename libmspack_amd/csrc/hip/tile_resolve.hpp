// tile_resolve.hpp -- LZ77 match resolution in an LDS tile, one LANE per MATCH (round 4; replaces the byte-per-lane
// position-space resolver of spec_queue.hpp on the frame-parallel paths).
//
// What the copy loop of the reference does one byte at a time (lzxd.c:613-651, mszipd.c:262-296) is, for a frame whose
// literals are already in place, a list of (position, length, distance) records in position order.  The output bytes
// the current records touch are held in LDS -- a TILE of 4 KiB (+ room for a match that starts in its last byte) filled
// from the output buffer with 16-byte rows and written back the same way, so the output leaves the wave as whole rows --
// and a batch of 64 records is copied by 64 lanes at once:
//   * a lane whose match is at most TR_NB bytes long and does not overlap itself ("short": 96 % of the matches of the
//     bench corpus) reads its 16 source bytes as five aligned dwords + v_alignbyte and writes them with one masked
//     update per destination dword (LDS atomics: neighbouring matches share dwords);
//   * what makes the copies depend on each other -- a source that is an EARLIER match's destination -- is tracked with
//     one NOT-FINAL bit per byte of the frame (4 KiB of LDS): set when a match is queued, cleared when it has been
//     copied.  A lane copies when no byte of its source has the bit set, and a batch is done in as many rounds as its
//     longest dependency chain (2-3 on text-like data);
//   * sources below the tile but inside the frame are final in the output buffer (earlier tiles were written back):
//     their loads are issued before the LDS rounds start and consumed after them;
//   * long, self-overlapping (distance < length) or tile-straddling sources are copied by the whole wave, one match at a
//     time (the pattern-period trick of lzx_copy_match for distance < 64).
// DEFERRAL.  A frame is resolved by the wave that parsed it, possibly before the frames below it are complete
// (`lowfinal` false).  A match that reads bytes below the frame, one whose distance is not known yet (LZX: a repeat of an
// R0-R2 value that comes from the previous frame), and -- transitively, through the bits that stay set -- every match that
// reads bytes such a match would have written, is not copied but DEFERRED: tr_batch hands back their mask, the caller
// keeps their records (compacted, in position order) and runs them through the same code a second time once the frames
// below are final.  In that second pass every match becomes ready (the oldest queued one always is).
// Contract: records in position order, non-overlapping, 2 <= length <= 257, 1 <= distance <= position; every record of
// a batch starts inside [T0, T0 + TR_TILE); `out` is 16-byte aligned, T0 and F0 multiples of 16 / 32.
#pragma once
#include "wave_common.hpp"

#define TR_TILE 4096u
#define TR_SLACK 272u                      /* a match that starts in the tile's last byte ends 256 bytes behind it */
#define TR_BYTES (TR_TILE + TR_SLACK)
#define TR_ROWS (TR_BYTES / 16u)           /* 273 rows of 16 bytes */
#define TR_NB 16u                          /* a lane copies up to this many bytes of its own match */
#define TR_FRAME 32768u                    /* bytes the not-final map covers: an LZX frame, an MSZIP block */
#ifndef TR_COLLAPSE
#define TR_COLLAPSE 1                      /* chains of matches that read each other are collapsed before the rounds (tr_batch) */
#endif

struct __align__(16) TileLds {
  u32 tile[TR_BYTES / 4u + 8u];            /* (+32 bytes: a lane's five source / destination dwords may look past the end) */
  u32 nf[TR_FRAME / 32u + 4u];             /* bit b: byte F0 + b belongs to a match that is queued (or deferred): not final */
};
struct TileState {
  u32 T0;                                  /* output position of tile byte 0 (a multiple of 16) */
  u32 hi;                                  /* end of the last match copied into the tile: what a flush has to write */
  bool live;
};

// The output buffer is read and written with GLOBAL instructions here (address space 1), not through gld / gst: a flat
// load counts in lgkmcnt too, so every LDS read behind it would wait for the memory round trip that the batch's LDS
// rounds are there to hide.
#ifdef MSPACK_WAVE_EMU
template <typename T> __device__ __forceinline__ T tr_gld(const T *p) { return *p; }
template <typename T> __device__ __forceinline__ void tr_gst(T *p, T v) { *p = v; }
#define TR_OPAQUE5(a, b, c, d, e) do { } while (0)
#else
template <typename T> __device__ __forceinline__ T tr_gld(const T *p) { return *(const __attribute__((address_space(1))) T *) p; }
template <typename T> __device__ __forceinline__ void tr_gst(T *p, T v) { *(__attribute__((address_space(1))) T *) p = v; }
typedef unsigned int tr_v4u_ __attribute__((ext_vector_type(4)));
typedef unsigned int tr_v2u_ __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint2 tr_gld(const uint2 *p) { const tr_v2u_ t = *(const __attribute__((address_space(1))) tr_v2u_ *) p; return make_uint2(t.x, t.y); }
__device__ __forceinline__ void tr_gst(uint2 *p, uint2 v) { tr_v2u_ t; t.x = v.x; t.y = v.y; *(__attribute__((address_space(1))) tr_v2u_ *) p = t; }
__device__ __forceinline__ uint4 tr_gld(const uint4 *p) { const tr_v4u_ t = *(const __attribute__((address_space(1))) tr_v4u_ *) p; return make_uint4(t.x, t.y, t.z, t.w); }
__device__ __forceinline__ void tr_gst(uint4 *p, uint4 v) { tr_v4u_ t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w; *(__attribute__((address_space(1))) tr_v4u_ *) p = t; }
/* the registers are "rewritten" here: what is computed from them cannot be moved in front of this point (the compiler
 * otherwise hoists a loaded value's first use -- and with it the wait for the load -- out of the loop that hides the load) */
#define TR_OPAQUE5(a, b, c, d, e) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e))
#endif

#ifdef LZX_PIPE_TRACE     /* analysis builds: s_memrealtime ticks per part of tr_batch, summed by the caller */
#define TR_T0() unsigned long long trt_ = __builtin_amdgcn_s_memrealtime()
#define TR_T(k) do { const unsigned long long n_ = __builtin_amdgcn_s_memrealtime(); if (ph) ph[k] += (u32)(n_ - trt_); trt_ = n_; } while (0)
#else
#define TR_T0() do { } while (0)
#define TR_T(k) do { } while (0)
#endif
#ifdef LZX_MARKS
#define TR_MARK(name) asm volatile("; MARK " name)
#else
#define TR_MARK(name) do { } while (0)
#endif

__device__ __forceinline__ void tr_clear_map(TileLds &L, const u32 lane)
{
  for (u32 w = lane; w < (u32)(sizeof(L.nf) / 4u); w += WAVE) L.nf[w] = 0u;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
}

// rows [T0, rd_end) of the output -> tile (rd_end: end of the readable range, e.g. the frame's end)
__device__ __forceinline__ void tr_fill(TileLds &L, const u8 *out, const u32 T0, const u32 rd_end, const u32 lane)
{
  const uint4 *src = (const uint4 *)(out + T0);
  uint4 *dst = (uint4 *) L.tile;
  u32 nfull = rd_end > T0 ? (rd_end - T0) >> 4 : 0u;
  const u32 nb = (rd_end > T0 && nfull < TR_ROWS) ? (rd_end - T0) & 15u : 0u;
  if (nfull > TR_ROWS) nfull = TR_ROWS;
  constexpr int NR = (int)((TR_ROWS + WAVE - 1u) / WAVE);
  uint4 v[NR];
  // (all the rows' loads are in flight before the first LDS store)
#pragma unroll
  for (int k = 0; k < NR; k++) { const u32 r = (u32) k * WAVE + lane; v[k] = make_uint4(0u, 0u, 0u, 0u); if (r < nfull) v[k] = tr_gld(src + r); }
  u32 tailb = 0;
  if (lane < nb) tailb = tr_gld(out + T0 + nfull * 16u + lane);
#pragma unroll
  for (int k = 0; k < NR; k++) { const u32 r = (u32) k * WAVE + lane; if (r < nfull) dst[r] = v[k]; }
  if (lane < nb) ((u8 *) L.tile)[nfull * 16u + lane] = (u8) tailb;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
}

// tile -> output, bytes [T0, upto): whole rows, the last partial row byte by byte (the bytes behind `upto` may belong
// to somebody who is writing them right now)
__device__ __forceinline__ void tr_flush(TileLds &L, u8 *out, const u32 T0, const u32 upto, const u32 lane)
{
  if (upto <= T0) return;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  const u32 nfull = (upto - T0) >> 4, nb = (upto - T0) & 15u;
  uint4 *dst = (uint4 *)(out + T0);
  const uint4 *src = (const uint4 *) L.tile;
  for (u32 r = lane; r < nfull; r += WAVE) tr_gst(dst + r, src[r]);
  if (lane < nb) tr_gst(out + T0 + nfull * 16u + lane, ((const u8 *) L.tile)[nfull * 16u + lane]);
}

// bits [b, e) of the map set or cleared by the lanes with `act` (lane-local; the lanes' ranges are disjoint)
__device__ __forceinline__ void tr_bits(u32 *map, const bool act, const u32 b, const u32 e, const bool set)
{
  if (act) {
    u32 bb = b;
    while (bb < e) {
      const u32 w = bb >> 5, o = bb & 31u;
      u32 nb = 32u - o; if (nb > e - bb) nb = e - bb;
      const u32 m = (nb == 32u ? 0xFFFFFFFFu : ((1u << nb) - 1u)) << o;
      if (set) atomicOr(&map[w], m); else atomicAnd(&map[w], ~m);
      bb += nb;
    }
  }
}
// is any bit of [b, e) set?  (lane-local)
__device__ __forceinline__ bool tr_any(const u32 *map, u32 b, const u32 e)
{
  while (b < e) {
    const u32 w = b >> 5, o = b & 31u;
    u32 nb = 32u - o; if (nb > e - b) nb = e - b;
    const u32 m = (nb == 32u ? 0xFFFFFFFFu : ((1u << nb) - 1u)) << o;
    if (map[w] & m) return true;
    b += nb;
  }
  return false;
}

// one match copied by the whole wave: source bytes below the tile come from the output buffer, the others from the
// tile; a distance below 64 that is shorter than the match is a pattern of that period (cf. lzx_copy_match)
__device__ __forceinline__ void tr_copy_one(TileLds &L, const u8 *out, const u32 T0, const u32 P, const u32 Ln, const u32 Of,
                                            const u32 lane)
{
  u8 *const tb = (u8 *) L.tile;
  const u32 S = P - Of, pr = P - T0;
  const bool per = Of < Ln && Of < WAVE;
  u32 r = lane, step = 64u;
  if (per) {
    u32 s = Of << 5, ss = Of << 5;
#pragma unroll
    for (int k = 0; k < 6; k++) { const u32 t = r - s; r = t < r ? t : r; s >>= 1; }               // lane mod Of
#pragma unroll
    for (int k = 0; k < 6; k++) { const u32 t = step - ss; step = t < step ? t : step; ss >>= 1; }  // 64 mod Of
  }
  for (u32 i = lane; i < Ln; i += WAVE) {
    // (distance >= 64 or >= length: a 64-byte step only reads bytes that earlier steps or matches have completed)
    const u32 sp = S + (per ? r : i);
    u32 b;
    if (sp >= T0) b = tb[sp - T0]; else b = tr_gld(out + sp);
    tb[pr + i] = (u8) b;
    if (per) { r += step; if (r >= Of) r -= Of; }
  }
}

// 16 source bytes v0..v3 -> the lane's destination dwords, masked (m[j]: the bytes of dword dw + j that are the match's)
__device__ __forceinline__ void tr_store(TileLds &L, const bool act, const u32 dw, const u32 dsh, const u32 *m,
                                         const u32 v0, const u32 v1, const u32 v2, const u32 v3)
{
  const u32 sh = 32u - 8u * dsh;                                  // 8 .. 32
  const u32 D0 = v0 << (8u * dsh);
  const u32 D1 = (u32)((((u64) v1 << 32) | v0) >> sh), D2 = (u32)((((u64) v2 << 32) | v1) >> sh);
  const u32 D3 = (u32)((((u64) v3 << 32) | v2) >> sh), D4 = (u32)(((u64) v3) >> sh);
  if (act) {
    atomicAnd(&L.tile[dw], ~m[0]); atomicOr(&L.tile[dw], D0 & m[0]);
    if (m[1]) { atomicAnd(&L.tile[dw + 1u], ~m[1]); atomicOr(&L.tile[dw + 1u], D1 & m[1]); }
  }
  if (ballot(act && m[2] != 0u)) {
    if (act && m[2]) { atomicAnd(&L.tile[dw + 2u], ~m[2]); atomicOr(&L.tile[dw + 2u], D2 & m[2]); }
    if (act && m[3]) { atomicAnd(&L.tile[dw + 3u], ~m[3]); atomicOr(&L.tile[dw + 3u], D3 & m[3]); }
    if (act && m[4]) { atomicAnd(&L.tile[dw + 4u], ~m[4]); atomicOr(&L.tile[dw + 4u], D4 & m[4]); }
  }
}

// One batch: the lanes with `ism` hold matches in position order (pos: output position, all of them inside
// [T0, T0 + TR_TILE) and inside the frame that starts at F0; len; off = distance, 1 <= off <= pos -- checked by the
// caller, except for the lanes with `unknown`: their distance is not known yet).
//   lowfinal   the bytes below F0 are final in the output buffer (else a match that reads them is deferred)
//   setbits    queue the batch's matches in the not-final map (false: their bits are set already -- the deferred
//              matches' second pass)
// Returns the lanes whose match was NOT copied (deferred; their bits stay set).
__device__ __forceinline__ u64 tr_batch(TileLds &L, const u8 *out, const u32 T0, const u32 F0, const bool lowfinal, const bool setbits,
                                        const bool ism, const bool unknown, const u32 pos, const u32 len, const u32 off,
                                        const u32 lane, u32 *ph = nullptr)
{
  TR_MARK("tr_batch_begin");
  TR_T0();
  const u32 pr = pos - T0, pf = pos - F0;
  u32 s0 = pos - off;
  if (setbits) tr_bits(L.nf, ism, pf, pf + len, true);
  // ---- CHAINS.  On record-like data match after match reads the match before it ("literal + 3 bytes at distance 4", over
  // and over): rounds would copy such a batch one link at a time (19.5 rounds per batch on the bench corpus, 40 on its
  // binary third -- profiles/round4_tile_resolver_experiment.txt).  But a match whose source lies INSIDE one earlier,
  // not self-overlapping match's destination reads what that match reads, shifted: it takes over that match's source --
  // and, by pointer jumping, its source's source -- until its source is something that is not a queued match's bytes.
  // Blocker = the last lane whose position is <= my source's first byte (a binary search over the sorted positions, six
  // ds_bpermute); then log2(chain length) jump steps of three ds_bpermute.  Left with ~2 rounds per batch. ----
  if (TR_COLLAPSE) {
    const bool cand0 = ism && !unknown && len <= TR_NB && off >= len && s0 >= T0;
    bool blocked = false;
    if (cand0 && s0 >= F0) {
      const u32 sf0 = s0 - F0, w = sf0 >> 5, o = sf0 & 31u;
      const u64 x = ((((u64) L.nf[w + 1u]) << 32) | L.nf[w]) >> o;
      blocked = (x & ((1ull << len) - 1ull)) != 0ull;
    }
    if (ballot(blocked)) {
      const u32 key = ism ? pos : 0xFFFFFFFFu;
      u32 cnt = 0;                                                // lanes whose position is <= s0 (positions ascend with the lane)
#pragma unroll
      for (u32 step = 32u; step >= 1u; step >>= 1) {
        const u32 probe = cnt + step - 1u;
        const u32 pk = (u32) __builtin_amdgcn_ds_bpermute((int)((probe & 63u) << 2), (int) key);
        if (probe < 64u && pk <= s0) cnt += step;
      }
      // a link needs: a blocker in this batch, not self-overlapping, of known distance, whose destination holds my source
      const u32 meta = len | ((ism && !unknown && off >= len) ? 0x10000u : 0u);
      const u32 bl_ = cnt ? cnt - 1u : 0u;
      const u32 bpos = (u32) __builtin_amdgcn_ds_bpermute((int)(bl_ << 2), (int) key);
      const u32 bmeta = (u32) __builtin_amdgcn_ds_bpermute((int)(bl_ << 2), (int) meta);
      const bool link = blocked && cnt != 0u && (bmeta & 0x10000u) && s0 + len <= bpos + (bmeta & 0xFFFFu);
      u32 blk = link ? bl_ : 64u, dd = s0 - bpos, src = s0;
      for (int it = 0; it < 6; it++) {
        if (!ballot(blk < 64u)) break;
        const u32 a = (blk & 63u) << 2;
        const u32 bsrc = (u32) __builtin_amdgcn_ds_bpermute((int) a, (int) src);
        const u32 bblk = (u32) __builtin_amdgcn_ds_bpermute((int) a, (int) blk);
        const u32 bdd = (u32) __builtin_amdgcn_ds_bpermute((int) a, (int) dd);
        if (blk < 64u) { src = bsrc + dd; dd = bdd + dd; blk = bblk; }
      }
      // (still linked after six doublings: cannot be -- a chain inside 64 lanes is shorter; such a lane keeps its own source)
      if (link && blk >= 64u) s0 = src;
    }
  }
  const u32 eoff = pos - s0;                                      // the distance in effect (>= off: a redirected source lies further back)
  // never copied in this pass: distance unknown, or a source byte below the frame while those are not final
  const bool predef = ism && (unknown || (!lowfinal && s0 < F0));
  const bool live = ism && !predef;
  const bool below = live && s0 + len <= T0;                      // the whole source lies below the tile
  const bool inside = live && s0 >= T0;
  const bool shortm = live && len <= TR_NB && eoff >= len && (below || inside);
  const bool slow = live && !shortm;
  const u32 sr = s0 - T0;                                         // (inside) source offset in the tile
  const u32 sf = s0 - F0;                                         // source offset in the frame (s0 >= F0)
  // destination dwords of a short match and their byte masks
  const u32 dsh = pr & 3u, dw = pr >> 2;
  const u32 be = shortm ? (((1u << len) - 1u) << dsh) : 0u;       // one bit per destination byte, from the first dword's byte 0
  u32 m[5];
#pragma unroll
  for (int j = 0; j < 5; j++) { const u32 n = (be >> (4 * j)) & 15u; m[j] = ((n * 0x204081u) & 0x01010101u) * 0xFFu; }
  // short matches with sources below the tile: the part inside the frame was written back by earlier tiles and is final
  // unless its bits are still set (deferred bytes); loads issued now, consumed behind the LDS rounds
  bool gl = shortm && below;
  bool gdef = false;
  if (gl) {
    const u32 lo = s0 >= F0 ? sf : 0u, hi = s0 + len > F0 ? s0 + len - F0 : 0u;
    if (lo < hi) {
      const u32 w = lo >> 5, o = lo & 31u;
      const u64 x = ((((u64) L.nf[w + 1u]) << 32) | L.nf[w]) >> o;
      gdef = (x & ((1ull << (hi - lo)) - 1ull)) != 0ull;
    }
  }
  gl = gl && !gdef;
  const u64 gmask = ballot(gl);
  u32 g0 = 0, g1 = 0, g2 = 0, g3 = 0, g4 = 0;
  if (gl) {
    const u32 *gp = (const u32 *)(out + (s0 & ~3u));
    const u32 need = (s0 & 3u) + len;                              // bytes from the first dword's byte 0
    g0 = tr_gld(gp);
    if (need > 4u) g1 = tr_gld(gp + 1);
    if (need > 8u) g2 = tr_gld(gp + 2);
    if (need > 12u) g3 = tr_gld(gp + 3);
    if (need > 16u) g4 = tr_gld(gp + 4);
  }
  u64 defer = ballot(predef || gdef);
  u64 todo = ballot(live && !gdef);
  const u64 insh = ballot(shortm && inside);
  const u64 slowm = ballot(slow);
  bool gdone = gmask == 0ull;
  TR_T(0);
  while (todo) {
    // (a) short matches whose source lies in the tile: the ones with no not-final source byte copy now
    const u64 cand = todo & insh;
    if (cand) {
      bool rdy = false;
      if (lane_in(cand)) {
        const u32 w = sf >> 5, o = sf & 31u;
        const u64 x = ((((u64) L.nf[w + 1u]) << 32) | L.nf[w]) >> o;
        rdy = (x & ((1ull << len) - 1ull)) == 0ull;
      }
      const u64 rm = ballot(rdy);
      if (rm) {
        TR_MARK("tr_round_begin");
        u32 v0 = 0, v1 = 0, v2 = 0, v3 = 0;
        const u32 a = sr >> 2, sh = sr & 3u;
        if (rdy) {
          const u32 d0 = L.tile[a], d1 = L.tile[a + 1u], d2 = L.tile[a + 2u];
          v0 = (u32) __builtin_amdgcn_alignbyte(d1, d0, sh); v1 = (u32) __builtin_amdgcn_alignbyte(d2, d1, sh);
        }
        if (ballot(rdy && len > 8u)) {
          if (rdy) {
            const u32 d2 = L.tile[a + 2u], d3 = L.tile[a + 3u], d4 = L.tile[a + 4u];
            v2 = (u32) __builtin_amdgcn_alignbyte(d3, d2, sh); v3 = (u32) __builtin_amdgcn_alignbyte(d4, d3, sh);
          }
        }
        tr_store(L, rdy, dw, dsh, m, v0, v1, v2, v3);
        tr_bits(L.nf, rdy, pf, pf + len, false);
        todo &= ~rm;
        TR_MARK("tr_round_end");
        TR_T(1);
        continue;
      }
    }
    // (b) the sources from below the tile have arrived
    if (!gdone) {
      TR_OPAQUE5(g0, g1, g2, g3, g4);
      const u32 sh = s0 & 3u;
      const u32 v0 = (u32) __builtin_amdgcn_alignbyte(g1, g0, sh), v1 = (u32) __builtin_amdgcn_alignbyte(g2, g1, sh);
      const u32 v2 = (u32) __builtin_amdgcn_alignbyte(g3, g2, sh), v3 = (u32) __builtin_amdgcn_alignbyte(g4, g3, sh);
      tr_store(L, gl, dw, dsh, m, v0, v1, v2, v3);
      tr_bits(L.nf, gl, pf, pf + len, false);
      todo &= ~gmask;
      gdone = true;
      TR_T(2);
      continue;
    }
    // (c) nothing short is ready.  A long / self-overlapping / straddling match whose source (the part below its own
    // first byte) holds no not-final byte is copied by the whole wave; if there is none, nothing that is still queued
    // can ever become ready in this pass: the rest is deferred.
    {
      bool srdy = false;
      if (lane_in(todo & slowm)) {
        const u32 lo = s0 >= F0 ? sf : 0u;
        const u32 se = s0 + len < pos ? s0 + len : pos;            // (a self-overlapping match reads its own first bytes)
        const u32 hi = se > F0 ? se - F0 : 0u;
        srdy = !tr_any(L.nf, lo, hi);
      }
      const u64 sm = ballot(srdy);
      if (!sm) { defer |= todo; TR_T(3); break; }
      const u32 j = (u32) __ffsll((long long) sm) - 1u;
      const u32 P = rdl(pos, j), Ln = rdl(len, j), Of = rdl(eoff, j);
      tr_copy_one(L, out, T0, P, Ln, Of, lane);
      {
        // its bits, a word per lane
        const u32 b0 = P - F0, e0 = b0 + Ln, w = (b0 >> 5) + lane;
        if (w <= ((e0 - 1u) >> 5)) {
          const u32 lo = w == (b0 >> 5) ? (b0 & 31u) : 0u, hi = w == ((e0 - 1u) >> 5) ? ((e0 - 1u) & 31u) + 1u : 32u;
          const u32 mk = (hi - lo == 32u ? 0xFFFFFFFFu : ((1u << (hi - lo)) - 1u)) << lo;
          atomicAnd(&L.nf[w], ~mk);
        }
      }
      todo &= ~(1ull << j);
      TR_T(3);
    }
  }
  TR_MARK("tr_batch_end");
  return defer;
}
