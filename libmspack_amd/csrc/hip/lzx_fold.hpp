// lzx_fold.hpp -- the per-FOLDER chain of LZ77 copies cut down to one gather pass per frame (mspack_lzx_fold, shim.hip).
// Included by lzx_kernel.hpp (plain LZX build only); reference loops it replaces: lzxd.c:613-646 (the match copy), :565-586 (R0-R2).
//
// What a folder of ordinary data is bound by: frame f's matches copy bytes of frames < f, so the
// frames' resolve tasks (lzx_pipe_resolve) run one after the other, and one of them is ~250 us of dependent steps -- which
// match covers this byte, where does that byte come from, is the source itself a match byte of this chunk -- for 32 KiB:
// 130 MB/s, below one CPU core.  Of those steps only ONE depends on the frames below: reading the source bytes.  So:
//   (1) off the chain, every frame on its own wave (mspack_lzx_fold: one wave per CU, the whole frame's SOURCE MAP in LDS,
//       128 KiB): R0-R2 along the frame's records with the three values at the frame's first byte as PLACEHOLDERS (the LRU
//       only moves offsets around); every byte's direct source into the map; pointer jumping inside the frame until every byte
//       points at a literal of this frame, at a byte of an EARLIER frame, or at a placeholder match;
//   (2) a short chain for R0-R2: a frame's (R0, R1, R2) behind its last match is a function of the three in front of it that
//       step (1) has computed -- published the moment the frame below has published its own (a few us per frame, not a resolve);
//   (3) bytes that come from literals of the frame itself are written at once; what stays on the folder's chain is ONE gather
//       pass over the bytes whose sources lie in earlier frames -- 64 loads in flight, no dependent LDS step, no match logic.
// The parse side is unchanged (lzx_pipe_parse: literals in place, one record per match); the same records, the same checks
// (lzxd.c:613-634) and the same hand-over to the serial path (rs_* in the unit's first record) as lzx_pipe_resolve.
// shim.hip decides per launch which of the two runs (few long units: this; many short ones: lzx_pipe_resolve).
#pragma once

#define LZX_SYM0 0xFFFFFFF0u                      /* "R0 / R1 / R2 as they are at the frame's first byte" (no offset is that large) */
#define LZX_IS_SYM(v_) ((v_) >= LZX_SYM0)
#define LZX_RST_OPEN 0u
#define LZX_RST_VALID 1u
#define LZX_RST_ENDED 2u
typedef FoldLds LzxFoldLds;

// the fold task of frame f of unit u, run by the FOLD_WAVES waves of one workgroup (mspack_lzx_fold's workgroups pull frame slots in
// order: a task only waits for lower frames of its unit, i.e. for earlier tickets).  Wave 0 reads the records and talks to the
// other tasks; what it decides reaches the task's other waves through L->ctl (written before a barrier, read behind it).
__device__ void lzx_fold_frame(const mspack_hip_unit &u, const u32 f, u8 *out_arena, LzxFrameRec *urecs, const uint2 *pool_base, LzxFoldLds *L)
{
  const u32 lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
  u8 *const out = out_arena + u.out_off;
  const u32 rf = u.reset_frames;
  const u32 nreal = (u.out_len + LZX_FRAME - 1u) / LZX_FRAME;
  const u32 wsize = 1u << u.window_bits;
  LzxFrameRec *rec = &urecs[f];
  LzxFrameRec *pr = rec - 1;
  const bool first = rf ? (f % rf) == 0u : f == 0u;
  u32 fsz = u.out_len - f * LZX_FRAME; if (fsz > LZX_FRAME) fsz = LZX_FRAME;
  const u32 frame_pos = f * LZX_FRAME;
  const u32 wbase = frame_pos & ~(wsize - 1u);
  // (wave 0's own: the other waves never look at these)
  u32 n_rec = 0, bytes = 0, end_bit = 0, prev_end = 0;
  bool bad = false, whole = false;
  u32 R0 = 1, R1 = 1, R2 = 1;
  FT0();
  if (wid == 0u) {
    // ---- the frame's record (the parse tasks ran in the launch before this one: every status is final) ----
    const u32 st = rfl(gld(&rec->status));
    bad = st != LZX_ST_EMITTED;
    if (f != 0u) {
      const u32 pst = rfl(gld(&pr->status));
      if (pst == LZX_ST_EMITTED) prev_end = (rfl(gld(&pr->end_bit)) + 15u) & ~15u; else bad = true;   // (the chain ends below anyway)
    }
    if (!bad) {
      n_rec = rfl(gld(&rec->n_tokens)); bytes = rfl(gld(&rec->bytes_done)); end_bit = rfl(gld(&rec->end_bit));
      bad = rfl(gld(&rec->frame_start_bit)) != prev_end || bytes > fsz || n_rec > REC_CHUNK * REC_CHUNKS;
    }
    R0 = first ? 1u : LZX_SYM0; R1 = first ? 1u : LZX_SYM0 + 1u; R2 = first ? 1u : LZX_SYM0 + 2u;
    u32 lim0 = 0xFFFFFFFFu, lim1 = 0xFFFFFFFFu, lim2 = 0xFFFFFFFFu;     // the largest offset each placeholder may turn out to be
    bool tagged = false;
    if (!bad) {
      fold_init(L, frame_pos, bytes, lane);
      // the literals of the frame's first cache line (kept in the record: lzx_parse_emit)
      const u32 ne = rfl(gld(&rec->n_edge));
      for (u32 i = lane; i < ne; i += WAVE)
        if ((gld(&rec->edge_mask[i >> 5]) >> (i & 31u)) & 1u) gst(out + frame_pos + i, gld(&rec->edge_lit[i]));
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      // ---- (1) the records, 64 at a time: R0-R2 with placeholders, the reference's checks, every match byte's direct source ----
      u32 th = 0;
      uint2 cur0 = make_uint2(0u, 0u), cur1 = cur0, cur2 = cur0, cur3 = cur0;
      if (n_rec) {
        const uint2 *g0 = rec_group(pool_base, rec->chunk, 0u);
        if (lane < n_rec) cur0 = gld(g0 + lane);
        if (64u + lane < n_rec) cur1 = gld(g0 + 64u + lane);
        if (128u + lane < n_rec) cur2 = gld(g0 + 128u + lane);
        if (192u + lane < n_rec) cur3 = gld(g0 + 192u + lane);
      }
      while (th < n_rec && !bad) {
        uint2 nx0 = make_uint2(0u, 0u), nx1 = nx0, nx2 = nx0, nx3 = nx0;
        if (th + 256u < n_rec) {
          const uint2 *g1 = rec_group(pool_base, rec->chunk, th + 256u);
          const u32 tb = th + 256u + lane;
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
          const u32 vmoff = lzx_lru_batch(ism, lane, which, c1, R0, R1, R2);
          // lzxd.c:613-634 for an offset that is known; for a placeholder: the largest value that will pass (every check is
          // "the offset is 0" or "the offset is larger than ..."), taken when the value arrives
          const u32 wp = opos - wbase;
          const bool sym = ism && LZX_IS_SYM(vmoff);
          const u32 fl = wp > frame_pos ? wp : frame_pos;          // !LZX_BAD_SOURCE: off <= wp, or off <= written (and off - wp <= wsize: implied by off <= wsize)
          u32 maxoff = wsize < opos ? wsize : opos; maxoff = fl < maxoff ? fl : maxoff;
          bool b_ = ism && (wp + olen > wsize || opos < frame_pos || opos + olen > frame_pos + bytes || olen == 0u);
          b_ = b_ || (ism && !sym && (vmoff == 0u || vmoff > maxoff));
          if (ballot(b_)) { bad = true; break; }
          if (ballot(sym)) {
            tagged = true;
#pragma unroll
            for (u32 q = 0; q < 3u; q++) {
              u32 m = (sym && vmoff == LZX_SYM0 + q) ? maxoff : 0xFFFFFFFFu;
#pragma unroll
              for (u32 dlt = 1; dlt < WAVE; dlt <<= 1) { const u32 o = (u32) __builtin_amdgcn_ds_bpermute((int)((lane ^ dlt) << 2), (int) m); m = o < m ? o : m; }
              m = rfl(m);
              if (q == 0u) lim0 = m < lim0 ? m : lim0; else if (q == 1u) lim1 = m < lim1 ? m : lim1; else lim2 = m < lim2 ? m : lim2;
            }
          }
          // the batch's matches into the map
          fold_fill_batch(L, frame_pos, ism, n, opos - frame_pos, olen, sym ? (FOLD_TAG | (vmoff - LZX_SYM0)) : vmoff, lane);
          th += n;
        }
        cur0 = nx0; cur1 = nx1; cur2 = nx2; cur3 = nx3;
      }
    }
    FT(0);
    // ---- (2) R0-R2: in front of the frame = behind the frame below; behind it = what the records made of the placeholders ----
    u32 iR0 = 1, iR1 = 1, iR2 = 1;
    if (!first) {
      u32 ps = lzx_status_load(&pr->rst);
      for (u32 tries = 0; ps == LZX_RST_OPEN && tries < (1u << 24); tries++) { __builtin_amdgcn_s_sleep(2); ps = lzx_status_load(&pr->rst); }
      if (ps == LZX_RST_VALID) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        iR0 = rfl(gld(&pr->rR0)); iR1 = rfl(gld(&pr->rR1)); iR2 = rfl(gld(&pr->rR2));
      }
      else bad = true;                                   // (the chain ends below: nothing of this frame will be used)
    }
    FT(9);
    // lzxd.c:613-634 for the matches that used a placeholder, now that the values are here
    if (!bad && ((lim0 != 0xFFFFFFFFu && (iR0 == 0u || iR0 > lim0)) || (lim1 != 0xFFFFFFFFu && (iR1 == 0u || iR1 > lim1)) ||
                 (lim2 != 0xFFFFFFFFu && (iR2 == 0u || iR2 > lim2)))) bad = true;
    if (LZX_IS_SYM(R0)) R0 = R0 == LZX_SYM0 ? iR0 : (R0 == LZX_SYM0 + 1u ? iR1 : iR2);
    if (LZX_IS_SYM(R1)) R1 = R1 == LZX_SYM0 ? iR0 : (R1 == LZX_SYM0 + 1u ? iR1 : iR2);
    if (LZX_IS_SYM(R2)) R2 = R2 == LZX_SYM0 ? iR0 : (R2 == LZX_SYM0 + 1u ? iR1 : iR2);
    whole = !bad && bytes == fsz;
    if (lane == 0 && whole) { rec->rR0 = R0; rec->rR1 = R1; rec->rR2 = R2; }
    lzx_status_publish(&rec->rst, whole ? LZX_RST_VALID : LZX_RST_ENDED, lane);
    // ---- the placeholders' matches get their sources ----
    if (!bad && tagged) {
      for (u32 b = lane; b < bytes; b += WAVE) {
        const u32 s = L->S[b];
        if (s & FOLD_TAG) { const u32 q = s & 3u; L->S[b] = frame_pos + b - (q == 0u ? iR0 : (q == 1u ? iR1 : iR2)); }
      }
    }
#if defined(MSPACK_WAVE_EMU)                             /* emulator analysis runs: which frames took this path */
    if (lane == 0 && getenv("MSPACK_EMU_FOLD_TRACE"))
      fprintf(stderr, "lzx_fold_frame: frame %u: %u records, %u bytes, placeholders %d, ok %d\n", f, n_rec, bytes, (int) tagged, (int) !bad);
#endif
    if (lane == 0) { L->ctl[1] = bad ? 0u : 1u; L->ctl[2] = bytes; }
    FT(0);
  }
  fold_barrier();
  // ---- all waves: pointer jumping until nothing moves; (3a) bytes that come from literals of this frame: final at once ----
  bool ok = L->ctl[1] != 0u;
  const u32 nb = L->ctl[2];
  if (ok) {
    fold_jump_all(L, frame_pos, nb, wid, lane);
    FT(1);
    fold_write_own(L, out, frame_pos, nb, wid, lane);
    FT(2);
  }
  // ---- (3b) the folder's chain, in three steps, each as early as its sources allow (fold_common.hpp): what comes from more than two
  // frames below once the frame THREE below is final, what comes from the frame two below once that one is, and -- the only step
  // on the chain -- what comes from the frame right below ----
  u32 nl = 0;
  for (u32 step = 0; step < 3u; step++) {
    const u32 dist = 3u - step;                           // wait for frame f - dist
    if (wid == 0u) {
      u32 pch = LZX_CH_DONE;
      if ((ok || step == 2u) && f >= dist) pch = lzx_chain_wait(&(rec - dist)->chain, false);
      if (lane == 0) L->ctl[3u + step] = pch;
      FT(step == 2u ? 5 : 3);
    }
    fold_barrier();
    const u32 pch = L->ctl[3u + step];
    if (step == 2u && pch != LZX_CH_DONE) {
      if (wid == 0u) { lzx_status_publish(&rec->chain, LZX_CH_ENDED, lane); FTFLUSH(); }
      return;
    }
    if (pch != LZX_CH_DONE) { ok = false; bad = true; }   // (the chain ended down there: it ends below this frame too)
    if (ok) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      if (step == 0u) nl = fold_write_early(L, out, frame_pos, nb, wid, lane);
      else if (nl) fold_write_late(L, out, frame_pos, nb, nl, step - 1u, wid, lane);
      FT(step == 2u ? 6 : 4);
    }
  }
  // (every wave's stores out of the door before wave 0 says so)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#ifndef MSPACK_WAVE_EMU
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  fold_barrier();
  if (wid != 0u) return;
  FT(6);
  u32 eR0 = 1, eR1 = 1, eR2 = 1;                          // R0-R2 in front of this frame, for the serial path if the chain ends here
  if (!first) { eR0 = rfl(gld(&pr->cR0)); eR1 = rfl(gld(&pr->cR1)); eR2 = rfl(gld(&pr->cR2)); }
  const bool done = whole && !bad;                        // (bad may have turned up on the chain: a frame below that never finished)
  if (lane == 0) {
    if (done) { rec->cR0 = R0; rec->cR1 = R1; rec->cR2 = R2; }
    if (!done || f + 1u == nreal) {
      LzxFrameRec *r0 = &urecs[0];
      const bool partial = !bad && !done;
      r0->rs_frame = done ? f + 1u : f; r0->rs_partial = partial ? 1u : 0u;
      r0->rs_P = done ? (f + 1u) * LZX_FRAME : (partial ? frame_pos + bytes : frame_pos);
      r0->rs_next_bit = done ? ((end_bit + 15u) & ~15u) : (partial ? end_bit : prev_end);
      r0->rs_R0 = bad ? eR0 : R0; r0->rs_R1 = bad ? eR1 : R1; r0->rs_R2 = bad ? eR2 : R2;
      r0->rs_valid = 1u;
    }
  }
  lzx_status_publish(&rec->chain, done ? LZX_CH_DONE : LZX_CH_ENDED, lane);
  FT(7);
  FTFLUSH();
}
