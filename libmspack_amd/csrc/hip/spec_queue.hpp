// spec_queue.hpp -- deferred LZ77 match resolution shared by the speculative decode paths (LZX, MSZIP).
//
// Decode rounds only QUEUE their matches (position, offset<<9 | length) in LDS and raise a flag (one bit) at the
// match's start position in a small ring; literals go straight to the output.  The queue is resolved in
// position space, 64 output bytes per pass, one byte per lane: the number of start flags at or below a
// lane's byte (ballot + mbcnt) is the index of the match that may cover it, and the byte is copied as
// out[b] = out[b - offset] -- LZ77 byte semantics, so overlapping matches (offset < length) need no
// special case and every pass is ONE byte gather and ONE coalesced 64-byte store.  A source byte inside
// the current 64-byte chunk that is itself a match byte is not in memory yet: such lanes follow the
// source's own pointer (pointer jumping with ds_bpermute, log steps).  Chunks are resolved in address
// order, so everything below the chunk is final.  What a pass costs is the memory round trip of its gather
// (the sources were stored a moment ago: the load comes back from L2), so SPQ_GROUP chunks are set up together --
// a source inside an earlier chunk of the group takes over that byte's source -- and their loads are in flight
// at once, while the next group is being set up.
//
// Contract: positions are relative to `out`; matches are pushed in position order, never overlap, and
// offset <= position (the source lies inside `out`); length <= 511; a round may be pushed only if it
// keeps (newP - (Pf & ~63)) <= SPQ_RING and mcount <= SPQ_CAP -- otherwise the caller resolves
// everything (fin = true) first.
#pragma once
#include "wave_common.hpp"

#ifndef SPQ_CAP
#define SPQ_CAP 160
#endif
#define SPQ_RING 2048u            /* positions the start-flag ring covers (one BIT per position) */
#ifndef SPQ_SPAN
#define SPQ_SPAN 512u             /* resolve when this many output bytes are pending */
#endif
#define SPQ_GROUP 4               /* 64-byte chunks resolved per memory round trip */

#ifdef LZX_MARKS        /* analysis builds: static instruction counts between marks (tools/count_isa.py) */
#define SPQ_MARK(name) asm volatile("; MARK " name)
#else
#define SPQ_MARK(name) do { } while (0)
#endif
#ifdef SPQ_TIMERS       /* analysis builds: where a resolve's cycles go (block 0's wave only; read back by mspack_hip_debug_counters) */
__device__ unsigned long long spq_tm[8];
#define SPQ_T0() unsigned long long spq_x_ = __builtin_amdgcn_s_memtime()
#define SPQ_T(k) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); if (blockIdx.x == 0 && threadIdx.x == 0) spq_tm[k] += n_ - spq_x_; spq_x_ = n_; } while (0)
#else
#define SPQ_T0() do { } while (0)
#define SPQ_T(k) do { } while (0)
#endif

struct SpecQueueLds {
  uint2 mlist[SPQ_CAP];           /* queued matches, sorted by position */
  u64 mbits[SPQ_RING / 64u];      /* bit (position mod ring) set where a queued match starts */
};
struct SpecQueue {
  u32 Pf;                         /* everything below Pf is final in memory */
  u32 mcount;                     /* queued matches */
  u32 ja;                         /* queue entries that start below Pf (0 or 1) */
};

__device__ __forceinline__ void spq_init(SpecQueueLds &l, SpecQueue &q, u32 P, u32 lane) {
  q.Pf = P; q.mcount = 0; q.ja = 0;
  if (lane < SPQ_RING / 64u) l.mbits[lane] = 0ull;
}
__device__ __forceinline__ bool spq_due(const SpecQueue &q, u32 P) {
  return P - q.Pf >= SPQ_SPAN || q.mcount > SPQ_CAP - 64u;
}

// which match covers byte c + lane, and where does that byte finally come from (sources inside the chunk itself
// are followed to their own sources: pointer jumping, log steps).  inmask = the lanes whose byte is a match byte.
__device__ __forceinline__ void spq_cover(SpecQueueLds &l, SpecQueue &q, const u32 c, const u32 lane,
                                          u64 &inmask, u32 &ptr, bool &ext)
{
  const u32 b = c + lane;
  const u32 wq = (c & (SPQ_RING - 1u)) >> 6;
  const u64 smv = l.mbits[wq];                                           // the chunk's 64 start flags: one word, same for all lanes
  if (lane == 0u) l.mbits[wq] = 0ull;
  const u64 sm = ((u64) rfl((u32)(smv >> 32)) << 32) | rfl((u32) smv);
  const u32 cnt = __builtin_amdgcn_mbcnt_hi((u32)(sm >> 32), __builtin_amdgcn_mbcnt_lo((u32) sm, 0u)) + (lane_in(sm) ? 1u : 0u);
  const int j = (int)(q.ja + cnt) - 1;
  const uint2 mr = l.mlist[j < 0 ? 0 : j];
  const u32 ml = mr.y & 511u;
  const bool inm = j >= 0 && (b - mr.x) < ml;
  ptr = b - (mr.y >> 9);
  q.ja += (u32) __popcll(sm);
  ext = (ballot(inm && mr.x + ml > c + 64u) >> 63) != 0ull;      // does the last byte's match run on?
  inmask = ballot(inm);
  if (ballot(inm && ptr >= c)) {
    for (;;) {
      u32 tl = (ptr - c) & 63u;
      u32 tp = (u32) __builtin_amdgcn_ds_bpermute((int)(tl << 2), (int) ptr);
      bool follow = inm && ptr >= c && ((inmask >> tl) & 1ull);
      if (!ballot(follow)) break;
      if (follow) ptr = tp;
    }
  }
}

// a group of up to SPQ_GROUP consecutive chunks from c on: afterwards no match byte of the group has its source
// inside the group's own match bytes (a source in an EARLIER chunk of the group takes over that byte's source,
// which is final already) -- so all the group's loads can be in flight at once
struct SpqGroup { u64 inmask[SPQ_GROUP]; u32 ptr[SPQ_GROUP]; u32 c; u32 nch; };
__device__ __forceinline__ void spq_cover_group(SpecQueueLds &l, SpecQueue &q, const u32 c, const u32 climit,
                                                const u32 lane, SpqGroup &G, bool &ext)
{
  G.c = c;
  u32 nch = (climit - c + 63u) >> 6; if (nch > SPQ_GROUP) nch = SPQ_GROUP;
  G.nch = nch;
#pragma unroll
  for (int g = 0; g < SPQ_GROUP; g++) {
    G.inmask[g] = 0ull; G.ptr[g] = 0u;
    if ((u32) g < nch) {
      spq_cover(l, q, c + 64u * g, lane, G.inmask[g], G.ptr[g], ext);
      if (g > 0) {
        const bool near = lane_in(G.inmask[g]) && G.ptr[g] >= c;
        if (ballot(near)) {
          const u32 tl = G.ptr[g] & 63u, jj = (G.ptr[g] - c) >> 6;
          u32 np = G.ptr[g];
#pragma unroll
          for (int j = 0; j < g; j++) {
            const u32 tp = (u32) __builtin_amdgcn_ds_bpermute((int)(tl << 2), (int) G.ptr[j]);
            if (near && jj == (u32) j && ((G.inmask[j] >> tl) & 1ull)) np = tp;
          }
          G.ptr[g] = np;
        }
      }
    }
  }
}

// resolve the queue up to position P: whole 64-byte chunks only unless `fin`
// (stores at or above `clip` are dropped: for callers whose last match may overshoot the output)
__device__ __forceinline__ void spq_resolve(SpecQueueLds &l, SpecQueue &q, u8 *const out, const u32 P,
                                            const bool fin, const u32 lane, const u32 clip = 0xFFFFFFFFu)
{
  u32 c = q.Pf & ~63u;
  const u32 climit = fin ? P : c + ((P - c) & ~63u);
  if (c < climit) {
    bool ext = false;
    SpqGroup G;
    SPQ_T0();
    spq_cover_group(l, q, c, climit, lane, G, ext);
    SPQ_T(0);
    for (;;) {
      SPQ_MARK("spq_group_begin");
      // the group's loads go out together; the next group is set up while they are in flight; then the stores
      u32 val[SPQ_GROUP];
#pragma unroll
      for (int g = 0; g < SPQ_GROUP; g++) { val[g] = 0; if (lane_in(G.inmask[g])) val[g] = (u32) gld(out + G.ptr[g]); }
      const SpqGroup cur = G;
      c += 64u * cur.nch;
      const bool more = c < climit;
      SPQ_T(1);
      if (more) spq_cover_group(l, q, c, climit, lane, G, ext);
      SPQ_T(0);
#ifdef SPQ_TIMERS
      { u32 acc_ = 0;
#pragma unroll
        for (int g = 0; g < SPQ_GROUP; g++) acc_ += val[g];
        asm volatile("s_waitcnt vmcnt(0)" ::"v"(acc_)); }          // (the loads' wait on its own)
      SPQ_T(2);
      if (blockIdx.x == 0 && threadIdx.x == 0) { spq_tm[4] += 1; spq_tm[5] += cur.nch; }
#endif
#pragma unroll
      for (int g = 0; g < SPQ_GROUP; g++) {
        const u32 b = cur.c + 64u * g + lane;
        if (lane_in(cur.inmask[g]) && b < clip) gst(out + b, (u8) val[g]);
      }
      SPQ_T(3);
      SPQ_MARK("spq_group_end");
      if (!more) break;
    }
    SPQ_T(7);
    if (!fin) {
      // keep the match that runs on into the next chunk (if any) and the ones that start above
      q.Pf = c;
      const u32 keep = q.ja - (ext ? 1u : 0u);
      const u32 nrem = q.mcount - keep;                  // <= 33: the span left is below 64 bytes
      if (keep) {
        uint2 mv = make_uint2(0u, 0u);
        if (lane < nrem) mv = l.mlist[keep + lane];
        __builtin_amdgcn_wave_barrier();
        if (lane < nrem) l.mlist[lane] = mv;
      }
      q.mcount = nrem; q.ja = ext ? 1u : 0u;
    }
    SPQ_T(6);
  }
  if (fin) { q.Pf = P; q.mcount = 0; q.ja = 0; }
}

// ---- runs ----------------------------------------------------------------------------------------------------------
// A batch of matches that follow each other without a literal in between and share ONE offset is a RUN [s, e):
//     out[b] = out[b - off]   for all b in [s, e)     ==     out[b] = out[s - off + (b - s) mod off]
// -- every source lies below s.  So once everything below s is final the run has no dependency inside it, however long it is:
// no memory round trip per 256 bytes (the position-space passes above: ~1.4 us per group, 170 MB/s for ONE folder), just
// stores.  That is what long stretches of zeros and of repeated records look like after LZ77 with a maximal match length
// of 257 / 258 (lzxd.c:588, mszipd.c:62-66) -- the reference's own large-files.test (a 64-byte line repeated for 2 GiB:
// ~127 matches of 257 bytes at offset 64 per frame) is the pure case.  
#ifndef SPQ_RUN_MIN
#define SPQ_RUN_MIN 512u            /* a batch takes the run path when it spans at least this many bytes */
#endif
// which matches of a batch (lanes 0..n-1, in position order) END a run that is worth the fill path: bit l set = the run that ends
// with lane l spans at least SPQ_RUN_MIN bytes; seg_first = the lane the run of a lane's match begins with.  A run is a maximal
// stretch of matches that follow each other without a gap and share one offset (one match alone -- at most 258 bytes -- never
// qualifies).  A batch may hold several: the large-files MSZIP blocks open with one match at distance 320 or 256, then 126 at 64.
__device__ __forceinline__ u64 spq_find_runs(const bool ism, const u32 n, const u32 opos, const u32 olen, const u32 off, const u32 lane,
                                             u32 &seg_first)
{
  seg_first = lane;
#ifdef SPQ_NO_RUNS                 /* analysis builds: everything through the queue */
  return 0ull;
#endif
  // (a run of >= SPQ_RUN_MIN bytes is made of LONG matches -- encoders take the longest match there is --: without two of at
  // least 64 bytes the batch is ordinary data and pays three instructions here, not twenty)
  { const u64 longm = ballot(ism && olen >= 64u); if (!(longm & (longm - 1ull))) return 0ull; }
  const u32 prev_end = (u32) __builtin_amdgcn_ds_bpermute((int)(((lane - 1u) & 63u) << 2), (int)(opos + olen));
  const u32 prev_off = (u32) __builtin_amdgcn_ds_bpermute((int)(((lane - 1u) & 63u) << 2), (int) off);
  const bool cont = ism && lane != 0u && prev_end == opos && prev_off == off;      // continues the match of the lane below
  const u64 contm = ballot(cont);
  if (!contm) return 0ull;                                                          // (ordinary data: no two matches in a row)
  const u64 startm = ballot(ism && !cont);                                          // lanes that begin a run
  const u64 below = startm & ((lane == 63u) ? ~0ull : ((2ull << lane) - 1ull));     // ... at or below this lane
  seg_first = below ? 63u - (u32) __clzll((long long) below) : 0u;
  const u32 first_pos = (u32) __builtin_amdgcn_ds_bpermute((int)(seg_first << 2), (int) opos);
  const bool last = ism && (lane + 1u == n || !((contm >> ((lane + 1u) & 63u)) & 1ull) || lane == 63u);
  return ballot(last && seg_first != lane && opos + olen - first_pos >= SPQ_RUN_MIN && off != 0u && off <= first_pos);
}
// write the run [s, e) with period `off` (everything below s is final in memory); stores at or above `clip` are dropped
// (inlined: as a real call it cost ~3 us per run -- callee-saved registers through scratch --, a third of a large-files block)
__device__ __forceinline__ void spq_fill_run(u8 *const out, const u32 s, const u32 e, const u32 off, const u32 lane,
                                             const u32 clip = 0xFFFFFFFFu)
{
  const u8 *const src = out + (s - off);
#if defined(SPQ_RUN_TRACE) && defined(MSPACK_WAVE_EMU)     /* emulator analysis builds: which runs took this path */
  if (lane == 0) fprintf(stderr, "spq_fill_run: [%u, %u) period %u\n", s, e, off);
#endif
  if (off <= 64u) {
    // the whole period sits in one register across the wave: byte (b - s) mod off comes from a lane, not from memory
    u32 k = lane % off;                                  // (one division per run)
    const u32 pat = (u32) gld(src + k);
    const u32 step = 64u % off;
    u32 cb = s;
    if (step == 0u) {
      // the period divides the wave: every lane writes ONE value all the way down (zeros, a 64-byte line: large-files.test)
      const u32 v = (u32) __builtin_amdgcn_ds_bpermute((int)(k << 2), (int) pat);
      for (; cb + 64u <= e && cb + 64u <= clip; cb += 64u) gst(out + cb + lane, (u8) v);
      if (cb < e && cb + lane < e && cb + lane < clip) gst(out + cb + lane, (u8) v);
      return;
    }
    // four chunks per pass: their lane look-ups are in flight together (one LDS latency per 256 bytes, not per 64)
    u32 step4 = step * 4u; while (step4 >= off) step4 -= off;
    for (; cb + 256u <= e && cb + 256u <= clip; cb += 256u) {
      u32 k1 = k + step; if (k1 >= off) k1 -= off;
      u32 k2 = k1 + step; if (k2 >= off) k2 -= off;
      u32 k3 = k2 + step; if (k3 >= off) k3 -= off;
      const u32 v0 = (u32) __builtin_amdgcn_ds_bpermute((int)(k << 2), (int) pat), v1 = (u32) __builtin_amdgcn_ds_bpermute((int)(k1 << 2), (int) pat);
      const u32 v2 = (u32) __builtin_amdgcn_ds_bpermute((int)(k2 << 2), (int) pat), v3 = (u32) __builtin_amdgcn_ds_bpermute((int)(k3 << 2), (int) pat);
      u8 *const o = out + cb + lane;
      gst(o, (u8) v0); gst(o + 64, (u8) v1); gst(o + 128, (u8) v2); gst(o + 192, (u8) v3);
      k += step4; if (k >= off) k -= off;
    }
    for (; cb < e; cb += 64u) {
      const u32 b = cb + lane;
      const u32 v = (u32) __builtin_amdgcn_ds_bpermute((int)(k << 2), (int) pat);
      if (b < e && b < clip) gst(out + b, (u8) v);
      k += step; if (k >= off) k -= off;
    }
  }
  else {
    // sources below s, all of them final: eight chunks' loads in flight at once, then their stores
    u32 k = lane;                                        // (b - s) mod off for b = cb + lane; lane < 64 < off
    for (u32 cb = s; cb < e; cb += 512u) {
      u32 val[8], kk = k;
#pragma unroll
      for (int g = 0; g < 8; g++) {
        val[g] = (cb + 64u * g + lane < e) ? (u32) gld(src + kk) : 0u;
        kk += 64u; if (kk >= off) kk -= off;
      }
#pragma unroll
      for (int g = 0; g < 8; g++) {
        const u32 b = cb + 64u * g + lane;
        if (b < e && b < clip) gst(out + b, (u8) val[g]);
      }
      k = kk;
    }
  }
}

// one lane-parallel push: lanes with `ism` hold matches of ranks 0..n-1 in position order
__device__ __forceinline__ void spq_push(SpecQueueLds &l, SpecQueue &q, const bool ism, const u32 rank,
                                         const u32 n, const u32 pos, const u32 off, const u32 len)
{
  if (ism) {
    l.mlist[q.mcount + rank] = make_uint2(pos, (off << 9) | len);
    atomicOr((unsigned long long *) &l.mbits[(pos & (SPQ_RING - 1u)) >> 6], 1ull << (pos & 63u));
  }
  q.mcount += n;
}

// queue the matches of the lanes in `mm` (position order; newP = the end of the last of them): a push must keep every start flag
// inside the ring, so the matches that end inside it are taken, the queue is resolved up to the first one that does not, and on
__device__ __forceinline__ void spq_push_batch(SpecQueueLds &l, SpecQueue &q, u8 *const out, u64 mm, const u32 opos, const u32 off,
                                               const u32 olen, const u32 newP, const u32 lane)
{
  if (!mm) return;
  if (q.mcount + (u32) __popcll(mm) > SPQ_CAP) spq_resolve(l, q, out, rdl(opos, (u32) __ffsll((long long) mm) - 1u), true, lane);
  bool im = lane_in(mm);
  for (;;) {
    const u32 limit = (q.Pf & ~63u) + SPQ_RING;
    const u64 fit = newP <= limit ? mm : ballot(im && opos + olen <= limit);
    if (fit) {
      const u32 rank = __builtin_amdgcn_mbcnt_hi((u32)(fit >> 32), __builtin_amdgcn_mbcnt_lo((u32) fit, 0u));
      spq_push(l, q, lane_in(fit), rank, (u32) __popcll(fit), opos, off, olen);
      mm &= ~fit;
      im = lane_in(mm);
    }
    if (!mm) break;
    spq_resolve(l, q, out, rdl(opos, (u32) __ffsll((long long) mm) - 1u), true, lane);
  }
}

// a batch of n matches (lanes 0..n-1): its runs through the fill path, everything else into the queue, in position order
// (inlined: as a real call the caller saved its registers around it -- the resolve task's scratch use tripled)
__device__ __forceinline__ void spq_push_runs(SpecQueueLds &l, SpecQueue &q, u8 *const out, const bool ism, const u32 n,
                                              const u32 opos, const u32 olen, const u32 off, const u32 lane)
{
  u32 seg_first;
  u64 runs = spq_find_runs(ism, n, opos, olen, off, lane, seg_first);
  u64 rest = ballot(ism);
  while (runs) {
    const u32 b = (u32) __ffsll((long long) runs) - 1u, a = rdl(seg_first, b);
    const u64 pre = rest & ((1ull << a) - 1ull);
    if (pre) spq_push_batch(l, q, out, pre, opos, off, olen, rdl(opos, a), lane);
    const u32 rs = rdl(opos, a), re = rdl(opos + olen, b);
    spq_resolve(l, q, out, rs, true, lane);                 // everything below the run final
    spq_fill_run(out, rs, re, rdl(off, a), lane);
    q.Pf = re;
    rest &= (b == 63u) ? 0ull : ~((2ull << b) - 1ull);
    runs &= runs - 1ull;
  }
  if (rest) spq_push_batch(l, q, out, rest, opos, off, olen, rdl(opos + olen, n - 1u), lane);
}
