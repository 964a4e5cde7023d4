// spec_queue.hpp -- deferred LZ77 match resolution shared by the speculative decode paths (LZX, MSZIP).
//
// Decode rounds only QUEUE their matches (position, offset<<9 | length) in LDS and raise a flag at the
// match's start position in a small ring; literals go straight to the output.  The queue is resolved in
// position space, 64 output bytes per pass, one byte per lane: the number of start flags at or below a
// lane's byte (ballot + mbcnt) is the index of the match that may cover it, and the byte is copied as
// out[b] = out[b - offset] -- LZ77 byte semantics, so overlapping matches (offset < length) need no
// special case and every pass is ONE byte gather and ONE coalesced 64-byte store.  A source byte inside
// the current 64-byte chunk that is itself a match byte is not in memory yet: such lanes follow the
// source's own pointer (pointer jumping with ds_bpermute, log steps).  Chunks are resolved in address
// order, so everything below the chunk is final.  The passes are software-pipelined: chunk k's load
// is in flight while chunk k+1 is set up.
//
// Contract: positions are relative to `out`; matches are pushed in position order, never overlap, and
// offset <= position (the source lies inside `out`); length <= 511; a round may be pushed only if it
// keeps (newP - (Pf & ~63)) <= SPQ_RING and mcount <= SPQ_CAP -- otherwise the caller resolves
// everything (fin = true) first.
#pragma once
#include "wave_common.hpp"

#define SPQ_CAP 160
#define SPQ_RING 512u
#ifndef SPQ_SPAN
#define SPQ_SPAN 128u             /* resolve when this many output bytes are pending */
#endif

struct SpecQueueLds {
  uint2 mlist[SPQ_CAP];           /* queued matches, sorted by position */
  u8  mflag[SPQ_RING];            /* 1 at (position mod ring) where a queued match starts */
};
struct SpecQueue {
  u32 Pf;                         /* everything below Pf is final in memory */
  u32 mcount;                     /* queued matches */
  u32 ja;                         /* queue entries that start below Pf (0 or 1) */
};

__device__ __forceinline__ void spq_init(SpecQueueLds &l, SpecQueue &q, u32 P, u32 lane) {
  q.Pf = P; q.mcount = 0; q.ja = 0;
  ((u32 *) l.mflag)[lane] = 0; ((u32 *) l.mflag)[64u + lane] = 0;
}
__device__ __forceinline__ bool spq_due(const SpecQueue &q, u32 P) {
  return P - q.Pf >= SPQ_SPAN || q.mcount > SPQ_CAP - 64u;
}

// which match covers byte c + lane, and where does that byte finally come from
__device__ __forceinline__ void spq_cover(SpecQueueLds &l, SpecQueue &q, const u32 c, const u32 lane,
                                          bool &inm, u32 &ptr, bool &ext)
{
  const u32 b = c + lane;
  const u32 fi = b & (SPQ_RING - 1u);
  const u32 f = l.mflag[fi];
  l.mflag[fi] = 0;
  const u64 sm = ballot(f != 0u);
  const u32 cnt = __builtin_amdgcn_mbcnt_hi((u32)(sm >> 32), __builtin_amdgcn_mbcnt_lo((u32) sm, 0u)) + (f != 0u ? 1u : 0u);
  const int j = (int)(q.ja + cnt) - 1;
  const uint2 mr = l.mlist[j < 0 ? 0 : j];
  const u32 ml = mr.y & 511u;
  inm = j >= 0 && (b - mr.x) < ml;
  ptr = b - (mr.y >> 9);
  q.ja += (u32) __popcll(sm);
  ext = (ballot(inm && mr.x + ml > c + 64u) >> 63) != 0ull;      // does the last byte's match run on?
  const u64 inmask = ballot(inm);
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

// resolve the queue up to position P: whole 64-byte chunks only unless `fin`
// (stores at or above `clip` are dropped: for callers whose last match may overshoot the output)
__device__ __forceinline__ void spq_resolve(SpecQueueLds &l, SpecQueue &q, u8 *const out, const u32 P,
                                            const bool fin, const u32 lane, const u32 clip = 0xFFFFFFFFu)
{
  u32 c = q.Pf & ~63u;
  const u32 climit = fin ? P : c + ((P - c) & ~63u);
  if (c < climit) {
    bool inm, ext; u32 ptr;
    spq_cover(l, q, c, lane, inm, ptr, ext);
    for (;;) {
      u32 val = 0; if (inm) val = (u32) out[ptr];
      const u32 bcur = c + lane; const bool icur = inm;
      c += 64u;
      const bool more = c < climit;
      if (more) spq_cover(l, q, c, lane, inm, ptr, ext);
      if (icur && bcur < clip) out[bcur] = (u8) val;
      if (!more) break;
    }
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
  }
  if (fin) { q.Pf = P; q.mcount = 0; q.ja = 0; }
}

// one lane-parallel push: lanes with `ism` hold matches of ranks 0..n-1 in position order
__device__ __forceinline__ void spq_push(SpecQueueLds &l, SpecQueue &q, const bool ism, const u32 rank,
                                         const u32 n, const u32 pos, const u32 off, const u32 len)
{
  if (ism) {
    l.mlist[q.mcount + rank] = make_uint2(pos, (off << 9) | len);
    l.mflag[pos & (SPQ_RING - 1u)] = 1;
  }
  q.mcount += n;
}
