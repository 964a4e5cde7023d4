// fold_common.hpp -- a 32 KiB frame's SOURCE MAP in LDS: what mspack_lzx_fold / mspack_mszip_fold (shim.hip) share.
//
// The per-folder chain of LZ77 copies (lzxd.c:613-646, mszipd.c:262-296) cut down to one gather pass per frame: off the chain a
// wave writes, for every byte of ITS frame, the position that byte is copied from (fold_fill_batch: the frame's match records, 64
// at a time, in position space), follows the pointers inside the frame until every byte points at a literal of the frame or at a
// byte of an EARLIER frame (fold_jump), and writes the bytes that come from the frame's own literals (fold_write_own); on the chain
// -- once everything below the frame is final -- it gathers the rest (fold_write_ext: FOLD_DEPTH x 64 loads in flight, no dependent
// LDS step, no match logic).  DESIGN.md section 8.1; the codec halves are lzx_fold.hpp and mszip_kernel.hpp (zip_fold_block).
#pragma once
#include "wave_common.hpp"

#define FOLD_FRAME 32768u
#define FOLD_TAG 0x80000000u                      /* a map entry that waits for a value (LZX: a placeholder R0 / R1 / R2): TAG | which */
#ifndef FOLD_DEPTH
#define FOLD_DEPTH 48                             /* gathers in flight in the chain's passes (x 64 lanes; a wave can have 63 memory instructions outstanding) */
#endif

// A fold task is run by FOLD_WAVES waves (one workgroup) that share the map: what costs time in the chain's passes is the round trip
// of scattered byte gathers to memory another XCD has just written (~8 us per batch of 48 x 64, measured), and a wave cannot have
// more than 63 memory instructions in flight -- eight waves have eight times that (measured: 35.0 / 22.1 / 15.0 / 11.7 / 11.2 ms for a
// 512-frame folder with 1 / 2 / 4 / 8 / 16 waves).  Wave 0 does what is serial (the records, the
// hand-offs), all waves do what is per byte.  (The wavefront emulator models one wave per workgroup: there the task runs on one.)
#if defined(MSPACK_WAVE_EMU)
#define FOLD_WAVES 1u
#define fold_barrier() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront")
#else
#ifndef FOLD_WAVES
#define FOLD_WAVES 8u
#endif
#define fold_barrier() __syncthreads()
#endif
#define FOLD_THREADS (64u * FOLD_WAVES)

#if defined(FOLD_NT_STORES) && !defined(MSPACK_WAVE_EMU)      /* analysis builds: the fold tasks' byte stores marked non-temporal */
__device__ __forceinline__ void fold_st(u8 *p, u8 v) { __builtin_nontemporal_store(v, p); }
#else
__device__ __forceinline__ void fold_st(u8 *p, u8 v) { gst(p, v); }
#endif

#ifdef FOLD_TRACE      /* analysis builds: ticks (s_memrealtime, 100 MHz) per phase of the fold tasks, summed over all tasks:
                          0 records -> map, 1 jumps, 2 own bytes, 3 wait for the frame two below, 4 early gather + list, 5 wait for
                          the frame below, 6 late gather, 7 publish, 8 tasks, 9 R0-R2 wait; read by mspack_hip_debug_fold_phases */
__device__ unsigned long long g_fold_phase[16];
#define FT0() unsigned long long ft_ = __builtin_amdgcn_s_memrealtime(); u32 fta_[10] = { 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 1u, 0u }
#define FT(k) do { const unsigned long long n_ = __builtin_amdgcn_s_memrealtime(); fta_[k] += (u32)(n_ - ft_); ft_ = n_; } while (0)
#define FTFLUSH() do { if (threadIdx.x == 0) for (int k_ = 0; k_ < 10; k_++) if (fta_[k_]) atomicAdd(&g_fold_phase[k_], (unsigned long long) fta_[k_]); } while (0)
#else
#define FT0() do { } while (0)
#define FT(k) do { } while (0)
#define FTFLUSH() do { } while (0)
#endif

struct __align__(16) FoldLds {
  u32 S[FOLD_FRAME];                              /* byte b of the frame finally comes from position S[b] of the unit's output */
  u64 start[FOLD_FRAME / 64u + 1u];               /* bit b: a match starts at byte b of the frame */
  uint2 bl[64];                                   /* the batch at hand: (position in the frame, length) ... */
  u32 bv[64];                                     /* ... and its distance, or TAG | which */
  u32 ctl[16];                                    /* what wave 0 tells the task's other waves (each word written before a barrier, read behind it) */
};
// wave w's share of [0, n): whole 256-byte groups
__device__ __forceinline__ void fold_share(const u32 n, const u32 wid, u32 &lo, u32 &hi)
{
  const u32 per = (((n + 255u) >> 8) + FOLD_WAVES - 1u) / FOLD_WAVES * 256u;
  lo = wid * per < n ? wid * per : n;
  hi = lo + per < n ? lo + per : n;
}

// every byte its own source (a literal), no match starts anywhere.  base = the position of the frame's first byte in the unit's output
__device__ __forceinline__ void fold_init(FoldLds *L, const u32 base, const u32 bytes, const u32 lane)
{
  for (u32 b = lane; b < bytes; b += WAVE) L->S[b] = base + b;
  for (u32 w = lane; w < FOLD_FRAME / 64u + 1u; w += WAVE) L->start[w] = 0ull;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
}

// a batch of n matches (lanes 0..n-1, in position order, not overlapping): fp = position in the frame, olen = length, val = distance
// (or TAG | which): every match byte's direct source into the map, in position space -- the start bits of the 64 bytes at hand
// rank the batch's list
__device__ __forceinline__ void fold_fill_batch(FoldLds *L, const u32 base, const bool ism, const u32 n, const u32 fp, const u32 olen,
                                                const u32 val, const u32 lane)
{
  if (ism) {
    L->bl[lane] = make_uint2(fp, olen);
    L->bv[lane] = val;
    atomicOr((unsigned long long *) &L->start[fp >> 6], 1ull << (fp & 63u));
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  const u32 lo = rdl(fp, 0u), hi = rdl(fp + olen, n - 1u);
  u32 ja = 0;
  for (u32 c = lo; c < hi; c += 64u) {
    const u32 w = c >> 6, sh = c & 63u;
    const u64 a0 = L->start[w], a1 = L->start[w + 1u];
    const u64 W = sh ? (a0 >> sh) | (a1 << (64u - sh)) : a0;         // start bits of bytes c .. c + 63 (the same for every lane)
    const u64 Wu = ((u64) rfl((u32)(W >> 32)) << 32) | rfl((u32) W);
    const u32 cnt = __builtin_amdgcn_mbcnt_hi((u32)(Wu >> 32), __builtin_amdgcn_mbcnt_lo((u32) Wu, 0u)) + (lane_in(Wu) ? 1u : 0u);
    const u32 j = ja + cnt - 1u;                                    // (c starts at a match: cnt >= 1 for every lane)
    const uint2 e = L->bl[j & 63u];
    const u32 v = L->bv[j & 63u];
    const u32 b = c + lane;
    if (b < hi && b - e.x < e.y) L->S[b] = (v & FOLD_TAG) ? v : base + b - v;
    ja += (u32) __popcll(Wu);
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
}

// one pass of pointer jumping over [0, n) of the map, in address order (sources lie below their bytes, so what a byte reads has
// usually been through this pass already: one pass resolves every chain whose links are longer than the pass's group).
// Returns true when a pointer moved.
__device__ __forceinline__ bool fold_jump(FoldLds *L, const u32 base, const u32 c0, const u32 n, const u32 lane)
{
  bool moved = false;
  for (u32 c = c0; c < n; c += 256u) {
    u32 s[4], t[4];
#pragma unroll
    for (int g = 0; g < 4; g++) { const u32 b = c + 64u * g + lane; s[g] = b < n ? L->S[b] : 0u; }
#pragma unroll
    for (int g = 0; g < 4; g++) {
      const u32 b = c + 64u * g + lane;
      // an in-frame pointer that is not the byte itself (a literal) and not a placeholder: look where IT points
      const bool in = b < n && !(s[g] & FOLD_TAG) && s[g] >= base && s[g] != base + b;
      t[g] = in ? L->S[s[g] - base] : 0u;
      // (a source that waits for a value is a root for now: keep pointing at it; a source that is a literal points at itself)
      const bool mv = in && !(t[g] & FOLD_TAG) && t[g] != s[g];
      if (mv) L->S[b] = t[g];
      moved = moved || mv;
    }
  }
  return ballot(moved) != 0ull;
}
// ... until nothing moves anywhere (a chain of n links inside one 256-byte group needs ~log2 n passes).  All the task's waves, each
// over its share; "something moved" goes through three flag words in turn (the one of pass p + 1 is cleared during pass p, when the
// last readers of its previous use -- pass p - 2 -- are two barriers behind).  Returns the passes it took
__device__ __forceinline__ u32 fold_jump_all(FoldLds *L, const u32 base, const u32 n, const u32 wid, const u32 lane)
{
  u32 lo, hi;
  fold_share(n, wid, lo, hi);
  if (wid == 0u && lane < 3u) L->ctl[8u + lane] = 0u;
  fold_barrier();
  u32 passes = 0;
  for (; passes < 64u; passes++) {
    const u32 w = 8u + passes % 3u, wn = 8u + (passes + 1u) % 3u;
    if (wid == 0u && lane == 0u) L->ctl[wn] = 0u;
    const bool moved = fold_jump(L, base, lo, hi, lane);
    if (moved && lane == 0u) L->ctl[w] = 1u;
    fold_barrier();
    if (L->ctl[w] == 0u) break;
  }
  return passes + 1u;
}

// bytes that come from literals of this frame: final at once
__device__ __forceinline__ void fold_write_own(const FoldLds *L, u8 *const out, const u32 base, const u32 n, const u32 wid, const u32 lane)
{
  u32 c0, bytes;
  fold_share(n, wid, c0, bytes);
  for (u32 c = c0; c < bytes; c += 512u) {
    u32 s[8], v[8];
#pragma unroll
    for (int g = 0; g < 8; g++) { const u32 b = c + 64u * g + lane; s[g] = b < bytes ? L->S[b] : 0u; }
#pragma unroll
    for (int g = 0; g < 8; g++) { const u32 b = c + 64u * g + lane; v[g] = (b < bytes && s[g] >= base && s[g] != base + b) ? (u32) gld(out + s[g]) : 0u; }
#pragma unroll
    for (int g = 0; g < 8; g++) { const u32 b = c + 64u * g + lane; if (b < bytes && s[g] >= base && s[g] != base + b) fold_st(out + base + b, (u8) v[g]); }
  }
}
// The folder's chain.  A byte that comes from below the frame waits for the frames below to be final -- but only for the one its
// source lies in.  Measured on the hardware with ONE pass behind the frame below: ~76 us per link (sixteen dependent round trips to
// memory that another XCD has just written); so the gathers are taken in three steps, each as early as its sources allow:
//   fold_write_early   everything more than two frames below is final (the frame three below is): gathers what comes from there,
//                      and compacts the rest -- sources in the 64 KiB right below the frame -- into a list at the bottom of the
//                      wave's share of the map: position in the frame << 16 | source position in those 64 KiB; in place (entry k
//                      is written when at least k bytes have been read).  Returns the list's length;
//   fold_write_late(0) the frame two below is final: the list's entries that come from it;
//   fold_write_late(1) the frame right below is final: the rest -- the only step left on the folder's chain.
__device__ __forceinline__ u32 fold_write_early(FoldLds *L, u8 *const out, const u32 base, const u32 n, const u32 wid, const u32 lane)
{
  // (every wave over its share of the frame; its list at the bottom of its share)
  const u32 org = base - 2u * FOLD_FRAME;                          // (wraps for the first two frames: only differences are used)
  const u32 late_lo = base >= 2u * FOLD_FRAME ? org : 0u;
  u32 c0, bytes;
  fold_share(n, wid, c0, bytes);
  u32 nl = c0;
  for (u32 c = c0; c < bytes; c += 64u * FOLD_DEPTH) {
    u32 s[FOLD_DEPTH], v[FOLD_DEPTH];
#pragma unroll
    for (int g = 0; g < FOLD_DEPTH; g++) { const u32 b = c + 64u * g + lane; s[g] = b < bytes ? L->S[b] : 0xFFFFFFFFu; }
#pragma unroll
    for (int g = 0; g < FOLD_DEPTH; g++) v[g] = s[g] < late_lo ? (u32) gld(out + s[g]) : 0u;
#pragma unroll
    for (int g = 0; g < FOLD_DEPTH; g++) {
      const u32 b = c + 64u * g + lane;
      const bool late = s[g] >= late_lo && s[g] < base;
      const u64 m = ballot(late);
      if (m) {
        const u32 rank = __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32) m, 0u));
        if (late) L->S[nl + rank] = (b << 16) | ((s[g] - org) & 0xFFFFu);
        nl += (u32) __popcll(m);
      }
    }
#pragma unroll
    for (int g = 0; g < FOLD_DEPTH; g++) { const u32 b = c + 64u * g + lane; if (s[g] < late_lo) fold_st(out + base + b, (u8) v[g]); }
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  return nl - c0;
}
// the list's entries whose sources lie in the frame two below (which = 0) / right below (which = 1): FOLD_DEPTH x 64 gathers in flight
__device__ __forceinline__ void fold_write_late(const FoldLds *L, u8 *const out, const u32 base, const u32 n, const u32 nl_, const u32 which,
                                                const u32 wid, const u32 lane)
{
  const u32 org = base - 2u * FOLD_FRAME;
  u32 c0, c1;
  fold_share(n, wid, c0, c1);
  const u32 nl = c0 + nl_;
  for (u32 k0 = c0; k0 < nl; k0 += 64u * FOLD_DEPTH) {
    u32 e[FOLD_DEPTH], v[FOLD_DEPTH];
#pragma unroll
    for (int g = 0; g < FOLD_DEPTH; g++) {
      const u32 k = k0 + 64u * g + lane;
      const u32 x = k < nl ? L->S[k] : 0xFFFFFFFFu;
      e[g] = (x != 0xFFFFFFFFu && ((x >> 15) & 1u) == which) ? x : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int g = 0; g < FOLD_DEPTH; g++) v[g] = e[g] != 0xFFFFFFFFu ? (u32) gld(out + (org + (e[g] & 0xFFFFu))) : 0u;
#pragma unroll
    for (int g = 0; g < FOLD_DEPTH; g++) if (e[g] != 0xFFFFFFFFu) fold_st(out + base + (e[g] >> 16), (u8) v[g]);
  }
}
