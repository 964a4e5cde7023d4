// shim.hip -- kernels' entry point + the extern "C" ABI declared in include/mspack_hip.h.
// Host code elsewhere in the library is plain C and reaches HIP only through these functions.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <thread>
#include <chrono>
#include <vector>
#include <algorithm>
#include <array>
#include "wave_common.hpp"
#include "spec_queue.hpp"
#include "fold_common.hpp"
// lzx_kernel.hpp is compiled twice: plain LZX (CAB, CHM) and LZX DELTA (OAB) -- see its header
namespace lzxn {
#include "lzx_kernel.hpp"
}
#define LZX_DELTA 1
namespace lzxd {
#include "lzx_kernel.hpp"
}
#undef LZX_DELTA
#define LZX_PARSE_ONLY 1
#ifndef LZX_LIT_RING
#define LZX_LIT_RING 1024u      /* bytes of literals a parse wave stages in LDS: they leave as whole 16-byte rows (a power of two) */
#endif
#ifndef LZX_STAGE_WORDS
#define LZX_STAGE_WORDS 768u    /* 3 KiB of a frame's input per pass + the 1 KiB literal ring: the pipe's 10 KiB LDS block (16 waves per CU).
                                   Measured (profiles/round4_ring_variants.txt): 4 KiB stage + 2 KiB ring at 12 waves per CU 3.00 / 5.87 ms
                                   (4096 / 8192 intervals), 2 KiB + 2 KiB at 16 waves 3.06 / 5.92, this 2.89 / 5.58 */
#endif
namespace lzxp {
#include "lzx_kernel.hpp"
}
#undef LZX_PARSE_ONLY
static_assert(sizeof(lzxp::LzxFrameRec) == sizeof(lzxn::LzxFrameRec), "one record layout");
#include "mszip_kernel.hpp"
#include "qtm_kernel.hpp"
#include "lzss_kernel.hpp"

// One wavefront == one workgroup == one unit.  blockIdx -> unit through the optional launch order
// (longest unit first keeps the tail of the batch short).  One kernel per codec (their register
// budgets differ a lot); a block whose unit belongs to another codec exits at once.
__device__ __forceinline__ bool pick_unit(const mspack_hip_unit *units, const u32 *order, u32 n_units,
                                          u32 kind, u32 &ui)
{
  u32 b = blockIdx.x;
  if (b >= n_units) return false;
  ui = rfl(order ? order[b] : b);
  return units[ui].kind == kind;
}

// ---- LZX / MSZIP work scratch (d_frame_scratch of the C ABI), n = n_frames_total + 1 frame slots ----------------
//   int32  meta[n]        per frame: intel_filesize to apply in the E8 pass (0 = none)
//   u32    frame_unit[n]  per frame slot: the unit it belongs to when a parse wave should take it, else ~0
//   u32    hdr[256]       per launch (up to 32 concurrent ones) 8 words: [0] = most, [1] = fewest frames of a unit with a
//                         frame table, [2] = ticket counter of mspack_lzx_pipe, [4] = chunks handed out of the launch's pool
//   LzxFrameRec recs[n]   what the parse wave of that frame assumed and found (lzx_kernel.hpp), incl. its chunk list
//   uint2  pool[n * REC_POOL_PER_SLOT * REC_CHUNK]   the frames' match records (wave_common.hpp: RecPool): 48 KiB per slot
//                         on average instead of round 3's 128 KiB worst case per slot; a launch uses the part that
//                         belongs to its slot range
struct LzxScratch { int32_t *meta; u32 *frame_unit; u32 *hdr; lzxn::LzxFrameRec *recs; uint2 *pool; size_t bytes; };
// n_rec_slots: frame slots that can hold a record + records -- all of them for a caller's own scratch (the size
// mspack_hip_frame_scratch_bytes states); the host path numbers the units that carry a table first and gives only those
// a record and a share of the pool (a batch of OAB blocks or of folders without tables needs the 4-byte meta words only)
#define REC_SLOT_RECORDS ((size_t) REC_POOL_PER_SLOT * REC_CHUNK)
__host__ __device__ static inline LzxScratch lzx_scratch(void *base, size_t n_frames_total, size_t n_rec_slots)
{
  const size_t n0 = n_frames_total + 1, n = n_rec_slots + 1, a = 255;
  const size_t o_fu = (n0 * 4 + a) & ~a, o_hdr = o_fu + ((n * 4 + a) & ~a), o_rec = o_hdr + 1024,
               o_pool = (o_rec + n * sizeof(lzxn::LzxFrameRec) + a) & ~a;
  LzxScratch L;
  const uintptr_t b = (uintptr_t) base;                  // (a NULL base only asks for the size: no arithmetic on a null POINTER)
  L.meta = (int32_t *) b; L.frame_unit = (u32 *)(b + o_fu); L.hdr = (u32 *)(b + o_hdr); L.recs = (lzxn::LzxFrameRec *)(b + o_rec);
  L.pool = (uint2 *)(b + o_pool);
  L.bytes = o_pool + n * REC_SLOT_RECORDS * sizeof(uint2);
  return L;
}

// one unit's frame slots: which of them get a parse wave, and the launch's minimum / maximum frames per unit
__device__ __forceinline__ void frame_map_unit(const mspack_hip_unit &u, const u32 ui, u32 *frame_unit, lzxn::LzxFrameRec *recs,
                                               u32 *hdr, const u32 kind)
{
  const u32 nreal = (u.out_len + LZX_FRAME - 1u) / LZX_FRAME;
  const bool usable = (u.flags & MSPACK_HIP_UF_FRAME_TABLE) != 0u && !(kind == MSPACK_HIP_KIND_MSZIP && (u.flags & (MSPACK_HIP_UF_MSZIP_REPAIR | MSPACK_HIP_UF_MSZIP_KWAJ)));
  // frame slots of a unit: LZX out_len/32768 + 1 (one spare for the look-ahead frame), MSZIP with a table one per block
  const u32 nslots = kind == MSPACK_HIP_KIND_LZX ? u.out_len / LZX_FRAME + 1u : (usable ? nreal : 0u);
  if (threadIdx.x == 0) {                   // (a plain look first: 4096 atomics on one word take 0.1 ms)
    const u32 v = usable ? nreal : 0u;
    if (hdr[0] < v) atomicMax(&hdr[0], v);
    if (hdr[1] > v) atomicMin(&hdr[1], v);
  }
  // (units without a usable table own no record slots -- the host path numbers them behind the last slot that has a
  // record: nothing of theirs is written here; frame_unit[] is preset to ~0 for the launch's slot range)
  if (!usable) return;
  for (u32 f = threadIdx.x; f < nslots; f += 64u) {
    frame_unit[u.frame_base + f] = f < nreal ? ui : 0xFFFFFFFFu;
    recs[u.frame_base + f].status = 0u;
    if (kind == MSPACK_HIP_KIND_MSZIP) ((ZipBlockRec *) &recs[u.frame_base + f])->fold = 0u;
  }
  if (threadIdx.x == 0 && kind == MSPACK_HIP_KIND_MSZIP) atomicAdd(&hdr[5], 1u);      // (units with a table: mspack_mszip_fold's rule)
}

// the same for mspack_lzx_pipe, one unit per LANE (4096 one-wave blocks with two atomics each on the same words took
// 0.19 ms): frame slots -> unit, record status words cleared, most / fewest frames per unit reduced per wave first
__global__ __launch_bounds__(64)
void mspack_lzx_pipe_map(const mspack_hip_unit *units, const u32 *order, u32 n_units, u32 *frame_unit,
                         lzxn::LzxFrameRec *recs, u32 *ctl)
{
  const u32 j = blockIdx.x * 64u + threadIdx.x;
  u32 fr = 0;                                                    // real frames of a unit whose frames get parse tasks
  bool other = false;
  if (j < n_units) {
    const u32 ui = order ? order[j] : j;
    const mspack_hip_unit u = units[ui];
    if (u.kind == MSPACK_HIP_KIND_LZX) {
      const bool usable = (u.flags & MSPACK_HIP_UF_FRAME_TABLE) != 0u;
      const u32 nreal = (u.out_len + LZX_FRAME - 1u) / LZX_FRAME, nslots = u.out_len / LZX_FRAME + 1u;
      // (a unit without a table owns no record slots: the host path numbers such units behind the last slot that has a
      // record, so nothing of theirs may be written -- frame_unit[] is preset to ~0 for the launch's slot range)
      if (usable) {
        for (u32 f = 0; f < nslots; f++) {
          frame_unit[u.frame_base + f] = f < nreal ? ui : 0xFFFFFFFFu;
          recs[u.frame_base + f].status = 0u;
          recs[u.frame_base + f].chain = 0u;
          recs[u.frame_base + f].rst = 0u;
        }
        recs[u.frame_base].rs_valid = 0u;
        // (what decides between lzx_pipe_resolve and mspack_lzx_fold: how many units carry a table; positions beyond 2^31 do not fit the fold's map)
        atomicAdd(&ctl[5], 1u);
        if (u.out_len > 0x7FFF0000u) atomicOr(&ctl[6], 1u);
      }
      fr = usable ? nreal : 0u;
    }
    else other = true;
  }
  const u32 live = j < n_units ? 1u : 0u;
  u32 mx = fr, mn = (live && !other) ? fr : (other ? 0u : 0xFFFFFFFFu);
  for (int o = 32; o >= 1; o >>= 1) {
    const u32 a = (u32) __builtin_amdgcn_ds_bpermute((int)(((threadIdx.x + (u32) o) & 63u) << 2), (int) mx);
    const u32 b = (u32) __builtin_amdgcn_ds_bpermute((int)(((threadIdx.x + (u32) o) & 63u) << 2), (int) mn);
    mx = a > mx ? a : mx; mn = b < mn ? b : mn;
  }
  if (threadIdx.x == 0) { atomicMax(&ctl[0], mx); atomicMin(&ctl[1], mn); }
}

// which frame slots get a parse wave: the real frames of LZX units that carry a frame table
__global__ __launch_bounds__(64)
void mspack_lzx_frame_map(const mspack_hip_unit *units, const u32 *order, u32 n_units, u32 *frame_unit,
                          lzxn::LzxFrameRec *recs, u32 *hdr, u32 kind)
{
  u32 ui;
  if (!pick_unit(units, order, n_units, kind, ui)) { if (threadIdx.x == 0 && hdr[1] != 0u) atomicMin(&hdr[1], 0u); return; }
  frame_map_unit(units[ui], ui, frame_unit, recs, hdr, kind);
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4)))
void mspack_decode_lzx(const mspack_hip_unit *units, const u32 *order, u32 n_units,
                       const u8 *in_arena, u8 *out_arena, mspack_hip_result *results,
                       int32_t *frame_meta, const lzxn::LzxFrameRec *recs, const uint2 *toks, u32 resume)
{
  __shared__ lzxn::LzxShared sh;
  u32 ui;
  if (!pick_unit(units, order, n_units, MSPACK_HIP_KIND_LZX, ui)) return;
  const mspack_hip_unit u = units[ui];
  mspack_hip_result *res = &results[ui];
  const u32 lane = threadIdx.x;
  lzxn::lzx_decode_unit(u, in_arena, out_arena, frame_meta, res, &sh, recs, toks, resume != 0u);
  // E8 translation, frame by frame, once the unit no longer needs its window (lzxd.c:706-736)
  if (frame_meta) {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    u32 produced = rfl(res->out_len);
    u32 nfr = (produced + LZX_FRAME - 1u) / LZX_FRAME;
    for (u32 f = 0; f < nfr; f++) {
      int32_t fs = (int32_t) rfl((u32) frame_meta[u.frame_base + f]);
      if (fs == 0) continue;
      // the frame size the decoder saw: full frames except the last one of the stream
      u32 fsize = u.out_len - f * LZX_FRAME; if (fsize > LZX_FRAME) fsize = LZX_FRAME;
      lzxn::lzx_e8_frame(out_arena + u.out_off + (size_t) f * LZX_FRAME, fsize,
                         (int32_t)((u32) u.e8_base + f * LZX_FRAME), fs, lane);
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// mspack_lzx_pipe -- headers, parse and match resolution of a launch's LZX units as ONE dependency-driven launch.
//
// Persistent waves pull TICKETS from a counter; a ticket is the task of one FRAME of one unit:
//   parse (lzx_pipe_parse):     header chain link (waits for the previous frame's code lengths), tables, tokens: literals
//                               stored in place, one record per match;
//   resolve (lzx_pipe_resolve): waits until the unit's previous frame is complete, checks that this frame continues it,
//                               resolves R0-R2 along the records and copies the matches, publishes the frame as complete.
// Tickets are handed out in an order in which every task only waits for tasks with EARLIER tickets -- frame-major (all first
// frames in launch order, longest unit first, then all second frames, ...) when all units have the same number of frames,
// else in frame-slot order (a unit's frames in a row).  No co-residency is assumed: a ticket is pulled by a running wave,
// so whatever a task waits for is held by a live wave or done.  Round 3 had a separate COMMIT task per unit behind the
// unit's last parse task; its ~1-1.8 ms chain was what a launch ended with (waves busy 0.77 of the span).  Now all tasks
// are alike, a wave that took a long first frame takes a short second frame (launch order is longest first in every
// section), and a unit of many frames has the parse of frame f + k running beside the copies of frame f.
// Hand-off: payload by plain stores, agent-scope release, relaxed status store; the reader polls the status relaxed, then
// one agent-scope acquire (lzx_kernel.hpp).  The first frame that is not a complete regular one ends its unit's chain and
// says where serial decoding resumes; mspack_decode_lzx (launched behind the pipe) finishes every unit.
// ---------------------------------------------------------------------------------------------------
// ---- few units of many frames: the folder's chain as one gather pass per frame (lzx_fold.hpp) ----
// Decided per launch from what the map kernel counted (ctl[0] = most frames of a unit with a table, ctl[5] = such units, ctl[6] =
// some unit is too long for the map's positions): policy 0 never, 1 when it pays, 2 whenever it can (tests).  Where it pays: the
// resolve tasks of lzx_pipe_resolve fill 16 waves per CU and cost ~0.25 ms per frame ON a unit's chain; the fold tasks fill ONE wave
// per CU (128 KiB of LDS each) and leave ~15 us per frame on the chain -- so: long units, and too few of them to fill the chip.
#ifndef LZX_FOLD_MIN_FRAMES
#define LZX_FOLD_MIN_FRAMES 4u
#endif
#ifndef LZX_FOLD_MAX_UNITS
#define LZX_FOLD_MAX_UNITS 128u
#endif
// (measured, tools/fold_policy_sweep.py, profiles/round6_fold_policy.txt: n folders of f frames, LZX, resolve tasks / fold tasks, ms:
//  4 x 256: 68.5 / 13.4; 16 x 64: 20.9 / 8.3; 32 x 32: 12.2 / 6.5; 64 x 16: 8.1 / 6.0; 128 x 8: 6.2 / 5.8; 128 x 4: 2.7 / 3.1;
//  256 x 8: 8.3 / 9.0 -- so: at most 128 units, and eight frames in the longest, or at least four when every frame gets a task of its
//  own at once; MSZIP folders gain at 128 x 4 too (2.6 / 2.2): four blocks)
#ifndef LZX_FOLD_LONG_FRAMES
#define LZX_FOLD_LONG_FRAMES 8u
#endif
__device__ __forceinline__ bool lzx_fold_on(const u32 *ctl, const u32 policy, const u32 n_slots, const bool mszip)
{
  if (policy == 0u || rfl(ctl[6]) != 0u || rfl(ctl[0]) == 0u) return false;
  if (policy >= 2u) return true;
  const u32 fmax = rfl(ctl[0]);
  if (rfl(ctl[5]) > LZX_FOLD_MAX_UNITS || fmax < LZX_FOLD_MIN_FRAMES) return false;
  return mszip || fmax >= LZX_FOLD_LONG_FRAMES || n_slots <= 256u;
}
// (a workgroup of FOLD_WAVES waves per task: fold_common.hpp; wave 0 pulls the tickets)
#define FOLD_TICKET(counter)                                                      \
  if (threadIdx.x == 0) sh.ctl[0] = atomicAdd(counter, 1u);                      \
  fold_barrier();                                                                 \
  const u32 t = rfl(sh.ctl[0]);                                                   \
  fold_barrier();                                    /* (the word is free again) */
// A unit whose matches are long RUNS (the reference's large-files.test: ~127 matches of 257 bytes per frame, one line repeated) is
// better off with lzx_pipe_resolve: its run fill writes such a frame without reading anything back (spec_queue.hpp), while a fold
// task would gather every byte (measured: 1.55-1.78 GB/s against 1.2).  Decided per unit from its FIRST frame's record -- every task
// of the unit, in both kernels, reads the same final words: at least 16 matches, and 96 bytes of output or more per match.
__device__ __forceinline__ bool lzx_unit_runs(const lzxn::LzxFrameRec *r0)
{
  u32 st = lzxn::lzx_status_load(&r0->status);
  // (the unit's first parse task has the unit's earliest ticket: a live wave holds it, or it is done)
  for (u32 tries = 0; (st == LZX_ST_NONE || st == LZX_ST_CLAIMED || st == LZX_ST_HEADER) && tries < (1u << 24); tries++) {
    __builtin_amdgcn_s_sleep(8);
    st = lzxn::lzx_status_load(&r0->status);
  }
  if (st != LZX_ST_EMITTED) return false;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  const u32 n = rfl(gld(&r0->n_tokens)), b = rfl(gld(&r0->bytes_done));
  return n >= 16u && b >= 96u * n;
}
__global__ __launch_bounds__(FOLD_THREADS)
void mspack_lzx_fold(const mspack_hip_unit *units, u32 slot_lo, u32 n_slots, u8 *out_arena, const u32 *frame_unit, u32 *ctl,
                     lzxn::LzxFrameRec *recs, const uint2 *toks, u32 fold_policy)
{
  __shared__ lzxn::LzxFoldLds sh;
  if (!lzx_fold_on(ctl, fold_policy, n_slots, false)) return;
  for (;;) {
    FOLD_TICKET(&ctl[7])
    if (t >= n_slots) break;
    const u32 slot = slot_lo + t;
    const u32 ui = rfl(frame_unit[slot]);
    if (ui == 0xFFFFFFFFu) continue;
    const mspack_hip_unit u = units[ui];
    if (u.kind != MSPACK_HIP_KIND_LZX || !(u.flags & MSPACK_HIP_UF_FRAME_TABLE)) continue;
    if (fold_policy == 1u && lzx_unit_runs(&recs[u.frame_base])) continue;      // (resolved by the pipe's own tasks; policy 2 folds these too: tests)
    lzxn::lzx_fold_frame(u, slot - u.frame_base, out_arena, &recs[u.frame_base], toks, &sh);
    fold_barrier();                                   // the next task reuses the LDS
  }
}

union LzxPipeLds { lzxp::LzxShared p; lzxn::LzxResolveLds r; };
static_assert(sizeof(LzxPipeLds) <= 10240, "16 waves per CU");

// the two halves of a task are real calls: each gets its own register allocation (inlined into the ticket loop they spill)
__device__ __attribute__((noinline)) u32 lzx_pipe_task_parse(const mspack_hip_unit *up, const u32 f, const u8 *in_arena, u8 *out_arena,
                                                             lzxn::LzxFrameRec *recs, uint2 *pool, u32 *pool_head, const u32 pool_chunks,
                                                             lzxp::LzxShared *sh, const u32 spec, const u32 stream)
{
  const mspack_hip_unit u = *up;
  RecPool rp; rp.base = pool; rp.head = pool_head; rp.cap = pool_chunks;
  return lzxp::lzx_pipe_parse(u, up, f, in_arena, out_arena, (lzxp::LzxFrameRec *) &recs[u.frame_base], rp, sh, stream != 0u, spec != 0u);
}
// (the rest of a frame whose first block ended inside it: one frame in a few hundred)
__device__ __attribute__((noinline)) void lzx_pipe_task_tail(const mspack_hip_unit *up, const u32 f, const u8 *in_arena, u8 *out_arena,
                                                             lzxn::LzxFrameRec *recs, uint2 *pool, u32 *pool_head, const u32 pool_chunks,
                                                             lzxp::LzxShared *sh)
{
  RecPool rp; rp.base = pool; rp.head = pool_head; rp.cap = pool_chunks;
  lzxp::lzx_pipe_parse_tail(up, f, in_arena, out_arena, (lzxp::LzxFrameRec *) &recs[rfl(up->frame_base)], rp, sh);
}
// (a frame's block header read ahead of the header chain, while the frame below is not that far: lzx_kernel.hpp)
__device__ __attribute__((noinline)) u32 lzx_pipe_task_spec(const mspack_hip_unit *up, const u32 f, const u8 *in_arena, const lzxn::LzxFrameRec *recs,
                                                            lzxp::LzxShared *sh)
{
  const u32 rf = rfl((u32) up->reset_frames);
  if (rf ? (f % rf) == 0u : f == 0u) return 0u;                 // (a frame that starts a reset interval has no chain below it)
  const lzxn::LzxFrameRec *pr = &recs[rfl(up->frame_base) + f - 1u];
  const u32 ps = lzxn::lzx_status_load(&pr->status);
  if (ps != LZX_ST_NONE && ps != LZX_ST_CLAIMED) return 0u;     // the frame below is there: nothing to wait for, nothing to guess
  const u32 in_len = rfl(up->in_len);
  const u32 fo = rfl(((const u32 *)(in_arena + (size_t) rfl(up->in_chunk) * 4u))[f]);
  if (fo >= in_len || in_len - fo <= 64u) return 0u;
  return rfl(lzxp::lzx_pipe_spec_header(up, fo, in_arena, sh) ? 1u : 0u);
}
__device__ __attribute__((noinline)) void lzx_pipe_task_resolve(const mspack_hip_unit *up, const u32 f, u8 *out_arena, lzxn::LzxFrameRec *recs,
                                                                uint2 *toks, lzxn::LzxResolveLds *rl, const bool merged)
{
  const mspack_hip_unit u = *up;
  lzxn::lzx_pipe_resolve(u, f, out_arena, &recs[u.frame_base], toks, rl, merged);
}
// (the same where the launch has wave slots to spare: the frame's records taken up while it is parsed -- lzx_kernel.hpp)
__device__ __attribute__((noinline)) void lzx_pipe_task_resolve_stream(const mspack_hip_unit *up, const u32 f, u8 *out_arena, lzxn::LzxFrameRec *recs,
                                                                       uint2 *toks, lzxn::LzxResolveLds *rl)
{
  const mspack_hip_unit u = *up;
  lzxn::lzx_pipe_resolve_stream(u, f, out_arena, &recs[u.frame_base], toks, rl);
}

#ifdef LZX_PIPE_TRACE      /* analysis builds: one line per ticket = start, end (s_memrealtime, 100 MHz), task, time waited */
__device__ unsigned long long g_pipe_trace[4 << 16];
#endif
#ifndef LZX_PIPE_WAVES_PER_EU
#define LZX_PIPE_WAVES_PER_EU 4
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(LZX_PIPE_WAVES_PER_EU)))
void mspack_lzx_pipe(const mspack_hip_unit *units, const u32 *order, u32 n_units, u32 slot_lo, u32 n_slots,
                     const u8 *in_arena, u8 *out_arena, mspack_hip_result *results, int32_t *frame_meta,
                     const u32 *frame_unit, u32 *ctl, lzxn::LzxFrameRec *recs, uint2 *toks, u32 pool_chunks, u32 fold_policy, u32 order_mode)
{
  __shared__ LzxPipeLds sh;
  const u32 lane = threadIdx.x;
  // all units carry a table and have the same number of frames F (CHM reset intervals): 2 * F sections of n_units tickets,
  // each in launch order (longest unit first) -- P(frame 0), ..., P(frame F-1), R(frame 0), ..., R(frame F-1).  A task's
  // dependencies lie at least a section back, so with more units than waves nobody waits (one task per frame -- parse +
  // resolve -- was measured first: the waves that finish the short first frames early take the LONGEST units' second
  // frames and then sit on their slots until those units' first frames are through: 253 us waited per task, headline 3.25 ms).
  // Otherwise: one ticket per frame slot, parse + resolve by the same wave, a unit's frames in a row.
  const u32 Fmax = rfl(ctl[0]), Fmin = rfl(ctl[1]);
  // few units of many frames: this launch only PARSES (a unit's frames in a row: the header chain); mspack_lzx_fold, launched behind
  // it, does what lzx_pipe_resolve would have done (lzx_fold.hpp)
  const bool fold = lzx_fold_on(ctl, fold_policy, n_slots, false);
  // control word 3 (the host's): this launch has a wave for every ticket and nothing runs beside it -- resolve tasks take their
  // frames up while they are parsed (lzx_pipe_resolve_stream)
  const u32 stream_ok = rfl(ctl[3]);
  const u32 F = (Fmax != 0u && Fmax == Fmin && !fold) ? Fmax : 0u;
  const u32 T = F ? 2u * n_units * F : n_slots;
  const u32 stream = (stream_ok != 0u && F != 0u && T <= gridDim.x) ? 1u : 0u;      // (every ticket finds a wave at once)
  // Ticket order of a uniform launch (round 6; measured: profiles/round6_ticket_order.txt).  Every order is correct -- a task only
  // ever waits for earlier tickets --; what differs is who runs beside whom.  Level order (P(f0) | P(f1) | R(f0) | R(f1), §4.1c) is
  // right when the launch has about as many units as the chip has waves: nobody waits.  When every ticket finds a wave at once
  // (T <= gridDim.x: 1024 intervals) the order only says which tasks share a CU, and level order gives a CU sixteen tasks of ONE
  // kind -- unit-major (a unit's tasks in a row) mixes them: 1.53 -> 1.40 ms.  With many more units than waves, sections that
  // alternate P(f_k) and R(f_k-1) keep parse and resolve waves side by side through the launch: 8192 intervals 5.34 -> 5.24 ms.
  // order_mode: 0 level, 1 mixed sections, 2 unit-major; 3 (the default) = by the launch's shape.
  u32 mode = order_mode;
  if (mode >= 3u) mode = T <= gridDim.x ? 2u : (2u * n_units >= 3u * gridDim.x ? 1u : 0u);
  for (;;) {
    u32 t = 0;
    if (lane == 0) t = atomicAdd(&ctl[2], 1u);
    t = rfl(t);
    if (t >= T) break;
    u32 ui = 0xFFFFFFFFu, f = 0;
    bool do_parse = true, do_resolve = !fold;
    if (F) {
      u32 ix;
      if (mode == 1u) {
        // P(f0) | P(f1) and R(f0) alternating | ... | R(f_last): parse and resolve tasks side by side on every CU
        if (t < n_units) { ix = t; f = 0u; do_resolve = false; }
        else {
          const u32 t1 = t - n_units, k = t1 / (2u * n_units) + 1u;
          if (k >= F) { ix = t1 - 2u * n_units * (F - 1u); f = F - 1u; do_parse = false; }
          else {
            const u32 w = t1 % (2u * n_units);
            ix = w >> 1;
            if (w & 1u) { f = k - 1u; do_parse = false; } else { f = k; do_resolve = false; }
          }
        }
      }
      else if (mode == 2u) {
        // a unit's tasks in a row: P(f0) .. P(f_last), R(f0) .. R(f_last) (a launch whose tickets all run at once: the order only
        // says which tasks share a CU)
        ix = t / (2u * F);
        const u32 w = t % (2u * F);
        if (w < F) { f = w; do_resolve = false; } else { f = w - F; do_parse = false; }
      }
      else {
        const u32 sct = t / n_units;
        ix = t % n_units;
        if (sct < F) { f = sct; do_resolve = false; } else { f = sct - F; do_parse = false; }
      }
      ui = rfl(order ? order[ix] : ix);
    }
    else {
      const u32 slot = slot_lo + t;
      ui = rfl(frame_unit[slot]);
      if (ui != 0xFFFFFFFFu) f = slot - rfl(units[ui].frame_base);
    }
    if (ui == 0xFFFFFFFFu) continue;
    const mspack_hip_unit *up = &units[ui];
    if (rfl((u32) up->kind) != MSPACK_HIP_KIND_LZX || !(rfl(up->flags) & MSPACK_HIP_UF_FRAME_TABLE)) continue;
#ifdef LZX_PIPE_TRACE
    const unsigned long long tr0 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) lzxn::g_pipe_wait[blockIdx.x & 0xFFFFu] = 0;
#endif
    if (do_parse) {
#ifndef LZX_NO_SPEC_HEADER
      const u32 spec = lzx_pipe_task_spec(up, f, in_arena, recs, &sh.p);
#else
      const u32 spec = 0u;
#endif
      if (lzx_pipe_task_parse(up, f, in_arena, out_arena, recs, toks, &ctl[4], pool_chunks, &sh.p, spec, stream))
        lzx_pipe_task_tail(up, f, in_arena, out_arena, recs, toks, &ctl[4], pool_chunks, &sh.p);
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");      // the resolver reuses the LDS
      if (fold) do_resolve = fold_policy == 1u && lzx_unit_runs(&recs[rfl(up->frame_base)]);    // (a unit of long runs keeps its resolve tasks)
    }
    if (do_resolve) {
      if (stream && !do_parse) lzx_pipe_task_resolve_stream(up, f, out_arena, recs, toks, &sh.r);
      else lzx_pipe_task_resolve(up, f, out_arena, recs, toks, &sh.r, do_parse);
    }
#ifdef LZX_PIPE_TRACE
    if (lane == 0 && t < (1u << 16)) {
      g_pipe_trace[4u * t] = tr0; g_pipe_trace[4u * t + 1u] = __builtin_amdgcn_s_memrealtime();
      g_pipe_trace[4u * t + 2u] = ((unsigned long long) ui << 32) | (f << 1) | (do_parse ? 0u : 1u);
      g_pipe_trace[4u * t + 3u] = lzxn::g_pipe_wait[blockIdx.x & 0xFFFFu] | ((unsigned long long) blockIdx.x << 40);
    }
#endif
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");      // the next task reuses the LDS
  }
}

// LZX DELTA units (OAB blocks): the same decoder compiled with LZX_DELTA (17.4 KiB of LDS: 9 units per CU)
__global__ __launch_bounds__(64)
void mspack_decode_lzxd(const mspack_hip_unit *units, const u32 *order, u32 n_units,
                        const u8 *in_arena, u8 *out_arena, mspack_hip_result *results,
                        int32_t *frame_meta)
{
  __shared__ lzxd::LzxShared sh;
  u32 ui;
  if (!pick_unit(units, order, n_units, MSPACK_HIP_KIND_LZX_DELTA, ui)) return;
  const mspack_hip_unit u = units[ui];
  mspack_hip_result *res = &results[ui];
  const u32 lane = threadIdx.x;
  lzxd::lzx_decode_unit(u, in_arena, out_arena, frame_meta, res, &sh);
  if (frame_meta) {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    u32 produced = rfl(res->out_len);
    u32 nfr = (produced + LZX_FRAME - 1u) / LZX_FRAME;
    for (u32 f = 0; f < nfr; f++) {
      int32_t fs = (int32_t) rfl((u32) frame_meta[u.frame_base + f]);
      if (fs == 0) continue;
      u32 fsize = u.out_len - f * LZX_FRAME; if (fsize > LZX_FRAME) fsize = LZX_FRAME;
      lzxd::lzx_e8_frame(out_arena + u.out_off + (size_t) f * LZX_FRAME, fsize,
                         (int32_t)((u32) u.e8_base + f * LZX_FRAME), fs, lane);
    }
  }
}

// one parse wave per CFDATA block of the MSZIP units that carry a frame table (mszip_kernel.hpp: "Block-level parse
// parallelism")
__global__ __launch_bounds__(64)
void mspack_mszip_parse(const mspack_hip_unit *units, const u32 *order, u32 n_units, u32 slot_lo, u32 n_slots,
                        const u8 *in_arena, u8 *out_arena, const u32 *frame_unit, u32 *hdr, lzxn::LzxFrameRec *recs, uint2 *toks, u32 pool_chunks)
{
  __shared__ MszipShared sh;
  if (blockIdx.x >= n_slots) return;
  u32 slot = slot_lo + blockIdx.x;
  const u32 F = rfl(hdr[0]);
  if (F != 0u && F == rfl(hdr[1])) {
    const u32 j = blockIdx.x / F, f = blockIdx.x % F;
    if (j >= n_units) return;
    const u32 uj = rfl(order ? order[j] : j);
    slot = units[uj].frame_base + f;
  }
  const u32 ui = rfl(frame_unit[slot]);
  if (ui == 0xFFFFFFFFu) return;
  const mspack_hip_unit u = units[ui];
  RecPool rp; rp.base = toks; rp.head = &hdr[4]; rp.cap = pool_chunks;
  zip_parse_block(u, slot - u.frame_base, in_arena, out_arena, (ZipBlockRec *) &recs[slot], rp, &sh);
}

// the copies of a launch's MSZIP blocks as fold tasks (zip_fold_block; the rule is mspack_lzx_fold's: few folders of many blocks)
__global__ __launch_bounds__(FOLD_THREADS)
void mspack_mszip_fold(const mspack_hip_unit *units, u32 slot_lo, u32 n_slots, u8 *out_arena, const u32 *frame_unit, u32 *hdr,
                       lzxn::LzxFrameRec *recs, const uint2 *toks, u32 fold_policy)
{
  __shared__ FoldLds sh;
  if (!lzx_fold_on(hdr, fold_policy, n_slots, true)) return;
  for (;;) {
    FOLD_TICKET(&hdr[7])
    if (t >= n_slots) break;
    const u32 slot = slot_lo + t;
    const u32 ui = rfl(frame_unit[slot]);
    if (ui == 0xFFFFFFFFu) continue;
    const mspack_hip_unit u = units[ui];
    {
      // (a folder of long runs -- its first block says -- stays with zip_run_tokens' run fill: lzx_unit_runs above)
      const ZipBlockRec *r0 = (const ZipBlockRec *) &recs[u.frame_base];
      const u32 n0 = rfl(gld(&r0->n_tokens)), b0 = rfl(gld(&r0->total_out));
      if (fold_policy == 1u && rfl(gld(&r0->status)) == 1u && n0 >= 16u && b0 >= 96u * n0) continue;
    }
    zip_fold_block(u, slot - u.frame_base, out_arena, (ZipBlockRec *) &recs[u.frame_base], toks, &sh);
    fold_barrier();
  }
}

__global__ __launch_bounds__(64)
void mspack_decode_mszip(const mspack_hip_unit *units, const u32 *order, u32 n_units,
                         const u8 *in_arena, u8 *out_arena, mspack_hip_result *results,
                         const lzxn::LzxFrameRec *recs, const uint2 *toks)
{
  __shared__ MszipShared sh;
  u32 ui;
  if (!pick_unit(units, order, n_units, MSPACK_HIP_KIND_MSZIP, ui)) return;
  const mspack_hip_unit u = units[ui];
  mszip_decode_unit(u, in_arena, out_arena, &results[ui], &sh, (const ZipBlockRec *) recs, toks);
}
static_assert(sizeof(ZipBlockRec) == sizeof(lzxn::LzxFrameRec), "MSZIP and LZX share the work scratch");

__global__ __launch_bounds__(64)
void mspack_decode_qtm(const mspack_hip_unit *units, const u32 *order, u32 n_units,
                       const u8 *in_arena, u8 *out_arena, mspack_hip_result *results)
{
  __shared__ QtmShared sh;
  u32 ui;
  if (!pick_unit(units, order, n_units, MSPACK_HIP_KIND_QUANTUM, ui)) return;
  const mspack_hip_unit u = units[ui];
  if ((u.flags & MSPACK_HIP_UF_QTM_MARKS) && u.ref_len) return;          // (mspack_decode_qtm_marks' unit)
  qtm_decode_unit<false>(u, in_arena, out_arena, &results[ui], &sh);
}
// the same for the units that carry marks (MSPACK_HIP_UF_QTM_MARKS: what requests ending at the marked positions hold back) -- the
// cabinet driver's Quantum folders; launched behind mspack_decode_qtm over the same list, each kernel leaves the other's units alone
__global__ __launch_bounds__(64)
void mspack_decode_qtm_marks(const mspack_hip_unit *units, const u32 *order, u32 n_units,
                             const u8 *in_arena, u8 *out_arena, mspack_hip_result *results)
{
  __shared__ QtmShared sh;
  u32 ui;
  if (!pick_unit(units, order, n_units, MSPACK_HIP_KIND_QUANTUM, ui)) return;
  const mspack_hip_unit u = units[ui];
  if (!((u.flags & MSPACK_HIP_UF_QTM_MARKS) && u.ref_len)) return;
  qtm_decode_unit<true>(u, in_arena, out_arena, &results[ui], &sh);
}

__global__ __launch_bounds__(64)
void mspack_decode_lzss(const mspack_hip_unit *units, const u32 *order, u32 n_units,
                        const u8 *in_arena, u8 *out_arena, mspack_hip_result *results)
{
  u32 ui;
  if (!pick_unit(units, order, n_units, MSPACK_HIP_KIND_LZSS, ui)) return;
  const mspack_hip_unit u = units[ui];
  lzss_decode_unit(u, in_arena, out_arena, &results[ui]);
}

__global__ __launch_bounds__(64)
void mspack_decode_kwaj_lzh(const mspack_hip_unit *units, const u32 *order, u32 n_units,
                            const u8 *in_arena, u8 *out_arena, mspack_hip_result *results)
{
  __shared__ LzhShared sh;
  u32 ui;
  if (!pick_unit(units, order, n_units, MSPACK_HIP_KIND_KWAJ_LZH, ui)) return;
  const mspack_hip_unit u = units[ui];
  kwaj_lzh_decode_unit(u, in_arena, out_arena, &results[ui], &sh);
}

// cabd_checksum (cabd.c:1462-1479) of a unit's input bytes with seed 0 -> result.in_next.  The XOR of the unit's dwords taken
// at its own (byte) alignment equals the byte-aligned window of the XORs of the ALIGNED dwords around it -- alignbyte is
// linear over XOR --, so every lane XORs aligned dwords (coalesced), and the two ends are fixed up once.
__global__ __launch_bounds__(64)
void mspack_xorsum(const mspack_hip_unit *units, const u32 *order, u32 n_units, const u8 *in_arena, mspack_hip_result *results)
{
  u32 ui;
  if (!pick_unit(units, order, n_units, MSPACK_HIP_KIND_XORSUM, ui)) return;
  const mspack_hip_unit u = units[ui];
  const u32 lane = threadIdx.x;
  const u8 *p = in_arena + u.in_off;
  const u32 nd = u.in_len >> 2, sh = (u32)((size_t) p & 3u);
  const u32 *w = (const u32 *)(p - sh);                  // aligned dwords; w[nd] exists (the arena's slack) when sh != 0
  u32 a = 0;
  for (u32 j = lane; j < nd; j += WAVE) a ^= w[j];
  for (int o = 32; o >= 1; o >>= 1) a ^= (u32) __builtin_amdgcn_ds_bpermute((int)(((lane ^ (u32) o) & 63u) << 2), (int) a);
  if (lane == 0) {
    u32 sum = a;
    if (sh && nd) {
      const u32 b = a ^ w[0] ^ w[nd];                     // the XOR of w[1 .. nd]
      sum = __builtin_amdgcn_alignbyte(b, a, sh);
    }
    const u8 *t = p + (size_t) nd * 4u;
    u32 tail = 0;
    switch (u.in_len & 3u) {
    case 3: tail |= (u32) *t++ << 16;   /* fall through */
    case 2: tail |= (u32) *t++ << 8;    /* fall through */
    case 1: tail |= *t;
    }
    mspack_hip_result r;
    r.err = ERR_OK; r.flags = 0; r.out_len = 0; r.in_used = u.in_len; r.good_len = 0; r.in_next = sum ^ tail;
    results[ui] = r;
  }
}

// ---------------------------------------------------------------------------------------------------
// Host side of the C ABI.
// ---------------------------------------------------------------------------------------------------
#include <mutex>
#include <condition_variable>
#include <atomic>
#define MSPK_MAX_DEV_CACHE 64
static thread_local char g_err[256] = "";
static int fail(hipError_t e, const char *what) {
  snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
  return -(int) e;
}
#define CK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(e_, #call); } while (0)

// one launch per codec over a COMPACT list of that codec's units (order[0..n) = unit indices).  LZX units
// that carry a frame table get their frames parsed by one wave each first (slots [slot_lo, slot_lo + n_slots)
// of the work scratch belong to this launch).
static int env_int(const char *name, int dflt, int lo, int hi);
static const bool g_no_frames = getenv("MSPACK_HIP_NO_FRAME_PARSE") != nullptr;     // experiments: serial path only
// MSPACK_HIP_FOLD: 0 = a folder's copies always through lzx_pipe_resolve, 1 (default) = through mspack_lzx_fold when the launch is few
// long units, 2 = whenever the units allow it (tests, A/B runs)
// MSPACK_HIP_STREAM_RESOLVE=0: resolve tasks never take frames up while they are parsed (A/B runs)
static const int g_stream_resolve = getenv("MSPACK_HIP_STREAM_RESOLVE") ? atoi(getenv("MSPACK_HIP_STREAM_RESOLVE")) : 1;     // (2: also in launches that run beside others -- A/B runs)
static const u32 g_fold_policy = getenv("MSPACK_HIP_FOLD") ? (u32) atoi(getenv("MSPACK_HIP_FOLD")) : 1u;
// MSPACK_HIP_TICKET_ORDER (A/B runs, tests): 0 level order, 1 mixed sections, 2 unit-major; 3 (default): by the launch's shape --
// unit-major when every ticket finds a wave at once, mixed sections from 1.5 x as many units as waves on, level order in between
static const u32 g_ticket_order = getenv("MSPACK_HIP_TICKET_ORDER") ? (u32) atoi(getenv("MSPACK_HIP_TICKET_ORDER")) : 3u;
// persistent waves of mspack_lzx_pipe: as many as the device holds at once (nothing depends on that number being right)
// (cached per device: mspack_hip_decode_batch_multi runs one host thread per device)
static unsigned lzx_pipe_waves()
{
  static std::mutex mu;
  static unsigned cache[MSPK_MAX_DEV_CACHE] = { 0 };
  int dev = 0, per_cu = 0; hipDeviceProp_t pr;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MSPK_MAX_DEV_CACHE) return 4096u;
  std::lock_guard<std::mutex> lock(mu);
  if (!cache[dev]) {
    if (hipGetDeviceProperties(&pr, dev) != hipSuccess) return 4096u;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mspack_lzx_pipe, 64, 0);
    if (e != hipSuccess || per_cu < 1) per_cu = 16;
    { const char *ev = getenv("MSPACK_HIP_PIPE_WAVES_PER_CU"); if (ev && atoi(ev) > 0) per_cu = atoi(ev); }
    cache[dev] = (unsigned) pr.multiProcessorCount * (unsigned) per_cu;
  }
  return cache[dev];
}
// waves of mspack_lzx_fold: one per CU (its LDS block is most of a CU's)
static unsigned lzx_fold_waves()
{
  int dev = 0; hipDeviceProp_t pr;
  static std::mutex mu;
  static unsigned cache[MSPK_MAX_DEV_CACHE] = { 0 };
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MSPK_MAX_DEV_CACHE) return 256u;
  std::lock_guard<std::mutex> lock(mu);
  if (!cache[dev]) {
    if (hipGetDeviceProperties(&pr, dev) != hipSuccess) return 256u;
    cache[dev] = (unsigned) pr.multiProcessorCount;
  }
  return cache[dev];
}
// a kernel launch whose status is RETURNED (hipLaunchKernelGGL leaves it in the thread's "last error", which is whoever's:
// an application's stale error made round 4's entry points fail, and clearing it on entry was the application's to do)
#include <tuple>
#include <utility>
template <typename... P, typename... A>
static hipError_t launch(void (*kernel)(P...), dim3 grid, dim3 block, hipStream_t st, A... a)
{
  static_assert(sizeof...(P) == sizeof...(A), "one argument per kernel parameter");
  std::tuple<P...> vals{ (P) a... };
#ifdef MSPACK_WAVE_EMU             /* tests/emu: the kernel is a host function, a launch runs it on the emulator's wave threads */
  (void) st;
  emu_launch(grid, block, [=]() { std::apply(kernel, vals); });
  return hipSuccess;
#else
  void *args[sizeof...(P)];
  size_t i = 0;
  std::apply([&](auto &... v) { ((args[i++] = (void *) &v), ...); }, vals);
  return hipLaunchKernel((const void *) kernel, grid, block, args, 0, st);
#endif
}
#define LK(call) do { const hipError_t e_ = (call); if (e_ != hipSuccess) return e_; } while (0)
#ifdef MSPACK_HOST_CHECK
hipError_t hostcheck_launch_kind(unsigned kind, const mspack_hip_unit *d_units, const uint32_t *d_order, size_t n,
                                 const void *d_in, void *d_out, mspack_hip_result *d_results, hipStream_t st);
#endif

static hipError_t launch_kind(unsigned kind, const mspack_hip_unit *d_units, const uint32_t *d_order, size_t n,
                              const void *d_in, void *d_out, mspack_hip_result *d_results, void *d_fm, size_t n_frames_total,
                              size_t slot_lo, size_t n_slots, hipStream_t st, bool frame_tables = true, unsigned launch_ix = 0,
                              size_t n_rec_slots = (size_t) -1, bool alone = true)
{
  if (n_rec_slots == (size_t) -1) n_rec_slots = n_frames_total;
  if (n == 0) return hipSuccess;
#ifdef MSPACK_HOST_CHECK      /* tests/hostcheck: the HOST half of this file under real sanitizers -- no kernel runs, a CPU stand-in takes the launch's place in the stream */
  return hostcheck_launch_kind(kind, d_units, d_order, n, d_in, d_out, d_results, st);
#endif
  const dim3 grid((unsigned) n), block(64);
  const u8 *const in = (const u8 *) d_in;
  u8 *const out = (u8 *) d_out;
  static const u32 hdr_init[8] = { 0u, 0xFFFFFFFFu, 0u, 0u, 0u, 0u, 0u, 0u };
  static const u32 hdr_init_stream[8] = { 0u, 0xFFFFFFFFu, 0u, 1u, 0u, 0u, 0u, 0u };
  switch (kind) {
  case MSPACK_HIP_KIND_LZX: {
    LzxScratch L = lzx_scratch(d_fm, n_frames_total, n_rec_slots);
    const bool frames = d_fm != nullptr && n_slots != 0 && !g_no_frames && frame_tables;
    // launches of one batch that run on different streams (host path, several chunks) have their own control words, and
    // their own part of the record pool: the part that belongs to their frame slots
    u32 *hdr = L.hdr + 8u * (launch_ix & 15u);
    uint2 *pool = L.pool + slot_lo * REC_SLOT_RECORDS;
    const u32 pool_chunks = (u32)(n_slots * REC_POOL_PER_SLOT);
    if (frames) {
      // one dependency-driven launch: parse and resolve tasks from a ticket counter (mspack_lzx_pipe)
      LK(hipMemsetAsync(L.frame_unit + slot_lo, 0xFF, n_slots * sizeof(u32), st));
      // (resolve tasks that take their frames up while they are parsed: only where every ticket finds a wave at once -- a resolve
      // wave that has started holds its slot until its frame's parse task is through -- and no other launch runs beside this one)
      const bool stream = (alone && g_stream_resolve != 0) || g_stream_resolve >= 2;      // (and the kernel knows how many tickets the launch has)
      LK(hipMemcpyAsync(hdr, stream ? hdr_init_stream : hdr_init, sizeof(hdr_init), hipMemcpyHostToDevice, st));
      LK(launch(mspack_lzx_pipe_map, dim3((unsigned)((n + 63) / 64)), block, st, d_units, d_order, (u32) n, L.frame_unit, L.recs, hdr));
      const size_t tickets = 2u * n_slots;
      // A launch that runs beside other chunks' launches asks for a third as many waves as it has tickets (MSPACK_HIP_CHUNK_WAVE_DIV): with
      // a wave for every ticket it ran unit-major, its resolve waves waiting on their slots for the parse waves -- slots the next chunk's
      // launch could use (the first chunk's 586 intervals were through after 1.9 ms instead of the 1.25 they take alone).  Headline batch
      // to the host 7.56-7.75 -> 7.38-7.48 ms, 1024 intervals to the device / host 2.16-2.24 / 3.38-3.47 -> 2.10-2.17 / 3.30-3.36 ms; 2 and 4
      // within 0.05 ms of 3 (profiles/round6_jobs.txt)
      static const size_t wave_div = (size_t) env_int("MSPACK_HIP_CHUNK_WAVE_DIV", 3, 1, 16);
      const unsigned waves = (unsigned) std::min<size_t>(alone ? tickets : std::max<size_t>(64, tickets / wave_div), lzx_pipe_waves());
      LK(launch(mspack_lzx_pipe, dim3(waves), block, st, d_units, d_order, (u32) n, (u32) slot_lo, (u32) n_slots, in, out, d_results,
                L.meta, L.frame_unit, hdr, L.recs, pool, pool_chunks, g_fold_policy, g_ticket_order));
      // few long units: the frames' copies as fold tasks, one wave per CU (the kernel decides from what the map kernel counted and
      // leaves at once otherwise; a launch of more units than the rule allows is not even asked)
      if (g_fold_policy >= 2u || (g_fold_policy == 1u && n <= LZX_FOLD_MAX_UNITS && n_slots >= LZX_FOLD_MIN_FRAMES))
        LK(launch(mspack_lzx_fold, dim3((unsigned) std::min<size_t>(n_slots, lzx_fold_waves())), dim3(FOLD_THREADS), st, d_units, (u32) slot_lo, (u32) n_slots,
                  out, L.frame_unit, hdr, L.recs, pool, g_fold_policy));
      // what the pipe leaves: the last bytes of every unit's input (the EOF-exact reader's), the look-ahead frame, frames
      // that are not one regular block, errors, E8, the results -- the unit kernel, resuming where each unit's chain of frames ended
      LK(launch(mspack_decode_lzx, grid, block, st, d_units, d_order, (u32) n, in, out, d_results, L.meta, L.recs, pool, 1u));
      break;
    }
    LK(launch(mspack_decode_lzx, grid, block, st, d_units, d_order, (u32) n, in, out, d_results, d_fm ? L.meta : (int32_t *) nullptr,
              (const lzxn::LzxFrameRec *) nullptr, (const uint2 *) nullptr, 0u));
    break; }
  case MSPACK_HIP_KIND_LZX_DELTA:
    LK(launch(mspack_decode_lzxd, grid, block, st, d_units, d_order, (u32) n, in, out, d_results, (int32_t *) d_fm)); break;
  case MSPACK_HIP_KIND_MSZIP: {
    LzxScratch L = lzx_scratch(d_fm, n_frames_total, n_rec_slots);
    const bool frames = d_fm != nullptr && n_slots != 0 && !g_no_frames && frame_tables;
    uint2 *pool = nullptr;
    u32 pool_chunks = 0;
    if (frames) {
      // one parse wave per CFDATA block first (mszip_kernel.hpp: "Block-level parse parallelism"); a pipe of the LZX kind
      // was measured slower here (profiles/round3_mszip.txt: the blocks' parse tasks do not depend on each other)
      u32 *hdr = L.hdr + 8u * (16u + (launch_ix & 15u));
      pool = L.pool + slot_lo * REC_SLOT_RECORDS;
      pool_chunks = (u32)(n_slots * REC_POOL_PER_SLOT);
      LK(hipMemsetAsync(L.frame_unit + slot_lo, 0xFF, n_slots * sizeof(u32), st));
      LK(hipMemcpyAsync(hdr, hdr_init, sizeof(hdr_init), hipMemcpyHostToDevice, st));
      LK(launch(mspack_lzx_frame_map, grid, block, st, d_units, d_order, (u32) n, L.frame_unit, L.recs, hdr, (u32) MSPACK_HIP_KIND_MSZIP));
      LK(launch(mspack_mszip_parse, dim3((unsigned) n_slots), block, st, d_units, d_order, (u32) n, (u32) slot_lo, (u32) n_slots, in, out,
                L.frame_unit, hdr, L.recs, pool, pool_chunks));
      if (g_fold_policy >= 2u || (g_fold_policy == 1u && n <= LZX_FOLD_MAX_UNITS && n_slots >= LZX_FOLD_MIN_FRAMES))
        LK(launch(mspack_mszip_fold, dim3((unsigned) std::min<size_t>(n_slots, lzx_fold_waves())), dim3(FOLD_THREADS), st, d_units, (u32) slot_lo, (u32) n_slots,
                  out, L.frame_unit, hdr, L.recs, pool, g_fold_policy));
    }
    LK(launch(mspack_decode_mszip, grid, block, st, d_units, d_order, (u32) n, in, out, d_results,
              frames ? L.recs : (lzxn::LzxFrameRec *) nullptr, pool));
    break; }
  case MSPACK_HIP_KIND_QUANTUM:
    LK(launch(mspack_decode_qtm, grid, block, st, d_units, d_order, (u32) n, in, out, d_results));
    LK(launch(mspack_decode_qtm_marks, grid, block, st, d_units, d_order, (u32) n, in, out, d_results)); break;
  case MSPACK_HIP_KIND_LZSS:
    LK(launch(mspack_decode_lzss, grid, block, st, d_units, d_order, (u32) n, in, out, d_results)); break;
  case MSPACK_HIP_KIND_KWAJ_LZH:
    LK(launch(mspack_decode_kwaj_lzh, grid, block, st, d_units, d_order, (u32) n, in, out, d_results)); break;
  case MSPACK_HIP_KIND_XORSUM:
    LK(launch(mspack_xorsum, grid, block, st, d_units, d_order, (u32) n, in, d_results)); break;
  default: break;
  }
  return hipSuccess;
}
#undef LK

extern "C" {

#ifdef SPQ_TIMERS
/* analysis builds only: read (and clear) the resolve cycle counters of block 0's wave */
int mspack_hip_debug_counters(unsigned long long *out8) {
  unsigned long long z[8] = {0};
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(spq_tm), sizeof(z)) != hipSuccess) return -1;
  return hipMemcpyToSymbol(HIP_SYMBOL(spq_tm), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif
#ifdef QTM_TIMERS
int mspack_hip_debug_qtm_timers(unsigned long long *out8) {
  return hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_qtm_tm), 64) == hipSuccess ? 0 : -1;
}
#endif
#ifdef FOLD_TRACE
int mspack_hip_debug_fold_phases(unsigned long long *out16) {
  unsigned long long z[16] = {0};
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_fold_phase), sizeof(z)) != hipSuccess) return -1;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_fold_phase), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif
#ifdef LZX_PIPE_TRACE
int mspack_hip_debug_pipe_trace(unsigned long long *out, size_t n_words) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pipe_trace), n_words * 8) == hipSuccess ? 0 : -1;
}
/* ticks (100 MHz) per phase summed over all waves: lzxp:: phases 0-8 (parse task), lzxn:: phases 9-11 (commit task); cleared on read */
int mspack_hip_debug_pipe_phases(unsigned long long *out32) {
  unsigned long long z[16] = {0};
  if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(lzxp::g_pipe_phase), sizeof(z)) != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out32 + 16, HIP_SYMBOL(lzxn::g_pipe_phase), sizeof(z)) != hipSuccess) return -1;
  hipMemcpyToSymbol(HIP_SYMBOL(lzxp::g_pipe_phase), z, sizeof(z));
  return hipMemcpyToSymbol(HIP_SYMBOL(lzxn::g_pipe_phase), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif
const char *mspack_hip_version(void) { return "mspack-hip 0.3 (gfx950; LZX/LZX-DELTA/Quantum/MSZIP batch decode)"; }
const char *mspack_hip_last_error(void) { return g_err; }

int mspack_hip_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) { fail(e, "hipGetDeviceCount"); return 0; }
  return n;
}
int mspack_hip_set_device(int device) { CK(hipSetDevice(device)); return 0; }

size_t mspack_hip_frame_scratch_bytes(size_t n_frames_total) { return lzx_scratch(nullptr, n_frames_total, n_frames_total).bytes; }

int mspack_hip_decode_batch_device(const mspack_hip_unit *d_units, const uint32_t *d_order,
                                   size_t n_units, const void *d_in, size_t in_bytes,
                                   void *d_out, size_t out_bytes, mspack_hip_result *d_results,
                                   void *d_frame_scratch, size_t n_frames_total, unsigned kind_mask,
                                   void *stream)
{
  (void) in_bytes; (void) out_bytes;
  if (n_units == 0) return 0;
  if ((kind_mask & 0xFEu) == 0) kind_mask |= 0xFEu;     // bit k = units of kind k may be present
  // the caller's unit table lives on the device, so the kinds cannot be compacted here: every codec in the
  // mask gets the whole grid and blocks of other kinds leave at once.  Callers with mixed batches pass one
  // order list per codec and a one-bit mask (what the host-buffer entry points below do).
  for (unsigned k = 1; k <= MSPACK_HIP_KIND_XORSUM; k++)
    if (kind_mask & (1u << k))
      CK(launch_kind(k, d_units, d_order, n_units, d_in, d_out, d_results, d_frame_scratch, n_frames_total, 0, n_frames_total,
                     (hipStream_t) stream, (kind_mask & MSPACK_HIP_MASK_FRAME_TABLES) != 0u));
  return 0;
}

double mspack_hip_time_batch_device(const mspack_hip_unit *d_units, const uint32_t *d_order,
                                    size_t n_units, const void *d_in, size_t in_bytes,
                                    void *d_out, size_t out_bytes, mspack_hip_result *d_results,
                                    void *d_frame_scratch, size_t n_frames_total, unsigned kind_mask,
                                    void *stream, int iters)
{
  hipEvent_t e0, e1;
  float ms = 0;
  if (iters < 1) iters = 1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1.0;
  hipEventRecord(e0, (hipStream_t) stream);
  for (int i = 0; i < iters; i++) {
    int rc = mspack_hip_decode_batch_device(d_units, d_order, n_units, d_in, in_bytes, d_out, out_bytes,
                                            d_results, d_frame_scratch, n_frames_total, kind_mask, stream);
    if (rc) { hipEventDestroy(e0); hipEventDestroy(e1); return -1.0; }
  }
  hipEventRecord(e1, (hipStream_t) stream);
  if (hipEventSynchronize(e1) != hipSuccess) { hipEventDestroy(e0); hipEventDestroy(e1); return -1.0; }
  hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0); hipEventDestroy(e1);
  return (double) ms / iters;
}

} // extern "C"

// ---------------------------------------------------------------------------------------------------
// Host-buffer path: a persistent context per device (device arenas and pinned staging grown on demand,
// never freed per call; four streams) and a chunked pipeline.  The batch is cut into up to MSPACK_HIP_NCHUNKS
// chunks of units that are contiguous in the caller's arenas:
//     copy-in stream :  H2D chunk 0, H2D chunk 1, ...                      (one after the other: the link's rate)
//     two compute streams, chunk c on stream c mod 2: wait for chunk c's H2D -> one launch per codec over the
//                       chunk's compact unit lists (a launch's waves leave as its queue runs dry, the next chunk's
//                       launch -- on the other stream -- fills the slots they free: no tail between chunks)
//     copy-out stream:  wait for chunk c's launches -> D2H chunk c          (PCIe is full duplex)
// so the copy of chunk c+1 overlaps the decode of chunk c and the copy-back of chunk c the decode of chunk c+1.
// Round 6's end (profiles/round6_jobs.txt): the chunks' shares GROW to the device (1 : 1 : 2 : 4: the last chunk's launches end the call
// and should fill the chip) and begin with half a share to the host (the copy back is the long leg); a batch that is small beside the
// chip uses all compute streams either way; an LZX launch beside other chunks' launches asks for a third of its tickets' waves; and the
// whole pipeline can run on a thread of its own and hand its chunks over as they come back (JobProgress, mspack_hip_decode_batch_begin).
// (Four streams = four hardware queues: with more, two streams share a queue and a copy waits behind another
// chunk's kernel -- what profiles/round2_hostpath_streams.txt shows for its third chunk.)
// ---------------------------------------------------------------------------------------------------
#define MSPK_MAX_DEV 16
#define MSPK_MAX_STREAMS 8
#define MSPK_MAX_CHUNKS 8
struct DevBuf { void *p = nullptr; size_t cap = 0; };
struct DevCtx {
  std::mutex mu;
  bool ready = false;
  int ns = 0;
  hipStream_t st[MSPK_MAX_STREAMS];      // [0] copy-in (and everything of a one-chunk call), [1] copy-out, [2] [3] compute
  hipEvent_t ev_in[MSPK_MAX_CHUNKS], ev_done[MSPK_MAX_CHUNKS], ev_back[MSPK_MAX_CHUNKS];      // chunk c: input there / launches through / output back
  int max_chunks = 1, n_compute = 2;
  DevBuf d_in, d_out, d_units, d_order, d_res, d_fm;
  DevBuf h_stage;                       // pinned: results + (optionally) the output on its way to pageable memory
};
static DevCtx g_ctx[MSPK_MAX_DEV];

static int env_int(const char *name, int dflt, int lo, int hi) {
  const char *e = getenv(name);
  int v = e ? atoi(e) : dflt;
  return v < lo ? lo : (v > hi ? hi : v);
}

static hipError_t grow(DevBuf &b, size_t need, bool pinned) {
  if (need <= b.cap) return hipSuccess;
  hipError_t e;
  if (b.p) { hipDeviceSynchronize(); e = pinned ? hipHostFree(b.p) : hipFree(b.p); b.p = nullptr; b.cap = 0; if (e != hipSuccess) return e; }
  size_t cap = need + need / 4 + 4096;
  e = pinned ? hipHostMalloc(&b.p, cap, hipHostMallocDefault) : hipMalloc(&b.p, cap);
  if (e != hipSuccess) { b.p = nullptr; return e; }
  b.cap = cap;
  return hipSuccess;
}

static void host_path_account(double plan_ms, double issue_ms, double drain_ms);
// What a job (mspack_hip_decode_batch_begin) lets its caller see of a batch that is still running: which chunk a unit went into, and
// how many chunks are through -- their bytes in the caller's output buffer, their units' results written.  Chunks finish in order.
struct JobProgress {
  std::mutex mu; std::condition_variable cv;
  bool planned = false;                 // chunk_of is filled in
  std::vector<uint32_t> chunk_of;       // the caller's unit index -> chunk
  size_t done = 0;                      // chunks complete
  bool finished = false; int rc = 0;    // the call has returned (rc); nothing is promised about chunks >= done when rc != 0
};
struct Chunk {
  size_t a, b;                          // local unit range [a, b)
  uint64_t in_lo, in_hi, out_lo, out_hi;
  size_t order_off[8], order_n[8];      // per kind: slice of the order array
  size_t fm_lo, fm_n;
  bool has_ftab;                        // some LZX unit of the chunk carries a frame table
};

// bytes below out_off that belong to the unit, and the room it may write past out_len
static inline uint64_t unit_below(const mspack_hip_unit &u) {
  return (u.kind == MSPACK_HIP_KIND_LZSS || u.kind == MSPACK_HIP_KIND_KWAJ_LZH) ? 4096u
       : (u.kind == MSPACK_HIP_KIND_LZX_DELTA ? u.ref_len : 0u);
}
static inline uint64_t unit_above(const mspack_hip_unit &u) {
  if (u.kind == MSPACK_HIP_KIND_LZX && (u.flags & MSPACK_HIP_UF_LZX_LOG))                   // the reset log, where MSZIP's would be
    return ((((uint64_t) u.out_len + 32768u + 15u) & ~15ull) - u.out_len) + 4u + 4u * (uint64_t) u.ref_len;
  if (u.kind == MSPACK_HIP_KIND_QUANTUM && (u.flags & MSPACK_HIP_UF_QTM_MARKS) && u.ref_len)   // the log of its marks
    return ((((uint64_t) u.out_len + 15u) & ~15ull) - u.out_len) + 4u * (uint64_t) u.ref_len;
  if (u.kind != MSPACK_HIP_KIND_MSZIP) return 0u;
  uint64_t a = 32768u;
  if ((u.flags & MSPACK_HIP_UF_MSZIP_REPAIR) && (u.flags & MSPACK_HIP_UF_MSZIP_LOG))      // the repair log behind the slack
    a = ((((uint64_t) u.out_len + 32768u + 15u) & ~15ull) - u.out_len) + 4u + 8u * (uint64_t)(u.e8_base > 0 ? u.e8_base : 0);
  return a;
}
static inline bool unit_has_ftab(const mspack_hip_unit &u) {
  if (!(u.flags & MSPACK_HIP_UF_FRAME_TABLE)) return false;
  if (u.kind == MSPACK_HIP_KIND_LZX) return true;
  return u.kind == MSPACK_HIP_KIND_MSZIP && !(u.flags & (MSPACK_HIP_UF_MSZIP_REPAIR | MSPACK_HIP_UF_MSZIP_KWAJ));
}
static inline uint64_t unit_ftab_bytes(const mspack_hip_unit &u) { return (((uint64_t) u.out_len + 32767u) / 32768u) * 4u; }
// a table the unit reads out of the input arena besides its stream (in_chunk * 4: a frame / block table, a Quantum unit's marks)
static inline bool unit_side_table(const mspack_hip_unit &u, uint64_t &lo, uint64_t &hi) {
  if (unit_has_ftab(u)) { lo = (uint64_t) u.in_chunk * 4u; hi = lo + unit_ftab_bytes(u); return true; }
  if (u.kind == MSPACK_HIP_KIND_QUANTUM && (u.flags & MSPACK_HIP_UF_QTM_MARKS) && u.ref_len) {
    lo = (uint64_t) u.in_chunk * 4u; hi = lo + 4u * (uint64_t) u.ref_len; return true;
  }
  return false;
}
static inline size_t unit_frames(const mspack_hip_unit &u) {
  if (u.kind == MSPACK_HIP_KIND_LZX || u.kind == MSPACK_HIP_KIND_LZX_DELTA) return (size_t) u.out_len / 32768u + 1u;
  if (u.kind == MSPACK_HIP_KIND_MSZIP && unit_has_ftab(u)) return ((size_t) u.out_len + 32767u) / 32768u;   // one per CFDATA block
  return 0u;
}

// ---- page-locked host ranges and the copies that touch them ---------------------------------------------------------
// ROOT CAUSE of round 4's intermittent "GPU batch decode failed ... invalid argument" (DESIGN.md sec. 8h):
// the runtime treats EVERY host address inside a registered range as that registration's memory, and a copy whose host
// side starts inside a registration and ends beyond it is refused with hipErrorInvalidValue.  Rounds 3-4 rounded their
// registrations OUTWARD to whole pages, so the first and last page also held whatever the allocator had put next to the
// caller's buffer -- e.g. this file's own std::vector<> of launch orders, 4 KiB that began in the arena's last page and
// ended behind it (glibc serves a 24 MiB arena from the brk heap once an mmap'd block of that size has been freed: the
// SECOND decompressor of a process).  Rules since:
//   (1) a registration never holds a byte outside the range its owner passed (whole pages INSIDE it);
//   (2) every copy between the device and host memory is cut at the boundaries of the registrations this library made
//       (mspack_hip_pin's registry + the call's own), so that no piece straddles one.
struct PinRange { uintptr_t ra, rb; const void *user; };
static std::mutex g_pin_mu;
static std::vector<PinRange> g_pins;
static const uintptr_t MSPK_PAGE = 4096u;
static inline bool inner_pages(const void *p, size_t bytes, uintptr_t &ra, uintptr_t &rb) {
  ra = ((uintptr_t) p + MSPK_PAGE - 1u) & ~(MSPK_PAGE - 1u);
  rb = ((uintptr_t) p + bytes) & ~(MSPK_PAGE - 1u);
  return rb > ra;
}
// the boundaries of known registrations strictly inside (lo, hi), ascending
static void pin_cuts(uintptr_t lo, uintptr_t hi, const PinRange *extra, int n_extra, std::vector<uintptr_t> &cuts) {
  cuts.clear();
  auto add = [&](const PinRange &r) { if (r.ra > lo && r.ra < hi) cuts.push_back(r.ra); if (r.rb > lo && r.rb < hi) cuts.push_back(r.rb); };
  { std::lock_guard<std::mutex> lock(g_pin_mu); for (const PinRange &r : g_pins) add(r); }
  for (int i = 0; i < n_extra; i++) add(extra[i]);
  std::sort(cuts.begin(), cuts.end());
}
// hipMemcpyAsync with the HOST side cut at registration boundaries
static hipError_t copy_cut(void *dst, const void *src, size_t n, hipMemcpyKind kind, hipStream_t st,
                           const PinRange *extra = nullptr, int n_extra = 0) {
  if (!n) return hipSuccess;
  const uintptr_t h = kind == hipMemcpyHostToDevice ? (uintptr_t) src : (uintptr_t) dst;
  std::vector<uintptr_t> cuts;
  pin_cuts(h, h + n, extra, n_extra, cuts);
  uintptr_t at = h;
  for (size_t i = 0; i <= cuts.size(); i++) {
    const uintptr_t to = i < cuts.size() ? cuts[i] : h + n;
    if (to <= at) continue;
    const hipError_t e = hipMemcpyAsync((char *) dst + (at - h), (const char *) src + (at - h), to - at, kind, st);
    if (e != hipSuccess) return e;
    at = to;
  }
  return hipSuccess;
}

// `sel` lists the unit indices this device handles (NULL = all n_sel units).  host_out != NULL: outputs are
// copied back into it; dev_out != NULL: the caller's DEVICE buffer receives them (out_off relative to it).
// per_unit_back: copy the outputs back unit by unit (a sharded call whose shards' output spans interleave)
static int pipeline_on_current_device(int dev, mspack_hip_unit *units, const uint32_t *sel, size_t n_sel,
                                      const void *in, size_t in_bytes, void *host_out, void *dev_out,
                                      size_t out_bytes, mspack_hip_result *results, char *errbuf, size_t errcap,
                                      bool per_unit_back = false, JobProgress *pg = nullptr)
{
  if (n_sel == 0) return 0;
  if (dev < 0 || dev >= MSPK_MAX_DEV) { snprintf(errbuf, errcap, "device index %d out of range", dev); return -1; }
  DevCtx &cx = g_ctx[dev];
  std::lock_guard<std::mutex> lock(cx.mu);
  hipError_t e;
  int rc = 0;
#define TRY(call) do { e = (call); if (e != hipSuccess) { snprintf(errbuf, errcap, "%s: %s", #call, hipGetErrorString(e)); rc = -(int) e; goto done; } } while (0)
  static const bool trace = getenv("MSPACK_HIP_TRACE") != nullptr;
  auto tnow = []() { return std::chrono::steady_clock::now(); };
  auto tms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return std::chrono::duration<double, std::milli>(b - a).count(); };
  auto t0 = tnow(), t1 = t0, t2 = t0, t3 = t0;

  // ---- plan: units in arena order, cut into chunks ----
  std::vector<uint32_t> idx(n_sel);
  for (size_t i = 0; i < n_sel; i++) idx[i] = sel ? sel[i] : (uint32_t) i;
  std::stable_sort(idx.begin(), idx.end(), [&](uint32_t x, uint32_t y) { return units[x].in_off < units[y].in_off; });
  std::vector<mspack_hip_unit> local(n_sel);
  bool monotone = !per_unit_back;
  uint64_t in_lo = ~0ull, in_hi = 0, out_lo = ~0ull, out_hi = 0, prev_hi = 0, in_sum = 0;
  size_t n_frames = 0, n_rec_slots = 0;
  for (size_t i = 0; i < n_sel; i++) {
    mspack_hip_unit &u = local[i];
    u = units[idx[i]];
    if (u.kind != MSPACK_HIP_KIND_LZX_DELTA && !(u.kind == MSPACK_HIP_KIND_LZX && (u.flags & MSPACK_HIP_UF_LZX_LOG)) &&
        !(u.kind == MSPACK_HIP_KIND_QUANTUM && (u.flags & MSPACK_HIP_UF_QTM_MARKS))) u.ref_len = 0;
    if (u.kind > MSPACK_HIP_KIND_XORSUM) { snprintf(errbuf, errcap, "unit %u: unknown kind %u", idx[i], u.kind); return -1; }
    if (u.kind == MSPACK_HIP_KIND_XORSUM) {                // reads its input, owns no output
      if (u.out_len) { snprintf(errbuf, errcap, "unit %u: a checksum unit has no output", idx[i]); return -1; }
      if (u.in_off + u.in_len > in_bytes) { snprintf(errbuf, errcap, "unit outside arena"); return -1; }
      in_lo = std::min<uint64_t>(in_lo, u.in_off); in_hi = std::max<uint64_t>(in_hi, u.in_off + u.in_len);
      continue;
    }
    // kind 0 = "no codec": the unit is carried along, no kernel takes it, its result says MSPACK_ERR_ARGS
    const uint64_t below = unit_below(u);
    if (below > u.out_off) { snprintf(errbuf, errcap, "unit's lower region outside arena"); return -1; }
    const uint64_t lo = u.out_off - below, hi = u.out_off + u.out_len + unit_above(u);
    if (u.in_off + u.in_len > in_bytes || hi > out_bytes) { snprintf(errbuf, errcap, "unit outside arena"); return -1; }
    if (i && lo < prev_hi) monotone = false;
    prev_hi = hi;
    in_lo = std::min<uint64_t>(in_lo, u.in_off); in_hi = std::max<uint64_t>(in_hi, u.in_off + u.in_len);
    {
      uint64_t tl, th;
      if (unit_side_table(u, tl, th)) {
        if (th > in_bytes) { snprintf(errbuf, errcap, "unit's table outside arena"); return -1; }
        if (u.kind == MSPACK_HIP_KIND_QUANTUM && (u.out_off & 3u)) { snprintf(errbuf, errcap, "unit %u: a Quantum unit with marks needs out_off %% 4 == 0", idx[i]); return -1; }
        in_lo = std::min(in_lo, tl); in_hi = std::max(in_hi, th);
      }
    }
    out_lo = std::min(out_lo, lo); out_hi = std::max(out_hi, hi);
    in_sum += u.in_len;
  }
  // frame slots: the units that carry a usable frame / block table first -- only their slots hold records and tokens
  for (int pass = 0; pass < 2; pass++) {
    for (size_t i = 0; i < n_sel; i++) {
      mspack_hip_unit &u = local[i];
      if ((pass == 0) != unit_has_ftab(u)) continue;
      u.frame_base = (uint32_t) n_frames; units[idx[i]].frame_base = (uint32_t) n_frames;
      n_frames += unit_frames(u);
    }
    if (pass == 0) n_rec_slots = n_frames;
  }
  in_lo &= ~15ull;                                     // keep the units' alignment
  if (out_lo > out_hi) out_lo = out_hi = 0;            // (checksum units only: nothing is written)
  if (dev_out) out_lo = 0;                             // the caller's device buffer is addressed as is
  const size_t in_span = (size_t)(in_hi - in_lo), out_span = (size_t)(out_hi - out_lo);

  if (!cx.ready) {
    cx.n_compute = env_int("MSPACK_HIP_NCOMPUTE", 4, 1, MSPK_MAX_STREAMS - 2);
    cx.ns = 2 + cx.n_compute;
    cx.max_chunks = env_int("MSPACK_HIP_NCHUNKS", 4, 1, MSPK_MAX_CHUNKS);
    // The runtime maps a process's streams onto a few hardware queues PER PRIORITY LEVEL (four by default), and streams that
    // share a queue run one after the other -- whichever library created them: inside a process that has streams of its own
    // (bench.py: torch's) the copy-in stream landed on a compute stream's queue and every chunk's copy waited for the chunk
    // before it (to the device 8.2 ms instead of 4.7, profiles/round3_hostpath.txt).  So the three roles live on three
    // priority levels, i.e. in three queue pools: compute streams high (a pool of their own: the chunks' launches run side
    // by side), copy-in normal, copy-out low.
    int prio_lo = 0, prio_hi = 0;
    TRY(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));           // (least, greatest): numerically high = low priority
    for (int i = 0; i < cx.ns; i++) {
      const int pr = i == 0 ? (prio_lo + prio_hi) / 2 : (i == 1 ? prio_lo : prio_hi);
      TRY(hipStreamCreateWithPriority(&cx.st[i], hipStreamNonBlocking, pr));
    }
    for (int i = 0; i < MSPK_MAX_CHUNKS; i++) {
      TRY(hipEventCreateWithFlags(&cx.ev_in[i], hipEventDisableTiming));
      TRY(hipEventCreateWithFlags(&cx.ev_done[i], hipEventDisableTiming));
      TRY(hipEventCreateWithFlags(&cx.ev_back[i], hipEventDisableTiming));
    }
    cx.ready = true;
  }
  {
    // chunks: arena-contiguous runs of units; enough of them to overlap the copies with the decode, each
    // big enough to be worth a launch.  Outputs that interleave (not monotone) are copied back unit by unit.
    // (a chunk: >= 8 MiB of input -- a copy of >= 150 us -- and >= 256 units)
    static const size_t chunk_bytes = (size_t) env_int("MSPACK_HIP_CHUNK_BYTES", 8 << 20, 1, 1 << 30);
    static const size_t chunk_units = (size_t) env_int("MSPACK_HIP_CHUNK_UNITS", 256, 1, 1 << 30);
    bool has_qtm = false;
    size_t want = monotone ? std::min<size_t>((size_t) cx.max_chunks, std::max<size_t>(1, in_sum / chunk_bytes)) : 1;
    {
      // (units that decode: checksum units ride along and are no reason to cut)
      size_t n_dec = 0;
      for (size_t i = 0; i < n_sel; i++) {
        if (local[i].kind != MSPACK_HIP_KIND_XORSUM) n_dec++;
        if (local[i].kind == MSPACK_HIP_KIND_QUANTUM) has_qtm = true;
      }
      want = std::min(want, std::max<size_t>(1, n_dec / chunk_units));
    }
    std::vector<Chunk> chunks;
    {
      // shares of the input bytes.  MSPACK_HIP_CHUNK_SHAPE: 0 equal; 1 = 1 : 1 : 2 : 4 ... (to the device: the default -- the small
      // chunks get the launches going while most of the input is still on its way, and the LAST chunk, whose launches end the call,
      // is the large one that fills the chip: headline 4.64-4.84 -> 4.05-4.15 ms, every growing shape within 0.1 ms of it, falling
      // ones and more than four chunks slower; 1024 and 16 384 units: no difference -- profiles/round6_jobs.txt; round 3 had it at
      // 4.55 against 4.74 and kept the equal shares); 2 = a first chunk of half a share (to the host: the default -- the copy-back,
      // the longest leg, starts as soon as the first chunk is through; 1 there: 8.2 against 7.6 ms); 3, 4: a x1.5 ramp, falling shares
      // (sweeps)
      static const int shape_env = getenv("MSPACK_HIP_CHUNK_SHAPE") ? env_int("MSPACK_HIP_CHUNK_SHAPE", 0, 0, 4) : -1;
      const int shape = shape_env >= 0 ? shape_env : (host_out ? 2 : 1);
      uint64_t wsum = 0, w[MSPK_MAX_CHUNKS];
      // (MSPACK_HIP_CHUNK_WEIGHTS="1,1,2,4": the shares spelled out -- sweeps)
      static const char *const w_env = getenv("MSPACK_HIP_CHUNK_WEIGHTS");
      uint64_t w_given[MSPK_MAX_CHUNKS]; size_t n_given = 0;
      if (w_env) for (const char *q = w_env; *q && n_given < MSPK_MAX_CHUNKS; ) { const long v = strtol(q, (char **) &q, 10); w_given[n_given++] = v > 0 ? (uint64_t) v : 1u; while (*q == ',' || *q == ' ') q++; }
      for (size_t k = 0; k < want; k++) {
        static const uint64_t ramp[MSPK_MAX_CHUNKS] = { 4, 6, 9, 13, 20, 30, 45, 67 };          // (3: every chunk half as large again)
        static const uint64_t fall[MSPK_MAX_CHUNKS] = { 8, 6, 4, 3, 2, 2, 1, 1 };               // (4: the last chunks -- whose launches end the call -- small)
        w[k] = (k < n_given) ? w_given[k] : shape == 4 ? fall[k] : shape == 3 ? ramp[k] : shape == 1 ? (k >= 2 ? (uint64_t) 2 << (k - 1) : 2) : (shape == 2 && k == 0 && want >= 3 ? 1 : 2);
        wsum += w[k];
      }
      // (a unit weighs what it reads that the NEXT unit does not start inside: a CHM's intervals are all given "to the end of the
      // file" as input, chmd.c:1146-1149 -- by in_len alone config 3's four chunks held 77, 174, 227 and 546 of its 1024 intervals)
      auto weight = [&](size_t i) -> uint64_t {
        uint64_t wgt = local[i].in_len;
        for (size_t j = i + 1; j < n_sel; j++) {
          if (local[j].kind == MSPACK_HIP_KIND_XORSUM) continue;
          if (local[j].in_off > local[i].in_off && local[j].in_off - local[i].in_off < wgt) wgt = local[j].in_off - local[i].in_off;
          break;
        }
        return wgt;
      };
      uint64_t w_sum = 0;
      for (size_t i = 0; i < n_sel; i++) if (local[i].kind != MSPACK_HIP_KIND_XORSUM) w_sum += weight(i);
      size_t a = 0; uint64_t acc = 0, upto = 0;
      for (size_t i = 0; i < n_sel; i++) {
        if (local[i].kind != MSPACK_HIP_KIND_XORSUM) acc += weight(i);           // (the checksum units ride along)
        const uint64_t goal = (uint64_t)((double) w_sum * (double)(upto + w[chunks.size()]) / (double) wsum);
        if (i + 1 == n_sel || (acc >= goal && chunks.size() + 1 < want)) {
          Chunk c; c.a = a; c.b = i + 1; upto += w[chunks.size()]; chunks.push_back(c); a = i + 1;
        }
      }
    }
    // per chunk: spans, per-kind launch lists (longest compressed unit first: the slowest chain starts first)
    std::vector<uint32_t> order(n_sel);
    size_t op = 0;
    uint64_t ci_prev_hi = out_lo == ~0ull ? 0 : out_lo;
    for (Chunk &c : chunks) {
      c.in_lo = ~0ull; c.in_hi = 0; c.out_lo = ~0ull; c.out_hi = 0;
      c.fm_lo = ~(size_t) 0; c.fm_n = 0; c.has_ftab = false;
      for (size_t i = c.a; i < c.b; i++) {
        const mspack_hip_unit &u = local[i];
        c.in_lo = std::min<uint64_t>(c.in_lo, u.in_off); c.in_hi = std::max<uint64_t>(c.in_hi, u.in_off + u.in_len);
        { uint64_t tl, th; if (unit_side_table(u, tl, th)) { c.in_lo = std::min(c.in_lo, tl); c.in_hi = std::max(c.in_hi, th); } }
        if (unit_has_ftab(u)) {
          c.has_ftab = true;
          c.fm_lo = std::min<size_t>(c.fm_lo, u.frame_base);          // (the chunk's table units' slots are contiguous)
          c.fm_n += unit_frames(u);
        }
        if (u.kind == MSPACK_HIP_KIND_XORSUM) continue;
        c.out_lo = std::min<uint64_t>(c.out_lo, u.out_off - unit_below(u));
        c.out_hi = std::max<uint64_t>(c.out_hi, u.out_off + u.out_len + unit_above(u));
      }
      if (c.out_lo > c.out_hi) c.out_lo = c.out_hi = (ci_prev_hi);         // (a chunk of checksum units only: an empty span)
      ci_prev_hi = c.out_hi;
      if (c.fm_lo == ~(size_t) 0) c.fm_lo = 0;
      c.in_lo &= ~15ull;
      for (unsigned k = 1; k <= MSPACK_HIP_KIND_XORSUM; k++) {
        c.order_off[k] = op;
        for (size_t i = c.a; i < c.b; i++) if (local[i].kind == k) order[op++] = (uint32_t) i;
        c.order_n[k] = op - c.order_off[k];
        std::stable_sort(order.begin() + c.order_off[k], order.begin() + op, [&](uint32_t x, uint32_t y) {
          return local[x].in_len + (local[x].out_len >> 2) > local[y].in_len + (local[y].out_len >> 2); });
      }
    }
    if (pg) {
      std::lock_guard<std::mutex> lk(pg->mu);
      for (size_t ci = 0; ci < chunks.size(); ci++) for (size_t i = chunks[ci].a; i < chunks[ci].b; i++) pg->chunk_of[idx[i]] = (uint32_t) ci;
      pg->planned = true;
      pg->cv.notify_all();
    }
    for (size_t i = 0; i < n_sel; i++) {
      { uint64_t tl, th; if (unit_side_table(local[i], tl, th)) local[i].in_chunk -= (uint32_t)(in_lo >> 2); }      // in_lo is a multiple of 16
      local[i].in_off -= in_lo;
      if (local[i].kind != MSPACK_HIP_KIND_XORSUM) local[i].out_off -= out_lo;
    }

    // ---- buffers (persistent) ----
    TRY(grow(cx.d_in, in_span + 64, false));
    if (!dev_out) TRY(grow(cx.d_out, out_span + 64, false));
    TRY(grow(cx.d_units, n_sel * sizeof(mspack_hip_unit), false));
    TRY(grow(cx.d_order, n_sel * sizeof(uint32_t), false));
    TRY(grow(cx.d_res, n_sel * sizeof(mspack_hip_result), false));
    TRY(grow(cx.d_fm, lzx_scratch(nullptr, n_frames, n_rec_slots).bytes, false));
    // (pinned staging: the results, and room for the few output bytes that lie outside every page-locked range -- below)
    const size_t stage_res = (n_sel * sizeof(mspack_hip_result) + 255u) & ~(size_t) 255u;
    const size_t STAGE_PIECE = 8192u, STAGE_SLOTS = 4u * MSPK_MAX_CHUNKS;
    TRY(grow(cx.h_stage, stage_res + STAGE_PIECE * STAGE_SLOTS, true));
    u8 *const d_in = (u8 *) cx.d_in.p;
    u8 *const d_out = dev_out ? (u8 *) dev_out : (u8 *) cx.d_out.p;
    mspack_hip_unit *const d_units = (mspack_hip_unit *) cx.d_units.p;
    uint32_t *const d_order = (uint32_t *) cx.d_order.p;
    mspack_hip_result *const d_res = (mspack_hip_result *) cx.d_res.p;
    mspack_hip_result *const h_res = (mspack_hip_result *) cx.h_stage.p;
    t1 = tnow();

    // ---- issue: tables, then every chunk's copy on the copy-in stream and its launches on a compute stream ----
    const bool one = chunks.size() == 1;                 // one chunk: everything in order on one stream, no events
    // compute streams in use: all of them when the output stays on the device (the chunks' launches side by side: the
    // last one ends earliest), two when it goes back to the host (the chunks then finish one after the other and the
    // copy-back, the longest leg, starts early) -- measured, profiles/round3_hostpath.txt
    // (A Quantum unit is one long serial chain: a launch of them takes as long as its slowest folder however few there are.
    // Chunks that hold some must not queue behind each other on one compute stream: all streams then, also to the host --
    // with 16 384 checksum units beside 512 folders config 4 was cut into four chunks on two streams: 794 ms instead of 416)
    // (... unless the whole batch is small beside the chip -- config 3's 1024 intervals are 4096 tickets for 4096 waves: its four
    // chunks' launches then fit side by side, and on two streams the second pair only waited: to the host 4.2 -> 3.3 ms,
    // tools/sessions/round6_sessions.md: session AB; the headline batch on four streams: slower, as it was)
    static const int ncomp_host = env_int("MSPACK_HIP_NCOMP_HOST", 0, 0, MSPK_MAX_STREAMS - 2);
    const size_t few = (ncomp_host > 0) ? (size_t) ncomp_host : (n_frames <= 6144u ? (size_t) cx.n_compute : 2u);
    const size_t n_comp = (host_out && !has_qtm) ? std::min<size_t>(few, (size_t) cx.n_compute) : (size_t) cx.n_compute;
    hipStream_t st_in = cx.st[0], st_out = one ? cx.st[0] : cx.st[1];
    TRY(hipMemcpyAsync(d_units, local.data(), n_sel * sizeof(mspack_hip_unit), hipMemcpyHostToDevice, st_in));
    TRY(hipMemcpyAsync(d_order, order.data(), n_sel * sizeof(uint32_t), hipMemcpyHostToDevice, st_in));
    TRY(hipMemsetAsync(cx.d_fm.p, 0, (n_frames + 1) * sizeof(int32_t), st_in));
    TRY(hipMemsetAsync(d_in + in_span, 0, 64, st_in));
    // The copies back are issued by a second thread.  A copy into PAGEABLE memory holds its calling thread and (measured,
    // profiles/round3_hostpath.txt) does not start before every stream of the device has drained, so this thread first
    // page-locks each chunk's part of the caller's buffer (hipHostRegister -- while the main thread is inside the H2D
    // copies and the first launches run), after which chunk c's D2H is a plain DMA that starts the moment chunk c's
    // launches have ended, next to the H2D of later chunks (PCIe is full duplex).  The pages are released before the
    // call returns.  A buffer that cannot be registered (already pinned by its owner, or the runtime refuses) is
    // copied the ordinary way.  Only whole pages INSIDE the bytes this call writes are locked (see PinRange above); what is
    // left over at the two ends of the span (less than a page each; nothing for a page-aligned buffer such as the C
    // drivers') goes through the pinned staging buffer and is copied into place at the end.
    std::atomic<size_t> issued{0};                       // chunks whose ev_done has been recorded
    std::atomic<bool> stop{false};
    hipError_t back_err = hipSuccess;
    double pin_ms = 0.0, unpin_ms = 0.0;
    std::thread back;
    struct Pins {                                          // page-locked ranges of the caller's buffers (this call's own)
      std::vector<PinRange> r; int n = 0;
      bool lock_one(uintptr_t ra, uintptr_t rb) {
        if (rb <= ra) return false;
        // (someone else's registration -- the caller's own hipHostRegister / hipHostMalloc -- is left alone: asking the runtime
        // to register bytes it has registered already fails after it has walked the pages, and is not this library's to undo)
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, (const void *) ra) == hipSuccess) { if (at.type == hipMemoryTypeHost) return false; }
        else (void) hipGetLastError();
        if (hipHostRegister((void *) ra, rb - ra, hipHostRegisterDefault) != hipSuccess) { (void) hipGetLastError(); return false; }
        r.push_back(PinRange{ ra, rb, nullptr }); n = (int) r.size();
        return true;
      }
      // [ra, rb) minus every range of mspack_hip_pin's registry: the library never asks the runtime to register a byte that
      // lies inside a registration it knows about
      bool lock(uintptr_t ra, uintptr_t rb) {
        if (rb <= ra) return false;
        std::vector<PinRange> known;
        { std::lock_guard<std::mutex> lock(g_pin_mu); for (const PinRange &k : g_pins) if (k.rb > ra && k.ra < rb) known.push_back(k); }
        std::sort(known.begin(), known.end(), [](const PinRange &x, const PinRange &y) { return x.ra < y.ra; });
        bool any = false;
        uintptr_t at = ra;
        for (const PinRange &k : known) {
          if (k.ra > at) any = lock_one(at, k.ra) || any;
          if (k.rb > at) at = k.rb;
        }
        if (at < rb) any = lock_one(at, rb) || any;
        return any;
      }
      void release() { for (const PinRange &x : r) if (hipHostUnregister((void *) x.ra) != hipSuccess) (void) hipGetLastError(); r.clear(); n = 0; }
      ~Pins() { if (n) { hipDeviceSynchronize(); release(); } }      // (an error path: nothing may still be writing them)
    } pins, pins_in;
    struct Staged { void *host; size_t off, n; };
    // (written by the thread that issues the copies back; read behind it: chunk ci's pieces are staged[.. staged_upto[ci]), final once
    // back_issued says the chunk's copies are on the stream)
    Staged staged[4u * MSPK_MAX_CHUNKS];
    size_t n_staged = 0, staged_upto[MSPK_MAX_CHUNKS] = { 0 };
    double tr_locked[MSPK_MAX_CHUNKS] = { 0 }, tr_h2d[MSPK_MAX_CHUNKS] = { 0 };      // (trace: ms after the call began)
    std::atomic<size_t> back_issued{0};                    // chunks whose copies back are on st_out, ev_back recorded behind them
    std::atomic<bool> back_ended{false};
    // one span of the output, device -> caller's memory on st_out: cut at the boundaries of every registration this library
    // knows; pieces outside all of them that are small go through the pinned staging buffer (no pageable copy in the way)
    auto copy_out = [&](uintptr_t lo, uintptr_t hi, const u8 *d_src) -> hipError_t {
      std::vector<uintptr_t> cuts;
      pin_cuts(lo, hi, pins.r.data(), pins.n, cuts);
      uintptr_t at = lo;
      for (size_t i = 0; i <= cuts.size(); i++) {
        const uintptr_t to = i < cuts.size() ? cuts[i] : hi;
        if (to <= at) continue;
        bool locked = false;
        for (int k = 0; k < pins.n && !locked; k++) locked = at >= pins.r[k].ra && to <= pins.r[k].rb;
        hipError_t ce;
        if (!locked && pins.n && to - at <= STAGE_PIECE && n_staged < STAGE_SLOTS) {
          const size_t off = stage_res + STAGE_PIECE * n_staged;
          ce = hipMemcpyAsync((char *) cx.h_stage.p + off, d_src + (at - lo), to - at, hipMemcpyDeviceToHost, st_out);
          staged[n_staged++] = Staged{ (void *) at, off, (size_t)(to - at) };
        }
        else ce = hipMemcpyAsync((void *) at, d_src + (at - lo), to - at, hipMemcpyDeviceToHost, st_out);
        if (ce != hipSuccess) return ce;
        at = to;
      }
      return hipSuccess;
    };
    struct Joiner { std::thread &t; std::atomic<bool> &stop; ~Joiner() { if (t.joinable()) { stop.store(true); t.join(); } } } joiner{back, stop};
    // (LZX DELTA units read their reference data out of the caller's output buffer while this call runs: no locking of it then)
    bool refs_in_out = false;
    for (size_t i = 0; i < n_sel && !refs_in_out; i++) refs_in_out = local[i].kind == MSPACK_HIP_KIND_LZX_DELTA && local[i].ref_len != 0u;
    static const bool pin_out_env = env_int("MSPACK_HIP_PIN_OUT", 1, 0, 1) != 0;
    // (a buffer that is page-locked already -- the drivers' arenas out of mspack_hip_stage_alloc, a caller's hipHostMalloc -- needs
    // no lock, and ASKING for one is not free: the runtime walks the pages before it notices: ~3 ms per 64 MB chunk, on the
    // copy-back's critical path.  Pins::lock_one asks the runtime whose memory a range is before it asks for the lock, per range)
    const bool pin_out = pin_out_env && !refs_in_out;
    // The INPUT is not locked here by default (MSPACK_HIP_PIN_IN=1 does it, one range per call): for a caller's warm buffer the
    // runtime's pageable path is as fast as the lock costs (to the host 7.4 -> 8.1 ms on the headline batch); for an arena that was
    // just written -- the C drivers' gather -- it runs at 5-6 GB/s, and those callers lock their arena themselves (mspack_hip_pin).
    static const bool pin_in = env_int("MSPACK_HIP_PIN_IN", 0, 0, 1) != 0;
    bool back_started = false;
    if (host_out && !one) try {
      back = std::thread([&]() {
        hipError_t be = hipSetDevice(dev);
        const uintptr_t base = (uintptr_t) host_out;
        uintptr_t span_a, span_b;                            // the whole pages inside the bytes this call writes
        const bool any = inner_pages((const void *)(base + out_lo), out_span, span_a, span_b);
        for (size_t ci = 0; ci < chunks.size() && be == hipSuccess; ci++) {
          const Chunk &c = chunks[ci];
          // chunk ci's pages: from the first page boundary at or behind its first byte to the first one at or behind its
          // end (the last chunk: the last one inside the span) -- disjoint from its neighbours' ranges
          uintptr_t ra = (base + c.out_lo + MSPK_PAGE - 1u) & ~(MSPK_PAGE - 1u), rb = (base + c.out_hi + MSPK_PAGE - 1u) & ~(MSPK_PAGE - 1u);
          if (ra < span_a) ra = span_a;
          if (rb > span_b || ci + 1 == chunks.size()) rb = span_b;
          auto r0 = tnow();
          if (pin_out && any) pins.lock(ra, rb);
          pin_ms += tms(r0, tnow());
          tr_locked[ci] = tms(t0, tnow());
          while (issued.load(std::memory_order_acquire) <= ci) { if (stop.load(std::memory_order_relaxed)) return; std::this_thread::yield(); }
          be = hipStreamWaitEvent(st_out, cx.ev_done[ci], 0);
          if (be == hipSuccess) be = copy_out(base + c.out_lo, base + c.out_hi, d_out + (c.out_lo - out_lo));
          if (be == hipSuccess && (pg || trace)) be = hipEventRecord(cx.ev_back[ci], st_out);
          if (be == hipSuccess) { staged_upto[ci] = n_staged; back_issued.store(ci + 1, std::memory_order_release); }
        }
        back_err = be;
        back_ended.store(true, std::memory_order_release);
      });
      back_started = true;
    } catch (...) { back_started = false; }      // (no helper thread: the copies back are issued below, in this thread)
    if (pin_in && in_span >= ((size_t) 4 << 20)) {
      // (ONE range, the whole pages inside what the copies below read: the chunks' input ranges may overlap -- frame tables
      // behind the streams)
      uintptr_t ra, rb;
      if (inner_pages((const char *) in + in_lo, in_span, ra, rb)) pins_in.lock(ra, rb);
    }
    // (what the copies so far have brought: ONE interval -- the chunks' input ranges ascend and may overlap: a CHM's intervals all
    // read "to the end of the file", chmd.c:1146-1149, so its first chunk's range is the whole arena and the later chunks' ranges
    // lie inside it; the copies run one after the other on st_in, and a chunk's launches wait for the event behind ITS copy)
    uint64_t cov_lo = 0, cov_hi = 0;
    for (size_t ci = 0; ci < chunks.size(); ci++) {
      const Chunk &c = chunks[ci];
      hipStream_t st = one ? cx.st[0] : cx.st[2 + ci % n_comp];
      {
        uint64_t lo = c.in_lo, hi = c.in_hi;
        if (cov_hi > cov_lo && lo >= cov_lo && lo <= cov_hi) { lo = std::min(hi, cov_hi); cov_hi = std::max(cov_hi, hi); }
        else { cov_lo = lo; cov_hi = hi; }
        TRY(copy_cut(d_in + (lo - in_lo), (const char *) in + lo, (size_t)(hi - lo), hipMemcpyHostToDevice, st_in,
                     pins_in.r.data(), pins_in.n));
      }
      if (host_out)
        for (size_t i = c.a; i < c.b; i++)               // LZX DELTA reference data sits below the unit's output
          if (local[i].ref_len && local[i].kind == MSPACK_HIP_KIND_LZX_DELTA)
            TRY(copy_cut(d_out + local[i].out_off - local[i].ref_len,
                         (const char *) host_out + out_lo + local[i].out_off - local[i].ref_len, local[i].ref_len,
                         hipMemcpyHostToDevice, st_in));
      tr_h2d[ci] = tms(t0, tnow());
      if (!one) { TRY(hipEventRecord(cx.ev_in[ci], st_in)); TRY(hipStreamWaitEvent(st, cx.ev_in[ci], 0)); }
      for (unsigned k = 1; k <= MSPACK_HIP_KIND_XORSUM; k++)
        TRY(launch_kind(k, d_units, d_order + c.order_off[k], c.order_n[k], d_in, d_out, d_res, cx.d_fm.p, n_frames, c.fm_lo, c.fm_n, st,
                        c.has_ftab, (unsigned) ci, n_rec_slots, one));
      TRY(hipMemcpyAsync(h_res + c.a, d_res + c.a, (c.b - c.a) * sizeof(mspack_hip_result), hipMemcpyDeviceToHost, st));
      if (!one) { TRY(hipEventRecord(cx.ev_done[ci], st)); issued.store(ci + 1, std::memory_order_release); }
    }
    t2 = tnow();
    // ---- copy-back, chunk by chunk on the copy-out stream (each copy waits for its own chunk's launches only) ----
    if (host_out && one) {
      const Chunk &c = chunks[0];
      if (monotone)
        TRY(copy_out((uintptr_t) host_out + c.out_lo, (uintptr_t) host_out + c.out_hi, d_out + (c.out_lo - out_lo)));
      else
        for (size_t i = c.a; i < c.b; i++) {
          const size_t nb = (size_t) local[i].out_len + ((local[i].flags & (MSPACK_HIP_UF_MSZIP_LOG | MSPACK_HIP_UF_LZX_LOG | MSPACK_HIP_UF_QTM_MARKS)) ? (size_t) unit_above(local[i]) : 0u);   // (a unit's log lies behind its slack)
          const uintptr_t lo = (uintptr_t) host_out + out_lo + local[i].out_off;
          TRY(copy_out(lo, lo + nb, d_out + local[i].out_off));
        }
    }
    if (host_out && !one && !back_started) {
      // (the helper thread could not be created: plain copies, chunk by chunk, behind each chunk's launches)
      for (size_t ci = 0; ci < chunks.size(); ci++) {
        const Chunk &c = chunks[ci];
        TRY(hipStreamWaitEvent(st_out, cx.ev_done[ci], 0));
        TRY(copy_out((uintptr_t) host_out + c.out_lo, (uintptr_t) host_out + c.out_hi, d_out + (c.out_lo - out_lo)));
      }
    }
    auto hand_over = [&](size_t a, size_t b) {             // the results of local units [a, b) into the caller's array
      for (size_t i = a; i < b; i++) {
        results[idx[i]] = h_res[i];
        if (local[i].kind == 0) { memset(&results[idx[i]], 0, sizeof(mspack_hip_result)); results[idx[i]].err = ERR_ARGS; }
      }
    };
    size_t handed = 0, staged_done = 0;                    // units / staged pieces already in the caller's memory
    if ((pg || trace) && back_started) {
      // a job: chunk by chunk as the copies back end -- the caller (mspack_hip_job_wait_unit) takes a chunk's bytes while the later
      // chunks are still being decoded and copied
      for (size_t ci = 0; ci < chunks.size(); ci++) {
        while (back_issued.load(std::memory_order_acquire) <= ci && !back_ended.load(std::memory_order_acquire)) std::this_thread::yield();
        if (back_issued.load(std::memory_order_acquire) <= ci) break;          // (the thread gave up: its error is reported below)
        double tr_done = 0.0;
        if (trace) { TRY(hipEventSynchronize(cx.ev_done[ci])); tr_done = tms(t0, tnow()); }
        TRY(hipEventSynchronize(cx.ev_back[ci]));            // chunk ci's launches, its results' copy and its bytes' copies are through
        for (; staged_done < staged_upto[ci]; staged_done++) memcpy(staged[staged_done].host, (const char *) cx.h_stage.p + staged[staged_done].off, staged[staged_done].n);
        hand_over(chunks[ci].a, chunks[ci].b);
        handed = chunks[ci].b;
        if (trace) fprintf(stderr, "mspack_hip[dev %d]: chunk %zu of %zu (%zu units, %.1f MB out): input copied %.2f, output pages seen to %.2f, launches through %.2f, "
                           "handed over %.2f ms after the call began\n", dev, ci, chunks.size(),
                           chunks[ci].b - chunks[ci].a, (chunks[ci].out_hi - chunks[ci].out_lo) / 1e6, tr_h2d[ci], tr_locked[ci], tr_done, tms(t0, tnow()));
        if (pg) {
          { std::lock_guard<std::mutex> lk(pg->mu); pg->done = ci + 1; }
          pg->cv.notify_all();
        }
      }
    }
    if (back.joinable()) {
      back.join();
      if (back_err != hipSuccess) TRY(back_err);
    }
    for (int i = 0; i < cx.ns; i++) TRY(hipStreamSynchronize(cx.st[i]));
    for (; staged_done < n_staged; staged_done++) memcpy(staged[staged_done].host, (const char *) cx.h_stage.p + staged[staged_done].off, staged[staged_done].n);
    { auto r0 = tnow(); pins.release(); pins_in.release(); unpin_ms = tms(r0, tnow()); }
    hand_over(handed, n_sel);
    t3 = tnow();
    host_path_account(tms(t0, t1), tms(t1, t2), tms(t2, t3));
    if (trace)
      fprintf(stderr, "mspack_hip[dev %d]: %zu units in %zu chunks (%d streams): plan+alloc %.2f ms, issue (H2D %.1f MB) %.2f ms, "
              "drain (D2H %.1f MB) %.2f ms (page-locking %.2f ms beside the issue, release %.2f ms)\n", dev, n_sel, chunks.size(), cx.ns,
              tms(t0, t1), in_span / 1e6, tms(t1, t2), host_out ? out_span / 1e6 : 0.0, tms(t2, t3), pin_ms, unpin_ms);
  }
done:
  if (rc) for (int i = 0; i < cx.ns; i++) hipStreamSynchronize(cx.st[i]);
  return rc;
#undef TRY
}

static int current_device() { int d = 0; if (hipGetDevice(&d) != hipSuccess) d = 0; return d; }

static std::mutex g_stats_mu;
static double g_stats_ms[4] = { 0, 0, 0, 0 };
static void host_path_account(double plan_ms, double issue_ms, double drain_ms) {
  std::lock_guard<std::mutex> lock(g_stats_mu);
  g_stats_ms[0] += plan_ms; g_stats_ms[1] += issue_ms; g_stats_ms[2] += drain_ms; g_stats_ms[3] += 1.0;
}

extern "C" {

int mspack_hip_decode_batch(mspack_hip_unit *units, size_t n_units, const void *in, size_t in_bytes,
                            void *out, size_t out_bytes, mspack_hip_result *results)
{
  return pipeline_on_current_device(current_device(), units, nullptr, n_units, in, in_bytes, out, nullptr, out_bytes,
                                    results, g_err, sizeof(g_err));
}

int mspack_hip_decode_batch_to_device(mspack_hip_unit *units, size_t n_units, const void *in, size_t in_bytes,
                                      void *d_out, size_t out_bytes, mspack_hip_result *results)
{
  return pipeline_on_current_device(current_device(), units, nullptr, n_units, in, in_bytes, nullptr, d_out, out_bytes,
                                    results, g_err, sizeof(g_err));
}

// ---- jobs: the same batch, handed over chunk by chunk while it runs (include/mspack_hip.h) ----
struct mspack_hip_job {
  std::thread th;
  JobProgress pg;
  char err[256];
};

mspack_hip_job *mspack_hip_decode_batch_begin(mspack_hip_unit *units, size_t n_units, const void *in, size_t in_bytes,
                                              void *out, size_t out_bytes, mspack_hip_result *results)
{
  static const bool off = env_int("MSPACK_HIP_JOBS", 1, 0, 1) == 0;          // (A/B runs: every caller takes its synchronous way)
  if (off || !units || !results || !out) return nullptr;
  mspack_hip_job *job = nullptr;
  try {
    job = new mspack_hip_job();
    job->err[0] = 0;
    job->pg.chunk_of.assign(n_units, 0u);
    const int dev = current_device();
    job->th = std::thread([=]() {
      int rc;
      if (hipSetDevice(dev) != hipSuccess) { (void) hipGetLastError(); snprintf(job->err, sizeof(job->err), "hipSetDevice(%d) failed", dev); rc = -1; }
      else rc = pipeline_on_current_device(dev, units, nullptr, n_units, in, in_bytes, out, nullptr, out_bytes, results,
                                           job->err, sizeof(job->err), false, &job->pg);
      { std::lock_guard<std::mutex> lk(job->pg.mu); job->pg.rc = rc; job->pg.finished = true; }
      job->pg.cv.notify_all();
    });
  } catch (...) { delete job; return nullptr; }            // (no thread, no memory: the caller takes the synchronous call)
  return job;
}

int mspack_hip_job_wait_unit(mspack_hip_job *job, size_t i)
{
  if (!job) return -1;
  JobProgress &pg = job->pg;
  std::unique_lock<std::mutex> lk(pg.mu);
  if (i >= pg.chunk_of.size()) return -1;
  pg.cv.wait(lk, [&]() { return pg.finished || (pg.planned && pg.done > pg.chunk_of[i]); });
  if (pg.planned && pg.done > pg.chunk_of[i]) return 0;    // (its chunk came through, whatever became of the later ones)
  if (pg.rc) { snprintf(g_err, sizeof(g_err), "%s", job->err); return pg.rc; }
  return 0;                                                // finished without an error: everything is there
}

int mspack_hip_job_end(mspack_hip_job *job)
{
  if (!job) return -1;
  if (job->th.joinable()) job->th.join();
  const int rc = job->pg.rc;
  if (rc) snprintf(g_err, sizeof(g_err), "%s", job->err);
  delete job;
  return rc;
}

int mspack_hip_decode_batch_multi(mspack_hip_unit *units, size_t n_units, const void *in,
                                  size_t in_bytes, void *out, size_t out_bytes,
                                  mspack_hip_result *results, int n_devices)
{
  int have = mspack_hip_device_count();
  if (n_devices > have) n_devices = have;
  if (n_devices > MSPK_MAX_DEV) n_devices = MSPK_MAX_DEV;
  const bool force_shards = getenv("MSPACK_HIP_FORCE_SHARDS") != nullptr;   // tests: exercise the sharded path on one GPU
  int n_shards = n_devices;
  if (force_shards) n_shards = env_int("MSPACK_HIP_FORCE_SHARDS", 2, 1, MSPK_MAX_DEV);
  if (n_shards <= 1 || n_units < 2) return mspack_hip_decode_batch(units, n_units, in, in_bytes, out, out_bytes, results);
  if (n_devices < 1) { snprintf(g_err, sizeof(g_err), "no HIP device"); return -1; }
  // static sharding, no inter-device traffic: units in arena order are cut into n_shards CONTIGUOUS ranges of
  // about equal compressed size, so that every device stages one contiguous span of each arena
  std::vector<uint32_t> idx(n_units);
  for (size_t i = 0; i < n_units; i++) idx[i] = (uint32_t) i;
  std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return units[a].in_off < units[b].in_off; });
  uint64_t total = 0;
  for (size_t i = 0; i < n_units; i++) total += (uint64_t) units[i].in_len + (units[i].out_len >> 2) + 256u;
  std::vector<std::vector<uint32_t>> shard(n_shards);
  {
    uint64_t acc = 0; int s = 0;
    for (size_t i = 0; i < n_units; i++) {
      shard[s].push_back(idx[i]);
      acc += (uint64_t) units[idx[i]].in_len + (units[idx[i]].out_len >> 2) + 256u;
      if (s + 1 < n_shards && acc * n_shards >= total * (uint64_t)(s + 1)) s++;
    }
  }
  // every shard copies its whole output span back with one copy -- valid only if the spans do not interleave, i.e. if
  // the outputs ascend with the inputs over the WHOLE batch; otherwise the shards copy back unit by unit
  bool ascending = true;
  {
    uint64_t prev_hi = 0;
    for (size_t i = 0; i < n_units && ascending; i++) {
      const mspack_hip_unit &u = units[idx[i]];
      if (u.kind == MSPACK_HIP_KIND_XORSUM) continue;       // (no output)
      const uint64_t lo = u.out_off - std::min<uint64_t>(u.out_off, unit_below(u)), hi = u.out_off + u.out_len + unit_above(u);
      if (lo < prev_hi) ascending = false;
      prev_hi = std::max(prev_hi, hi);
    }
  }
  std::vector<int> rcs(n_shards, 0);
  std::vector<std::array<char, 256>> errs(n_shards);
  std::vector<std::thread> th;
  auto run_shard = [&](int sh) {
    const int dv = sh % n_devices;
    errs[sh][0] = 0;
    hipError_t e = hipSetDevice(dv);
    if (e != hipSuccess) { snprintf(errs[sh].data(), 256, "hipSetDevice(%d): %s", dv, hipGetErrorString(e)); rcs[sh] = -(int) e; return; }
    rcs[sh] = pipeline_on_current_device(dv, units, shard[sh].data(), shard[sh].size(), in, in_bytes, out, nullptr,
                                         out_bytes, results, errs[sh].data(), 256, !ascending);
  };
  th.reserve((size_t) n_shards);
  for (int sh = 0; sh < n_shards; sh++) {
    // (a thread that cannot be created must not throw through the C ABI: that shard runs here, after the others were started)
    try { th.emplace_back(run_shard, sh); } catch (...) { run_shard(sh); }
  }
  for (auto &t : th) t.join();
  for (int sh = 0; sh < n_shards; sh++)
    if (rcs[sh]) { snprintf(g_err, sizeof(g_err), "shard %d: %s", sh, errs[sh].data()); return rcs[sh]; }
  return 0;
}

// page-locked caller arenas (mspack_hip_pin / _unpin): the whole pages INSIDE [p, p + bytes) -- never a byte that is not the
// caller's (PinRange above).  A buffer that starts and ends on page boundaries is locked completely (the C drivers' arenas do:
// mspack_arena_alloc); of any other one the bytes before the first and behind the last boundary are copied the pageable way.
int mspack_hip_pin(const void *p, size_t bytes)
{
  uintptr_t ra, rb;
  if (!p || !bytes || !inner_pages(p, bytes, ra, rb)) return -1;
  const hipError_t e = hipHostRegister((void *) ra, rb - ra, hipHostRegisterPortable);       // (every device of the process: _multi)
  if (e != hipSuccess) { (void) hipGetLastError(); return (int) e; }
  std::lock_guard<std::mutex> lock(g_pin_mu);
  g_pins.push_back(PinRange{ ra, rb, p });
  return 0;
}
void mspack_hip_unpin(const void *p)
{
  uintptr_t base = 0;
  {
    std::lock_guard<std::mutex> lock(g_pin_mu);
    for (size_t i = 0; i < g_pins.size(); i++)
      if (g_pins[i].user == p) { base = g_pins[i].ra; g_pins.erase(g_pins.begin() + (long) i); break; }
  }
  if (base && hipHostUnregister((void *) base) != hipSuccess) (void) hipGetLastError();
}

// ---- the library's own page-locked staging memory ----------------------------------------------------------------------
// Locking a caller's arena per call (mspack_hip_pin) costs about as much as the copy it speeds up (hipHostRegister + Unregister of
// the 51 MB of config 2's payloads: ~8 of the 39 ms an extract-everything run took), and an output arena that has to be locked
// chunk by chunk while the copies back wait for it costs more.  So the drivers' big arenas can come from here: blocks of
// hipHostMalloc'ed memory that are KEPT when they are handed back and reused by the next batch -- the cost of locking is paid
// once per process, not once per call.  Bounded: MSPACK_HIP_PINNED_MB (default 1024) MiB in all; a request that does not fit
// returns NULL and the caller takes the sys->alloc + mspack_hip_pin way.  mspack_hip_release() gives the idle blocks back.
struct StageBlock { void *p; size_t cap; bool busy; unsigned long long used; };      // used: when it was last handed out or back (g_stage_clock)
static unsigned long long g_stage_clock = 0;
static std::mutex g_stage_mu;
static std::vector<StageBlock> g_stage;
static size_t g_stage_total = 0;
void *mspack_hip_stage_alloc(size_t bytes)
{
  static const size_t limit = (size_t) env_int("MSPACK_HIP_PINNED_MB", 1024, 0, 1 << 20) << 20;
  if (!bytes || bytes > limit) return nullptr;
  std::lock_guard<std::mutex> lock(g_stage_mu);
  StageBlock *best = nullptr;
  for (StageBlock &b : g_stage)
    if (!b.busy && b.cap >= bytes && b.cap <= bytes + bytes / 2 + (1u << 20) && (!best || b.cap < best->cap)) best = &b;
  if (best) { best->busy = true; best->used = ++g_stage_clock; return best->p; }
  // room?  idle blocks that fit nothing are given back first
  if (g_stage_total + bytes > limit) {
    for (size_t i = 0; i < g_stage.size() && g_stage_total + bytes > limit; )
      if (!g_stage[i].busy) { if (hipHostFree(g_stage[i].p) != hipSuccess) (void) hipGetLastError(); g_stage_total -= g_stage[i].cap; g_stage.erase(g_stage.begin() + (long) i); }
      else i++;
    if (g_stage_total + bytes > limit) return nullptr;
  }
  void *p = nullptr;
  const size_t cap = (bytes + ((size_t) 2 << 20) - 1) & ~(((size_t) 2 << 20) - 1);
  if (hipHostMalloc(&p, cap, hipHostMallocPortable) != hipSuccess || !p) { (void) hipGetLastError(); return nullptr; }
  g_stage.push_back(StageBlock{ p, cap, true, ++g_stage_clock });
  g_stage_total += cap;
  return p;
}
void mspack_hip_stage_free(void *p)
{
  if (!p) return;
  // what stays page-locked while nobody uses it is bounded too (a process that once opened a large cabinet kept
  // hundreds of MiB locked for good): MSPACK_HIP_PINNED_IDLE_MB, default 768 -- the arenas of the largest single cabinet among
  // BASELINE's configs (config 4: 190 MB in + 528 MB out) come back at once for the next one (with 512 its output arena was locked anew
  // on every open-and-extract: 175 ms of a 620 ms run, tools/sessions/round6_sessions.md: session V); beyond that the idle blocks that have been idle
  // LONGEST go back to the system (the largest first, as it was, gave config 4's output arena back whenever a process had opened
  // other cabinets before: the blocks just handed back are the ones the next cabinet of that size will ask for)
  static const size_t idle_limit = (size_t) env_int("MSPACK_HIP_PINNED_IDLE_MB", 768, 0, 1 << 20) << 20;
  std::lock_guard<std::mutex> lock(g_stage_mu);
  for (StageBlock &b : g_stage) if (b.p == p) { b.busy = false; b.used = ++g_stage_clock; break; }
  for (;;) {
    size_t idle = 0, big = (size_t) -1;
    for (size_t i = 0; i < g_stage.size(); i++)
      if (!g_stage[i].busy) { idle += g_stage[i].cap; if (big == (size_t) -1 || g_stage[i].used < g_stage[big].used) big = i; }
    if (idle <= idle_limit || big == (size_t) -1) break;
    if (hipHostFree(g_stage[big].p) != hipSuccess) (void) hipGetLastError();
    g_stage_total -= g_stage[big].cap;
    g_stage.erase(g_stage.begin() + (long) big);
  }
}
static void stage_release_idle()
{
  std::lock_guard<std::mutex> lock(g_stage_mu);
  for (size_t i = 0; i < g_stage.size(); )
    if (!g_stage[i].busy) { if (hipHostFree(g_stage[i].p) != hipSuccess) (void) hipGetLastError(); g_stage_total -= g_stage[i].cap; g_stage.erase(g_stage.begin() + (long) i); }
    else i++;
}

void mspack_hip_host_path_stats(double *ms4, int reset)
{
  std::lock_guard<std::mutex> lock(g_stats_mu);
  if (ms4) for (int i = 0; i < 4; i++) ms4[i] = g_stats_ms[i];
  if (reset) for (int i = 0; i < 4; i++) g_stats_ms[i] = 0.0;
}

// free every persistent context (device arenas, pinned staging, streams) of this process
void mspack_hip_release(void)
{
  int keep = current_device();
  stage_release_idle();
  for (int d = 0; d < MSPK_MAX_DEV; d++) {
    DevCtx &cx = g_ctx[d];
    std::lock_guard<std::mutex> lock(cx.mu);
    if (!cx.ready && !cx.d_in.p && !cx.h_stage.p) continue;
    if (hipSetDevice(d) != hipSuccess) continue;
    hipDeviceSynchronize();
    for (DevBuf *b : { &cx.d_in, &cx.d_out, &cx.d_units, &cx.d_order, &cx.d_res, &cx.d_fm }) { if (b->p) hipFree(b->p); b->p = nullptr; b->cap = 0; }
    if (cx.h_stage.p) { hipHostFree(cx.h_stage.p); cx.h_stage.p = nullptr; cx.h_stage.cap = 0; }
    if (cx.ready) {
      for (int i = 0; i < cx.ns; i++) hipStreamDestroy(cx.st[i]);
      for (int i = 0; i < MSPK_MAX_CHUNKS; i++) { hipEventDestroy(cx.ev_in[i]); hipEventDestroy(cx.ev_done[i]); hipEventDestroy(cx.ev_back[i]); }
    }
    cx.ready = false;
  }
  hipSetDevice(keep);
}

} // extern "C"
