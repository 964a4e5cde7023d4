// wave_common.hpp -- device-side building blocks shared by the LZX / MSZIP / Quantum kernels.
//
// Execution model: ONE WAVEFRONT (64 lanes, one 64-thread workgroup) decodes ONE unit.  Control
// flow, the bit buffer, positions and match parameters are wave-uniform and live in SGPRs (we force
// that with readfirstlane / readlane); the 64 lanes are used for
//   * the compressed-input window: each lane keeps one dword of the current 256-byte chunk in a
//     VGPR, the bit buffer is refilled with v_readlane (no memory latency on the symbol chain),
//     the following chunk is prefetched one chunk ahead;
//   * building canonical-Huffman decode tables in LDS (histogram, ballot-ranked counting sort,
//     one table entry per lane);
//   * LZ77 match copies / literal runs as coalesced byte stores.
// gfx950 only; no portability shims.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../../include/mspack_hip.h"

typedef uint8_t  u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;

// MSPACK_ERR_* values (mspack.h:485-507)
#define ERR_OK        0
#define ERR_ARGS      1
#define ERR_READ      3
#define ERR_DATAFORMAT 8
#define ERR_DECRUNCH 11

#define WAVE 64

// Accesses to the arenas and the work scratch on the hot paths go through gld / gst.  Pointers that live in structs or
// reach a real (non-inlined) device function have lost their address space, so these are FLAT instructions.  Casting them
// to address space 1 (global_load / global_store) was measured on the headline launch (round 3, same box, same session):
// 3.38 ms flat vs 3.49 ms global, 6.43 vs 6.77 ms at 8192 units -- the explicit form is SLOWER here, so it is off.
// streaming stores of the parse waves (literals, match records): analysis builds can mark them non-temporal
#if defined(MSPACK_NT_STORES) && !defined(MSPACK_WAVE_EMU)
template <typename T> __device__ __forceinline__ void gst_stream(T *p, T v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void gst_stream(uint2 *p, uint2 v) { __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y); }
#else
#define gst_stream gst
#endif
#if !defined(MSPACK_GLOBAL_ACCESS) || defined(MSPACK_WAVE_EMU)
template <typename T> __device__ __forceinline__ T gld(const T *p) { return *p; }
template <typename T> __device__ __forceinline__ void gst(T *p, T v) { *p = v; }
#else
template <typename T> __device__ __forceinline__ T gld(const T *p) { return *(const __attribute__((address_space(1))) T *) p; }
template <typename T> __device__ __forceinline__ void gst(T *p, T v) { *(__attribute__((address_space(1))) T *) p = v; }
typedef unsigned int gv2u_ __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint2 gld(const uint2 *p) { const gv2u_ t = *(const __attribute__((address_space(1))) gv2u_ *) p; return make_uint2(t.x, t.y); }
__device__ __forceinline__ void gst(uint2 *p, uint2 v) { gv2u_ t; t.x = v.x; t.y = v.y; *(__attribute__((address_space(1))) gv2u_ *) p = t; }
#endif
// write-through forms (`sc1`: the bytes go to memory at once and the line is dropped from the XCD's L2) for payload that only
// another workgroup reads: an agent-scope release (buffer_wbl2) writes back EVERY dirty line of the XCD's L2, so what a wave leaves
// dirty there is written out -- partially filled -- by whichever of the XCD's 512 waves publishes next (MI355X guide: stores of each
// flavour; profiles/round6_sc1_stores.txt).  MSPACK_SC1_RECORDS: the parse waves' match records; MSPACK_SC1_ROWS: their literal rows.
#if defined(MSPACK_SC1_RECORDS) && !defined(MSPACK_WAVE_EMU)
__device__ __forceinline__ void gst_record(uint2 *p, uint2 v) {
  __hip_atomic_store((unsigned long long *) p, ((unsigned long long) v.y << 32) | v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#else
__device__ __forceinline__ void gst_record(uint2 *p, uint2 v) { gst_stream(p, v); }
#endif
#if defined(MSPACK_SC1_ROWS) && !defined(MSPACK_WAVE_EMU)
typedef unsigned int gv4u_ __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void gst_row(uint4 *p, uint4 v) {
  gv4u_ t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
  asm volatile("flat_store_dwordx4 %0, %1 sc1" :: "v"(p), "v"(t) : "memory");
}
#else
#define gst_row(p_, v_) gst((p_), (v_))
#endif

__device__ __forceinline__ u32 rfl(u32 v) { return (u32) __builtin_amdgcn_readfirstlane((int) v); }
__device__ __forceinline__ u32 rdl(u32 v, u32 l) { return (u32) __builtin_amdgcn_readlane((int) v, (int) l); }
// write `val` into lane `l` of a per-lane register (this clang has no writelane builtin; the
// compare+select is two VALU ops and keeps everything visible to the scheduler)
#define wrl(old, val, l) ((threadIdx.x == (u32)(l)) ? (u32)(val) : (u32)(old))
__device__ __forceinline__ u64 ballot(bool p) { return __ballot(p); }
// is this lane's bit set in a wave-uniform mask?  (the mask goes straight into exec / vcc: no per-lane shift + test)
__device__ __forceinline__ bool lane_in(u64 uniform_mask) { return __builtin_amdgcn_inverse_ballot_w64(uniform_mask); }
__device__ __forceinline__ u32 lanemask_lt_popc(u64 m, u32 lane) {
  return __popcll(m & ((1ull << lane) - 1ull));
}

// ---------------------------------------------------------------------------------------------------
// Compressed-input window.  Byte offsets are relative to the unit's first compressed byte.  Reads
// beyond in_len return zero (the reference fabricates two zero bytes at EOF, readbits.h:196-208;
// whether a read beyond those is an ERR_READ is decided by the callers' position checks).
// ---------------------------------------------------------------------------------------------------
struct InWindow {
  const u8 *unit;     // arena + in_off (uniform); the arena has >= 8 bytes of readable slack
  u32 in_len;
  u32 eofs;           // zero bytes the reference fabricates at EOF before ERR_READ: 2, or 0 for
                      // a hard end (MSPACK_HIP_UF_HARD_EOF: the feeder's read failed, readbits.h:194)
  u32 origin;         // byte offset of dword index 0
  u32 wi;             // next dword index (relative to origin) to hand out
  u32 cur, nxt;       // per-lane: dword `lane` of the current / next 256-byte chunk

  __device__ __forceinline__ u32 load_chunk(u32 c, u32 lane) const {
    u32 o = origin + c * 256u + lane * 4u;
    u32 v = 0;
    if (o < in_len) {
      const u8 *p = unit + o;
      u64 a = (u64) p;
      u32 sh = (u32)(a & 3u);
      const u32 *q = (const u32 *)(a & ~3ull);
      u32 lo = gld(q);
      u32 hi = gld(q + 1);
      v = __builtin_amdgcn_alignbyte(hi, lo, sh);
      u32 rem = in_len - o;
      if (rem < 4u) v &= (1u << (8u * rem)) - 1u;
    }
    return v;
  }
  __device__ __forceinline__ void seek(u32 byte_pos, u32 lane) {
    origin = byte_pos; wi = 0;
    cur = load_chunk(0, lane);
    nxt = load_chunk(1, lane);
  }
  __device__ __forceinline__ u32 next_dword(u32 lane) {
    u32 d = rdl(cur, wi & 63u);
    wi++;
    if ((wi & 63u) == 0u) { cur = nxt; nxt = load_chunk((wi >> 6) + 1u, lane); }
    return d;
  }
  // single byte at absolute unit offset (zero beyond in_len); per-lane
  __device__ __forceinline__ u32 byte_at(u32 pos) const { return pos < in_len ? gld(unit + pos) : 0u; }
};

// ---------------------------------------------------------------------------------------------------
// Canonical Huffman tables in LDS.
//   tab[1<<P]  : direct lookup on the next P bits (MSB-first codes): sym | len<<10, 0 = "long code"
//   sorted[]   : symbols ordered by (length, symbol) -- the canonical order (readhuff.h:97-117)
//   limv / fov : per-lane registers; lane l (1..16) holds the exclusive upper bound of left-aligned
//                16-bit codes of length <= l, and first_code(l) | offs(l)<<16.  Long codes are
//                resolved by ONE wave-wide compare + ballot (which lane's limit is the first one
//                above the 16-bit peek) instead of a bit-by-bit tree walk (readhuff.h:58-65).
// Acceptance follows make_decode_table exactly: lengths <= ref_tablebits that fill the code space
// make longer codes unreachable and are accepted (readhuff.h:121-122); otherwise the code must be
// exactly complete (readhuff.h:108,147,175).
// ---------------------------------------------------------------------------------------------------
struct HuffRegs { u32 limv; u32 fov; };

// returns 0 accepted, 1 rejected, 2 accepted-but-empty (no symbol has a length)
// Table entries are symbol | length << SH: 10 bits of symbol in u16 entries everywhere, except alphabets
// above 1023 symbols (LZX DELTA main tree: 12 bits of symbol in u32 entries).
template <int P, int SH = 10, typename TabT = u16>
__device__ __forceinline__ int huff_build(const u8 *lens, int nsyms, int ref_tablebits, TabT *tab, u16 *sorted,
                          u32 *cnt_scratch /* >= 20 u32 in LDS */, HuffRegs &hr, u32 lane, bool lsb, u32 *n_sorted = nullptr)
{
  // 1. histogram of code lengths
  if (lane < 20u) cnt_scratch[lane] = 0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  for (int s = (int) lane; s < nsyms; s += WAVE) {
    u32 l = lens[s];
    if (l >= 1u && l <= 16u) atomicAdd(&cnt_scratch[l], 1u);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  u32 mycnt = (lane >= 1u && lane <= 16u) ? cnt_scratch[lane] : 0u;

  // 2. Kraft sums, canonical first codes / offsets / limits (wave-uniform scalar work)
  u32 kshort = 0, kall = 0;
  for (int l = 1; l <= 16; l++) {
    u32 c = rdl(mycnt, (u32) l);
    kall += c << (16 - l);
    if (l <= ref_tablebits) kshort = kall;
  }
  if (kall == 0) return 2;
  int maxl = 16;
  if (kshort > 65536u) return 1;
  if (kshort == 65536u) maxl = ref_tablebits;
  else if (kall != 65536u) return 1;

  u32 code = 0, n = 0, limv = 0, fov = 0, curv = 0;
  for (int l = 1; l <= 16; l++) {
    u32 c = (l <= maxl) ? rdl(mycnt, (u32) l) : 0u;
    fov  = wrl(fov, code | (n << 16), (u32) l);
    curv = wrl(curv, n, (u32) l);
    code += c; n += c;
    limv = wrl(limv, (l <= maxl) ? (code << (16 - l)) : 0x10000u, (u32) l);   // beyond maxl: never below the peek
    code <<= 1;
  }
  hr.limv = limv; hr.fov = fov;
  if (n_sorted) *n_sorted = n;                           // symbols that have a code (the sorted list's length)

  // 3. counting sort by (length, symbol): ballot-ranked, 64 symbols per step
  for (int base = 0; base < nsyms; base += WAVE) {
    int s = base + (int) lane;
    u32 l = (s < nsyms) ? lens[s] : 0u;
    if (l > (u32) maxl) l = 0;
    u64 todo = ballot(l != 0u);
    while (todo) {
      u32 leader = (u32) __ffsll((long long) todo) - 1u;
      u32 L = rdl(l, leader);
      u64 m = ballot(l == L);
      u32 c = rdl(curv, L);
      if (l == L) sorted[c + lanemask_lt_popc(m, lane)] = (u16) s;
      curv = wrl(curv, c + (u32) __popcll(m), L);
      todo &= ~m;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  // 4. direct table, one entry per lane per step
  u32 lim[P + 1];
#pragma unroll
  for (int l = 1; l <= P; l++) lim[l] = rdl(limv, (u32) l);
  for (u32 e = lane; e < (1u << P); e += WAVE) {
    u32 peek16 = e << (16 - P);
    u32 len = 1;
#pragma unroll
    for (int l = 1; l <= P; l++) len += (peek16 >= lim[l]) ? 1u : 0u;
    // first/offs for this lane's length: fetch from the owning lane (all lanes active here)
    u32 lq = (len <= (u32) P) ? len : 0u;
    u32 fo = (u32) __builtin_amdgcn_ds_bpermute((int)(lq << 2), (int) fov);
    u32 idx = (fo >> 16) + ((peek16 >> (16 - lq)) - (fo & 0xFFFFu));
    u32 ent = 0;
    if (lq) ent = (u32) sorted[idx] | (lq << SH);
    u32 slot = e;
    if (lsb) slot = __brev(e) >> (32 - P);     // LSB-first streams index by the reversed code
    tab[slot] = (TabT) ent;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  return 0;
}

// Resolve a code longer than the direct table: one compare per lane + ballot.
// peek16 = next 16 bits, MSB-first, left-aligned in 16 bits.  Returns sym | len<<10 (0 if none).
template <int SH = 10>
__device__ __forceinline__ u32 huff_long(const HuffRegs &hr, const u16 *sorted, u32 peek16, u32 lane)
{
  u64 m = ballot(lane >= 1u && lane <= 16u && peek16 < hr.limv);
  if (m == 0) return 0;
  u32 len = (u32) __ffsll((long long) m) - 1u;
  u32 fo = rdl(hr.fov, len);
  u32 idx = (fo >> 16) + ((peek16 >> (16 - len)) - (fo & 0xFFFFu));
  return rfl((u32) sorted[idx]) | (len << SH);
}

// inclusive prefix sum over the 64 lanes with DPP row shifts / row broadcasts (no LDS traffic)
__device__ __forceinline__ u32 wave_incl_scan(u32 x)
{
  u32 v = x;
  v += (u32) __builtin_amdgcn_update_dpp(0, (int) v, 0x111, 0xf, 0xf, false);   // row_shr:1
  v += (u32) __builtin_amdgcn_update_dpp(0, (int) v, 0x112, 0xf, 0xf, false);   // row_shr:2
  v += (u32) __builtin_amdgcn_update_dpp(0, (int) v, 0x114, 0xf, 0xf, false);   // row_shr:4
  v += (u32) __builtin_amdgcn_update_dpp(0, (int) v, 0x118, 0xf, 0xf, false);   // row_shr:8
  v += (u32) __builtin_amdgcn_update_dpp(0, (int) v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1,3
  v += (u32) __builtin_amdgcn_update_dpp(0, (int) v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2,3
  return v;
}

// inclusive prefix maximum over the 64 lanes (values are unsigned; 0 is the identity)
__device__ __forceinline__ u32 wave_incl_max(u32 x)
{
  u32 v = x, t;
  t = (u32) __builtin_amdgcn_update_dpp(0, (int) v, 0x111, 0xf, 0xf, false); v = t > v ? t : v;
  t = (u32) __builtin_amdgcn_update_dpp(0, (int) v, 0x112, 0xf, 0xf, false); v = t > v ? t : v;
  t = (u32) __builtin_amdgcn_update_dpp(0, (int) v, 0x114, 0xf, 0xf, false); v = t > v ? t : v;
  t = (u32) __builtin_amdgcn_update_dpp(0, (int) v, 0x118, 0xf, 0xf, false); v = t > v ? t : v;
  t = (u32) __builtin_amdgcn_update_dpp(0, (int) v, 0x142, 0xa, 0xf, false); v = t > v ? t : v;
  t = (u32) __builtin_amdgcn_update_dpp(0, (int) v, 0x143, 0xc, 0xf, false); v = t > v ? t : v;
  return v;
}

// ---------------------------------------------------------------------------------------------------
// Match records of the frame-parallel paths (LZX frames, MSZIP blocks): a POOL per launch instead of a worst-case region
// per frame slot.  A 32 KiB frame can hold 16384 matches (128 KiB of records), the frames of the bench corpus hold ~4200,
// and round 3's scratch reserved the worst case for every slot: 4x the decoded bytes.  Now a launch's frames take chunks
// of REC_CHUNK records from one pool as their parse needs them (one atomic per chunk); a frame's chunk list -- at most
// REC_CHUNKS entries -- lives in its record.  The pool holds REC_POOL_PER_SLOT chunks per frame slot (48 KiB: 1.5x the
// decoded bytes); a frame that finds it empty is simply not parsed ahead (the serial path decodes it).
// Record index j of a frame lives at pool[chunk[j / REC_CHUNK] + j % REC_CHUNK]; readers work in batches of 64 records
// that start at multiples of 64, i.e. inside one chunk.
// ---------------------------------------------------------------------------------------------------
#define REC_CHUNK 1024u
#define REC_CHUNKS 16u
#define REC_POOL_PER_SLOT 6u
struct RecPool { uint2 *base; u32 *head; u32 cap; };          /* records, chunks handed out so far (device counter), chunks in the pool */
struct RecWriter {
  RecPool pool;
  u32 *ctab;                       /* LDS: REC_CHUNKS words, the frame's chunk list (record index of each chunk's first record) */
  u32 *gtab;                       /* the same in the frame's record (global memory), for the reader */
  u32 n_chunks;
  __device__ __forceinline__ void begin(const RecPool &p, u32 *lds_tab, u32 *rec_tab) { pool = p; ctab = lds_tab; gtab = rec_tab; n_chunks = 0; }
  // the LDS table put aside / taken back when something else needs its room in between (lzx_pipe_parse: a block header in the
  // middle of a frame; `keep`: REC_CHUNKS words of LDS that survive it)
  __device__ __forceinline__ void save(u32 *keep, const u32 lane) const { if (lane < REC_CHUNKS) keep[lane] = ctab[lane]; }
  __device__ __forceinline__ void restore(const u32 *keep, const u32 lane) {
    if (lane < REC_CHUNKS) ctab[lane] = keep[lane];
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  }
  // room for the records with index < upto?  (wave-uniform; false: the pool is empty or the frame has more than 16384)
  __device__ __forceinline__ bool ensure(const u32 upto, const u32 lane) {
    bool grew = false;
    while (n_chunks * REC_CHUNK < upto) {
      if (n_chunks >= REC_CHUNKS) return false;
      u32 c = 0;
      if (lane == 0) c = atomicAdd(pool.head, 1u);
      c = rfl(c);
      if (c >= pool.cap) return false;
      if (lane == 0) { ctab[n_chunks] = c * REC_CHUNK; gtab[n_chunks] = c * REC_CHUNK; }
      n_chunks++; grew = true;
    }
    if (grew) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    return true;
  }
  __device__ __forceinline__ uint2 *at(const u32 j) const { return pool.base + ctab[j >> 10] + (j & (REC_CHUNK - 1u)); }
};
// reader side: the records [g, g + 256) of a frame, g a multiple of 256, lie in one chunk: their common base (wave-uniform)
__device__ __forceinline__ const uint2 *rec_group(const uint2 *pool_base, const u32 *chunk, const u32 g) {
  return pool_base + rfl(gld(chunk + (g >> 10))) + (g & (REC_CHUNK - 1u));
}
// reader side: record j of a frame whose chunk list is `chunk` (global memory)
__device__ __forceinline__ const uint2 *rec_at(const uint2 *pool_base, const u32 *chunk, const u32 j) {
  return pool_base + gld(chunk + (j >> 10)) + (j & (REC_CHUNK - 1u));
}

