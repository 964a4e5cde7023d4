// lzss_kernel.hpp -- the two small single-stream codecs of SZDD / KWAJ files: one wavefront per stream.
//
//   LZSS ........ lzss_decompress (libmspack/mspack/lzssd.c:36-91): SZDD files, KWAJ method 2, MS Help.
//   KWAJ LZH .... lzh_decompress / lzh_read_lens (libmspack/mspack/kwajd.c:432-563): KWAJ method 3.
//
// Both use a 4096-byte ring filled with spaces.  Here the ring is the output buffer itself ("linear
// window", as for the other codecs): the caller leaves 4096 bytes of room BELOW the unit's output, the
// kernel fills them with spaces, positions are biased by 4096, and a ring reference becomes a plain
// distance: LZSS names the ring position of the source, d = ((pos - mpos - 1) & 4095) + 1; LZH names the
// distance directly (0 meaning 4096).  Byte-serial copy semantics (window[pos] = window[mpos], both
// advancing) are those of an overlapping LZ77 copy.
// These streams carry no length: decoding ends where the input ends (out_len of the unit is the room
// available; the result's out_len is what the stream produced).
#pragma once
#include "wave_common.hpp"

#define LZSS_WINDOW 4096u
#define KWAJ_P 9                       /* KWAJ_TABLEBITS (kwaj.h:50) */

// copy `len` bytes at distance d (1..4096) to position P of the biased buffer, dropping what lies beyond cap
__device__ __forceinline__ void lzss_copy(u8 *buf, u32 P, u32 d, u32 len, u32 cap, u32 lane)
{
  for (u32 k = lane; k < len; k += WAVE) {
    u32 kk = k;
    while (kk >= d) kk -= d;                             // periodic source: only bytes that existed before
    if (P + k < cap) buf[P + k] = buf[P - d + kk];
  }
}

// ---- LZSS ------------------------------------------------------------------------------------------------
// One control byte and its eight items per step: lane i (< 8) owns item i; its byte offset inside the
// group follows from the control bits below it (a literal takes one byte, a match two).
__device__ void lzss_decode_unit(const mspack_hip_unit &u, const u8 *in_arena, u8 *out_arena, mspack_hip_result *res)
{
  const u32 lane = threadIdx.x;
  const u8 *in = in_arena + u.in_off;
  const u32 in_len = u.in_len, mode = u.window_bits;     // 0 EXPAND, 1 MSHELP, 2 QBASIC (lzssd.c:49-51)
  u8 *buf = out_arena + u.out_off - LZSS_WINDOW;         // biased: output byte k lives at buf[4096 + k]
  const u32 cap = LZSS_WINDOW + u.out_len;
  if (mode > 2u) { if (lane == 0) { res->err = ERR_ARGS; res->flags = 0; res->out_len = 0; res->in_used = 0; res->good_len = 0; res->in_next = 0; } return; }
  for (u32 k = lane; k < LZSS_WINDOW; k += WAVE) buf[k] = 0x20;
  const u32 start = LZSS_WINDOW - (mode == 2u ? 18u : 16u);   // ring position of the first output byte
  const u32 invert = (mode == 1u) ? 0xFFu : 0u;
  u32 ip = 0, P = LZSS_WINDOW;
  for (;;) {
    if (ip >= in_len) break;
    const u32 c = ((u32) in[ip] ^ invert) & 0xFFu;       // wave-uniform
    const u32 below = (~c) & ((1u << (lane & 7u)) - 1u) & 0xFFu;
    const bool it = lane < 8u;
    const u32 ioff = ip + 1u + (lane & 7u) + (u32) __popc(below);
    const bool lit = (c >> (lane & 7u)) & 1u;
    const u32 need = lit ? 1u : 2u;
    const bool have = it && ioff + need <= in_len;       // ENSURE_BYTES before every byte (lzssd.c:15-28)
    u32 b0 = 0, b1 = 0;
    if (have) { b0 = in[ioff]; if (!lit) b1 = in[ioff + 1u]; }
    // the first incomplete item ends the stream; items before it are complete
    const u64 okm = ballot(have) & 0xFFull;
    u32 nitems = (u32) __builtin_ctzll(~okm);            // leading complete items (0..8)
    const u32 olen = (it && lane < nitems) ? (lit ? 1u : (b1 & 0x0Fu) + 3u) : 0u;
    const u32 incl = wave_incl_scan(olen);
    const u32 opos = P + incl - olen;
    if (olen && lit && opos < cap) buf[opos] = (u8) b0;
    // matches in order: a later one may read what an earlier one of this group wrote
    const u64 mm = ballot(olen != 0u && !lit);
    for (u64 m1 = mm; m1; m1 &= m1 - 1ull) {
      u32 j = (u32) __ffsll((long long) m1) - 1u;
      u32 pj = rdl(opos, j), lj = rdl(olen, j), mp = rdl(b0, j) | ((rdl(b1, j) & 0xF0u) << 4);
      u32 ring = (start + (pj - LZSS_WINDOW)) & 4095u;
      u32 d = ((ring - mp - 1u) & 4095u) + 1u;
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      lzss_copy(buf, pj, d, lj, cap, lane);
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    P += rdl(incl, 63);
    if (nitems < 8u) { ip = in_len; break; }
    ip = rdl(ioff + need, 7);
  }
  if (lane == 0) {
    res->err = ERR_OK; res->flags = 0; res->out_len = P - LZSS_WINDOW; res->good_len = P - LZSS_WINDOW;
    res->in_used = ip; res->in_next = 0;
  }
}

// ---- KWAJ LZH --------------------------------------------------------------------------------------------
struct __align__(16) LzhShared {
  u16 tab[5][1 << KWAJ_P];
  u16 sorted[5][256];
  u8  lens[5][256 + 64];
  u32 cnt[20];
};

struct LzhBits {                       // MSB-first bits, fed a byte at a time (kwajd.c:374-381)
  const u8 *in; u32 in_len, ip;
  u64 bb; int bl;
  int input_end;                       // 8 * zero bytes fed after the end of the input (kwajd.c:548-563)
  __device__ __forceinline__ void ensure(int n) {
    while (bl < n) {
      u32 byte = 0;
      if (ip < in_len) byte = in[ip++]; else input_end += 8;
      bb |= (u64) byte << (56 - bl);
      bl += 8;
    }
  }
  // READ_BITS_SAFE (kwajd.c:400-404): true = the read used bits from beyond the end
  __device__ __forceinline__ bool bits(int n, u32 &v) {
    ensure(n);
    v = n ? (u32)(bb >> (64 - n)) : 0u;
    bb <<= n; bl -= n;
    return input_end && bl < input_end;
  }
};

// READ_HUFFSYM_SAFE: returns 0 ok, 1 ran past the end (stop with OK), 2 invalid code (DATAFORMAT)
__device__ __forceinline__ int lzh_sym(LzhBits &b, const u16 *tab, const u16 *sorted, const HuffRegs &hr, u32 lane, u32 &sym)
{
  b.ensure(16);
  u32 e = rfl((u32) tab[(u32)(b.bb >> (64 - KWAJ_P))]);
  if (e == 0) { e = huff_long(hr, sorted, (u32)(b.bb >> 48), lane); if (e == 0) return 2; }
  u32 l = e >> 10;
  b.bb <<= l; b.bl -= (int) l;
  sym = e & 1023u;
  return (b.input_end && b.bl < b.input_end) ? 1 : 0;
}

__device__ void kwaj_lzh_decode_unit(const mspack_hip_unit &u, const u8 *in_arena, u8 *out_arena,
                                     mspack_hip_result *res, LzhShared *sh)
{
  const u32 lane = threadIdx.x;
  u8 *buf = out_arena + u.out_off - LZSS_WINDOW;
  const u32 cap = LZSS_WINDOW + u.out_len;
  LzhBits b;
  b.in = in_arena + u.in_off; b.in_len = u.in_len; b.ip = 0; b.bb = 0; b.bl = 0; b.input_end = 0;
  for (u32 k = lane; k < LZSS_WINDOW; k += WAVE) buf[k] = 0x20;
  HuffRegs hr[5];
  static const u16 nsyms[5] = { 16, 16, 32, 64, 256 };
  u32 types[6], v, P = LZSS_WINDOW, lit_run = 0;
  int err = ERR_OK;
  bool stop = false;

  for (int i = 0; i < 6 && !stop; i++) { if (b.bits(4, v)) stop = true; types[i] = v; }
  for (int t = 0; t < 5 && !stop && !err; t++) {
    // lzh_read_lens (kwajd.c:497-546): serial by nature (each length is coded against the previous one)
    const u32 n = nsyms[t];
    u8 *lens = sh->lens[t];
    u32 c = 0, sel;
    for (u32 k = lane; k < 256u + 64u; k += WAVE) lens[k] = 0;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    switch (types[t]) {
    case 0:
      c = (n == 16u) ? 4u : (n == 32u) ? 5u : (n == 64u) ? 6u : 8u;
      for (u32 k = lane; k < n; k += WAVE) lens[k] = (u8) c;
      break;
    case 1:
      if (b.bits(4, c)) { stop = true; break; }
      if (lane == 0) lens[0] = (u8) c;
      for (u32 i = 1; i < n; i++) {
        if (b.bits(1, sel)) { stop = true; break; }
        if (sel) {
          if (b.bits(1, sel)) { stop = true; break; }
          if (sel == 0) c = (c + 1u) & 0xFFu;
          else if (b.bits(4, c)) { stop = true; break; }
        }
        if (lane == 0) lens[i] = (u8) c;
      }
      break;
    case 2:
      if (b.bits(4, c)) { stop = true; break; }
      if (lane == 0) lens[0] = (u8) c;
      for (u32 i = 1; i < n; i++) {
        if (b.bits(2, sel)) { stop = true; break; }
        if (sel == 3u) { if (b.bits(4, c)) { stop = true; break; } }
        else c = (c + sel - 1u) & 0xFFu;
        if (lane == 0) lens[i] = (u8) c;
      }
      break;
    case 3:
      for (u32 i = 0; i < n; i++) { if (b.bits(4, c)) { stop = true; break; } if (lane == 0) lens[i] = (u8) c; }
      break;
    default: break;                                      // types 4..15: the lengths stay as they are (zero)
    }
    if (stop) break;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    if (huff_build<KWAJ_P>(lens, (int) n, KWAJ_P, sh->tab[t], sh->sorted[t], sh->cnt, hr[t], lane, false)) err = ERR_DATAFORMAT;
  }
  while (!stop && !err && !b.input_end) {
    u32 len, sym;
    int r = lzh_sym(b, sh->tab[lit_run ? 1 : 0], sh->sorted[lit_run ? 1 : 0], hr[lit_run ? 1 : 0], lane, len);
    if (r) { if (r == 2) err = ERR_DATAFORMAT; break; }
    if (len > 0u) {
      len += 2u; lit_run = 0;
      r = lzh_sym(b, sh->tab[3], sh->sorted[3], hr[3], lane, sym);
      if (r) { if (r == 2) err = ERR_DATAFORMAT; break; }
      u32 off = sym << 6;
      if (b.bits(6, v)) break;
      off |= v;
      u32 d = off ? off : LZSS_WINDOW;
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      lzss_copy(buf, P, d, len, cap, lane);
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      P += len;
    }
    else {
      r = lzh_sym(b, sh->tab[2], sh->sorted[2], hr[2], lane, len);
      if (r) { if (r == 2) err = ERR_DATAFORMAT; break; }
      len++;
      lit_run = (len == 32u) ? 0u : 1u;
      while (len-- > 0u) {
        r = lzh_sym(b, sh->tab[4], sh->sorted[4], hr[4], lane, sym);
        if (r) break;
        if (lane == 0 && P < cap) buf[P] = (u8) sym;
        P++;
      }
      if (r) { if (r == 2) err = ERR_DATAFORMAT; break; }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  if (lane == 0) {
    res->err = err; res->flags = 0; res->out_len = P - LZSS_WINDOW; res->good_len = P - LZSS_WINDOW;
    res->in_used = b.ip; res->in_next = 0;
  }
}
