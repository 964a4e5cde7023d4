/* oracle/oracle_huff.h -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Canonical Huffman decoding for the CPU restatement.  The reference builds a direct table plus
 * a binary-tree spill area (readhuff.h:83-176); the table *layout* is not observable, only
 *   (a) which sets of code lengths are accepted, and
 *   (b) which symbol a bit pattern decodes to once accepted.
 * (a): lengths 1..tablebits are placed first; if they alone fill the code space the builder
 *      returns success WITHOUT looking at longer codes (readhuff.h:121-122) -- those longer codes
 *      can then never be decoded.  Otherwise the whole code must be exactly complete
 *      (readhuff.h:147,175); over-subscription at any point is an error (readhuff.h:108,147).
 * (b): codes are assigned in (length, symbol) order, MSB first (readhuff.h:97-117,144-172).
 */
#ifndef MSPACK_AMD_ORACLE_HUFF_H
#define MSPACK_AMD_ORACLE_HUFF_H
#include <stdint.h>
#include <string.h>

#define OH_MAXLEN 16

typedef struct oh_table {
  uint32_t limit[OH_MAXLEN + 2]; /* exclusive upper bound of left-aligned 16-bit codes per length */
  uint16_t first[OH_MAXLEN + 2]; /* first canonical code of each length                          */
  uint16_t offs[OH_MAXLEN + 2];  /* index in sorted[] of the first symbol of each length         */
  uint16_t sorted[2576 + 64];    /* symbols ordered by (length, symbol)                          */
} oh_table;

/* acceptance only (no table); mirrors oh_build's decision but with the exact short-circuit */
static int oh_accepts(const uint8_t *lens, int nsyms, int tablebits)
{
  uint64_t kshort = 0, kall = 0;
  int s;
  for (s = 0; s < nsyms; s++) {
    int l = lens[s];
    if (l < 1 || l > OH_MAXLEN) continue;
    kall += 1ull << (OH_MAXLEN - l);
    if (l <= tablebits) kshort += 1ull << (OH_MAXLEN - l);
  }
  if (kshort > (1ull << OH_MAXLEN)) return 0;
  if (kshort == (1ull << OH_MAXLEN)) return 1;
  return kall == (1ull << OH_MAXLEN);
}

/* returns 0 = accepted, 1 = rejected; exactly the reference's accept set */
static int oh_build(oh_table *t, const uint8_t *lens, int nsyms, int tablebits)
{
  uint32_t count[OH_MAXLEN + 2];
  uint32_t kshort = 0, code = 0;
  int l, s, n = 0, maxl = OH_MAXLEN;

  if (!oh_accepts(lens, nsyms, tablebits)) return 1;
  memset(count, 0, sizeof(count));
  for (s = 0; s < nsyms; s++) if (lens[s] >= 1 && lens[s] <= OH_MAXLEN) count[lens[s]]++;
  for (l = 1; l <= tablebits && l <= OH_MAXLEN; l++) kshort += count[l] << (OH_MAXLEN - l);
  if (kshort == (1u << OH_MAXLEN)) maxl = tablebits;   /* longer codes are never reachable */
  for (l = 1; l <= OH_MAXLEN; l++) {
    t->first[l] = (uint16_t) code;
    t->offs[l]  = (uint16_t) n;
    if (l <= maxl) {
      for (s = 0; s < nsyms; s++) if (lens[s] == l) t->sorted[n++] = (uint16_t) s;
      code += count[l];
    }
    t->limit[l] = (l <= maxl) ? (code << (OH_MAXLEN - l)) : 0;
    code <<= 1;
  }
  return 0;
}

/* decode one symbol from a left-aligned 16-bit peek; returns symbol, *len = code length.
 * Only valid on an accepted table (every 16-bit pattern then resolves). */
static inline int oh_decode(const oh_table *t, uint32_t peek16, int *len)
{
  int l;
  for (l = 1; l <= OH_MAXLEN; l++) {
    if (peek16 < t->limit[l]) {
      *len = l;
      return t->sorted[t->offs[l] + ((peek16 >> (OH_MAXLEN - l)) - t->first[l])];
    }
  }
  *len = 0;
  return -1;
}
#endif
