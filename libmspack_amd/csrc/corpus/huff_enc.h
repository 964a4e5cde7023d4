/* huff_enc.h -- length-limited canonical Huffman code construction for the corpus encoders
 * (test/bench infrastructure).  Produces a COMPLETE prefix code (Kraft sum exactly 1), which is
 * what the reference's table builder demands (readhuff.h:121-122,175): Huffman depths from the
 * two-queue method, then the JPEG Annex-K style depth limiter, then lengths handed out by
 * descending frequency. */
#ifndef MSPACK_AMD_HUFF_ENC_H
#define MSPACK_AMD_HUFF_ENC_H
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { uint32_t freq; uint16_t sym; } he_item;
static int he_cmp(const void *a, const void *b) {
  const he_item *x = (const he_item *) a, *y = (const he_item *) b;
  if (x->freq != y->freq) return x->freq < y->freq ? -1 : 1;
  return (int) x->sym - (int) y->sym;
}

/* freq[0..n) -> lens[0..n) with max length `limit`.  n <= 1024. */
static void he_build_lengths(const uint32_t *freq, int n, int limit, uint8_t *lens)
{
  he_item leaves[1024];
  uint64_t weight[2048];
  int parent[2048], depth[2048], bits[64];
  int used = 0, i, nodes, q1, q2, l;

  memset(lens, 0, (size_t) n);
  for (i = 0; i < n; i++) if (freq[i]) { leaves[used].freq = freq[i]; leaves[used].sym = (uint16_t) i; used++; }
  if (used == 0) return;
  if (used == 1) {                                   /* a lone symbol still needs a complete code */
    int other = (leaves[0].sym == 0) ? 1 : 0;
    lens[leaves[0].sym] = 1; lens[other] = 1;
    return;
  }
  qsort(leaves, (size_t) used, sizeof(he_item), he_cmp);
  for (i = 0; i < used; i++) weight[i] = leaves[i].freq;
  nodes = used; q1 = 0; q2 = used;
  while (nodes < 2 * used - 1) {
    int pick[2], k;
    for (k = 0; k < 2; k++) {
      if (q1 < used && (q2 >= nodes || weight[q1] <= weight[q2])) pick[k] = q1++;
      else pick[k] = q2++;
    }
    weight[nodes] = weight[pick[0]] + weight[pick[1]];
    parent[pick[0]] = parent[pick[1]] = nodes;
    nodes++;
  }
  depth[nodes - 1] = 0;
  for (i = nodes - 2; i >= 0; i--) depth[i] = depth[parent[i]] + 1;
  memset(bits, 0, sizeof(bits));
  for (i = 0; i < used; i++) bits[depth[i] > 62 ? 62 : depth[i]]++;
  for (l = 62; l > limit; l--) {                     /* push over-deep leaves up, keeping Kraft == 1 */
    while (bits[l] > 0) {
      int j = l - 2;
      while (bits[j] == 0) j--;
      bits[l] -= 2; bits[l - 1] += 1; bits[j + 1] += 2; bits[j] -= 1;
    }
  }
  /* most frequent symbols get the shortest codes: leaves[] is ascending by freq */
  i = used - 1;
  for (l = 1; l <= limit; l++) { int c = bits[l]; while (c-- > 0) lens[leaves[i--].sym] = (uint8_t) l; }
}

/* canonical codes (MSB-first), ordered by (length, symbol) as readhuff.h:97-117 assigns them */
static void he_assign_codes(const uint8_t *lens, int n, uint16_t *codes)
{
  uint32_t next[18], count[18];
  int i, l;
  memset(count, 0, sizeof(count));
  for (i = 0; i < n; i++) count[lens[i]]++;
  count[0] = 0; next[0] = 0; next[1] = 0;
  for (l = 2; l <= 17; l++) next[l] = (next[l - 1] + count[l - 1]) << 1;
  for (i = 0; i < n; i++) codes[i] = lens[i] ? (uint16_t) next[lens[i]]++ : 0;
}
#endif
