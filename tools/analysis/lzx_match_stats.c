/* lzx_match_stats.c -- analysis tool (not product, not test): what the matches of the bench corpus look like.
 * Hooks the oracle's match site.  Build: gcc -O2 -I. tools/analysis/lzx_match_stats.c libmspack_amd/csrc/corpus/*.c -lpthread -lm
 * Prints per frame index: tokens, matches, match bytes; bytes whose chain of sources leaves the frame ("ext-derived");
 * runs of such bytes; per batch of 64 matches: matches whose source overlaps an earlier match of the batch; length stats. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
static void stats_match(uint32_t wpos, uint32_t len, uint32_t off, uint32_t slot);
#define ORACLE_MATCH_HOOK(p, l, o, s) stats_match(p, l, o, s)
#include "../../oracle/lzx_oracle.c"
#include "../../libmspack_amd/csrc/corpus/corpus.h"

#define MAXF 64
static uint64_t n_match[MAXF], n_mbytes[MAXF], n_ext[MAXF], n_extruns[MAXF], n_frames_seen[MAXF], n_rep[MAXF], n_extm[MAXF], n_extm_direct[MAXF];
static uint64_t len_hist[260];
static uint64_t c_depth_sum[3], c_hist8[3], c_jumps[3], b_depth_sum, b_depth_hist[66], b_total, b_dep, b_maxlen_sum, b_batches, b_bytes, b_dep_tile[4], b_long16, b_long32;
static uint64_t tile_dep[4], n_selfov, n_selfov_short, n_selfov16, n_in4k, n_in4k_b, n_in2k, n_len16;
/* per unit state */
static uint32_t *extsrc;     /* per output byte of the unit: 0 = final inside its frame, else 1 + ultimate ext source */
static uint32_t unit_bytes;
static uint32_t bp[64], bl[64], bo[64]; static int bn;
static void flush_batch(void) {
  if (!bn) return;
  uint32_t mx = 0, sum = 0; int dep = 0;
  int depth[64], maxd = 0;
  for (int i = 0; i < bn; i++) {
    uint32_t s0 = bp[i] - bo[i], s1 = s0 + bl[i];
    int dd = 0;
    for (int j = 0; j < i; j++) if (s1 > bp[j] && s0 < bp[j] + bl[j] && depth[j] > dd) dd = depth[j];
    depth[i] = dd + 1; if (depth[i] > maxd) maxd = depth[i];
  }
  b_depth_sum += maxd; b_depth_hist[maxd > 65 ? 65 : maxd]++;
  /* containment links collapsed first (pointer jumping): variant 0 = blocker anywhere in the batch, 1 = among the 3 lanes before, 2 = the lane before only */
  for (int var = 0; var < 3; var++) {
    uint32_t es[64]; int dep2[64], md = 0, jumps = 0;
    for (int i = 0; i < bn; i++) {
      uint32_t s0 = bp[i] - bo[i];
      int hops = 0;
      if (bo[i] >= bl[i]) for (;;) {
        int j, found = -1;
        for (j = i - 1; j >= 0 && (var == 0 || j >= i - (var == 1 ? 3 : 1)); j--)
          if (s0 >= bp[j] && s0 + bl[i] <= bp[j] + bl[j] && bo[j] >= bl[j]) { found = j; break; }
        if (found < 0) break;
        s0 -= bo[found]; hops++;
        if (s0 > bp[i]) break;
      }
      es[i] = s0; if (hops > jumps) jumps = hops;
      int dd = 0;
      uint32_t s1 = s0 + bl[i];
      for (int j = 0; j < i; j++) if (s1 > bp[j] && s0 < bp[j] + bl[j] && dep2[j] > dd) dd = dep2[j];
      dep2[i] = dd + 1; if (dep2[i] > md) md = dep2[i];
    }
    c_depth_sum[var] += md; if (md <= 4) c_hist8[var]++; c_jumps[var] += jumps;
  }
  for (int i = 0; i < bn; i++) {
    if (bl[i] > mx) mx = bl[i]; sum += bl[i];
    uint32_t s0 = bp[i] - bo[i], s1 = s0 + bl[i];
    int d = 0;
    for (int j = 0; j < i && !d; j++) if (s1 > bp[j] && s0 < bp[j] + bl[j]) d = 1;
    if (bo[i] < bl[i]) d = 0 + d;   /* self overlap alone is fine for a byte-serial lane */
    dep += d;
  }
  b_total += bn; b_dep += dep; b_maxlen_sum += mx; b_batches++; b_bytes += sum; bn = 0;
}
static void stats_match(uint32_t wpos, uint32_t len, uint32_t off, uint32_t slot) {
  uint32_t f = wpos / 32768u; if (f >= MAXF) f = MAXF - 1;
  uint32_t fpos = f * 32768u;
  n_match[f]++; n_mbytes[f] += len; if (slot < 3) n_rep[f]++;
  len_hist[len > 258 ? 258 : len]++;
  if (len > 16) b_long16++; if (len > 32) b_long32++;
  if (off < len) { n_selfov++; if (len <= 8) n_selfov_short++; if (len <= 16) n_selfov16++; }
  if (len <= 16) n_len16++;
  { uint32_t T0 = wpos & ~4095u; if (off <= wpos && wpos - off >= T0) n_in4k++; T0 = wpos & ~2047u; if (off <= wpos && wpos - off >= T0) n_in2k++; if (off < 4096) n_in4k_b++; }
  int any_ext = 0;
  for (uint32_t k = 0; k < len; k++) {
    uint32_t b = wpos + k;
    if (off > b) { extsrc[b] = 0; continue; }          /* (before the stream: zeros) */
    uint32_t s = b - off;
    uint32_t e = s < fpos ? 1u + s : extsrc[s];
    extsrc[b] = e;
    if (e) { n_ext[f]++; any_ext = 1; if (k == 0 || extsrc[b - 1] == 0 || extsrc[b - 1] + 1 != e) n_extruns[f]++; }
  }
  if (any_ext) n_extm[f]++;
  if (off <= wpos && wpos - off < fpos) n_extm_direct[f]++;
  bp[bn] = wpos; bl[bn] = len; bo[bn] = off; bn++;
  if (bn == 64) flush_batch();
}

int main(int argc, char **argv) {
  int n_units = argc > 1 ? atoi(argv[1]) : 64;
  size_t ub = argc > 2 ? (size_t) atol(argv[2]) : 65536;
  int reset = argc > 3 ? atoi(argv[3]) : 2;
  int kind = argc > 4 ? atoi(argv[4]) : MSPK_TEXT_MIX;
  int wbits = argc > 5 ? atoi(argv[5]) : 21;
  uint8_t *plain = malloc(ub), *comp = malloc(mspk_lzx_bound(ub) + 64), *out = malloc(ub);
  extsrc = calloc(ub, 4); unit_bytes = (uint32_t) ub;
  uint64_t tot_c = 0;
  for (int u = 0; u < n_units; u++) {
    mspk_gen_plaintext(0xB5EED ^ ((uint64_t) u * 0x9E3779B97F4A7C15ull), kind, plain, ub);
    mspk_lzx_opts o; memset(&o, 0, sizeof(o)); o.use_repeats = 1; o.lazy = 1;
    size_t c = mspk_lzx_encode(plain, ub, wbits, reset, &o, comp, mspk_lzx_bound(ub), NULL);
    memset(comp + c, 0, 64);
    memset(extsrc, 0, ub * 4);
    oracle_result r;
    oracle_lzx_decode(comp, c + 8, out, ub, ub, ub, wbits, reset, 0, &r);
    flush_batch();
    if (r.err || memcmp(out, plain, ub)) { printf("unit %d: decode mismatch err %d\n", u, r.err); return 1; }
    tot_c += c;
    for (uint32_t f = 0; f < (ub + 32767) / 32768 && f < MAXF; f++) n_frames_seen[f]++;
  }
  printf("units %d x %zu bytes, reset %d, kind %d, window %d: ratio %.3f\n", n_units, ub, reset, kind, wbits, (double) tot_c / ((double) n_units * ub));
  printf("frame  matches  match_bytes  rep%%  ext_bytes  ext%%ofmatchbytes  ext_runs  bytes/run  matches_with_ext%%  direct_ext_matches%%\n");
  for (int f = 0; f < MAXF; f++) if (n_frames_seen[f] && (f < 6 || f == MAXF - 1 || f % 8 == 0)) {
    double nf = (double) n_frames_seen[f];
    printf("%5d  %7.0f  %11.0f  %4.1f  %9.0f  %5.1f  %8.0f  %5.2f  %5.1f  %5.1f\n", f, n_match[f] / nf, n_mbytes[f] / nf, 100.0 * n_rep[f] / (n_match[f] + 1e-9), n_ext[f] / nf,
           100.0 * n_ext[f] / (n_mbytes[f] + 1e-9), n_extruns[f] / nf, n_ext[f] / (n_extruns[f] + 1e-9), 100.0 * n_extm[f] / (n_match[f] + 1e-9), 100.0 * n_extm_direct[f] / (n_match[f] + 1e-9));
  }
  printf("batches of 64 matches: %.1f bytes per batch, mean max length %.1f, matches depending on an earlier match of the batch %.1f%%\n",
         (double) b_bytes / b_batches, (double) b_maxlen_sum / b_batches, 100.0 * b_dep / b_total);
  printf("rounds a batch needs (longest chain of matches that read each other): mean %.2f; share of batches with <=2: %.1f%%, <=4: %.1f%%, <=8: %.1f%%, >16: %.1f%%\n", (double) b_depth_sum / b_batches,
         100.0 * (b_depth_hist[1] + b_depth_hist[2]) / b_batches, 100.0 * (b_depth_hist[1] + b_depth_hist[2] + b_depth_hist[3] + b_depth_hist[4]) / b_batches,
         ({ uint64_t a_ = 0; for (int k = 0; k <= 8; k++) a_ += b_depth_hist[k]; 100.0 * a_ / b_batches; }), ({ uint64_t a_ = 0; for (int k = 17; k < 66; k++) a_ += b_depth_hist[k]; 100.0 * a_ / b_batches; }));
  for (int var = 0; var < 3; var++) printf("  containment chains collapsed first (blocker %s): rounds mean %.2f, batches with <= 4 rounds %.1f%%, longest chain of hops mean %.1f\n", var == 0 ? "anywhere in the batch" : var == 1 ? "within 3 lanes" : "the lane before", (double) c_depth_sum[var] / b_batches, 100.0 * c_hist8[var] / b_batches, (double) c_jumps[var] / b_batches);
  printf("matches longer than 16: %.2f%%, longer than 32: %.2f%%\n", 100.0 * b_long16 / b_total, 100.0 * b_long32 / b_total);
  printf("self-overlapping (off < len): %.2f%% of matches; of those with len <= 8: %.2f%%, len <= 16: %.2f%%\n", 100.0 * n_selfov / b_total, 100.0 * n_selfov_short / b_total, 100.0 * n_selfov16 / b_total);
  printf("source starts inside the same aligned 4 KiB tile: %.1f%%, 2 KiB tile: %.1f%%; offset < 4096: %.1f%%\n", 100.0 * n_in4k / b_total, 100.0 * n_in2k / b_total, 100.0 * n_in4k_b / b_total);
  printf("length histogram (2..20): ");
  for (int l = 2; l <= 20; l++) printf("%d:%.1f%% ", l, 100.0 * len_hist[l] / b_total);
  printf("\n");
  return 0;
}
