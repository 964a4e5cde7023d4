/* textgen.c -- deterministic synthetic plaintext for the decode corpora (test/bench infrastructure).
 * Three families: English/HTML-like text from a Zipf-sampled pseudo-word
 * vocabulary, structured binary (LE32 counters / pointers / x86-like code with 0xE8 call sites),
 * and repeated fixed-layout records.  Everything derives from the 64-bit seed only. */
#include <string.h>
#include <stdlib.h>
#include <math.h>
#include <pthread.h>
#include "corpus.h"

typedef struct { uint64_t s; } rng_t;
static inline uint64_t rng_next(rng_t *r) {             /* splitmix64 */
  uint64_t z = (r->s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static inline uint32_t rng_below(rng_t *r, uint32_t n) { return (uint32_t)((rng_next(r) >> 32) * (uint64_t) n >> 32); }

/* ---- vocabulary (built once, fixed seed) --------------------------------------------------------- */
#define VOCAB 4096
static char vocab_word[VOCAB][14];
static uint8_t vocab_len[VOCAB];
static uint32_t vocab_cdf[VOCAB];
static pthread_once_t vocab_once = PTHREAD_ONCE_INIT;

static void vocab_init(void) {
  static const char *onset[] = { "b", "c", "d", "f", "g", "h", "l", "m", "n", "p", "r", "s", "t", "w",
                                 "st", "tr", "ch", "th", "sh", "pr", "pl", "gr", "br", "" };
  static const char *nucleus[] = { "a", "e", "i", "o", "u", "ea", "ou", "io", "ai", "ee" };
  static const char *coda[] = { "", "", "n", "r", "s", "t", "l", "m", "d", "ng", "nt", "st", "ck" };
  static const char *common[] = { "the", "of", "and", "to", "a", "in", "is", "that", "for", "it", "as",
                                  "with", "was", "on", "be", "by", "this", "are", "or", "from", "an",
                                  "at", "not", "which", "have", "can", "file", "data", "window", "help" };
  rng_t r = { 0x5EEDC0DEull };
  double sum = 0, acc = 0;
  int i;
  for (i = 0; i < VOCAB; i++) {
    char *w = vocab_word[i];
    if (i < (int)(sizeof(common) / sizeof(common[0]))) { strcpy(w, common[i]); }
    else {
      int syl = 1 + (int) rng_below(&r, 3) + (i > 1000), k;
      w[0] = 0;
      for (k = 0; k < syl && strlen(w) < 9; k++) {
        strcat(w, onset[rng_below(&r, 24)]);
        strcat(w, nucleus[rng_below(&r, 10)]);
        strcat(w, coda[rng_below(&r, 13)]);
      }
    }
    vocab_len[i] = (uint8_t) strlen(w);
    sum += 1.0 / pow((double)(i + 1), 1.05);
  }
  for (i = 0; i < VOCAB; i++) {
    acc += 1.0 / pow((double)(i + 1), 1.05);
    vocab_cdf[i] = (uint32_t)(acc / sum * 4294967295.0);
  }
  vocab_cdf[VOCAB - 1] = 0xFFFFFFFFu;
}
static int vocab_sample(rng_t *r) {
  uint32_t u = (uint32_t)(rng_next(r) >> 32);
  int lo = 0, hi = VOCAB - 1;
  while (lo < hi) { int mid = (lo + hi) >> 1; if (vocab_cdf[mid] < u) lo = mid + 1; else hi = mid; }
  return lo;
}

/* ---- generators: each fills out[0..n) ----------------------------------------------------------------- */
static size_t put_str(uint8_t *out, size_t pos, size_t n, const char *s) {
  while (*s && pos < n) out[pos++] = (uint8_t) *s++;
  return pos;
}

static void gen_english(rng_t *r, uint8_t *out, size_t n) {
  static const char *tags[] = { "<p>", "</p>\r\n", "<b>", "</b>", "<li>", "</li>\r\n", "<br>\r\n",
                                "<a href=\"", "<td class=\"cell\">", "</td>", "<h2>", "</h2>\r\n" };
  size_t pos = 0;
  int sentence_left = 0, cap = 1;
  while (pos < n) {
    int w = vocab_sample(r);
    uint32_t roll = rng_below(r, 100);
    if (sentence_left == 0) { sentence_left = 4 + (int) rng_below(r, 14); cap = 1; }
    if (roll < 4) { pos = put_str(out, pos, n, tags[rng_below(r, 12)]); if (roll == 0) pos = put_str(out, pos, n, "topic"); }
    if (pos < n) {
      size_t k, l = vocab_len[w];
      for (k = 0; k < l && pos < n; k++) {
        char c = vocab_word[w][k];
        if (cap && k == 0 && c >= 'a' && c <= 'z') c = (char)(c - 32);
        out[pos++] = (uint8_t) c;
      }
      cap = 0;
    }
    if (--sentence_left == 0) {
      pos = put_str(out, pos, n, (roll & 7) == 0 ? "?" : ".");
      pos = put_str(out, pos, n, (rng_below(r, 6) == 0) ? "\r\n" : " ");
    }
    else if (roll > 92) pos = put_str(out, pos, n, ", ");
    else pos = put_str(out, pos, n, " ");
  }
}

static void put_le32(uint8_t *out, size_t *pos, size_t n, uint32_t v) {
  int k;
  for (k = 0; k < 4 && *pos < n; k++) out[(*pos)++] = (uint8_t)(v >> (8 * k));
}

static void gen_binary(rng_t *r, uint8_t *out, size_t n) {
  static const uint8_t ops[] = { 0x8B, 0x89, 0x55, 0x5D, 0xC3, 0x83, 0xEC, 0x45, 0x4D, 0xFF, 0x50, 0x51,
                                 0x6A, 0x00, 0x74, 0x75, 0x33, 0xC0, 0x85, 0x8D };
  size_t pos = 0;
  uint32_t counter = (uint32_t) rng_next(r) & 0xFFFF, base = 0x00400000u + (rng_below(r, 256) << 12);
  while (pos < n) {
    uint32_t kind = rng_below(r, 10), k, cnt;
    if (kind < 4) {                              /* x86-like code with CALL rel32 */
      cnt = 24 + rng_below(r, 200);
      for (k = 0; k < cnt && pos < n; k++) {
        uint32_t roll = rng_below(r, 48);
        if (roll == 0) {
          out[pos++] = 0xE8;
          put_le32(out, &pos, n, (uint32_t)(int32_t)((int32_t) rng_below(r, 60000) - 30000));
        }
        else if (roll < 6) { out[pos++] = ops[rng_below(r, 20)]; if (pos < n) out[pos++] = (uint8_t) rng_below(r, 256); }
        else out[pos++] = ops[rng_below(r, 20)];
      }
    }
    else if (kind < 7) {                         /* table of increasing LE32 counters */
      uint32_t stride = 1u << rng_below(r, 6);
      cnt = 8 + rng_below(r, 120);
      for (k = 0; k < cnt && pos < n; k++) { put_le32(out, &pos, n, counter); counter += stride; }
    }
    else if (kind < 9) {                         /* pointer table: base + small offsets */
      cnt = 8 + rng_below(r, 64);
      for (k = 0; k < cnt && pos < n; k++) put_le32(out, &pos, n, base + (rng_below(r, 4096) & ~3u));
    }
    else {                                       /* zero / 0xFF padding */
      uint8_t fill = (rng_below(r, 2)) ? 0x00 : 0xFF;
      cnt = 16 + rng_below(r, 200);
      for (k = 0; k < cnt && pos < n; k++) out[pos++] = fill;
    }
  }
}

static void gen_records(rng_t *r, uint8_t *out, size_t n) {
  size_t pos = 0;
  while (pos < n) {
    uint8_t tmpl[160];
    uint32_t rec_len = 48 + rng_below(r, 100), nrec = 20 + rng_below(r, 200), i, k, id = rng_below(r, 100000);
    rng_t tr = { rng_next(r) };
    gen_english(&tr, tmpl, rec_len);
    for (k = 0; k < rec_len; k += 16) tmpl[k] = '|';
    for (i = 0; i < nrec && pos < n; i++) {
      uint32_t edits = rng_below(r, 4);
      char num[12];
      int dl = 0, v = (int)(id + i);
      for (k = 0; k < rec_len && pos + k < n; k++) out[pos + k] = tmpl[k];
      do { num[dl++] = (char)('0' + v % 10); v /= 10; } while (v && dl < 10);
      for (k = 0; k < (uint32_t) dl && 2 + k < rec_len && pos + 2 + k < n; k++) out[pos + 2 + k] = (uint8_t) num[dl - 1 - (int) k];
      while (edits--) { uint32_t at = rng_below(r, rec_len); if (pos + at < n) out[pos + at] = (uint8_t)('a' + rng_below(r, 26)); }
      pos += rec_len;
      if (pos < n) out[pos++] = '\n';
    }
  }
}

static void gen_random(rng_t *r, uint8_t *out, size_t n) {
  size_t pos = 0;
  while (pos < n) { uint64_t v = rng_next(r); int k; for (k = 0; k < 8 && pos < n; k++) out[pos++] = (uint8_t)(v >> (8 * k)); }
}

static void gen_repetitive(rng_t *r, uint8_t *out, size_t n) {
  uint8_t phrase[96];
  uint32_t pl = 24 + rng_below(r, 72);
  size_t pos = 0;
  gen_english(r, phrase, pl);
  while (pos < n) {
    uint32_t k;
    for (k = 0; k < pl && pos < n; k++) out[pos++] = phrase[k];
    if (rng_below(r, 40) == 0) phrase[rng_below(r, pl)] = (uint8_t)('A' + rng_below(r, 26));
  }
}

void mspk_gen_plaintext(uint64_t seed, int kind, uint8_t *out, size_t n) {
  rng_t r = { seed * 0x9E3779B97F4A7C15ull + 0x1234567ull };
  pthread_once(&vocab_once, vocab_init);
  switch (kind) {
  case MSPK_TEXT_ENGLISH:    gen_english(&r, out, n); break;
  case MSPK_TEXT_BINARY:     gen_binary(&r, out, n); break;
  case MSPK_TEXT_RECORDS:    gen_records(&r, out, n); break;
  case MSPK_TEXT_RANDOM:     gen_random(&r, out, n); break;
  case MSPK_TEXT_REPETITIVE: gen_repetitive(&r, out, n); break;
  default: {
    size_t pos = 0;
    while (pos < n) {
      size_t chunk = 2048 + rng_below(&r, 12288);
      uint32_t which = rng_below(&r, 3);
      if (chunk > n - pos) chunk = n - pos;
      if (which == 0) gen_english(&r, out + pos, chunk);
      else if (which == 1) gen_binary(&r, out + pos, chunk);
      else gen_records(&r, out + pos, chunk);
      pos += chunk;
    }
    break; }
  }
}
