/* containers.c -- minimal CAB and CHM writers for the synthetic corpora (test/bench infrastructure).
 * The reference has no container writers (cabc.c / chmc.c are stubs).  Layouts follow what its
 * readers parse: CAB cab.h:16-67 + cabd.c:317-628,1362-1479; CHM chm.h:17-92 + chmd.c:254-532,
 * 704-898 (quick-ref area), 1072-1267 (ControlData / ResetTable / SpanInfo). */
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include "corpus.h"

static void w16(uint8_t *p, unsigned v) { p[0] = (uint8_t) v; p[1] = (uint8_t)(v >> 8); }
static void w32(uint8_t *p, uint32_t v) { p[0] = (uint8_t) v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
static void w64(uint8_t *p, uint64_t v) { w32(p, (uint32_t) v); w32(p + 4, (uint32_t)(v >> 32)); }

static uint32_t cab_checksum(const uint8_t *d, uint32_t n, uint32_t ck) {
  uint32_t k = n >> 2, t = 0;
  while (k--) { ck ^= (uint32_t) d[0] | ((uint32_t) d[1] << 8) | ((uint32_t) d[2] << 16) | ((uint32_t) d[3] << 24); d += 4; }
  switch (n & 3) { case 3: t |= (uint32_t) *d++ << 16; /* fall through */ case 2: t |= (uint32_t) *d++ << 8; /* fall through */ case 1: t |= *d; }
  return ck ^ t;
}

size_t mspk_cab_write(const mspk_cab_folder *folders, int n_folders, const mspk_cab_file *files, int n_files,
                      uint8_t *dst, size_t cap)
{
  size_t pos = 0x24 + (size_t) n_folders * 8, files_off, i;
  int f, b;
  uint8_t *fold_hdr = dst + 0x24;
  for (i = 0; i < (size_t) n_files; i++) pos += 16 + strlen(files[i].name) + 1;
  for (f = 0; f < n_folders; f++) {
    size_t need = 0;
    for (b = 0; b < folders[f].n_blocks; b++) need += 8 + folders[f].block_comp[b];
    pos += need;
  }
  if (pos > cap || n_folders > 65535 || n_files > 65535) return 0;
  memset(dst, 0, 0x24);
  memcpy(dst, "MSCF", 4);
  w32(dst + 0x08, (uint32_t) pos);
  files_off = 0x24 + (size_t) n_folders * 8;
  w32(dst + 0x10, (uint32_t) files_off);
  dst[0x18] = 3; dst[0x19] = 1;
  w16(dst + 0x1A, (unsigned) n_folders); w16(dst + 0x1C, (unsigned) n_files);
  w16(dst + 0x1E, 0); w16(dst + 0x20, 0x1234); w16(dst + 0x22, 0);
  pos = files_off;
  for (i = 0; i < (size_t) n_files; i++) {
    size_t nl = strlen(files[i].name) + 1;
    w32(dst + pos, files[i].length); w32(dst + pos + 4, files[i].folder_offset);
    w16(dst + pos + 8, files[i].folder_index);
    w16(dst + pos + 10, (unsigned)((2026 - 1980) << 9 | 9 << 5 | 26));   /* date */
    w16(dst + pos + 12, (unsigned)(12 << 11));                             /* time */
    w16(dst + pos + 14, 0x20);                                             /* archive attribute */
    memcpy(dst + pos + 16, files[i].name, nl);
    pos += 16 + nl;
  }
  for (f = 0; f < n_folders; f++) {
    const uint8_t *src = folders[f].data;
    w32(fold_hdr + f * 8, (uint32_t) pos);
    w16(fold_hdr + f * 8 + 4, (unsigned) folders[f].n_blocks);
    w16(fold_hdr + f * 8 + 6, (unsigned) folders[f].comp_type);
    for (b = 0; b < folders[f].n_blocks; b++) {
      uint32_t cl = folders[f].block_comp[b], ul = folders[f].block_uncomp[b], ck;
      uint8_t *h = dst + pos;
      w16(h + 4, cl); w16(h + 6, ul);
      memcpy(h + 8, src, cl);
      ck = cab_checksum(h + 8, cl, 0);
      ck = cab_checksum(h + 4, 4, ck);
      w32(h, ck);
      src += cl; pos += 8 + cl;
    }
  }
  return pos;
}

/* ---- CHM ------------------------------------------------------------------------------------------------ */
typedef struct { const char *name; unsigned section; uint64_t offset, length; } dirent_t;

static int ci_cmp(const void *a, const void *b) {
  const unsigned char *x = (const unsigned char *)((const dirent_t *) a)->name, *y = (const unsigned char *)((const dirent_t *) b)->name;
  for (;; x++, y++) {
    int cx = tolower(*x), cy = tolower(*y);
    if (cx != cy) return cx - cy;
    if (!cx) return 0;
  }
}
static size_t put_encint(uint8_t *p, uint64_t v) {
  uint8_t tmp[10]; int n = 0; size_t k = 0;
  do { tmp[n++] = (uint8_t)(v & 0x7F); v >>= 7; } while (v);
  while (n-- > 0) p[k++] = (uint8_t)(tmp[n] | (n ? 0x80 : 0));
  return k;
}

static const char n_content[]  = "::DataSpace/Storage/MSCompressed/Content";
static const char n_control[]  = "::DataSpace/Storage/MSCompressed/ControlData";
static const char n_spaninfo[] = "::DataSpace/Storage/MSCompressed/SpanInfo";
static const char n_rtable[]   = "::DataSpace/Storage/MSCompressed/Transform/{7FC28940-9D31-11D0-9B27-00A0C91E9C7C}/InstanceData/ResetTable";

size_t mspk_chm_bound(size_t lzx_len, size_t n_frames, int n_files) {
  return lzx_len + n_frames * 8 + (size_t) n_files * 96 + 65536 + ((size_t) n_files / 20 + 8) * 4096;
}

size_t mspk_chm_write(const uint8_t *lzx, size_t lzx_len, const uint64_t *frame_off, size_t n_frames,
                      uint64_t uncomp_len, int window_bits, int reset_frames,
                      const mspk_chm_file *files, int n_files, uint8_t *dst, size_t cap)
{
  static const uint8_t guids[32] = {
    0x10, 0xFD, 0x01, 0x7C, 0xAA, 0x7B, 0xD0, 0x11, 0x9E, 0x0C, 0x00, 0xA0, 0xC9, 0x22, 0xE6, 0xEC,
    0x11, 0xFD, 0x01, 0x7C, 0xAA, 0x7B, 0xD0, 0x11, 0x9E, 0x0C, 0x00, 0xA0, 0xC9, 0x22, 0xE6, 0xEC };
  static const uint8_t itsp_guid[16] = { 0x6A, 0x92, 0x02, 0x5D, 0x2E, 0x21, 0xD0, 0x11, 0x9D, 0xF9, 0x00, 0xA0, 0xC9, 0x22, 0xE6, 0xEC };
  const size_t CHUNK = 4096;
  int n_ent = n_files + 4, i, n_chunks = 0, n_pmgl = 0, depth = 1;
  uint32_t index_root = 0xFFFFFFFFu;
  dirent_t *ents = (dirent_t *) calloc((size_t) n_ent, sizeof(*ents));
  uint8_t *chunks = NULL;
  size_t rt_len = 0x28 + n_frames * 8, sec0_len, dir_off = 0x78 + 0x54, sec0_off, pos, total;
  uint64_t control_off = 0, rtable_off = 0x1C, span_off = 0x1C + rt_len, content_off = span_off + 8;

  /* section 0 holds: ControlData, ResetTable, SpanInfo, Content (in this order) */
  sec0_len = (size_t) content_off + lzx_len + 16;
  for (i = 0; i < n_files; i++) { ents[i].name = files[i].name; ents[i].section = 1; ents[i].offset = files[i].offset; ents[i].length = files[i].length; }
  ents[n_files + 0].name = n_content;  ents[n_files + 0].offset = content_off; ents[n_files + 0].length = lzx_len;
  ents[n_files + 1].name = n_control;  ents[n_files + 1].offset = control_off; ents[n_files + 1].length = 0x1C;
  ents[n_files + 2].name = n_spaninfo; ents[n_files + 2].offset = span_off;    ents[n_files + 2].length = 8;
  ents[n_files + 3].name = n_rtable;   ents[n_files + 3].offset = rtable_off;  ents[n_files + 3].length = rt_len;
  qsort(ents, (size_t) n_ent, sizeof(*ents), ci_cmp);

  /* PMGL chunks */
  chunks = (uint8_t *) calloc(((size_t) n_ent / 8 + 2) * 2 + 4, CHUNK);
  {
    int e = 0;
    while (e < n_ent) {
      uint8_t *c = chunks + (size_t) n_chunks * CHUNK;
      size_t p = 0x14;
      int cnt = 0;
      uint16_t qr[1024]; int nqr = 0;
      memcpy(c, "PMGL", 4);
      while (e < n_ent) {
        size_t nl = strlen(ents[e].name), need = nl + 2 + 1 + 10 + 10;
        size_t qr_bytes = 2 + 2 * (size_t)((cnt + 1 + 4) / 5);
        if (p + need + qr_bytes + 8 > CHUNK) break;
        if (cnt && (cnt % 5) == 0) qr[nqr++] = (uint16_t)(p - 0x14);
        p += put_encint(c + p, nl);
        memcpy(c + p, ents[e].name, nl); p += nl;
        p += put_encint(c + p, ents[e].section);
        p += put_encint(c + p, ents[e].offset);
        p += put_encint(c + p, ents[e].length);
        cnt++; e++;
      }
      w16(c + CHUNK - 2, (unsigned) cnt);
      for (i = 0; i < nqr; i++) w16(c + CHUNK - 2 - 2 * (size_t)(i + 1), qr[i]);
      w32(c + 4, (uint32_t)(2 + 2 * (size_t) nqr));          /* quick-ref area size */
      w32(c + 8, 0);
      w32(c + 0x0C, n_chunks ? (uint32_t)(n_chunks - 1) : 0xFFFFFFFFu);
      w32(c + 0x10, 0xFFFFFFFFu);                              /* patched below */
      if (n_chunks) w32(chunks + (size_t)(n_chunks - 1) * CHUNK + 0x10, (uint32_t) n_chunks);
      n_chunks++;
    }
  }
  /* PMGI index levels over more than one PMGL chunk (chmd.c:560-598, 700-840): an entry is
   * (first name of a child chunk, child chunk number); levels are stacked until one chunk remains */
  {
    int lvl_first = 0, lvl_count = n_chunks;
    n_pmgl = n_chunks;
    while (lvl_count > 1) {
      int child = lvl_first, next_first = n_chunks;
      while (child < lvl_first + lvl_count) {
        uint8_t *c = chunks + (size_t) n_chunks * CHUNK;
        size_t p = 8;
        int cnt = 0;
        uint16_t qr[1024]; int nqr = 0;
        memcpy(c, "PMGI", 4);
        while (child < lvl_first + lvl_count) {
          /* first name of the child: its first entry (PMGL at 0x14, PMGI at 8) */
          const uint8_t *cc = chunks + (size_t) child * CHUNK;
          const uint8_t *q = cc + (cc[3] == 'L' ? 0x14 : 8);
          size_t nl = 0, k = 0;
          do { nl = (nl << 7) | (q[k] & 0x7F); } while (q[k++] & 0x80);
          size_t need = nl + 2 + 5, qr_bytes = 2 + 2 * (size_t)((cnt + 1 + 4) / 5);
          if (p + need + qr_bytes + 8 > CHUNK) break;
          if (cnt && (cnt % 5) == 0) qr[nqr++] = (uint16_t)(p - 8);
          p += put_encint(c + p, nl);
          memcpy(c + p, q + k, nl); p += nl;
          p += put_encint(c + p, (uint64_t) child);
          cnt++; child++;
        }
        w16(c + CHUNK - 2, (unsigned) cnt);
        for (i = 0; i < nqr; i++) w16(c + CHUNK - 2 - 2 * (size_t)(i + 1), qr[i]);
        w32(c + 4, (uint32_t)(2 + 2 * (size_t) nqr));
        n_chunks++;
      }
      lvl_first = next_first; lvl_count = n_chunks - next_first;
      depth++;
    }
    if (depth > 1) index_root = (uint32_t)(n_chunks - 1);
  }
  sec0_off = dir_off + (size_t) n_chunks * CHUNK;
  total = sec0_off + sec0_len;
  if (total > cap) { free(ents); free(chunks); return 0; }
  memset(dst, 0, total);
  /* ITSF */
  memcpy(dst, "ITSF", 4); w32(dst + 4, 3); w32(dst + 8, 0x60); w32(dst + 0x0C, 1);
  dst[0x10] = 0x12; dst[0x11] = 0x34; dst[0x12] = 0x56; dst[0x13] = 0x78;   /* timestamp, big-endian */
  w32(dst + 0x14, 0x409); memcpy(dst + 0x18, guids, 32);
  w64(dst + 0x38, 0x60); w64(dst + 0x40, 0x18); w64(dst + 0x48, 0x78); w64(dst + 0x50, 0x54 + (uint64_t) n_chunks * CHUNK);
  w64(dst + 0x58, sec0_off);
  /* header section 0 */
  w32(dst + 0x60, 0x1FE); w32(dst + 0x64, 0); w64(dst + 0x68, total); w32(dst + 0x70, 0); w32(dst + 0x74, 0);
  /* header section 1 = ITSP + chunks */
  pos = 0x78;
  memcpy(dst + pos, "ITSP", 4); w32(dst + pos + 4, 1); w32(dst + pos + 8, 0x54); w32(dst + pos + 0x0C, 0x0A);
  w32(dst + pos + 0x10, (uint32_t) CHUNK); w32(dst + pos + 0x14, 2); w32(dst + pos + 0x18, (uint32_t) depth);
  w32(dst + pos + 0x1C, index_root); w32(dst + pos + 0x20, 0); w32(dst + pos + 0x24, (uint32_t)(n_pmgl - 1));
  w32(dst + pos + 0x28, 0xFFFFFFFFu); w32(dst + pos + 0x2C, (uint32_t) n_chunks); w32(dst + pos + 0x30, 0x409);
  memcpy(dst + pos + 0x34, itsp_guid, 16); w32(dst + pos + 0x44, 0x54);
  w32(dst + pos + 0x48, 0xFFFFFFFFu); w32(dst + pos + 0x4C, 0xFFFFFFFFu); w32(dst + pos + 0x50, 0xFFFFFFFFu);
  memcpy(dst + dir_off, chunks, (size_t) n_chunks * CHUNK);
  /* section 0 */
  pos = sec0_off;
  w32(dst + pos, 6); memcpy(dst + pos + 4, "LZXC", 4); w32(dst + pos + 8, 2);
  w32(dst + pos + 0x0C, (uint32_t) reset_frames); w32(dst + pos + 0x10, (uint32_t)((1u << window_bits) / 32768u));
  w32(dst + pos + 0x14, (uint32_t) reset_frames); w32(dst + pos + 0x18, 0);
  pos = sec0_off + rtable_off;
  w32(dst + pos, 2); w32(dst + pos + 4, (uint32_t) n_frames); w32(dst + pos + 8, 8); w32(dst + pos + 0x0C, 0x28);
  w64(dst + pos + 0x10, uncomp_len); w64(dst + pos + 0x18, lzx_len); w64(dst + pos + 0x20, 0x8000);
  for (i = 0; (size_t) i < n_frames; i++) w64(dst + pos + 0x28 + 8 * (size_t) i, frame_off[i]);
  w64(dst + sec0_off + span_off, uncomp_len);
  memcpy(dst + sec0_off + content_off, lzx, lzx_len);
  free(ents); free(chunks);
  return total;
}
