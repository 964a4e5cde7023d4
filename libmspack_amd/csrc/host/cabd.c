/* cabd.c -- CAB driver of the libmspack-compatible API (include/mspack.h), on the GPU batch decoder.
 *
 * Mirrors the behaviour of the reference's cabd.c for the path open -> extract:
 *   header / folder / file parsing and its error codes ...... cabd.c:317-628
 *   extract() argument checks, in the reference's order ....... cabd.c:1075-1125
 *   CFDATA reader: sizes, reserve areas, checksum .............. cabd.c:1362-1479
 *   codec dispatch by comp_type, Quantum 0xFF trailer, LZX total length .. cabd.c:1226-1269,1327-1340
 *   READ errors of a codec are reported as the feeder's error .. cabd.c:1198,1205
 * but NOT its control flow: the reference streams one folder through one codec instance and
 * re-decodes from block 0 whenever an earlier file is requested (cabd.c:1142-1146).  Here the first
 * extract() that touches a cabinet gathers EVERY folder of that cabinet into one batch (one unit per
 * folder), decodes it on the GPU with mspack_hip_decode_batch[_multi], keeps the decoded folders,
 * and every extract() is then a slice of that result.  A unit's result tells how far it decoded
 * without error; files that end inside that prefix succeed exactly as they do in the reference
 * (which never decodes further than asked), later ones return the unit's error.
 * Cabinet sets: append()/prepend() join cabinets and merge a folder that
 * continues across them (reference cabd.c:870-1064); a merged folder is a list of (cabinet, offset)
 * segments, and a CFDATA block with uncompressed size 0 continues as the first block of the next
 * segment (cabd.c:1432-1455).  search() finds embedded cabinets with the reference's acceptance
 * rules (cabd.c:656-868).
 */
#include <stdlib.h>
#include <stdio.h>
#include "host_common.h"

#define CAB_BLOCKMAX   32768u
#define CAB_INPUTMAX   (CAB_BLOCKMAX + 6144u)
#define CAB_INPUTMAX_SALVAGE 65535u
#define CAB_LENGTHMAX  (CAB_BLOCKMAX * 65535u)
#define CAB_FOLDERMAX  65535u

struct cab_p;
struct out_store { unsigned char *base; size_t refs; };   /* one batch's output arena, kept as the folders' decoded bytes */
struct fseg {                         /* where a folder's CFDATA blocks live: one entry per cabinet */
  struct fseg *next;
  struct cab_p *cab;
  off_t offset;                       /* first CFDATA of this folder in that cabinet               */
};
struct folder_p {
  struct mscabd_folder base;
  struct fseg data;
  struct mscabd_file *merge_prev, *merge_next;
  /* decoded state */
  int decoded;
  unsigned char *dec;                 /* decoded bytes (good prefix valid): inside `store`, the output arena of the batch that
                                         decoded the folder -- the folders of a batch share it, the last one to go frees it */
  struct out_store *store;
  unsigned int total;                 /* sum of the blocks' uncompressed sizes                     */
  unsigned int good_len;              /* bytes that decode without error                           */
  unsigned int written;               /* bytes the codec handed to sys->write when asked for the whole folder (result.out_len):
                                         for Quantum less than good_len -- it writes when its window wraps, qtmd.c:420-428 */
  unsigned int n_frames_good;         /* LZX: complete frames in good_len                          */
  int dec_err;                        /* MSPACK_ERR_* of the unit                                  */
  int read_err;                       /* what the feeder would have reported for ERR_READ          */
  int hard_eof;                       /* the block chain ended with a read failure, not cleanly    */
  int cksum_on_host;                  /* a block of this folder failed its checksum on the device (below): gather it again, checksums
                                         verified while the blocks are read -- the chain then ends AT the bad block as the reference's does */
  unsigned int res_flags;
  /* MSZIP repair mode (MSCABD_PARAM_FIXMSZIP): the blocks the codec repaired, in stream order: rep[2 i] = output offset of
   * the block, rep[2 i + 1] = bytes lost (what mszipd tells sys->message, mszipd.c:427) */
  unsigned int rep_n;
  unsigned int *rep;
  /* the same for the feeder's "bad block checksum" warnings of such a folder (checksums are ignored in repair mode,
   * cabd.c:1408-1421): rep_ck[i] = the output offset at which the reference has read block i's header and complained --
   * the block whose decoding pulls the input chunk (MSCABD_PARAM_DECOMPBUF bytes) that block i starts in */
  unsigned int ck_n;
  unsigned int *rep_ck;
  /* Quantum: the request boundaries the folder's files imply (their first bytes and the bytes behind their last), ascending, and
   * what qtmd holds back at each -- the rest of the match that covers the request's last byte, which the NEXT call writes before it
   * decodes anything (qtmd.c:268-276) -- as the batch reported it (MSPACK_HIP_UF_QTM_MARKS; mark_log lies inside `store`) */
  unsigned int n_marks;
  unsigned int *marks;
  const unsigned char *mark_log;
  /* (decode_cabinet: how many of the cabinet's files lie in this folder and the first of them in list order -- one pass over the
   * file list for all folders, so that a folder's files are found without walking the whole list per folder) */
  unsigned int file_count;
  struct mscabd_file *first_file;
  /* the folder's batch while it is still running (mspack_hip.h: jobs): unit job_k of it -- not decoded yet, and not to be gathered
   * again; folder_settle() waits for the unit and takes its result over */
  struct cab_batch *job;
  size_t job_k;
};
struct cab_p {
  struct mscabd_cabinet base;
  int block_resv;
};
/* the CFDATA feeder of one folder (the subset of the reference's mscabd_decompress_state, cab.h:95-110,
 * that cabd_sys_read_block works on) */
/* CFDATA checksums verified ON THE DEVICE (cabd.c:1411-1417): the folders' payloads go to HBM for decoding
 * anyway, so the gather only notes every block part that carries a checksum -- where its payload lies in the input arena and
 * what the XOR of its dwords must be (the stored checksum with the header's cbData / cbUncomp word folded back out) -- and the batch
 * carries one MSPACK_HIP_KIND_XORSUM unit per part.  A part that fails makes its folder be gathered again the reference's way
 * (folder_p.cksum_on_host): rare, and then exact.  Not used where checksums are ignored and only talked about (salvage mode,
 * MSZIP repair mode): when the warning is said depends on the read order there. */
struct ck_part { size_t off; uint32_t len, want; unsigned int owner; };
struct ck_list { struct ck_part *p; size_t n, cap; int failed; };
struct blk_reader {
  struct ck_list *defer;              /* note the parts' checksums here instead of verifying them (gather only)          */
  size_t defer_base; unsigned int defer_owner;   /* arena offset of input[0]; index of the folder in the batch          */
  int quiet_cksum, bad_cksum;          /* do not say "bad block checksum" now: the caller notes it (bad_cksum) and says it later */
  struct folder_p *folder;
  struct fseg *seg;                   /* cabinet the next block header is read from                */
  struct mspack_file *fh;
  unsigned int block;                 /* blocks started                                            */
  unsigned char *input;               /* one (possibly reassembled) block                          */
  int borrowed;                       /* `input` is the caller's (gather: a place in the input arena), not reader_open's */
  unsigned int i_ptr, i_end;
};
struct cabd_p {
  struct mscab_decompressor base;
  struct mspack_system *system;
  int error, read_error;
  int searchbuf_size, fix_mszip, buf_size, salvage;
  int devices, cache_mb;
  /* stored (uncompressed) folders need no codec and are streamed exactly like the reference does it,
   * including what a later extract() sees after a failed one (cabd.c:1283-1345, 1530-1541) */
  struct blk_reader st;
  unsigned int st_offset;             /* bytes produced so far from st.folder                      */
  int st_active;
  struct folder_p *last_folder;       /* folder of the previous extract()                          */
  /* the reference's ONE decompressor as far as its answers go (cabd.c:1136-1175): it lives on across extract() calls while the
   * files of a folder are asked for at ascending offsets -- also after a call that FAILED: its codec then answers every further
   * call with the same error (lzxd.c / mszipd.c / qtmd.c: "if (x->error) return x->error") and writes nothing -- and it starts
   * over for another folder or for an offset below what it has written so far */
  struct folder_p *live_folder;
  unsigned int live_offset;           /* bytes the reference's decompressor has handed to cabd_sys_write (cabd.c:1347-1354) */
  int live_failed, live_err;          /* it is in its error state; what extract() returns from then on (READ already mapped) */
  /* the reference's decompressor as far as its MESSAGES go: which folder it is on, how far it has decoded (cabd.c:1136-1175:
   * another folder or an earlier offset starts it again from the folder's first block, and it says everything again) */
  struct folder_p *msg_folder;
  unsigned int msg_offset, msg_next, msg_next_ck;
};

static void stored_reset(struct cabd_p *self);

/* ---- small helpers -------------------------------------------------------------------------------- */
static unsigned int cab_checksum(const unsigned char *data, unsigned int bytes, unsigned int cksum) {
  unsigned int n = bytes >> 2, tail = 0;
#if defined(__BYTE_ORDER__) && __BYTE_ORDER__ == __ORDER_LITTLE_ENDIAN__
  {
    /* (the XOR of the dwords: four independent 64-bit lanes, folded at the end) */
    uint64_t a = 0, b = 0, c = 0, d = 0, w[4];
    while (n >= 8) {
      memcpy(w, data, 32);
      a ^= w[0]; b ^= w[1]; c ^= w[2]; d ^= w[3];
      data += 32; n -= 8;
    }
    a ^= b ^ c ^ d;
    cksum ^= (unsigned int) a ^ (unsigned int)(a >> 32);
  }
#endif
  while (n--) { cksum ^= rd_le32(data); data += 4; }
  switch (bytes & 3) {
  case 3: tail |= (unsigned int) *data++ << 16; /* fall through */
  case 2: tail |= (unsigned int) *data++ << 8;  /* fall through */
  case 1: tail |= *data;
  }
  return cksum ^ tail;
}

static char *read_cstring(struct mspack_system *sys, struct mspack_file *fh, int permit_empty, int *error) {
  off_t base = sys->tell(fh);
  char buf[256], *str;
  int len = sys->read(fh, buf, 256), i;
  if (len <= 0) { *error = MSPACK_ERR_READ; return NULL; }
  for (i = 0; i < len && buf[i]; i++) ;
  if (i == len || (i == 0 && !permit_empty)) { *error = MSPACK_ERR_DATAFORMAT; return NULL; }
  if (sys->seek(fh, base + (off_t)(i + 1), MSPACK_SYS_SEEK_START)) { *error = MSPACK_ERR_SEEK; return NULL; }
  if (!(str = (char *) sys->alloc(sys, (size_t) i + 1))) { *error = MSPACK_ERR_NOMEMORY; return NULL; }
  sys->copy(buf, str, (size_t) i + 1);
  *error = MSPACK_ERR_OK;
  return str;
}

static void batch_abandon(struct mspack_system *sys, struct cab_batch *B);
static void free_folder_cache(struct mspack_system *sys, struct folder_p *f) {
  if (f->job) batch_abandon(sys, f->job);                 /* (its batch is still running: to its end, nothing of it kept) */
  if (f->store && --f->store->refs == 0) { mspack_arena_free(sys, f->store->base); sys->free(f->store); }
  f->store = NULL; f->dec = NULL; f->decoded = 0;
  sys->free(f->rep); f->rep = NULL; f->rep_n = 0;
  sys->free(f->rep_ck); f->rep_ck = NULL; f->ck_n = 0;
  sys->free(f->marks); f->marks = NULL; f->n_marks = 0; f->mark_log = NULL;
}

/* ---- headers (reference cabd.c:317-628) ------------------------------------------------------------- */
static int read_files(struct mspack_system *sys, struct mspack_file *fh, struct cab_p *cab,
                      int num_folders, int num_files, int salvage)
{
  struct mscabd_file *tail = NULL;
  struct mscabd_folder *cur = NULL;
  unsigned int cur_idx = 0;
  unsigned char buf[16];
  int i, err;
  for (i = 0; i < num_files; i++) {
    struct mscabd_file *f;
    unsigned int fidx, x;
    if (sys->read(fh, buf, 16) != 16) return MSPACK_ERR_READ;
    if (!(f = (struct mscabd_file *) sys->alloc(sys, sizeof(*f)))) return MSPACK_ERR_NOMEMORY;
    f->next = NULL;
    f->length = rd_le32(buf);
    f->offset = rd_le32(buf + 4);
    fidx = rd_le16(buf + 8);
    f->attribs = (int) rd_le16(buf + 14);
    f->folder = NULL;
    if (fidx < 0xFFFD) {
      if ((int) fidx < num_folders) {
        /* (files come sorted by folder: walk on from the previous file's folder, not from the head of the list) */
        if (!cur || fidx < cur_idx) { cur = cab->base.folders; cur_idx = 0; }
        while (cur_idx < fidx && cur) { cur = cur->next; cur_idx++; }
        f->folder = cur;
      }
    }
    else {
      struct folder_p *fp;
      if (fidx == 0xFFFE || fidx == 0xFFFF) {            /* continued to next: lives in the last folder */
        struct mscabd_folder *fo = cab->base.folders;
        while (fo->next) fo = fo->next;
        f->folder = fo; fp = (struct folder_p *) fo;
        if (!fp->merge_next) fp->merge_next = f;
      }
      if (fidx == 0xFFFD || fidx == 0xFFFF) {            /* continued from previous: first folder       */
        f->folder = cab->base.folders; fp = (struct folder_p *) f->folder;
        if (!fp->merge_prev) fp->merge_prev = f;
      }
    }
    x = rd_le16(buf + 12);
    f->time_h = (char)(x >> 11); f->time_m = (char)((x >> 5) & 0x3F); f->time_s = (char)((x << 1) & 0x3E);
    x = rd_le16(buf + 10);
    f->date_d = (char)(x & 0x1F); f->date_m = (char)((x >> 5) & 0xF); f->date_y = (int)(x >> 9) + 1980;
    f->filename = read_cstring(sys, fh, 0, &err);
    if (err || !f->folder) {
      sys->free(f->filename); sys->free(f);
      if (salvage) continue;
      return err ? err : MSPACK_ERR_DATAFORMAT;
    }
    if (tail) tail->next = f; else cab->base.files = f;
    tail = f;
  }
  return MSPACK_ERR_OK;
}

static int read_headers(struct mspack_system *sys, struct mspack_file *fh, struct cab_p *cab,
                        off_t offset, int salvage, int quiet)
{
  unsigned char buf[64];
  int num_folders, num_files, folder_resv = 0, i, err;
  struct folder_p *fol = NULL, *tail = NULL;
  off_t files_off_hdr, files_off_real;

  cab->base.next = NULL; cab->base.files = NULL; cab->base.folders = NULL;
  cab->base.prevcab = cab->base.nextcab = NULL;
  cab->base.prevname = cab->base.nextname = cab->base.previnfo = cab->base.nextinfo = NULL;
  cab->base.base_offset = offset;
  cab->block_resv = 0;

  if (sys->seek(fh, offset, MSPACK_SYS_SEEK_START)) return MSPACK_ERR_SEEK;
  if (sys->read(fh, buf, 0x24) != 0x24) return MSPACK_ERR_READ;
  if (rd_le32(buf) != 0x4643534Du) return MSPACK_ERR_SIGNATURE;
  cab->base.length = rd_le32(buf + 0x08);
  cab->base.set_id = (unsigned short) rd_le16(buf + 0x20);
  cab->base.set_index = (unsigned short) rd_le16(buf + 0x22);
  files_off_hdr = (off_t) rd_le32(buf + 0x10);
  num_folders = (int) rd_le16(buf + 0x1A);
  if (num_folders == 0) { if (!quiet) sys->message(fh, "no folders in cabinet."); return MSPACK_ERR_DATAFORMAT; }
  num_files = (int) rd_le16(buf + 0x1C);
  if (num_files == 0) { if (!quiet) sys->message(fh, "no files in cabinet."); return MSPACK_ERR_DATAFORMAT; }
  if (buf[0x19] != 1 && buf[0x18] != 3) { if (!quiet) sys->message(fh, "WARNING; cabinet version is not 1.3"); }
  cab->base.flags = (int) rd_le16(buf + 0x1E);
  cab->base.header_resv = 0;
  if (cab->base.flags & MSCAB_HDR_RESV) {
    if (sys->read(fh, buf, 4) != 4) return MSPACK_ERR_READ;
    cab->base.header_resv = (unsigned short) rd_le16(buf);
    folder_resv = buf[2];
    cab->block_resv = buf[3];
    if (cab->base.header_resv > 60000 && !quiet) sys->message(fh, "WARNING; reserved header > 60000.");
    if (cab->base.header_resv && sys->seek(fh, (off_t) cab->base.header_resv, MSPACK_SYS_SEEK_CUR)) return MSPACK_ERR_SEEK;
  }
  if (cab->base.flags & MSCAB_HDR_PREVCAB) {
    cab->base.prevname = read_cstring(sys, fh, 0, &err); if (err) return err;
    cab->base.previnfo = read_cstring(sys, fh, 1, &err); if (err) return err;
  }
  if (cab->base.flags & MSCAB_HDR_NEXTCAB) {
    cab->base.nextname = read_cstring(sys, fh, 0, &err); if (err) return err;
    cab->base.nextinfo = read_cstring(sys, fh, 1, &err); if (err) return err;
  }
  for (i = 0; i < num_folders; i++) {
    if (sys->read(fh, buf, 8) != 8) return MSPACK_ERR_READ;
    if (folder_resv && sys->seek(fh, (off_t) folder_resv, MSPACK_SYS_SEEK_CUR)) return MSPACK_ERR_SEEK;
    if (!(fol = (struct folder_p *) sys->alloc(sys, sizeof(*fol)))) return MSPACK_ERR_NOMEMORY;
    memset(fol, 0, sizeof(*fol));
    fol->base.comp_type = (int) rd_le16(buf + 6);
    fol->base.num_blocks = rd_le16(buf + 4);
    fol->data.next = NULL;
    fol->data.cab = cab;
    fol->data.offset = offset + (off_t) rd_le32(buf);
    if (tail) tail->base.next = &fol->base; else cab->base.folders = &fol->base;
    tail = fol;
  }
  files_off_real = sys->tell(fh) - cab->base.base_offset;
  err = read_files(sys, fh, cab, num_folders, num_files, salvage);
  if (files_off_real != files_off_hdr) {
    if (!quiet) sys->message(fh, "WARNING; atypical files offset in header");
    if (salvage && files_off_hdr < (off_t) cab->base.length &&
        !sys->seek(fh, files_off_hdr + cab->base.base_offset, MSPACK_SYS_SEEK_START)) {
      struct mscabd_file *first = cab->base.files, *second;
      int err2 = read_files(sys, fh, cab, num_folders, num_files, salvage);
      second = cab->base.files;
      if (first && first != second) {
        struct mscabd_file *e = first;
        while (e->next) e = e->next;
        e->next = second; cab->base.files = first;
      }
      err = err ? err : err2;
    }
  }
  if (err) {
    if (salvage && cab->base.files) { if (!quiet) sys->message(fh, "WARNING; ignoring error %d while salvaging", err); }
    else return err;
  }
  if (!cab->base.files) return MSPACK_ERR_DATAFORMAT;
  return MSPACK_ERR_OK;
}

/* ---- public methods ------------------------------------------------------------------------------------ */
static void free_cab_strings(struct mspack_system *sys, struct mscabd_cabinet *c) {
  sys->free(c->prevname); sys->free(c->nextname); sys->free(c->previnfo); sys->free(c->nextinfo);
}

/* frees a cabinet, every cabinet joined to it by append/prepend (they share ONE files/folders list)
 * and every cabinet linked through ->next (reference cabd.c:240-307) */
static void cabd_close(struct mscab_decompressor *base, struct mscabd_cabinet *origcab)
{
  struct cabd_p *self = (struct cabd_p *) base;
  struct mspack_system *sys;
  if (!self) return;
  sys = self->system;
  self->error = MSPACK_ERR_OK;
  while (origcab) {
    struct mscabd_file *fi, *nfi;
    struct mscabd_folder *fo, *nfo;
    struct mscabd_cabinet *c, *nc, *nextc = origcab->next;
    for (fi = origcab->files; fi; fi = nfi) { nfi = fi->next; sys->free(fi->filename); sys->free(fi); }
    for (fo = origcab->folders; fo; fo = nfo) {
      struct fseg *sg, *nsg;
      nfo = fo->next;
      if (self->st_active && self->st.folder == (struct folder_p *) fo) stored_reset(self);
      if (self->last_folder == (struct folder_p *) fo) self->last_folder = NULL;
      if (self->live_folder == (struct folder_p *) fo) self->live_folder = NULL;
      if (self->msg_folder == (struct folder_p *) fo) self->msg_folder = NULL;
      free_folder_cache(sys, (struct folder_p *) fo);
      for (sg = ((struct folder_p *) fo)->data.next; sg; sg = nsg) { nsg = sg->next; sys->free(sg); }
      sys->free(fo);
    }
    for (c = origcab->prevcab; c; c = nc) { nc = c->prevcab; free_cab_strings(sys, c); sys->free(c); }
    for (c = origcab->nextcab; c; c = nc) { nc = c->nextcab; free_cab_strings(sys, c); sys->free(c); }
    free_cab_strings(sys, origcab);
    sys->free(origcab);
    origcab = nextc;
  }
}

static struct mscabd_cabinet *cabd_open(struct mscab_decompressor *base, const char *filename)
{
  struct cabd_p *self = (struct cabd_p *) base;
  struct mspack_system *sys;
  struct mspack_file *fh;
  struct cab_p *cab = NULL;
  if (!self) return NULL;
  sys = self->system;
  if (!(fh = sys->open(sys, filename, MSPACK_SYS_OPEN_READ))) { self->error = MSPACK_ERR_OPEN; return NULL; }
  if ((cab = (struct cab_p *) sys->alloc(sys, sizeof(*cab)))) {
    int err;
    memset(cab, 0, sizeof(*cab));
    cab->base.filename = filename;
    err = read_headers(sys, fh, cab, (off_t) 0, self->salvage, 0);
    if (err) { cabd_close(base, &cab->base); cab = NULL; }
    self->error = err;
  }
  else self->error = MSPACK_ERR_NOMEMORY;
  sys->close(fh);
  return (struct mscabd_cabinet *) cab;
}

/* search: scan a file for embedded cabinets (reference cabd.c:656-868).  A candidate is every "MSCF"
 * followed by 16 more header bytes; it is tried when its files offset lies inside its claimed length
 * and both stay within 32 bytes of the end of the file (salvage: the length may be garbage).  A
 * candidate that parses restarts the scan after its claimed length, one that does not restarts it
 * right after the signature.  The scan works on searchbuf_size pieces with a byte-wise state machine,
 * so signatures may straddle pieces. */
static int cabd_find(struct cabd_p *self, unsigned char *buf, struct mspack_file *fh, const char *filename,
                     off_t flen, off_t *firstlen, struct mscabd_cabinet **first)
{
  struct mspack_system *sys = self->system;
  struct mscabd_cabinet *link = NULL;
  off_t offset, length;
  unsigned int cablen = 0, foffset = 0;
  int state = 0;

  for (offset = 0; offset < flen; offset += length) {
    off_t i;
    length = flen - offset;
    if (length > (off_t) self->searchbuf_size) length = (off_t) self->searchbuf_size;
    if (sys->read(fh, buf, (int) length) != (int) length) return MSPACK_ERR_READ;
    if (offset == 0 && length >= 4 && rd_le32(buf) == 0x28635349u)
      sys->message(fh, "WARNING; found InstallShield header. Use unshield "
                       "(https://github.com/twogood/unshield) to unpack this file");
    for (i = 0; i < length; ) {
      unsigned char c = buf[i];
      if (state == 0) {                                   /* hunt for 'M' */
        while (i < length && buf[i] != 0x4D) i++;
        if (i < length) { i++; state = 1; }
        continue;
      }
      i++;
      switch (state) {
      case 1: state = (c == 0x53) ? 2 : 0; break;
      case 2: state = (c == 0x43) ? 3 : 0; break;
      case 3: state = (c == 0x46) ? 4 : 0; break;
      case 8:  cablen  = c;                       state++; break;
      case 9:  cablen |= (unsigned int) c << 8;   state++; break;
      case 10: cablen |= (unsigned int) c << 16;  state++; break;
      case 11: cablen |= (unsigned int) c << 24;  state++; break;
      case 16: foffset  = c;                      state++; break;
      case 17: foffset |= (unsigned int) c << 8;  state++; break;
      case 18: foffset |= (unsigned int) c << 16; state++; break;
      case 19: {
        off_t caboff = offset + i - 20;
        foffset |= (unsigned int) c << 24;
        offset = caboff + 4;                              /* where to go on if this is no cabinet */
        if (caboff == 0) *firstlen = (off_t) cablen;
        if (foffset < cablen && (caboff + (off_t) foffset) < (flen + 32) &&
            ((caboff + (off_t) cablen) < (flen + 32) || self->salvage)) {
          struct cab_p *cab = (struct cab_p *) sys->alloc(sys, sizeof(*cab));
          if (!cab) return MSPACK_ERR_NOMEMORY;
          memset(cab, 0, sizeof(*cab));
          cab->base.filename = filename;
          if (read_headers(sys, fh, cab, caboff, self->salvage, caboff > 0)) {
            cabd_close(&self->base, &cab->base);
          }
          else {
            if (!link) *first = &cab->base; else link->next = &cab->base;
            link = &cab->base;
            offset = caboff + (off_t) cablen;
          }
        }
        if (offset >= flen) return MSPACK_ERR_OK;
        if (sys->seek(fh, offset, MSPACK_SYS_SEEK_START)) return MSPACK_ERR_SEEK;
        length = 0; i = 0;                                /* leaves the piece loop; refill at `offset` */
        state = 0;
        break;
      }
      default: state++; break;                            /* bytes 4-7 and 12-15 are not looked at */
      }
    }
  }
  return MSPACK_ERR_OK;
}

static struct mscabd_cabinet *cabd_search(struct mscab_decompressor *base, const char *filename)
{
  struct cabd_p *self = (struct cabd_p *) base;
  struct mspack_system *sys;
  struct mspack_file *fh;
  struct mscabd_cabinet *cab = NULL;
  unsigned char *buf;
  off_t flen = 0, firstlen = 0;
  if (!self) return NULL;
  sys = self->system;
  if (!(buf = (unsigned char *) sys->alloc(sys, (size_t) self->searchbuf_size))) { self->error = MSPACK_ERR_NOMEMORY; return NULL; }
  if ((fh = sys->open(sys, filename, MSPACK_SYS_OPEN_READ))) {
    if (!(self->error = mspack_sys_filelen(sys, fh, &flen)))
      self->error = cabd_find(self, buf, fh, filename, flen, &firstlen, &cab);
    if (firstlen && firstlen != flen && (!cab || cab->base_offset == 0)) {
      if (firstlen < flen) sys->message(fh, "WARNING; possible %ld extra bytes at end of file.", (long)(flen - firstlen));
      else sys->message(fh, "WARNING; file possibly truncated by %ld bytes.", (long)(firstlen - flen));
    }
    sys->close(fh);
  }
  else self->error = MSPACK_ERR_OPEN;
  sys->free(buf);
  return cab;
}

/* ---- cabinet sets: append / prepend (reference cabd.c:870-1064) ----------------------------------------- */
/* may the last folder of the left cabinet be continued by the first folder of the right one? */
static int can_merge_folders(struct mspack_system *sys, struct folder_p *lfol, struct folder_p *rfol)
{
  struct mscabd_file *l, *r;
  int some = 0;
  if (lfol->base.comp_type != rfol->base.comp_type) return 0;
  if (lfol->base.num_blocks + rfol->base.num_blocks > CAB_FOLDERMAX) return 0;
  if (!lfol->merge_next || !rfol->merge_prev) return 0;
  /* the files continued out of the left folder should open the right folder, same order, same
   * offsets and lengths */
  for (l = lfol->merge_next, r = rfol->merge_prev; l; l = l->next, r = r->next)
    if (!r || l->offset != r->offset || l->length != r->length) break;
  if (!l) return 1;
  /* otherwise accept as soon as ONE continued file is listed on both sides, and name the others */
  for (l = lfol->merge_next; l; l = l->next) {
    for (r = rfol->merge_prev; r; r = r->next)
      if (l->offset == r->offset && l->length == r->length) break;
    if (r) some = 1;
    else sys->message(NULL, "WARNING; merged file %s not listed in both cabinets", l->filename);
  }
  return some;
}

static int cabd_merge(struct mscab_decompressor *base, struct mscabd_cabinet *lcab, struct mscabd_cabinet *rcab)
{
  struct cabd_p *self = (struct cabd_p *) base;
  struct mspack_system *sys;
  struct mscabd_cabinet *c;
  struct folder_p *lfol, *rfol;
  struct mscabd_file *fi;
  if (!self) return MSPACK_ERR_ARGS;
  sys = self->system;
  if (!lcab || !rcab || lcab == rcab) return self->error = MSPACK_ERR_ARGS;
  if (lcab->nextcab || rcab->prevcab) return self->error = MSPACK_ERR_ARGS;       /* already joined */
  for (c = lcab->prevcab; c; c = c->prevcab) if (c == rcab) return self->error = MSPACK_ERR_ARGS;
  for (c = rcab->nextcab; c; c = c->nextcab) if (c == lcab) return self->error = MSPACK_ERR_ARGS;
  if (lcab->set_id != rcab->set_id) sys->message(NULL, "WARNING; merged cabinets with differing Set IDs.");
  if (lcab->set_index > rcab->set_index) sys->message(NULL, "WARNING; merged cabinets with odd order.");

  lfol = (struct folder_p *) lcab->folders;
  while (lfol->base.next) lfol = (struct folder_p *) lfol->base.next;
  rfol = (struct folder_p *) rcab->folders;

  if (!lfol->merge_next && !rfol->merge_prev) {
    /* nothing continues across the join: chain the cabinets, folders and files */
    lcab->nextcab = rcab; rcab->prevcab = lcab;
    lfol->base.next = &rfol->base;
    for (fi = lcab->files; fi->next; fi = fi->next) ;
    fi->next = rcab->files;
  }
  else {
    struct fseg *seg, *tail;
    struct mscabd_file *prev, *nfi;
    struct folder_p *last;
    if (!can_merge_folders(sys, lfol, rfol)) return self->error = MSPACK_ERR_DATAFORMAT;
    if (!(seg = (struct fseg *) sys->alloc(sys, sizeof(*seg)))) return self->error = MSPACK_ERR_NOMEMORY;
    lcab->nextcab = rcab; rcab->prevcab = lcab;
    /* the right folder's segments continue the left folder; the split block is counted by both */
    for (tail = &lfol->data; tail->next; tail = tail->next) ;
    *seg = rfol->data; tail->next = seg; rfol->data.next = NULL;
    lfol->base.num_blocks += rfol->base.num_blocks - 1;
    /* the merged folder is continued further only by files of the right cabinet that live in
     * ANOTHER folder of it (a right folder that is itself both continued-from and continued-to keeps
     * the left side's list: its own entries are about to be deleted) */
    if (!rfol->merge_next || rfol->merge_next->folder != &rfol->base) lfol->merge_next = rfol->merge_next;
    free_folder_cache(sys, lfol);                       /* anything decoded before the join is stale */
    for (last = lfol; last->base.next; last = (struct folder_p *) last->base.next) ;
    last->base.next = rfol->base.next;
    for (fi = lcab->files; fi->next; fi = fi->next) ;
    fi->next = rcab->files;
    /* the right cabinet's copies of the continued files go away with its folder */
    for (prev = NULL, fi = lcab->files; fi; fi = nfi) {
      nfi = fi->next;
      if (fi->folder == &rfol->base) {
        if (prev) prev->next = nfi; else lcab->files = nfi;
        sys->free(fi->filename); sys->free(fi);
      }
      else prev = fi;
    }
    free_folder_cache(sys, rfol);
    if (self->st_active && (self->st.folder == rfol || self->st.folder == lfol)) stored_reset(self);
    if (self->last_folder == rfol) self->last_folder = NULL;
    if (self->live_folder == rfol || self->live_folder == lfol) self->live_folder = NULL;     /* (the merged folder is decoded anew) */
    if (self->msg_folder == rfol || self->msg_folder == lfol) self->msg_folder = NULL;       /* (and says everything again) */
    sys->free(rfol);
  }
  /* every cabinet of the set shows the same lists */
  for (c = lcab->prevcab; c; c = c->prevcab) { c->files = lcab->files; c->folders = lcab->folders; }
  for (c = lcab->nextcab; c; c = c->nextcab) { c->files = lcab->files; c->folders = lcab->folders; }
  return self->error = MSPACK_ERR_OK;
}

static int cabd_append(struct mscab_decompressor *base, struct mscabd_cabinet *cab, struct mscabd_cabinet *nextcab) {
  return cabd_merge(base, cab, nextcab);
}
static int cabd_prepend(struct mscab_decompressor *base, struct mscabd_cabinet *cab, struct mscabd_cabinet *prevcab) {
  return cabd_merge(base, prevcab, cab);
}

/* ---- the CFDATA block reader (reference cabd.c:1362-1459) ----------------------------------------------- */
#define CAB_INPUTBUF (CAB_INPUTMAX_SALVAGE + 8u)

static void reader_close(struct cabd_p *self, struct blk_reader *r) {
  if (r->fh) self->system->close(r->fh);
  if (!r->borrowed) self->system->free(r->input);
  memset(r, 0, sizeof(*r));
}

/* start reading a folder's blocks: MSPACK_ERR_OPEN / SEEK / NOMEMORY as extract() reports them */
static int reader_open(struct cabd_p *self, struct blk_reader *r, struct folder_p *fol, int borrowed) {
  struct mspack_system *sys = self->system;
  memset(r, 0, sizeof(*r));
  r->folder = fol; r->seg = &fol->data; r->borrowed = borrowed;
  if (!borrowed && !(r->input = (unsigned char *) sys->alloc(sys, CAB_INPUTBUF))) return MSPACK_ERR_NOMEMORY;
  if (!(r->fh = sys->open(sys, r->seg->cab->base.filename, MSPACK_SYS_OPEN_READ))) { reader_close(self, r); return MSPACK_ERR_OPEN; }
  if (sys->seek(r->fh, r->seg->offset, MSPACK_SYS_SEEK_START)) { reader_close(self, r); return MSPACK_ERR_SEEK; }
  return MSPACK_ERR_OK;
}

/* read one block into r->input[0, i_end), reassembling a block that is split over the cabinets of a set
 * (a part with uncompressed size 0 continues as the first block of the next cabinet).  On failure the
 * parts read so far STAY in the buffer, as in the reference. */
static int reader_block(struct cabd_p *self, struct blk_reader *r, unsigned int *ulen_out,
                        int ignore_cksum, int ignore_size)
{
  struct mspack_system *sys = self->system;
  r->i_ptr = r->i_end = 0;
  for (;;) {
    unsigned char hdr[8];
    unsigned int len, ulen, cksum, full;
    if (sys->read(r->fh, hdr, 8) != 8) return MSPACK_ERR_READ;
    if (r->seg->cab->block_resv && sys->seek(r->fh, (off_t) r->seg->cab->block_resv, MSPACK_SYS_SEEK_CUR)) return MSPACK_ERR_SEEK;
    len = rd_le16(hdr + 4); ulen = rd_le16(hdr + 6);
    full = r->i_end + len;
    if (full > CAB_INPUTMAX && (!ignore_size || full > CAB_INPUTMAX_SALVAGE)) return MSPACK_ERR_DATAFORMAT;
    if (ulen > CAB_BLOCKMAX && !ignore_size) return MSPACK_ERR_DATAFORMAT;
    if (sys->read(r->fh, r->input + r->i_end, (int) len) != (int) len) return MSPACK_ERR_READ;
    if ((cksum = rd_le32(hdr)) && r->defer && !ignore_cksum) {
      struct ck_list *L = r->defer;
      if (L->n == L->cap) {
        const size_t ncap = L->cap ? L->cap * 2 : 1024;
        struct ck_part *np = (struct ck_part *) sys->alloc(sys, ncap * sizeof(*np));
        if (np) { if (L->n) sys->copy(L->p, np, L->n * sizeof(*np)); sys->free(L->p); L->p = np; L->cap = ncap; }
        else L->failed = 1;
      }
      if (L->n < L->cap) {
        L->p[L->n].off = r->defer_base + r->i_end; L->p[L->n].len = len;
        L->p[L->n].want = cksum ^ rd_le32(hdr + 4);       /* (cab_checksum over one whole dword is an XOR) */
        L->p[L->n].owner = r->defer_owner; L->n++;
        cksum = 0;                                        /* (verified on the device, with the batch) */
      }
      /* else: the list could not grow -- THIS part is verified right here (it used to go unverified, and the
       * parts recorded so far were thrown away with it) */
    }
    if (cksum) {                                          /* every part carries its own checksum */
      unsigned int sum = cab_checksum(r->input + r->i_end, len, 0);
      if (cab_checksum(hdr + 4, 4, sum) != cksum) {
        if (!ignore_cksum) return MSPACK_ERR_CHECKSUM;
        if (r->quiet_cksum) r->bad_cksum = 1;
        else sys->message(r->fh, "WARNING; bad block checksum found");
      }
    }
    r->i_end += len;
    if (ulen) { *ulen_out = ulen; return MSPACK_ERR_OK; }
    sys->close(r->fh); r->fh = NULL;
    if (!(r->seg = r->seg->next)) {
      sys->message(NULL, "WARNING; ran out of cabinets in set. Are any missing?");
      return MSPACK_ERR_DATAFORMAT;
    }
    if (!(r->fh = sys->open(sys, r->seg->cab->base.filename, MSPACK_SYS_OPEN_READ))) return MSPACK_ERR_OPEN;
    if (sys->seek(r->fh, r->seg->offset, MSPACK_SYS_SEEK_START)) return MSPACK_ERR_SEEK;
  }
}

/* ---- stored folders: streamed (reference cabd.c:1283-1345 + noned_decompress 1530-1541) --------------- */
static void stored_reset(struct cabd_p *self) {
  if (self->st_active) reader_close(self, &self->st);
  self->st_active = 0; self->st_offset = 0;
}

/* the feeder as the "codec" sees it: returns the bytes delivered, -1 after a block error */
static int stored_read(struct cabd_p *self, unsigned char *buf, int bytes) {
  struct blk_reader *r = &self->st;
  int todo = bytes;
  while (todo > 0) {
    unsigned int avail = r->i_end - r->i_ptr;
    if (avail) {
      if (avail > (unsigned int) todo) avail = (unsigned int) todo;
      self->system->copy(r->input + r->i_ptr, buf, avail);
      r->i_ptr += avail; buf += avail; todo -= (int) avail;
    }
    else {
      unsigned int ulen;
      if (r->block++ >= r->folder->base.num_blocks) {     /* out of blocks */
        if (!self->salvage) self->read_error = MSPACK_ERR_DATAFORMAT;
        break;
      }
      if (!r->fh) { self->read_error = MSPACK_ERR_READ; return -1; }   /* the chain already broke */
      self->read_error = reader_block(self, r, &ulen, self->salvage, self->salvage);
      if (self->read_error) return -1;
    }
  }
  return bytes - todo;
}

/* produce `bytes` more bytes of the folder; out == NULL skips (the offset still advances) */
static int stored_run(struct cabd_p *self, unsigned int bytes, struct mspack_file *out, unsigned char *buf) {
  while (bytes > 0) {
    int run = (bytes > (unsigned int) self->buf_size) ? self->buf_size : (int) bytes;
    if (stored_read(self, buf, run) != run) return MSPACK_ERR_READ;
    self->st_offset += (unsigned int) run;
    if (out && self->system->write(out, buf, run) != run) return MSPACK_ERR_WRITE;
    bytes -= (unsigned int) run;
  }
  return MSPACK_ERR_OK;
}

static int stored_extract(struct cabd_p *self, struct folder_p *fol, struct mscabd_file *file,
                          unsigned int filelen, const char *filename)
{
  struct mspack_system *sys = self->system;
  struct mspack_file *fh;
  unsigned char *buf;
  if (!self->st_active || self->st.folder != fol || self->st_offset > file->offset) {
    int err;
    stored_reset(self);
    if ((err = reader_open(self, &self->st, fol, 0))) return self->error = err;
    self->st_active = 1; self->st_offset = 0;
    self->read_error = MSPACK_ERR_OK;                     /* lasts for the lifetime of a decompressor */
  }
  if (!(fh = sys->open(sys, filename, MSPACK_SYS_OPEN_WRITE))) return self->error = MSPACK_ERR_OPEN;
  self->error = MSPACK_ERR_OK;
  if (filelen) {
    if (!(buf = (unsigned char *) sys->alloc(sys, (size_t) self->buf_size))) self->error = MSPACK_ERR_NOMEMORY;
    else {
      int err = MSPACK_ERR_OK;
      if (file->offset > self->st_offset) err = stored_run(self, file->offset - self->st_offset, NULL, buf);
      self->error = (err == MSPACK_ERR_READ) ? self->read_error : err;
      if (!self->error) {
        err = stored_run(self, filelen, fh, buf);
        self->error = (err == MSPACK_ERR_READ) ? self->read_error : err;
      }
      sys->free(buf);
    }
  }
  sys->close(fh);
  return self->error;
}

/* ---- gather + batch decode ------------------------------------------------------------------------------ */
/* the batch's input arena: the folders' CFDATA payloads are READ straight into it, one folder behind the other (no buffer
 * per folder, no second copy); it grows by doubling */
struct in_arena { unsigned char *p; size_t len, cap; };

static int arena_room(struct mspack_system *sys, struct in_arena *a, size_t need)
{
  size_t ncap;
  unsigned char *n;
  if (a->len + need <= a->cap) return 1;
  ncap = a->cap * 2;
  if (ncap < a->len + need) ncap = a->len + need;
  if (!(n = (unsigned char *) mspack_arena_alloc(sys, ncap))) return 0;
  if (a->len) sys->copy(a->p, n, a->len);
  mspack_arena_free(sys, a->p);
  a->p = n; a->cap = ncap;
  return 1;
}

struct gathered {
  struct folder_p *fol;
  size_t in_off, len;                           /* codec input in the arena: payloads (+0xFF per block for Quantum), then 64
                                                   zero bytes */
  size_t tab_off;                               /* frames_ok: the frame table (uint32 per block) in the arena */
  unsigned int total;                           /* sum of uncompressed sizes of the blocks read     */
  int read_err; int hard_eof;
  uint32_t *boff; unsigned int nblk;            /* where every block's payload starts in the stream; frames_ok: every
                                                   block but the last holds exactly one 32 KiB frame  */
  int frames_ok;
  size_t marks_off; unsigned int n_marks;       /* Quantum: the folder's request boundaries as a table in the arena (gather_marks) */
  unsigned int *marks;
};

/* The request boundaries a Quantum folder's files imply -- cabd_extract asks its codec for the bytes in front of a file, then for
 * the file (cabd.c:1195-1218): requests end where files begin and where they end.  Ascending, without duplicates, inside
 * (0, total): appended to the arena as a uint32 table for MSPACK_HIP_UF_QTM_MARKS (mspack_hip.h).  Without memory for it the
 * folder decodes without marks (what a failing call leaves of its predecessor's last match is then not known: nothing). */
static int cmp_u32(const void *a, const void *b) { const unsigned int x = *(const unsigned int *) a, y = *(const unsigned int *) b; return x < y ? -1 : x > y; }
static void gather_marks(struct mspack_system *sys, struct cab_p *cab, struct gathered *g, struct in_arena *A)
{
  struct mscabd_file *f;
  size_t nf = 0, m = 0, i;
  unsigned int *v;
  g->n_marks = 0; g->marks = NULL; g->marks_off = 0;
  (void) cab;
  nf = g->fol->file_count;                     /* (counted by decode_cabinet in one pass over the list) */
  if (!nf || !(v = (unsigned int *) sys->alloc(sys, 2 * nf * sizeof(unsigned int)))) return;
  for (f = g->fol->first_file, i = 0; f && i < nf; f = f->next) {
    if ((struct folder_p *) f->folder != g->fol) continue;
    i++;
    if (f->offset > 0 && f->offset < g->total) v[m++] = f->offset;
    if (f->length < g->total && f->offset < g->total - f->length && f->offset + f->length > 0) v[m++] = f->offset + f->length;
  }
  qsort(v, m, sizeof(unsigned int), cmp_u32);
  for (i = 0, nf = 0; i < m; i++) if (!nf || v[i] != v[nf - 1]) v[nf++] = v[i];
  m = nf;
  if (!m || !arena_room(sys, A, 4 * m + 8)) { sys->free(v); return; }
  A->len = (A->len + 3) & ~(size_t) 3;
  memcpy(A->p + A->len, v, 4 * m);
  g->marks_off = A->len; g->n_marks = (unsigned int) m; g->marks = v;
  A->len += 4 * m;
}

/* walk the CFDATA chain of one folder (reference cabd.c:1283-1345 + 1362-1459), following it through
 * the cabinets of a set.  Returns MSPACK_ERR_OPEN / SEEK / NOMEMORY when nothing could be started (the arena is as it
 * was); every later failure ends the chain and is recorded as the feeder's error (g->read_err, g->hard_eof). */
static int gather_folder_once(struct cabd_p *self, struct gathered *g, struct in_arena *A, struct ck_list *ck, unsigned int owner)
{
  struct mspack_system *sys = self->system;
  struct folder_p *fol = g->fol;
  struct blk_reader r;
  const int method = fol->base.comp_type & 0x0F;
  const int ignore_cksum = self->salvage || (self->fix_mszip && method == MSCAB_COMP_MSZIP);
  const int ignore_size = self->salvage;
  const size_t len0 = A->len;
  uint32_t *qoff = NULL;               /* Quantum: where every block starts in the folder's stream (LZX / MSZIP: g->boff) */
  unsigned int qn = 0;
  unsigned int *bad = NULL, bad_n = 0, bad_cap = 0;   /* blocks whose checksum was ignored: indices here, published to the
                                                         folder as output offsets once the chain has been walked */
  int err;
  /* (a folder that is gathered again -- the batch it was in failed -- starts its list afresh) */
  sys->free(fol->rep_ck); fol->rep_ck = NULL; fol->ck_n = 0;
  g->len = 0; g->total = 0; g->read_err = MSPACK_ERR_OK; g->hard_eof = 0;
  g->nblk = 0; g->frames_ok = 1; g->tab_off = 0;
  if (!arena_room(sys, A, 16 + 64 + 32)) return MSPACK_ERR_NOMEMORY;
  while (A->len & 15) A->p[A->len++] = 0;
  g->in_off = A->len;
  g->boff = (method == MSCAB_COMP_LZX || method == MSCAB_COMP_MSZIP)
          ? (uint32_t *) sys->alloc(sys, ((size_t) fol->base.num_blocks + 1) * sizeof(uint32_t)) : NULL;
  if ((err = reader_open(self, &r, fol, 1))) {
    if (err != MSPACK_ERR_SEEK) { sys->free(g->boff); g->boff = NULL; A->len = len0; return err; }
    /* the reference fails extract() with SEEK before any decoding; keep it as this folder's error */
    memset(A->p + A->len, 0, 64); A->len += 64;
    g->read_err = MSPACK_ERR_SEEK; g->hard_eof = 1; g->frames_ok = 0;
    return MSPACK_ERR_OK;
  }
  /* ignored checksums (salvage mode; MSZIP repair mode) are complained about when the reference READS the block -- in the
   * extract() call whose decoding gets there, not when this driver gathers the cabinet's folders (rep_ck below) */
  if (ignore_cksum && !g->boff) qoff = (uint32_t *) sys->alloc(sys, ((size_t) fol->base.num_blocks + 1) * sizeof(uint32_t));
  r.quiet_cksum = ignore_cksum && (g->boff != NULL || qoff != NULL);
  while (r.block < fol->base.num_blocks) {
    unsigned int ulen = 0;
    r.block++;
    r.bad_cksum = 0;
    /* (a block, reassembled from the cabinets of a set or not, is at most CAB_INPUTBUF bytes: reader_block) */
    if (!arena_room(sys, A, (size_t) CAB_INPUTBUF + 1 + 64 + 16)) { reader_close(self, &r); sys->free(g->boff); g->boff = NULL; sys->free(qoff); sys->free(bad); A->len = len0; return MSPACK_ERR_NOMEMORY; }
    r.input = A->p + A->len;
    r.defer = (ck && !ignore_cksum && !fol->cksum_on_host && !ck->failed) ? ck : NULL;
    r.defer_base = A->len; r.defer_owner = owner;
    if ((err = reader_block(self, &r, &ulen, ignore_cksum, ignore_size))) { g->read_err = err; g->hard_eof = 1; break; }
    if (r.bad_cksum) {
      /* said when the reference would say it (cabd_extract): remember the block */
      if (bad_n == bad_cap) {
        const unsigned int ncap = bad_cap ? bad_cap * 2 : 16;
        unsigned int *nw = (unsigned int *) sys->alloc(sys, (size_t) ncap * sizeof(unsigned int));
        if (nw) {
          if (bad_n) sys->copy(bad, nw, (size_t) bad_n * sizeof(unsigned int));
          sys->free(bad); bad = nw; bad_cap = ncap;
        }
      }
      if (bad_n < bad_cap) bad[bad_n++] = g->boff ? g->nblk : qn;
    }
    if (qoff) qoff[qn++] = (uint32_t)(A->len - g->in_off);
    if (g->boff) {
      if (g->total % CAB_BLOCKMAX) g->frames_ok = 0;          /* an earlier block was not a whole frame */
      g->boff[g->nblk++] = (uint32_t)(A->len - g->in_off);
    }
    A->len += r.i_end;
    if (method == MSCAB_COMP_QUANTUM) A->p[A->len++] = 0xFF;
    g->total += ulen;
  }
  reader_close(self, &r);
  g->len = A->len - g->in_off;
  if (bad_n && (g->boff || qoff)) {
    /* block i is read when the codec's refill reaches the input chunk it starts in; that refill happens while the block
     * that holds the chunk's first byte is being decoded (every block but the last decodes to 32 KiB) */
    const size_t q = (size_t)(self->buf_size > 0 ? self->buf_size : 4096);
    const uint32_t *const off = g->boff ? g->boff : qoff;
    unsigned int i;
    for (i = 0; i < bad_n; i++) {
      const unsigned int b = bad[i];
      const size_t chunk_start = ((size_t) off[b] / q) * q;
      unsigned int t = b;
      while (t > 0 && (size_t) off[t] > chunk_start) t--;
      bad[i] = t * CAB_BLOCKMAX;
      if (method == MSCAB_COMP_QUANTUM && t < b) {
        /* lzxd and mszipd decode a whole frame / block before they hand any of it over: its first byte asked for is enough.
         * qtmd decodes as far as it is asked: WHERE in block t -- when it has used the block's input up to the chunk, taken
         * as the same share of the block's output (an estimate: which extract() call says it can be off when a file ends
         * right there) */
        const size_t clen = (size_t) off[t + 1] - off[t];
        const size_t ulen = g->total - (size_t) t * CAB_BLOCKMAX < CAB_BLOCKMAX ? g->total - (size_t) t * CAB_BLOCKMAX : CAB_BLOCKMAX;
        if (clen) bad[i] += (unsigned int)((chunk_start - off[t]) * ulen / clen);
      }
    }
    fol->rep_ck = bad; fol->ck_n = bad_n; bad = NULL;
  }
  sys->free(bad);
  sys->free(qoff);
  if (!g->hard_eof) g->read_err = self->salvage ? MSPACK_ERR_OK : MSPACK_ERR_DATAFORMAT;  /* ran out of blocks */
  else {
    /* the codec pulls buf_size bytes per read; a read that reaches the bad block fails as a whole,
     * so everything from the start of that read on is lost (cabd.c:1297-1324) */
    size_t q = (size_t)((self->buf_size + 1) & ~1);
    g->len -= g->len % q;
  }
  A->len = g->in_off + g->len;
  memset(A->p + A->len, 0, 64 + 16); A->len += 64;           /* (room for both: arena_room above / at the top) */
  /* LZX: every CFDATA block is one frame (cabd.c:1362-1479); MSZIP: every block is a deflate stream of its own
   * (mszipd.c:406-418).  The block sizes are the folder's frame table: the blocks' tokens are parsed by one
   * wavefront each (MSPACK_HIP_UF_FRAME_TABLE) before the folder's wavefront commits them */
  g->frames_ok = g->frames_ok && g->boff && !g->hard_eof &&
                 (method == MSCAB_COMP_LZX || (method == MSCAB_COMP_MSZIP && !self->fix_mszip)) &&
                 g->nblk >= 1 && (size_t) g->nblk * CAB_BLOCKMAX >= g->total;     /* (one block too: the lane parser beats the serial one) */
  if (g->frames_ok) {
    A->len = (A->len + 3) & ~(size_t) 3;
    if (!arena_room(sys, A, (size_t) g->nblk * 4 + 16)) g->frames_ok = 0;       /* (decodes without the table) */
    else { g->tab_off = A->len; memcpy(A->p + A->len, g->boff, (size_t) g->nblk * 4); A->len += (size_t) g->nblk * 4; }
  }
  return MSPACK_ERR_OK;
}

static int gather_folder(struct cabd_p *self, struct gathered *g, struct in_arena *A, struct ck_list *ck, unsigned int owner)
{
  const size_t ck_mark = ck ? ck->n : 0, len0 = A->len;
  int err = gather_folder_once(self, g, A, ck, owner);
  if (!err && g->hard_eof && ck && ck->n > ck_mark) {
    /* the chain broke behind blocks whose checksums nobody has verified yet (they were left to the device), and the input is cut
     * at the failing read: those parts no longer lie in the arena as they were read.  Once more, verifying while the blocks are
     * read -- what the reference does: a block whose checksum fails ends the chain first. */
    ck->n = ck_mark; A->len = len0;
    self->system->free(g->boff); g->boff = NULL;
    g->fol->cksum_on_host = 1;
    err = gather_folder_once(self, g, A, ck, owner);
  }
  return err;
}

/* one batch of a cabinet's folders: what it reads (units, the input arena with the block parts' checksums noted in it), what it
 * writes (results, the output arena = the folders' decoded bytes afterwards) and, while it runs as a job, the job */
struct cab_batch {
  struct mspack_system *sys;
  struct gathered *gs; mspack_hip_unit *units; mspack_hip_result *res; struct ck_list ck; struct in_arena A;
  unsigned char *out_arena; struct out_store *store;
  size_t n, nu, left;                  /* folders, units (folders + checksum parts), folders that have not settled yet */
  size_t *ck_lo;                       /* n + 1 entries: folder k's checksum parts are ck.p[ck_lo[k] .. ck_lo[k + 1]) -- the gather notes
                                          them folder by folder; NULL: not in that order (or no memory), every part is looked at */
  int pinned;
  mspack_hip_job *job;
};

/* folder k of the batch is through: its result and its place in the output arena become the folder's decoded state */
static void batch_take_folder(struct cab_batch *B, size_t k)
{
  struct mspack_system *sys = B->sys;
  struct gathered *gs = B->gs;
  const mspack_hip_unit *units = B->units;
  const mspack_hip_result *res = B->res;
  unsigned char *const out_arena = B->out_arena;
  struct folder_p *fp = gs[k].fol;
  int method = fp->base.comp_type & 0x0F;
  fp->total = gs[k].total; fp->read_err = gs[k].read_err; fp->hard_eof = gs[k].hard_eof;
  fp->store = B->store; B->store->refs++;
  fp->dec = out_arena + units[k].out_off;
  if (method >= 1 && method <= 3) {
    unsigned int g = res[k].good_len > gs[k].total ? gs[k].total : res[k].good_len;
    fp->good_len = g; fp->dec_err = res[k].err; fp->res_flags = res[k].flags;
    fp->written = res[k].out_len > gs[k].total ? gs[k].total : res[k].out_len;
    if (method == MSCAB_COMP_MSZIP && res[k].err == MSPACK_ERR_OK && !(units[k].flags & MSPACK_HIP_UF_MSZIP_REPAIR) &&
        res[k].in_next && res[k].in_next <= CAB_BLOCKMAX && res[k].out_len == gs[k].total) {
      /* mszipd never reads a CFDATA header's uncompressed size: a block is as long as its deflate stream (mszipd.c:377-460), and
       * what the folder's last block inflated to beyond the headers' sum is there for the files that ask for it (the unit's
       * slack holds those bytes, in_next says how many: DESIGN.md section 8g) */
      fp->total += res[k].in_next; fp->good_len = fp->total; fp->written = fp->total;
    }
    sys->free(fp->marks); fp->marks = NULL; fp->n_marks = 0; fp->mark_log = NULL;
    if (units[k].flags & MSPACK_HIP_UF_QTM_MARKS) {
      fp->marks = gs[k].marks; fp->n_marks = gs[k].n_marks; gs[k].marks = NULL;
      fp->mark_log = out_arena + units[k].out_off + (((size_t) gs[k].total + 15) & ~(size_t) 15);
    }
    if (units[k].flags & MSPACK_HIP_UF_MSZIP_LOG) {
      const unsigned char *lg = out_arena + units[k].out_off + (((size_t) gs[k].total + 32768 + 15) & ~(size_t) 15);
      unsigned int cnt = rd_le32(lg), i;
      if (cnt > (unsigned int) units[k].e8_base) cnt = (unsigned int) units[k].e8_base;
      sys->free(fp->rep); fp->rep = NULL; fp->rep_n = 0;
      if (cnt && (fp->rep = (unsigned int *) sys->alloc(sys, (size_t) cnt * 2 * sizeof(unsigned int)))) {
        for (i = 0; i < 2 * cnt; i++) fp->rep[i] = rd_le32(lg + 4 + 4 * (size_t) i);
        fp->rep_n = cnt;
      }
    }
  }
  else { fp->good_len = 0; fp->written = 0; fp->dec_err = MSPACK_ERR_DATAFORMAT; fp->res_flags = 0; }   /* cabd.c:1254 */
  fp->n_frames_good = fp->good_len / CAB_BLOCKMAX;
  fp->decoded = 1;
}

/* decode every not-yet-decoded folder of `cab` (budget permitting, `want` always) in ONE batch */
static int decode_cabinet(struct cabd_p *self, struct cab_p *cab, struct folder_p *want)
{
  struct mspack_system *sys = self->system;
  struct mscabd_folder *fo;
  struct gathered *gs;
  mspack_hip_unit *units;
  mspack_hip_result *res;
  struct in_arena A = { NULL, 0, 0 };
  struct ck_list ck = { NULL, 0, 0, 0 };
  struct cab_batch B;
  unsigned char *out_arena = NULL;
  size_t n = 0, k, out_bytes = 0, budget = (size_t) self->cache_mb << 20, used = 0, nu;
  int err = MSPACK_ERR_OK, rc, again = 0;

  size_t n_qtm_files = 0;                               /* files in Quantum folders: two marks each at most (gather_marks) */
  for (fo = cab->base.folders; fo; fo = fo->next) { n++; ((struct folder_p *) fo)->file_count = 0; ((struct folder_p *) fo)->first_file = NULL; }
  {
    struct mscabd_file *fi;
    for (fi = cab->base.files; fi; fi = fi->next) {
      struct folder_p *fp = (struct folder_p *) fi->folder;
      if (fp && !fp->file_count++) fp->first_file = fi;
      if (fp && (fp->base.comp_type & 0x0F) == MSCAB_COMP_QUANTUM) n_qtm_files++;
    }
  }
  gs = (struct gathered *) sys->alloc(sys, n * sizeof(*gs));
  units = NULL; res = NULL;                              /* (sized once the gather knows how many checksum units ride along) */
  /* (first guess for the arena: the cabinet's stated length, within reason -- it grows when that was wrong or the folders
   *  go on in other cabinets) */
  A.cap = (size_t) cab->base.length;
  if (A.cap > ((size_t) 256 << 20)) A.cap = (size_t) 256 << 20;
  /* (room for the Quantum folders' tables of marks too: an arena that has to grow is allocated anew -- page-locked -- and copied:
   * 170 ms for config 4's 190 MB when the tables' 131 KB did not fit the first guess) */
  A.cap += n * 96 + 65536 + 8 * n_qtm_files + 16 * n;
  A.p = (unsigned char *) mspack_arena_alloc(sys, A.cap);
  if (!gs || !A.p) { sys->free(gs); mspack_arena_free(sys, A.p); return MSPACK_ERR_NOMEMORY; }
  n = 0;
  for (fo = cab->base.folders; fo; fo = fo->next) {
    struct folder_p *fp = (struct folder_p *) fo;
    size_t est = (size_t) fo->num_blocks * CAB_BLOCKMAX;
    if (fp->decoded || fp->job) continue;
    if ((fo->comp_type & 0x0F) == MSCAB_COMP_NONE || fp->merge_prev) continue;   /* streamed / not extractable */
    if (fp != want && used + est > budget) continue;
    gs[n].fol = fp;
    {
      const size_t ck_mark = ck.n;
      err = gather_folder(self, &gs[n], &A, &ck, (unsigned int) n);
      if (err) ck.n = ck_mark;                             /* (the arena was rolled back: so are the parts noted in it) */
    }
    if (err == MSPACK_ERR_OPEN && fp != want) { err = MSPACK_ERR_OK; continue; }   /* that folder stays undecoded */
    if (err) break;
    gs[n].n_marks = 0; gs[n].marks = NULL; gs[n].marks_off = 0;
    if ((fo->comp_type & 0x0F) == MSCAB_COMP_QUANTUM) gather_marks(sys, cab, &gs[n], &A);
    used += est;
    n++;
  }
  if (!err && !arena_room(sys, &A, 64)) err = MSPACK_ERR_NOMEMORY;
  /* (a list that could not grow: the parts it holds are still verified with the batch, the others were verified while they were read) */
  nu = n + ck.n;
  if (!err) {
    units = (mspack_hip_unit *) sys->alloc(sys, (nu ? nu : 1) * sizeof(*units));
    res = (mspack_hip_result *) sys->alloc(sys, (nu ? nu : 1) * sizeof(*res));
    if (!units || !res) err = MSPACK_ERR_NOMEMORY;
  }
  if (err) { for (k = 0; k < n; k++) { sys->free(gs[k].boff); sys->free(gs[k].marks); } sys->free(gs); sys->free(units); sys->free(res); sys->free(ck.p); mspack_arena_free(sys, A.p); return err; }
  memset(A.p + A.len, 0, 64);

  /* the units: where gather_folder put their input, one stretch of the output arena each */
  memset(units, 0, nu * sizeof(*units));
  for (k = 0; k < ck.n; k++) {                             /* the block parts' checksums (mspack_hip.h: MSPACK_HIP_KIND_XORSUM) */
    units[n + k].kind = MSPACK_HIP_KIND_XORSUM;
    units[n + k].in_off = ck.p[k].off; units[n + k].in_len = ck.p[k].len;
  }
  for (k = 0; k < n; k++) {
    struct folder_p *fp = gs[k].fol;
    int method = fp->base.comp_type & 0x0F;
    units[k].in_off = gs[k].in_off; units[k].in_len = (uint32_t) gs[k].len;
    if (gs[k].frames_ok) units[k].in_chunk = (uint32_t)(gs[k].tab_off / 4);
    units[k].out_off = out_bytes; units[k].out_len = gs[k].total;
    out_bytes += ((size_t) gs[k].total + 32768 + 15) & ~(size_t) 15;
    if (self->fix_mszip && method == MSCAB_COMP_MSZIP) {        /* the repair log behind the unit's slack (mspack_hip.h) */
      units[k].e8_base = (int32_t)(fp->base.num_blocks + 8u);
      out_bytes += (4u + 8u * (size_t) units[k].e8_base + 15u) & ~(size_t) 15;
    }
    units[k].kind = (uint8_t)((method >= 1 && method <= 3) ? method : 0);   /* 0: no codec (cabd.c:1254), skipped */
    units[k].window_bits = (uint8_t)((fp->base.comp_type >> 8) & 0x1F);
    units[k].reset_frames = 0;
    units[k].flags = (gs[k].hard_eof ? MSPACK_HIP_UF_HARD_EOF : 0) |
                     ((self->fix_mszip && method == MSCAB_COMP_MSZIP) ? (MSPACK_HIP_UF_MSZIP_REPAIR | MSPACK_HIP_UF_MSZIP_LOG) : 0) |
                     (gs[k].frames_ok ? MSPACK_HIP_UF_FRAME_TABLE : 0);
    if (!gs[k].frames_ok) units[k].in_chunk = (uint32_t)((self->buf_size + 1) & ~1);    /* mszipd.c:348 */
    if (method == MSCAB_COMP_QUANTUM && gs[k].n_marks) {        /* the folder's request boundaries; their log behind the output */
      units[k].flags |= MSPACK_HIP_UF_QTM_MARKS;
      units[k].in_chunk = (uint32_t)(gs[k].marks_off / 4); units[k].ref_len = gs[k].n_marks;
      if (4 * (size_t) gs[k].n_marks + 16 > 32768) out_bytes += (4 * (size_t) gs[k].n_marks + 15) & ~(size_t) 15;
    }
  }
  out_arena = (unsigned char *) mspack_arena_alloc(sys, out_bytes + 64);
  if (!out_arena) err = MSPACK_ERR_NOMEMORY;
  else {
    size_t nhip = 0;
    for (k = 0; k < n; k++) if (units[k].kind != 0) nhip++;
    memset(res, 0, nu * sizeof(*res));
    B.sys = sys; B.gs = gs; B.units = units; B.res = res; B.ck = ck; B.A = A; B.out_arena = out_arena; B.n = n; B.nu = nu;
    B.store = NULL; B.job = NULL; B.left = 0; B.pinned = 0; B.ck_lo = NULL;
    if (nhip || ck.n) {                  /* (checksum units alone are a batch too: stored folders' parts -- ADVICE round 5) */
      /* the arena was written a moment ago: page-locked, its copy to the device is plain DMA (mspack_hip.h; advice only) */
      const int pinned = A.len >= ((size_t) 4 << 20) && !mspack_arena_is_locked(A.p) && mspack_hip_pin(A.p, mspack_arena_room(A.len + 64)) == 0;
      /* Several folders on one device: the batch runs as a job (mspack_hip.h) -- this extract() returns when ITS folder is through,
       * the caller writes the file while the other folders are decoded and copied back, and every later extract() waits for its
       * own folder only (folder_settle).  Everything the batch reads and writes belongs to it until then: struct cab_batch. */
      struct cab_batch *J = NULL;
      if (n >= 2 && self->devices <= 1 && (J = (struct cab_batch *) sys->alloc(sys, sizeof(*J)))) {
        *J = B; J->pinned = pinned;
        if ((J->store = (struct out_store *) sys->alloc(sys, sizeof(*J->store)))) {
          J->store->base = out_arena; J->store->refs = 1;             /* (the batch's own hold on the arena) */
          J->job = mspack_hip_decode_batch_begin(units, nu, A.p, A.len + 64, out_arena, out_bytes + 64, res);
        }
        if (!J->job) { sys->free(J->store); sys->free(J); J = NULL; }
      }
      if (J && ck.n && (J->ck_lo = (size_t *) sys->alloc(sys, (n + 1) * sizeof(size_t)))) {
        /* (a folder's parts by index: looking at all parts for every folder was 4096 x 4096 steps for config 2 -- 7 ms of 24) */
        size_t j = 0, f;
        int sorted = 1;
        for (f = 0; f <= n; f++) {
          J->ck_lo[f] = j;
          while (j < ck.n && ck.p[j].owner == (unsigned int) f) j++;
        }
        if (j != ck.n) sorted = 0;
        if (!sorted) { sys->free(J->ck_lo); J->ck_lo = NULL; }
      }
      if (J) {
        for (k = 0; k < n; k++) { gs[k].fol->job = J; gs[k].fol->job_k = k; }
        J->left = n;
        return MSPACK_ERR_OK;                                          /* (folder_settle takes it from here) */
      }
      /* kinds other than 1..3 are answered with MSPACK_ERR_ARGS by the kernels; fix them up below */
      rc = (self->devices > 1)
        ? mspack_hip_decode_batch_multi(units, nu, A.p, A.len + 64, out_arena, out_bytes + 64, res, self->devices)
        : mspack_hip_decode_batch(units, nu, A.p, A.len + 64, out_arena, out_bytes + 64, res);
      if (pinned) mspack_hip_unpin(A.p);
      if (rc) {
        sys->message(NULL, "GPU batch decode failed: %s", mspack_hip_last_error());
        err = MSPACK_ERR_DECRUNCH;
      }
    }
  }
  if (!err) {
    /* a block part whose payload is not what its header says: that folder is gathered again, the reference's way */
    for (k = 0; k < ck.n; k++)
      if (res[n + k].err != MSPACK_ERR_OK || res[n + k].in_next != ck.p[k].want) { gs[ck.p[k].owner].fol->cksum_on_host = 1; again = 1; }
  }
  if (!err && n) {
    /* the output arena stays: every folder's decoded bytes are where the batch put them (no copy per folder) */
    if (!(B.store = (struct out_store *) sys->alloc(sys, sizeof(*B.store)))) err = MSPACK_ERR_NOMEMORY;
    else { B.store->base = out_arena; B.store->refs = 0; }
    for (k = 0; k < n && !err; k++) {
      struct folder_p *fp = gs[k].fol;
      if (fp->cksum_on_host && ck.n) {                     /* (flagged just now?  It was decoded past a bad block: not kept) */
        size_t j; int mine = 0;
        for (j = 0; j < ck.n && !mine; j++) mine = ck.p[j].owner == (unsigned int) k;
        if (mine) continue;
      }
      batch_take_folder(&B, k);
    }
    if (B.store && B.store->refs) out_arena = NULL;        /* the folders own it now */
    else if (B.store) sys->free(B.store);
  }
  for (k = 0; k < n; k++) { sys->free(gs[k].boff); sys->free(gs[k].marks); }
  sys->free(gs); sys->free(units); sys->free(res); sys->free(ck.p); mspack_arena_free(sys, A.p); mspack_arena_free(sys, out_arena);
  /* (the folders flagged above defer nothing the second time: one more round at most) */
  if (!err && again) return decode_cabinet(self, cab, want);
  return err;
}

/* ---- a cabinet's batch as a job ---- */
/* the batch to its end and everything it held given back; the folders that have not taken their results stay undecoded */
static void batch_release(struct mspack_system *sys, struct cab_batch *B)
{
  size_t k;
  if (B->job) { (void) mspack_hip_job_end(B->job); B->job = NULL; }
  if (B->pinned) mspack_hip_unpin(B->A.p);
  for (k = 0; k < B->n; k++) {
    if (B->gs[k].fol) { B->gs[k].fol->job = NULL; B->gs[k].fol = NULL; }
    sys->free(B->gs[k].boff); sys->free(B->gs[k].marks);
  }
  sys->free(B->gs); sys->free(B->units); sys->free(B->res); sys->free(B->ck.p); sys->free(B->ck_lo); mspack_arena_free(sys, B->A.p);
  if (B->store && --B->store->refs == 0) { mspack_arena_free(sys, B->store->base); sys->free(B->store); }
  sys->free(B);
}
static void batch_abandon(struct mspack_system *sys, struct cab_batch *B) { batch_release(sys, B); }

/* a folder whose batch is still running: wait for its unit (and its blocks' checksum units), take the result over.  Returns 0 with
 * the folder decoded -- or flagged and NOT decoded (a block failed its checksum on the device: the caller has it gathered again,
 * the reference's way) --, or the error of a batch that failed as a whole */
static int folder_settle(struct cabd_p *self, struct folder_p *fp)
{
  struct mspack_system *sys = self->system;
  struct cab_batch *B = fp->job;
  const size_t k = fp->job_k;
  size_t j;
  int rc = mspack_hip_job_wait_unit(B->job, k), bad = 0;
  const size_t j_lo = B->ck_lo ? B->ck_lo[k] : 0, j_hi = B->ck_lo ? B->ck_lo[k + 1] : B->ck.n;
  for (j = j_lo; j < j_hi && !rc; j++)
    if (B->ck.p[j].owner == (unsigned int) k) {
      rc = mspack_hip_job_wait_unit(B->job, B->n + j);
      if (!rc && (B->res[B->n + j].err != MSPACK_ERR_OK || B->res[B->n + j].in_next != B->ck.p[j].want)) bad = 1;
    }
  if (rc) {
    /* the batch failed as a whole: what decode_cabinet says of a failed call -- none of its folders is decoded */
    batch_release(sys, B);
    sys->message(NULL, "GPU batch decode failed: %s", mspack_hip_last_error());
    return MSPACK_ERR_DECRUNCH;
  }
  if (bad) fp->cksum_on_host = 1;                          /* (decoded past a bad block: not kept) */
  else batch_take_folder(B, k);
  fp->job = NULL; B->gs[k].fol = NULL;
  if (--B->left == 0) batch_release(sys, B);
  return MSPACK_ERR_OK;
}

/* Quantum: what the codec holds back when a request ends at `pos` -- the batch's answer for the folder's marks (0: nothing, or not
 * known); QTM_MARK_FAILS: a request that ends there fails although the stream decodes beyond it (mspack_hip.h) */
#define QTM_MARK_FAILS 0xFFFFFFFFu
static unsigned int qtm_mark(const struct folder_p *fp, unsigned int pos)
{
  unsigned int lo = 0, hi = fp->n_marks;
  if (!fp->marks || !fp->mark_log) return 0;
  while (lo < hi) { const unsigned int mid = lo + (hi - lo) / 2; if (fp->marks[mid] < pos) lo = mid + 1; else hi = mid; }
  return (lo < fp->n_marks && fp->marks[lo] == pos) ? rd_le32(fp->mark_log + 4 * (size_t) lo) : 0;
}

/* can the reference produce bytes [0, end) of this folder, and with which error if not? */
static int folder_status(struct folder_p *fp, unsigned int end, int read_error, int *failed)
{
  int method = fp->base.comp_type & 0x0F, ok;
  *failed = 1;                        /* (a failure whose code is the feeder's may read MSPACK_ERR_OK in salvage mode) */
  if (fp->dec_err == MSPACK_ERR_OK && end > fp->total) {
    /* the request goes past everything the blocks hold.  After a failed block read the codec's
     * next refill fails (-> the feeder's error); after a clean end LZX knows the stream length and
     * gives up with DECRUNCH (lzxd.c:458-461,758-761), the others run into the end of input */
    if (fp->hard_eof) return read_error;
    return (method == MSCAB_COMP_LZX) ? MSPACK_ERR_DECRUNCH : read_error;
  }
  if (method == MSCAB_COMP_LZX) {
    /* lzxd decodes every frame up to and including frame end/32768 (one frame of look-ahead when
     * `end` is a frame multiple, lzxd.c:419) */
    unsigned int need_f = end / CAB_BLOCKMAX, nframes = (fp->total + CAB_BLOCKMAX - 1) / CAB_BLOCKMAX;
    if (fp->dec_err == MSPACK_ERR_OK) ok = 1;
    else if (fp->good_len >= fp->total) ok = (need_f < nframes);          /* only the look-ahead failed */
    else ok = (need_f < fp->n_frames_good);
  }
  else {
    ok = (end <= fp->good_len);
    /* (a request that ends inside a match which crosses the window's end, in front of that end: qtmd cannot serve it, qtmd.c:366-374
     * -- windows below the frame size, or damage) */
    if (ok && method == MSCAB_COMP_QUANTUM && qtm_mark(fp, end) == QTM_MARK_FAILS) return MSPACK_ERR_DECRUNCH;
  }
  if (ok) { *failed = 0; return MSPACK_ERR_OK; }
  if (fp->dec_err == MSPACK_ERR_OK) return read_error;
  return (fp->dec_err == MSPACK_ERR_READ) ? read_error : fp->dec_err;
}

/* what the reference's decompressor has WRITTEN when a call fails (its d->offset, cabd.c:1352): lzxd and mszipd hand over every
 * frame / block once it is decoded (lzxd.c:738-751, mszipd.c:440-452), so everything below the failing frame is out -- but no more
 * than the call asked for; qtmd writes when its window wraps and at the end of a call that succeeds (qtmd.c:420-428, 468-474) */
static unsigned int flushed_at_failure(struct folder_p *fp, unsigned int end)
{
  const int method = fp->base.comp_type & 0x0F;
  const unsigned int reached = (fp->dec_err == MSPACK_ERR_OK || fp->good_len > fp->total) ? fp->total : fp->good_len;
  unsigned int f;
  if (method == MSCAB_COMP_QUANTUM) {
    const unsigned int w = 1u << ((fp->base.comp_type >> 8) & 0x1F);
    /* (a stream that fails does so at the same symbol whatever the request: what the codec wrote for the whole folder -- its
     * window can wrap right in front of a failing frame trailer, one byte beyond good_len) */
    f = fp->dec_err != MSPACK_ERR_OK ? fp->written : reached / w * w;
    if (end <= fp->good_len && qtm_mark(fp, end) == QTM_MARK_FAILS) f = end / w * w;   /* (that request's own failure: the windows below it) */
  }
  else if (method == MSCAB_COMP_LZX && fp->dec_err != MSPACK_ERR_OK && fp->good_len < fp->total) f = fp->n_frames_good * CAB_BLOCKMAX;
  else f = (reached == fp->total) ? fp->total : reached / CAB_BLOCKMAX * CAB_BLOCKMAX;
  return f < end ? f : end;
}

static int cabd_extract(struct mscab_decompressor *base, struct mscabd_file *file, const char *filename)
{
  struct cabd_p *self = (struct cabd_p *) base;
  struct mspack_system *sys;
  struct mspack_file *fh;
  struct folder_p *fol;
  unsigned int filelen;

  if (!self) return MSPACK_ERR_ARGS;
  if (!file) return self->error = MSPACK_ERR_ARGS;
  sys = self->system;
  fol = (struct folder_p *) file->folder;

  /* the reference's argument checks, in its order (cabd.c:1090-1125) */
  if (file->offset > CAB_LENGTHMAX) return self->error = MSPACK_ERR_DATAFORMAT;
  filelen = file->length;
  if (filelen > (CAB_LENGTHMAX - file->offset)) {
    if (self->salvage) filelen = CAB_LENGTHMAX - file->offset;
    else return self->error = MSPACK_ERR_DATAFORMAT;
  }
  if (!fol || fol->merge_prev) {
    sys->message(NULL, "ERROR; file \"%s\" cannot be extracted, cabinet set is incomplete", file->filename);
    return self->error = MSPACK_ERR_DECRUNCH;
  }
  if (!self->salvage) {
    unsigned int maxlen = fol->base.num_blocks * CAB_BLOCKMAX;
    if (file->offset > maxlen || filelen > (maxlen - file->offset)) {
      sys->message(NULL, "ERROR; file \"%s\" cannot be extracted, cabinet set is incomplete", file->filename);
      return self->error = MSPACK_ERR_DECRUNCH;
    }
  }
  switch (fol->base.comp_type & 0x0F) {
  case MSCAB_COMP_NONE: case MSCAB_COMP_MSZIP: case MSCAB_COMP_QUANTUM: case MSCAB_COMP_LZX: break;
  default: return self->error = MSPACK_ERR_DATAFORMAT;                   /* cabd.c:1254 */
  }
  {
    int wb = (fol->base.comp_type >> 8) & 0x1F, m = fol->base.comp_type & 0x0F;
    /* lzxd_init / qtmd_init refuse these windows -> MSPACK_ERR_NOMEMORY (cabd.c:1256) */
    if ((m == MSCAB_COMP_LZX && (wb < 15 || wb > 21)) || (m == MSCAB_COMP_QUANTUM && (wb < 10 || wb > 21)))
      return self->error = MSPACK_ERR_NOMEMORY;
  }

  /* one decompression state per decompressor: another folder ends the stored folder's stream */
  if (self->last_folder != fol) stored_reset(self);
  self->last_folder = fol;
  if ((fol->base.comp_type & 0x0F) == MSCAB_COMP_NONE) { self->live_folder = NULL; self->msg_folder = NULL; return stored_extract(self, fol, file, filelen, filename); }

  /* (a folder whose batch is still running waits for its own unit; one that comes out of that flagged -- a block failed its
   * checksum on the device -- is gathered again, which may start the next batch) */
  {
    int tries;
    for (tries = 0; !fol->decoded && tries < 4; tries++) {
      int err = fol->job ? folder_settle(self, fol) : decode_cabinet(self, (struct cab_p *) fol->data.cab, fol);
      if (err) return self->error = err;
    }
  }
  self->read_error = fol->read_err;
  if (self->live_folder != fol || self->live_offset > file->offset) {    /* cabd.c:1136: another folder, or an earlier offset */
    self->live_folder = fol; self->live_offset = 0; self->live_failed = 0; self->live_err = MSPACK_ERR_OK;
  }

  if (!(fh = sys->open(sys, filename, MSPACK_SYS_OPEN_WRITE))) return self->error = MSPACK_ERR_OPEN;
  self->error = MSPACK_ERR_OK;
  /* What the codec would have said on the way: mszipd in repair mode reports every block it repairs when it decodes it
   * (mszipd.c:420-433), i.e. when the first byte of that block is asked for -- by this file, or by the skip to its offset.
   * The reference's decompressor starts over (and says it all again) for another folder or an earlier offset. */
  if (filelen && self->live_failed) ;       /* (the codec's sticky error: it reads nothing more and says nothing more) */
  else if ((fol->rep_n || fol->ck_n) && filelen) {
    const unsigned int end = file->offset + filelen;
    if (self->msg_folder != fol || self->msg_offset > file->offset) { self->msg_folder = fol; self->msg_next = 0; self->msg_next_ck = 0; }
    for (;;) {
      const int have_ck = self->msg_next_ck < fol->ck_n && fol->rep_ck[self->msg_next_ck] < end;
      const int have_rp = self->msg_next < fol->rep_n && fol->rep[2 * self->msg_next] < end;
      if (have_ck && (!have_rp || fol->rep_ck[self->msg_next_ck] <= fol->rep[2 * self->msg_next])) {
        /* the reference says this with the handle of the cabinet it is reading (cabd.c:1415: sys->message(d->infh, ...)): a
         * system that prints the file's name must find one.  This driver's gather closed the cabinet long ago, so it is opened
         * for the occasion (the folder's first cabinet; if that fails the line goes out without a handle) */
        struct mspack_file *mfh = sys->open(sys, fol->data.cab->base.filename, MSPACK_SYS_OPEN_READ);
        sys->message(mfh, "WARNING; bad block checksum found");
        if (mfh) sys->close(mfh);
        self->msg_next_ck++;
      }
      else if (have_rp) {
        sys->message(NULL, "MSZIP error, %u bytes of data lost.", fol->rep[2 * self->msg_next + 1]);
        self->msg_next++;
      }
      else break;
    }
    self->msg_offset = end;
  }
  else if (filelen) { self->msg_folder = fol; self->msg_offset = file->offset + filelen; self->msg_next = 0; self->msg_next_ck = 0; }
  if (filelen && self->live_failed) self->error = self->live_err;      /* (the codec's sticky error: nothing decoded, nothing written) */
  else if (filelen) {
    /* skip phase: getting to file->offset must itself be error free (cabd.c:1195-1199) */
    int failed = 0;
    int err = file->offset ? folder_status(fol, file->offset, self->read_error, &failed) : MSPACK_ERR_OK;
    unsigned int end = file->offset, wrote_to = 0;
    if (!err) {
      /* (a skip that failed with a code that reads OK -- salvage mode, out of blocks -- is followed by the emit call, which meets
       * the codec's sticky error: the same code again, nothing written) */
      if (!failed) {
        unsigned int have = fol->good_len > file->offset ? fol->good_len - file->offset : 0;
        if (file->offset > self->live_offset) self->live_offset = file->offset;      /* (the skip call's bytes were handed over) */
        if (have > filelen) have = filelen;
        end = file->offset + filelen;
        err = folder_status(fol, end, self->read_error, &failed);
        if (failed && (fol->base.comp_type & 0x0F) == MSCAB_COMP_QUANTUM) {
          /* a Quantum call that fails has written what its window wraps flushed, not everything it decoded (qtmd.c:420-428; the
           * wrap inside a match that crosses the window's end included, :358-390): what the codec wrote for the whole folder */
          const unsigned int w = 1u << ((fol->base.comp_type >> 8) & 0x1F);
          /* (a folder that decodes cleanly but is asked for more than it holds fails at the end of its input: the last window's
           * bytes were never flushed) */
          unsigned int wr = fol->dec_err == MSPACK_ERR_OK ? fol->total / w * w : fol->written;
          /* before it decodes anything the call hands over what its predecessor -- the skip to this file -- held back: the rest
           * of the match that covers the byte in front of the file (qtmd.c:268-276) */
          unsigned int carry = qtm_mark(fol, file->offset);
          if (end <= fol->good_len && qtm_mark(fol, end) == QTM_MARK_FAILS) wr = end / w * w;
          have = wr > file->offset ? wr - file->offset : 0;
          if (carry != QTM_MARK_FAILS && have < carry) have = carry;
          if (have > filelen) have = filelen;
          wrote_to = file->offset + have;
        }
        if (write_slice(sys, fh, fol->dec + file->offset, have) != MSPACK_ERR_OK) err = MSPACK_ERR_WRITE;
      }
    }
    self->error = err;
    if (failed) {
      const unsigned int fl = flushed_at_failure(fol, end);
      self->live_failed = 1; self->live_err = err;
      if (fl > self->live_offset) self->live_offset = fl;
      if (wrote_to > self->live_offset) self->live_offset = wrote_to;
    }
    else self->live_offset = file->offset + filelen;
  }
  sys->close(fh);
  return self->error;
}

static int cabd_param(struct mscab_decompressor *base, int param, int value)
{
  struct cabd_p *self = (struct cabd_p *) base;
  if (!self) return MSPACK_ERR_ARGS;
  switch (param) {
  case MSCABD_PARAM_SEARCHBUF: if (value < 4) return MSPACK_ERR_ARGS; self->searchbuf_size = value; break;
  case MSCABD_PARAM_FIXMSZIP:  self->fix_mszip = value; break;
  case MSCABD_PARAM_DECOMPBUF: if (value < 4) return MSPACK_ERR_ARGS; self->buf_size = value; break;
  case MSCABD_PARAM_SALVAGE:   self->salvage = value; break;
  case MSCABD_PARAM_HIP_DEVICES:  if (value < 1) return MSPACK_ERR_ARGS; self->devices = value; break;
  case MSCABD_PARAM_HIP_CACHE_MB: if (value < 1) return MSPACK_ERR_ARGS; self->cache_mb = value; break;
  default: return MSPACK_ERR_ARGS;
  }
  return MSPACK_ERR_OK;
}

static int cabd_error(struct mscab_decompressor *base) {
  struct cabd_p *self = (struct cabd_p *) base;
  return self ? self->error : MSPACK_ERR_ARGS;
}

struct mscab_decompressor *mspack_create_cab_decompressor(struct mspack_system *sys)
{
  struct cabd_p *self;
  if (!sys) sys = mspack_default_system;
  if (!mspack_valid_system(sys)) return NULL;
  if (!(self = (struct cabd_p *) sys->alloc(sys, sizeof(*self)))) return NULL;
  self->base.open = &cabd_open;
  self->base.close = &cabd_close;
  self->base.search = &cabd_search;
  self->base.extract = &cabd_extract;
  self->base.prepend = &cabd_prepend;
  self->base.append = &cabd_append;
  self->base.set_param = &cabd_param;
  self->base.last_error = &cabd_error;
  self->system = sys;
  self->error = MSPACK_ERR_OK; self->read_error = MSPACK_ERR_OK;
  self->searchbuf_size = 32768; self->fix_mszip = 0; self->buf_size = 4096; self->salvage = 0;
  self->devices = 1; self->cache_mb = 2048;
  memset(&self->st, 0, sizeof(self->st)); self->st_offset = 0; self->st_active = 0; self->last_folder = NULL;
  self->msg_folder = NULL; self->msg_offset = 0; self->msg_next = 0; self->msg_next_ck = 0;
  self->live_folder = NULL; self->live_offset = 0; self->live_failed = 0; self->live_err = MSPACK_ERR_OK;
  return &self->base;
}

void mspack_destroy_cab_decompressor(struct mscab_decompressor *base)
{
  struct cabd_p *self = (struct cabd_p *) base;
  if (self) { stored_reset(self); self->system->free(self); }
}
