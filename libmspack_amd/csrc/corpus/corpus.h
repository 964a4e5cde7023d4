/* corpus.h -- synthetic corpus generators (TEST / BENCH INFRASTRUCTURE, not the decode product).
 *
 * The reference ships no compressors (libmspack/mspack/lzxc.c, qtmc.c, mszipc.c, cabc.c, chmc.c
 * are stubs that return NULL, e.g. cabc.c:15-20), so every LZX / Quantum stream and every
 * CAB / CHM container the tests and bench.py decode is produced by the encoders declared here.
 * The encoder-side rules were derived from what the reference decoders accept
 * (lzxd.c:447-523,538-651; qtmd.c:92-123,292-442; chmd.c:1072-1267; cabd.c:1362-1479).
 */
#ifndef MSPACK_AMD_CORPUS_H
#define MSPACK_AMD_CORPUS_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- plaintext ------------------------------------------------------------------------------ */
enum {
  MSPK_TEXT_MIX = 0,        /* 1/3 each of the three below, switched every few KiB            */
  MSPK_TEXT_ENGLISH = 1,    /* Zipf-sampled pseudo-English / HTML-ish text                     */
  MSPK_TEXT_BINARY = 2,     /* LE32 counters, pointers, x86-like bytes incl. 0xE8 call sites   */
  MSPK_TEXT_RECORDS = 3,    /* repeated fixed-layout records with small per-record edits       */
  MSPK_TEXT_RANDOM = 4,     /* PRNG bytes (incompressible)                                     */
  MSPK_TEXT_REPETITIVE = 5  /* long matches (a short phrase repeated with rare edits)          */
};
void mspk_gen_plaintext(uint64_t seed, int kind, uint8_t *out, size_t n);

/* ---- LZX encoder ------------------------------------------------------------------------------ */
typedef struct mspk_lzx_opts {
  int block_mode;      /* 0 auto (verbatim vs aligned by cost), 1 verbatim, 2 aligned,
                          3 uncompressed, 4 cycle 1,2,3 per block                               */
  int block_size;      /* uncompressed bytes per block; 0 = 32768 (one block per frame)        */
  int chain_depth;     /* hash-chain search depth; 0 = 24                                      */
  int use_repeats;     /* 1 = code matches at R0/R1/R2 with slots 0..2                         */
  int lazy;            /* 1 = one-step lazy matching                                           */
  int32_t intel_filesize; /* != 0: set the E8 header and pre-translate x86 CALLs (lzxd.c:706-736
                             inverse) so that decode(encode(x)) == x                            */
  int32_t e8_base;     /* value of lzx->offset at the first byte of `src` (curpos origin)       */
  int delta;           /* 1 = LZX DELTA (lzxd.c:288-293,440-444,588-611): window 2^17..2^25, a 16-bit
                          chunk size before every frame, match lengths beyond 257; reset_frames 0 */
  const uint8_t *ref;  /* DELTA reference data the matches may reach into (lzxd.c:348-382)       */
  size_t ref_len;
} mspk_lzx_opts;

/* Encode src[0..n) as one LZX stream.  reset_frames > 0: encoder state (match history, Huffman
 * lengths, R0-R2, header bit) restarts every reset_frames*32768 bytes, as CHM content does;
 * reset_frames == 0: one continuous stream (a CAB folder).
 * frame_off (optional, n_frames+1 entries) receives the compressed byte offset at which each
 * 32 KiB frame starts (entry n_frames = total size): the CHM reset table / CAB CFDATA split.
 * Returns compressed size, or 0 if dst_cap was too small. */
size_t mspk_lzx_encode(const uint8_t *src, size_t n, int window_bits, int reset_frames,
                       const mspk_lzx_opts *opts, uint8_t *dst, size_t dst_cap,
                       uint64_t *frame_off);
size_t mspk_lzx_bound(size_t n);

/* ---- Quantum encoder -------------------------------------------------------------------------- */
/* Encode src[0..n) as one Quantum folder stream; frame_size[i] = compressed bytes of frame i
 * (each frame is one CFDATA payload; the 0xFF trailer is NOT stored).  Returns total size. */
size_t mspk_qtm_encode(const uint8_t *src, size_t n, int window_bits, int chain_depth,
                       uint8_t *dst, size_t dst_cap, uint32_t *frame_size);
size_t mspk_qtm_bound(size_t n);

/* ---- containers ------------------------------------------------------------------------------- */
typedef struct mspk_cab_folder {
  int comp_type;                 /* CFFOLDER typeCompress (cab.h:52-58 values)                  */
  const uint8_t *data;           /* concatenated CFDATA payloads                                */
  const uint32_t *block_comp;    /* payload size per block                                       */
  const uint32_t *block_uncomp;  /* uncompressed size per block                                  */
  int n_blocks;
} mspk_cab_folder;
typedef struct mspk_cab_file {
  const char *name; uint32_t length; uint32_t folder_offset; uint16_t folder_index;
} mspk_cab_file;
/* Writes a single cabinet (valid CFDATA checksums).  Returns size or 0 if cap too small. */
size_t mspk_cab_write(const mspk_cab_folder *folders, int n_folders,
                      const mspk_cab_file *files, int n_files, uint8_t *dst, size_t cap);

typedef struct mspk_chm_file {
  const char *name;              /* e.g. "/doc0001.html"                                         */
  uint64_t offset, length;       /* position in the uncompressed section-1 stream                */
} mspk_chm_file;
/* Writes a CHM (ITSF v3) whose section 1 is the given LZX stream.  frame_off has n_frames+1
 * entries as produced by mspk_lzx_encode.  Returns size or 0 if cap too small. */
size_t mspk_chm_write(const uint8_t *lzx, size_t lzx_len, const uint64_t *frame_off, size_t n_frames,
                      uint64_t uncomp_len, int window_bits, int reset_frames,
                      const mspk_chm_file *files, int n_files, uint8_t *dst, size_t cap);
size_t mspk_chm_bound(size_t lzx_len, size_t n_frames, int n_files);

/* ---- batch corpus (multi-threaded) ---------------------------------------------------------------- */
/* Generate n_units independent LZX units (each `unit_bytes` of plaintext of the given kind, seed =
 * base_seed mixed with the unit index), encoded with window_bits and reset every unit.
 * plain (n_units*unit_bytes) and comp (capacity comp_cap) are caller-allocated; comp_off/comp_len
 * receive each unit's slice.  Returns total compressed bytes, 0 on overflow. */
size_t mspk_corpus_lzx_units(uint64_t base_seed, int kind, int n_units, size_t unit_bytes,
                             int window_bits, const mspk_lzx_opts *opts, int n_threads,
                             uint8_t *plain, uint8_t *comp, size_t comp_cap,
                             uint64_t *comp_off, uint32_t *comp_len);
/* the same for units [first_unit, first_unit + n_units) of a larger global list (a rank's shard of a
 * strong-scaling corpus): unit i of the call gets the seed of global unit first_unit + i */
size_t mspk_corpus_lzx_units_at(uint64_t base_seed, uint64_t first_unit, int kind, int n_units, size_t unit_bytes,
                                int window_bits, const mspk_lzx_opts *opts, int n_threads,
                                uint8_t *plain, uint8_t *comp, size_t comp_cap,
                                uint64_t *comp_off, uint32_t *comp_len);
/* the same, and every unit's frame table written into `comp` behind the unit (tab_off[i] = its arena offset,
 * 4-byte aligned; one uint32 per 32 KiB frame: the frame's compressed offset from the unit's first byte) */
size_t mspk_corpus_lzx_units_ft(uint64_t base_seed, uint64_t first_unit, int kind, int n_units, size_t unit_bytes,
                                int window_bits, const mspk_lzx_opts *opts, int n_threads,
                                uint8_t *plain, uint8_t *comp, size_t comp_cap,
                                uint64_t *comp_off, uint32_t *comp_len, uint64_t *tab_off);

#ifdef __cplusplus
}
#endif
#endif
