/* mspack_hip.h -- C ABI of the MI355X batched LZX / Quantum / MSZIP decoder.
 *
 * This is the drop-in boundary for the reference's decompression hot path.  The reference exposes
 * that path only as pull-streams, one stream at a time:
 *     lzxd_init / lzxd_decompress / lzxd_free      libmspack/mspack/lzx.h:146-214
 *     qtmd_init / qtmd_decompress / qtmd_free      libmspack/mspack/qtm.h:92-122
 *     mszipd_init / mszipd_decompress / mszipd_free libmspack/mspack/mszip.h:85-120
 * driven by cabd_extract (cabd.c:1075-1214, codec dispatch cabd.c:1226-1269) and chmd_extract
 * (chmd.c:906-1041, chmd_init_decomp chmd.c:1072-1188).  A GPU decodes many independent streams
 * per launch, so the replacement ABI is "one batch of units in, one batch of results out":
 * a unit is exactly one reference stream, i.e. what ONE xxxd_init + xxxd_decompress(out_len)
 * pair would have decoded --
 *     LZX     : one CAB folder (reset_frames = 0) or one CHM reset interval (chmd.c:1147-1183)
 *     Quantum : one CAB folder (payloads + the 0xFF trailer cabd.c:1330-1332 adds per block)
 *     MSZIP   : one CAB folder (concatenated "CK" blocks; history persists, mszipd.c:267-268)
 * Results are bit-exact with the reference, including its MSPACK_ERR_* code per unit.
 *
 * Plain C: pointers and sizes only; no HIP or torch types appear in any signature (device
 * pointers and the stream are passed as void*).  The libmspack-compatible object API
 * (mspack_create_cab_decompressor & co.) that sits on top of this lives in mspack.h.
 */
#ifndef MSPACK_HIP_H
#define MSPACK_HIP_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* unit kinds: the CAB compression-method codes (cab.h:52-58, cabd.c:1232-1262) */
#define MSPACK_HIP_KIND_MSZIP   1
#define MSPACK_HIP_KIND_QUANTUM 2
#define MSPACK_HIP_KIND_LZX     3
#define MSPACK_HIP_KIND_LZX_DELTA 4   /* lzxd_init(is_delta = 1): OAB blocks (oabd.c:196, 346; lzxd.c:288-293) */
#define MSPACK_HIP_KIND_LZSS     5   /* lzss_decompress (lzssd.c:36-91): SZDD, KWAJ method 2; window_bits = mode
                                       (0 EXPAND, 1 MSHELP, 2 QBASIC).  The stream carries no length: out_len is
                                       the room available, the result's out_len what the stream produced.  The
                                       4096 bytes below out_off belong to the unit (its window pre-fill).        */
#define MSPACK_HIP_KIND_KWAJ_LZH 6   /* lzh_decompress (kwajd.c:432-563): KWAJ method 3; same conventions         */
#define MSPACK_HIP_KIND_XORSUM   7   /* cabd_checksum(data, bytes, 0) (cabd.c:1462-1479): the checksum of ONE CFDATA block's
                                       payload -- the XOR of its little-endian dwords and the reference's odd big-endian tail of
                                       bytes & 3 bytes.  Decodes nothing: out_len must be 0, no output region; the checksum comes
                                       back in result.in_next (err 0, in_used = in_len).  The cabinet driver puts one such unit
                                       per block part into the batch that decodes the folders: the bytes are in HBM anyway, and
                                       it compares with the CFDATA headers (cabd.c:1411-1417) when the results are back        */

#define MSPACK_HIP_MASK_FRAME_TABLES 0x80000000u   /* mspack_hip_decode_batch_device(kind_mask): see there */

/* result flags */
#define MSPACK_HIP_F_E8_APPLIED     1u  /* >=1 frame went through the E8 translation (lzxd.c:706-736) */
#define MSPACK_HIP_F_LOOKAHEAD_READ 2u  /* all bytes produced; the one-frame look-ahead of
                                           lzxd.c:419 then hit end of input (err = MSPACK_ERR_READ)  */
#define MSPACK_HIP_F_INTEL_HEADER   4u  /* an LZX interval header carried intel_filesize != 0        */
#define MSPACK_HIP_F_BLOCK_OPEN    16u  /* LZX: the unit's last frame ended inside a block (block_remaining != 0).
                                           At a reset point the reference only warns and goes on with that
                                           block (lzxd.c:424-431): the next interval is then NOT an
                                           independent unit, and drivers decode such a stream serially     */
#define MSPACK_HIP_F_FRAMES_ADOPTED 32u /* LZX, diagnostic: at least one frame's tokens came from a parse wavefront
                                           (MSPACK_HIP_UF_FRAME_TABLE); says nothing about the decoded bytes      */
#define MSPACK_HIP_F_OUT_FULL       8u  /* KWAJ-framed MSZIP: the next block did not fit out_len (give the
                                           unit more room and decode again)                          */

/* unit input flags */
#define MSPACK_HIP_UF_MSZIP_REPAIR  1u  /* mszipd repair mode (MSCABD_PARAM_FIXMSZIP, mszipd.c:420-437) */
#define MSPACK_HIP_UF_MSZIP_KWAJ    4u  /* MSZIP as KWAJ files frame it (mszipd_decompress_kwaj, mszipd.c:462-495):
                                           16-bit block length (0 ends the stream), 'C','K', one deflate stream;
                                           out_len is the room available, the result's out_len what was produced */
#define MSPACK_HIP_UF_FRAME_TABLE   8u  /* LZX / MSZIP: the container states where every 32 KiB frame (LZX) / CFDATA block
                                           (MSZIP; every block but the last holding 32768 bytes) of the unit starts in
                                           the compressed stream (a cabinet: the CFDATA sizes, cabd.c:1362-1479; a CHM: one
                                           reset-table entry per frame, chmd.c:1146-1149).  in_chunk * 4 is the byte
                                           offset, in the input arena, of a uint32 table with one entry per frame: its
                                           offset from in_off.  With it the frames' tokens are parsed by one wavefront
                                           each before the unit's own wavefront commits them; the table is a HINT -- a
                                           wrong one costs time, never correctness (the unit's wavefront checks every
                                           frame's bit position and falls back to decoding serially).  Ignored for MSZIP
                                           units in repair or KWAJ mode (in_chunk has its other meaning there)        */
#define MSPACK_HIP_UF_MSZIP_LOG    16u  /* with MSPACK_HIP_UF_MSZIP_REPAIR: report the repaired blocks (what the reference says through
                                           sys->message, mszipd.c:427).  e8_base = capacity N of a log the unit owns in the output
                                           arena at out_off + ((out_len + 32768 + 15) & ~15): uint32 count of repaired blocks
                                           (all of them, also beyond N), then N pairs (output offset of the block, bytes lost) */
#define MSPACK_HIP_UF_LZX_LOG      32u  /* MSPACK_HIP_KIND_LZX with reset_frames != 0: report the reset points that found a block still
                                           open -- where the reference says "WARNING; invalid reset interval detected during LZX
                                           decompression" (lzxd.c:423-431).  ref_len = capacity N of a log the unit owns in the
                                           output arena at out_off + ((out_len + 32768 + 15) & ~15): uint32 count (all of them,
                                           also beyond N), then N frame indices (counted from the unit's first frame) */
#define MSPACK_HIP_UF_QTM_MARKS    64u  /* MSPACK_HIP_KIND_QUANTUM: what qtmd holds back at request boundaries.  A qtmd_decompress() call decodes
                                           whole tokens: the match that covers the last byte asked for usually runs past it, and the
                                           bytes beyond stay in the window until the NEXT call, which hands them to ITS output before
                                           it decodes anything (qtmd.c:268-276) -- a next call that then FAILS has still written them.
                                           The token sequence does not depend on where requests end, so one decode of the whole stream
                                           can say it for every boundary a caller may use: in_chunk * 4 = byte offset, in the input
                                           arena, of a uint32 table of ref_len ascending positions p (0 < p < out_len; a cabinet: the
                                           offsets at which the folder's files begin); the unit owns a log of ref_len uint32 in the
                                           output arena at out_off + ((out_len + 15) & ~15): entry i = how many bytes beyond p_i were
                                           decoded when a request ending at p_i returns (0: a token ends exactly there -- or the
                                           stream failed before p_i was reached); 0xFFFFFFFF: a request ending at p_i FAILS in the
                                           reference (MSPACK_ERR_DECRUNCH) although the stream decodes beyond it -- p_i lies inside a
                                           match that crosses the window's end, in front of that end (qtmd.c:358-374).  out_off must
                                           be a multiple of 4 */
#define MSPACK_HIP_UF_HARD_EOF      2u  /* the feeder's read FAILED at in_len (sys->read < 0, e.g. a bad
                                           CFDATA block, cabd.c:1322-1324): ERR_READ at once, without the
                                           two fabricated zero bytes of a clean EOF (readbits.h:194-208) */

typedef struct mspack_hip_unit {
  uint64_t in_off;       /* byte offset of the unit's compressed bytes in the input arena          */
  uint64_t out_off;      /* byte offset of the unit's output in the output arena                   */
  uint32_t in_len;       /* compressed bytes available; reading past them yields the reference's
                            two fabricated zero bytes, then MSPACK_ERR_READ (readbits.h:192-214)   */
  uint32_t out_len;      /* bytes to produce == xxxd_decompress(out_bytes) == the stream's length  */
  uint32_t frame_base;   /* LZX: first slot of this unit in the per-frame scratch (see below)      */
  int32_t  e8_base;      /* LZX: lzx->offset at the unit's first byte (E8 curpos origin)           */
  uint8_t  kind;         /* MSPACK_HIP_KIND_*                                                      */
  uint8_t  window_bits;  /* LZX 15..21, LZX DELTA 17..25, Quantum 10..21, ignored for MSZIP         */
  uint16_t reset_frames; /* LZX: lzxd_init reset_interval in 32 KiB frames (0 = never, CAB)        */
  uint32_t flags;        /* MSPACK_HIP_UF_*                                                        */
  uint32_t ref_len;      /* LZX DELTA: bytes of reference data (lzxd_set_reference_data, lzxd.c:348-382)
                            that the caller placed in the output arena at [out_off - ref_len, out_off);
                            LZX with MSPACK_HIP_UF_LZX_LOG: capacity of the unit's log;
                            Quantum with MSPACK_HIP_UF_QTM_MARKS: entries of the unit's table of marks */
  uint32_t in_chunk;     /* MSZIP repair mode: input_buffer_size of mszipd_init (mszipd.c:338-375), i.e.
                            the chunking of the folder stream by the reference's feeder; where the
                            next block is looked for after a failed one depends on it (mszipd.c:404,
                            readbits.h:184-214).  0 = 4096 (the cabd default).
                            LZX with MSPACK_HIP_UF_FRAME_TABLE: arena offset / 4 of the frame table.
                            Quantum with MSPACK_HIP_UF_QTM_MARKS: arena offset / 4 of the table of marks.  Else ignored */
} mspack_hip_unit;

typedef struct mspack_hip_result {
  int32_t  err;          /* MSPACK_ERR_* exactly as the reference's decompress call returns        */
  uint32_t flags;        /* MSPACK_HIP_F_*                                                         */
  uint32_t out_len;      /* bytes produced (handed to sys->write in the reference)                 */
  uint32_t in_used;      /* compressed bytes the unit pulled (diagnostic)                          */
  uint32_t good_len;     /* bytes decoded before the failing point (== out_len when err == 0).  A
                            request that ends at or before good_len succeeds in the reference too
                            (it decodes no further than asked): LZX counts whole frames, MSZIP
                            whole blocks, Quantum the position of the failing symbol              */
  uint32_t in_next;      /* LZX: input byte offset (from in_off) right after the 16-bit realignment that
                            follows the last completely decoded non-empty frame (lzxd.c:695-697), i.e.
                            where a decoder that keeps going reads the next frame from.  The CHM driver
                            checks the reset table against it; 0 if no frame was completed.
                            MSZIP (err 0): bytes the last decoded block inflated to BEYOND out_len -- mszipd sizes a block
                            by its deflate stream, not by the CFDATA header (mszipd.c:377-460), and keeps those bytes for
                            its next call; they lie in the unit's 32 KiB of slack behind out_len                     */
} mspack_hip_result;

/* ---- library / device ----------------------------------------------------------------------- */
/* 0 on success, else a negative hipError_t.  All calls are per calling thread's current device
 * unless mspack_hip_set_device() is used. */
int  mspack_hip_device_count(void);
int  mspack_hip_set_device(int device);
const char *mspack_hip_version(void);
const char *mspack_hip_last_error(void);

/* ---- device-resident batch decode (the hot path proper) ---------------------------------------
 * All pointers are DEVICE pointers valid on the current device.  `stream` is a hipStream_t (NULL =
 * default stream).  The call is asynchronous with respect to the host.
 *   d_units    : n_units descriptors
 *   d_order    : optional launch order (unit indices, e.g. longest first); NULL = identity
 *   d_in       : input arena, in_bytes long (must be followed by >= 8 readable bytes of slack
 *                or in_bytes must leave 8 bytes of head-room inside the allocation)
 *   d_out      : output arena (every unit owns [out_off, out_off + out_len))
 *   d_results  : n_units results
 *   d_frame_scratch : >= mspack_hip_frame_scratch_bytes(total LZX frames incl. 1 spare per unit) bytes of LZX
 *                work space: per frame the E8 decision, and -- for units with a frame table -- the parse
 *                waves' records and a POOL of match records shared by the launch's frames: about 49.4 KiB per frame
 *                slot (1.4 KiB of record + 48 KiB of pool: 1.5x the decoded bytes; round 3 reserved the worst case
 *                of 16384 matches for every slot, 129 KiB).  A frame takes pool chunks of 1024 records as its parse
 *                needs them; when the pool runs dry (more than 6144 matches per frame on average over the launch)
 *                the frames that find it empty are decoded by the serial path -- slower, same result.  Contents need no
 *                initialisation; may be NULL if the batch has no LZX units (LZX then decodes serially only)
 *   kind_mask  : bit k set = units of kind k may be present; MSPACK_HIP_MASK_FRAME_TABLES set = LZX units may
 *                carry frame tables (MSPACK_HIP_UF_FRAME_TABLE): only then are the parse wavefronts launched.
 *                Was: bit k set = units of kind k may be present (one kernel per codec is launched;
 *                units of other kinds are skipped); 0 = all three codecs
 * MSZIP units need 32768 bytes of slack after out_len in their output region.
 * Units with a frame / block table: the parse wavefronts store literals into the unit's output region (for MSZIP incl. its
 * slack) before the unit is known to decode; the first result.out_len bytes are the decoded data, the rest of the region
 * is unspecified afterwards -- also when the unit fails.
 * Returns 0 or a negative hipError_t from the launch. */
int mspack_hip_decode_batch_device(const mspack_hip_unit *d_units, const uint32_t *d_order,
                                   size_t n_units, const void *d_in, size_t in_bytes,
                                   void *d_out, size_t out_bytes, mspack_hip_result *d_results,
                                   void *d_frame_scratch, size_t n_frames_total, unsigned kind_mask,
                                   void *stream);
size_t mspack_hip_frame_scratch_bytes(size_t n_frames_total);

/* ---- host-buffer entry points (what the C drivers in mspack.h use) -------------------------------------
 * Same semantics with HOST pointers.  Each device keeps a persistent context (device arenas, pinned staging,
 * its streams and their events, grown or created on demand, never freed per call).  The batch is cut into up to
 * MSPACK_HIP_NCHUNKS (default 4) chunks of arena-contiguous units (each >= 8 MiB of input and >= 256 units; to the device the
 * chunks grow 1 : 1 : 2 : 4 -- the small ones get the launches going while the input is on its way, the last one fills the chip --,
 * to the host the first chunk is half a share so that the copy back, the long leg, begins early); chunk c's
 * input is copied on the copy-in stream, its launches (one per codec over a compact list of that codec's units) run on
 * a compute stream (four of them when the output stays on the device, two when it goes back to the host), its decoded
 * span goes back on the copy-out stream -- so copies overlap decode in both directions; the three roles' streams have
 * three different priorities, i.e. hardware queues of their own.  While that runs, mspack_hip_decode_batch page-locks
 * the caller's output buffer chunk by chunk (hipHostRegister; released before it returns) so that the copy-back is a
 * DMA that need not wait for the other streams; a buffer that cannot be registered is copied the ordinary way (MSPACK_HIP_PIN_OUT=0: always).
 * `units[i].frame_base` is filled in by the call.  Synchronous.  Bytes of the output arena
 * BETWEEN units that lie inside a copied span (alignment padding, the MSZIP slack) are unspecified afterwards.
 * Thread-safe; calls that target the same device are serialised. */
int mspack_hip_decode_batch(mspack_hip_unit *units, size_t n_units, const void *in, size_t in_bytes,
                            void *out, size_t out_bytes, mspack_hip_result *results);

/* ---- jobs: mspack_hip_decode_batch, handed over chunk by chunk while it runs -------------------------------
 * The reference's API is one extract() per file (cabd.c:1075-1214, chmd.c:906-1041); a driver that decodes a whole container in
 * one batch made the FIRST extract() wait for all of it and the caller's sys->write of file 1 begin only then.  A job is the same
 * batch on a thread of the library's own: _begin() returns at once, _wait_unit(job, i) returns as soon as the chunk that holds
 * unit i is through -- units[i]'s bytes are in `out`, results[i] is written (so are those of every unit of that chunk and of the
 * chunks before it: chunks finish in arena order) -- while the later chunks are still decoded and copied back; _end() waits for
 * the rest, frees the job and returns what mspack_hip_decode_batch would have returned.  `units`, `in`, `out` and `results` are
 * the job's until _end(): nothing of them may be written or freed before, and of `out` / `results` only what a successful
 * _wait_unit has covered may be read.  _wait_unit returns 0, or the batch's error code once the batch has failed before the
 * unit's chunk came through (mspack_hip_last_error() says what).  A batch that is not cut into chunks (interleaved outputs, a
 * small batch) is handed over whole: _wait_unit then returns with _end's answer.  _begin returns NULL when no job could be
 * started (no thread, MSPACK_HIP_JOBS=0): the caller then makes the synchronous call.  One device (the calling thread's current
 * one); calls for the same device -- jobs or not -- are serialised: a second batch waits until the job's batch has returned. */
typedef struct mspack_hip_job mspack_hip_job;
mspack_hip_job *mspack_hip_decode_batch_begin(mspack_hip_unit *units, size_t n_units, const void *in, size_t in_bytes,
                                              void *out, size_t out_bytes, mspack_hip_result *results);
int mspack_hip_job_wait_unit(mspack_hip_job *job, size_t i);
int mspack_hip_job_end(mspack_hip_job *job);

/* Host input, DEVICE output: as above, but the decoded bytes stay in the caller's device buffer `d_out` on
 * the current device (unit out_off relative to it; LZX DELTA reference data must already be there). */
int mspack_hip_decode_batch_to_device(mspack_hip_unit *units, size_t n_units, const void *in, size_t in_bytes,
                                      void *d_out, size_t out_bytes, mspack_hip_result *results);

/* Shard the same host batch across `n_devices` GPUs (devices 0..n_devices-1), one host thread per
 * device, no inter-device traffic: the units, in arena order, are cut into n_devices
 * contiguous ranges of about equal compressed size, so every device stages one contiguous span of each
 * arena.  (MSPACK_HIP_FORCE_SHARDS=k cuts into k shards even on fewer devices -- tests of this path.) */
int mspack_hip_decode_batch_multi(mspack_hip_unit *units, size_t n_units, const void *in,
                                  size_t in_bytes, void *out, size_t out_bytes,
                                  mspack_hip_result *results, int n_devices);

/* Free every persistent per-device context of this process (arenas, pinned staging, streams). */
void mspack_hip_release(void);

/* ---- process-wide defaults for the object API (mspack.h) ----------------------------------------------
 * mscab_decompressor has set_param (MSCABD_PARAM_HIP_DEVICES / _HIP_CACHE_MB); the CHM, OAB, SZDD and KWAJ
 * objects of the reference API have no parameter call, so their drivers use these defaults:
 *   devices  : GPUs a driver's batches are sharded over (mspack_hip_decode_batch_multi); initial value from
 *              the environment variable MSPACK_HIP_DEVICES, else 1
 *   cache_mb : upper bound, in MiB, on decoded bytes a CHM object keeps in host memory (least recently used
 *              batches are dropped and decoded again on demand); MSPACK_HIP_CACHE_MB, else 1024
 * Values < 1 are rejected (return -1). */
int mspack_hip_set_default_devices(int n_devices);
int mspack_hip_default_devices(void);
int mspack_hip_set_cache_mb(int mb);
int mspack_hip_cache_mb(void);

/* ---- timing helper for bench.py (HIP events on the launch stream) ------------------------------- */
/* Runs `iters` back-to-back device-resident decodes and returns the average milliseconds per
 * decode measured with hipEvents on `stream`; < 0 on error. */
double mspack_hip_time_batch_device(const mspack_hip_unit *d_units, const uint32_t *d_order,
                                    size_t n_units, const void *d_in, size_t in_bytes,
                                    void *d_out, size_t out_bytes, mspack_hip_result *d_results,
                                    void *d_frame_scratch, size_t n_frames_total, unsigned kind_mask,
                                    void *stream, int iters);

/* ---- diagnostics: where the host-buffer entry points spend their time ------------------------------------
 * Milliseconds summed over every call of mspack_hip_decode_batch / _to_device / _multi in this process since the last
 * reset: ms4[0] planning + growing the persistent buffers, ms4[1] issuing (the H2D copies of pageable input are
 * synchronous: this is mostly H2D) and launching, ms4[2] waiting for the kernels and the copies back, ms4[3] the number
 * of calls.  ms4 may be NULL; reset != 0 clears the sums after reading.  (bench infrastructure: csrc/bench/api_bench.c) */
void mspack_hip_host_path_stats(double *ms4, int reset);

/* ---- optional: page-lock a caller's input arena ---------------------------------------------------------
 * The host-buffer entry points copy their input with the runtime's pageable path unless the caller's buffer is
 * page-locked; for an arena that was just written (the C drivers' gather) that path runs at 5-6 GB/s.  mspack_hip_pin()
 * page-locks the WHOLE PAGES INSIDE [p, p + bytes) so that copies out of them are plain DMA; mspack_hip_unpin(p) releases
 * them -- before the memory is freed.  Never a byte outside the caller's range: the runtime treats every host address inside
 * a locked range as part of it, and a copy that starts inside one and ends behind it fails (hipErrorInvalidValue) -- with
 * outward rounding that happened to whatever the allocator had placed beside the arena (round 4's intermittent "GPU batch
 * decode failed").  A buffer that starts and ends on page boundaries is locked completely; the entry
 * points cut their copies at the boundaries of the ranges locked through this call.  A caller who locks memory with
 * hipHostRegister() directly must follow the same rule: the locked range has to contain every byte of the in / out buffers
 * it overlaps.  Both calls are advice: pin returns 0 when the range is locked now, a non-zero code when it is not (no device,
 * already locked, less than a page, the runtime refuses) and the entry points work either way.  (Output buffers are handled
 * inside mspack_hip_decode_batch itself, chunk by chunk; a buffer that is used for many calls is better locked by its owner.) */
int  mspack_hip_pin(const void *p, size_t bytes);
void mspack_hip_unpin(const void *p);

/* ---- optional: arenas out of the library's own page-locked memory -------------------------------------------
 * Page-locking a buffer costs about what the copy it speeds up costs, so a caller that builds a fresh arena for every batch
 * (the C drivers do) gains little from mspack_hip_pin().  mspack_hip_stage_alloc() hands out page-locked memory (page aligned,
 * every device of the process can copy to and from it) from blocks the library KEEPS when mspack_hip_stage_free() returns
 * them: locked once per process, not once per call.  At most MSPACK_HIP_PINNED_MB MiB in all (environment; default 1024; 0 =
 * never): a request beyond that -- or without a device -- returns NULL, and the caller allocates the ordinary way.
 * mspack_hip_release() frees the idle blocks.  (The drivers use it for arenas of a MiB and more when the mspack_system they
 * were given allocates with the library's own default allocator -- a caller who supplies an allocator gets every byte from it.) */
void *mspack_hip_stage_alloc(size_t bytes);
void  mspack_hip_stage_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
