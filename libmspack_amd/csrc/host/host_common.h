/* host_common.h -- shared bits of the plain-C host drivers (cabd.c, chmd.c, system.c).
 * The drivers never touch HIP: they gather units, call the C ABI of include/mspack_hip.h and
 * serve slices of the decoded result through the caller's struct mspack_system. */
#ifndef MSPACK_AMD_HOST_COMMON_H
#define MSPACK_AMD_HOST_COMMON_H
#define _FILE_OFFSET_BITS 64
#include <stdint.h>
#include <string.h>
#include "mspack.h"
#include "mspack_hip.h"

extern struct mspack_system *mspack_default_system;
int mspack_valid_system(struct mspack_system *sys);
int mspack_sys_filelen(struct mspack_system *system, struct mspack_file *file, off_t *length);
/* A batch's input / output arena: memory from sys->alloc that starts on a page boundary and is whole pages long
 * (mspack_arena_room(bytes) of them), so that mspack_hip_pin(arena, mspack_arena_room(bytes)) page-locks exactly the arena;
 * released with mspack_arena_free(), never with sys->free directly.  On Linux, arenas of several MiB also carry the advice
 * that the fresh pages behind them may be huge ones (madvise MADV_HUGEPAGE: the arenas are written once, front to back, by
 * sys->read or the copy back from the device, and 4 KiB faults were most of that time). */
void *mspack_arena_alloc(struct mspack_system *sys, size_t bytes);
void mspack_arena_free(struct mspack_system *sys, void *arena);
size_t mspack_arena_room(size_t bytes);
/* does the arena already lie in page-locked memory (the library's staging blocks)?  Then there is nothing to mspack_hip_pin() */
int mspack_arena_is_locked(const void *arena);

static inline unsigned int rd_le16(const unsigned char *p) { return (unsigned int) p[0] | ((unsigned int) p[1] << 8); }
static inline unsigned int rd_le32(const unsigned char *p) {
  return (unsigned int) p[0] | ((unsigned int) p[1] << 8) | ((unsigned int) p[2] << 16) | ((unsigned int) p[3] << 24);
}
static inline unsigned int rd_be32(const unsigned char *p) {
  return (unsigned int) p[3] | ((unsigned int) p[2] << 8) | ((unsigned int) p[1] << 16) | ((unsigned int) p[0] << 24);
}
static inline int64_t rd_le64(const unsigned char *p) {
  return (int64_t)((uint64_t) rd_le32(p) | ((uint64_t) rd_le32(p + 4) << 32));
}

/* write `n` bytes through sys->write in pieces of at most 32 KiB (the reference's codecs hand
 * over one frame at a time); returns MSPACK_ERR_OK / MSPACK_ERR_WRITE */
static inline int write_slice(struct mspack_system *sys, struct mspack_file *fh, const unsigned char *p, size_t n) {
  while (n) {
    int run = n > 32768 ? 32768 : (int) n;
    if (sys->write(fh, (void *) p, run) != run) return MSPACK_ERR_WRITE;
    p += run; n -= (size_t) run;
  }
  return MSPACK_ERR_OK;
}
#endif
