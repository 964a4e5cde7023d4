/* system.c -- default stdio-backed struct mspack_system, version and self-test entry points.
 * Interface: reference mspack.h:191-262, 285-455; behaviour follows libmspack/mspack/system.c
 * (open modes, read/write returning -1 on stream error, alloc/free/copy). */
#include <stdio.h>
#include <stdlib.h>
#include <stdarg.h>
#include "host_common.h"
#ifdef __linux__
#include <sys/mman.h>
#endif

/* A batch's input / output arena: sys->alloc'd memory whose first byte lies on a page boundary and whose room is whole pages,
 * so that mspack_hip_pin() can page-lock ALL of it and nothing else -- a lock never holds a byte that is not the arena's
 * (include/mspack_hip.h: mspack_hip_pin; DESIGN.md sec. 8h: what happened when one did).  The block sys->alloc returned is
 * remembered in the word below the arena; mspack_arena_free() hands that block back to sys->free. */
#define ARENA_PAGE ((size_t) 4096)
#define ARENA_TAG_SYS    0x41726e61u          /* the block came from sys->alloc */
#define ARENA_TAG_STAGE  0x53746167u          /* ... from the library's page-locked staging memory (mspack_hip_stage_alloc) */
struct arena_hdr { void *raw; unsigned int tag, pad; };     /* lies right below the arena */
static void *std_alloc(struct mspack_system *self, size_t bytes);
void *mspack_arena_alloc(struct mspack_system *sys, size_t bytes) {
  const size_t room = (bytes + ARENA_PAGE - 1) & ~(ARENA_PAGE - 1);
  unsigned char *raw = NULL, *p;
  struct arena_hdr h;
  if (room < bytes || room + 2 * ARENA_PAGE < room) return NULL;
  h.tag = ARENA_TAG_SYS; h.pad = 0;
  /* big arenas of a caller who did not bring an allocator: page-locked memory the library keeps between batches (mspack_hip.h) */
  if (room >= ((size_t) 1 << 20) && sys->alloc == &std_alloc && (raw = (unsigned char *) mspack_hip_stage_alloc(room + ARENA_PAGE)))
    h.tag = ARENA_TAG_STAGE;
  if (!raw && !(raw = (unsigned char *) sys->alloc(sys, room + ARENA_PAGE + sizeof(struct arena_hdr)))) return NULL;
  p = (unsigned char *)(((uintptr_t) raw + sizeof(struct arena_hdr) + ARENA_PAGE - 1) & ~(uintptr_t)(ARENA_PAGE - 1));
  h.raw = raw;
  memcpy(p - sizeof(h), &h, sizeof(h));
#if defined(__linux__) && defined(MADV_HUGEPAGE)
  if (h.tag == ARENA_TAG_SYS) {
    static int on = -1;                 /* MSPACK_ARENA_HUGEPAGES=0 turns the advice off */
    if (on < 0) { const char *e = getenv("MSPACK_ARENA_HUGEPAGES"); on = !(e && e[0] == '0'); }
    if (on && room >= ((size_t) 4 << 20)) {
      const uintptr_t H = (uintptr_t) 2 << 20;
      const uintptr_t a = ((uintptr_t) p + H - 1) & ~(H - 1), b = ((uintptr_t) p + room) & ~(H - 1);
      if (b > a) (void) madvise((void *) a, (size_t)(b - a), MADV_HUGEPAGE);       /* (whole 2 MiB pages inside the block only) */
    }
  }
#endif
  return p;
}
void mspack_arena_free(struct mspack_system *sys, void *arena) {
  struct arena_hdr h;
  if (!arena) return;
  memcpy(&h, (unsigned char *) arena - sizeof(h), sizeof(h));
  if (h.tag == ARENA_TAG_STAGE) mspack_hip_stage_free(h.raw);
  else sys->free(h.raw);
}
int mspack_arena_is_locked(const void *arena) {
  struct arena_hdr h;
  if (!arena) return 0;
  memcpy(&h, (const unsigned char *) arena - sizeof(h), sizeof(h));
  return h.tag == ARENA_TAG_STAGE;
}
size_t mspack_arena_room(size_t bytes) { return (bytes + ARENA_PAGE - 1) & ~(ARENA_PAGE - 1); }

int mspack_version(int entity) {
  switch (entity) {
  case MSPACK_VER_MSCHMD: case MSPACK_VER_MSCABD: case MSPACK_VER_MSOABD:
  case MSPACK_VER_MSSZDDD: case MSPACK_VER_MSKWAJD:
    return 2;                       /* structure revisions this library is layout-compatible with */
  case MSPACK_VER_LIBRARY: case MSPACK_VER_SYSTEM:
    return 1;
  case MSPACK_VER_MSCABC: case MSPACK_VER_MSCHMC: case MSPACK_VER_MSLITD: case MSPACK_VER_MSLITC:
  case MSPACK_VER_MSHLPD: case MSPACK_VER_MSHLPC: case MSPACK_VER_MSSZDDC: case MSPACK_VER_MSKWAJC:
  case MSPACK_VER_MSOABC:
    return 0;                       /* not provided by this library */
  }
  return -1;
}

int mspack_sys_selftest_internal(int offt_size) {
  return (sizeof(off_t) == (size_t) offt_size) ? MSPACK_ERR_OK : MSPACK_ERR_SEEK;
}

int mspack_valid_system(struct mspack_system *sys) {
  return sys && sys->open && sys->close && sys->read && sys->write && sys->seek && sys->tell &&
         sys->message && sys->alloc && sys->free && sys->copy && (sys->null_ptr == NULL);
}

int mspack_sys_filelen(struct mspack_system *system, struct mspack_file *file, off_t *length) {
  off_t here;
  if (!system || !file || !length) return MSPACK_ERR_OPEN;
  here = system->tell(file);
  if (system->seek(file, 0, MSPACK_SYS_SEEK_END)) return MSPACK_ERR_SEEK;
  *length = system->tell(file);
  if (system->seek(file, here, MSPACK_SYS_SEEK_START)) return MSPACK_ERR_SEEK;
  return MSPACK_ERR_OK;
}

struct stdio_file { FILE *fp; const char *name; };

static struct mspack_file *std_open(struct mspack_system *self, const char *filename, int mode) {
  static const char *modes[4] = { "rb", "wb", "r+b", "ab" };
  struct stdio_file *f;
  (void) self;
  if (mode < 0 || mode > 3 || !filename) return NULL;
  if (!(f = (struct stdio_file *) malloc(sizeof(*f)))) return NULL;
  f->name = filename;
  if (!(f->fp = fopen(filename, modes[mode]))) { free(f); return NULL; }
  return (struct mspack_file *) f;
}
static void std_close(struct mspack_file *file) {
  struct stdio_file *f = (struct stdio_file *) file;
  if (f) { fclose(f->fp); free(f); }
}
static int std_read(struct mspack_file *file, void *buffer, int bytes) {
  struct stdio_file *f = (struct stdio_file *) file;
  if (f && buffer && bytes >= 0) {
    size_t n = fread(buffer, 1, (size_t) bytes, f->fp);
    if (!ferror(f->fp)) return (int) n;
  }
  return -1;
}
static int std_write(struct mspack_file *file, void *buffer, int bytes) {
  struct stdio_file *f = (struct stdio_file *) file;
  if (f && buffer && bytes >= 0) {
    size_t n = fwrite(buffer, 1, (size_t) bytes, f->fp);
    if (!ferror(f->fp)) return (int) n;
  }
  return -1;
}
static int std_seek(struct mspack_file *file, off_t offset, int mode) {
  struct stdio_file *f = (struct stdio_file *) file;
  int whence;
  if (!f) return -1;
  if (mode == MSPACK_SYS_SEEK_START) whence = SEEK_SET;
  else if (mode == MSPACK_SYS_SEEK_CUR) whence = SEEK_CUR;
  else if (mode == MSPACK_SYS_SEEK_END) whence = SEEK_END;
  else return -1;
  return fseeko(f->fp, offset, whence);
}
static off_t std_tell(struct mspack_file *file) {
  struct stdio_file *f = (struct stdio_file *) file;
  return f ? (off_t) ftello(f->fp) : 0;
}
static void std_message(struct mspack_file *file, const char *format, ...) {
  va_list ap;
  if (file) fprintf(stderr, "%s: ", ((struct stdio_file *) file)->name);
  va_start(ap, format); vfprintf(stderr, format, ap); va_end(ap);
  fputc('\n', stderr); fflush(stderr);
}
static void *std_alloc(struct mspack_system *self, size_t bytes) { (void) self; return malloc(bytes); }
static void std_free(void *p) { free(p); }
static void std_copy(void *src, void *dest, size_t bytes) { memcpy(dest, src, bytes); }

static struct mspack_system std_system = {
  &std_open, &std_close, &std_read, &std_write, &std_seek, &std_tell, &std_message,
  &std_alloc, &std_free, &std_copy, NULL
};
struct mspack_system *mspack_default_system = &std_system;

/* ---- process-wide driver defaults (include/mspack_hip.h) ------------------------------------------------ */
static int dflt_devices = 0, dflt_cache_mb = 0;
static int env_int(const char *name, int fallback) {
  const char *e = getenv(name);
  int v = e ? atoi(e) : 0;
  return v >= 1 ? v : fallback;
}
int mspack_hip_set_default_devices(int n) { if (n < 1) return -1; dflt_devices = n; return 0; }
int mspack_hip_default_devices(void) { if (!dflt_devices) dflt_devices = env_int("MSPACK_HIP_DEVICES", 1); return dflt_devices; }
int mspack_hip_set_cache_mb(int mb) { if (mb < 1) return -1; dflt_cache_mb = mb; return 0; }
int mspack_hip_cache_mb(void) { if (!dflt_cache_mb) dflt_cache_mb = env_int("MSPACK_HIP_CACHE_MB", 1024); return dflt_cache_mb; }
