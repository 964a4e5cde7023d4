// tests/hostcheck/hostcheck_main.cpp -- TEST INFRASTRUCTURE ONLY.  Scenarios for the host half of shim.hip under real ASan / TSan
// (see hostcheck_runtime.cpp).  usage: hostcheck <scenario> ; exit code 0 = outputs equal the plaintext, no violation of the
// runtime's modelled rules; the sanitizer's own exit code otherwise.  Environment (MSPACK_HIP_CHUNK_BYTES, _CHUNK_UNITS,
// _FORCE_SHARDS, ...) is the caller's: the library reads most of it once per process.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <string>
#include <thread>
#include <unistd.h>
#include "../../include/mspack_hip.h"
#include "../../include/mspack.h"
#include "../../libmspack_amd/csrc/corpus/corpus.h"

extern "C" int hostcheck_violations(void);
extern "C" int mspack_hip_pin(const void *p, size_t bytes);
extern "C" void mspack_hip_unpin(const void *p);
extern "C" void *mspack_hip_stage_alloc(size_t bytes);
extern "C" void mspack_hip_stage_free(void *p);

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "hostcheck: %s:%d: check failed: %s (%s)\n", __FILE__, __LINE__, #c, mspack_hip_last_error()); exit(3); } } while (0)

struct Batch {
  std::vector<uint8_t> plain, comp;
  std::vector<mspack_hip_unit> units;
  size_t out_bytes = 0;
};
// n LZX-21 units of `ub` bytes (a CHM's reset intervals: reset every two frames), back to back, outputs back to back
static Batch lzx_batch(uint64_t seed, int n, size_t ub, bool tables)
{
  Batch B;
  B.plain.resize((size_t) n * ub);
  B.comp.resize(mspk_lzx_bound(ub) * (size_t) n + 4096 + (tables ? (size_t) n * 64 : 0));
  std::vector<uint64_t> off(n), tab(n);
  std::vector<uint32_t> len(n);
  size_t tot = tables ? mspk_corpus_lzx_units_ft(seed, 0, MSPK_TEXT_MIX, n, ub, 21, nullptr, 4, B.plain.data(), B.comp.data(), B.comp.size() - 64, off.data(), len.data(), tab.data())
                      : mspk_corpus_lzx_units(seed, MSPK_TEXT_MIX, n, ub, 21, nullptr, 4, B.plain.data(), B.comp.data(), B.comp.size() - 64, off.data(), len.data());
  CHECK(tot != 0);
  B.comp.resize(tot + 64);
  B.units.resize(n);
  size_t fb = 0;
  for (int i = 0; i < n; i++) {
    mspack_hip_unit &u = B.units[i];
    memset(&u, 0, sizeof(u));
    u.in_off = off[i]; u.in_len = len[i] + 4; u.out_off = (uint64_t) i * ub; u.out_len = (uint32_t) ub;
    u.kind = MSPACK_HIP_KIND_LZX; u.window_bits = 21; u.reset_frames = 2; u.frame_base = (uint32_t) fb;
    if (tables) { u.flags |= MSPACK_HIP_UF_FRAME_TABLE; u.in_chunk = (uint32_t)(tab[i] / 4); }
    fb += ub / 32768 + 1;
  }
  B.out_bytes = (size_t) n * ub;
  return B;
}
static void decode_and_compare(Batch &B, const uint8_t *in, size_t in_bytes, uint8_t *out, size_t out_room, int multi)
{
  std::vector<mspack_hip_result> res(B.units.size());
  std::vector<mspack_hip_unit> u = B.units;
  int rc = multi ? mspack_hip_decode_batch_multi(u.data(), u.size(), in, in_bytes, out, out_room, res.data(), multi)
                 : mspack_hip_decode_batch(u.data(), u.size(), in, in_bytes, out, out_room, res.data());
  CHECK(rc == 0);
  for (size_t i = 0; i < res.size(); i++) CHECK(res[i].err == 0 && res[i].out_len == B.units[i].out_len);
  CHECK(memcmp(out, B.plain.data(), B.out_bytes) == 0);
}

// the caller page-locks PART of its arenas: every copy crosses a registration's boundary (tests/test_gpu_hostpath.py::
// test_copies_are_cut_at_pin_boundaries, at the size it has on the hardware).  With MSPACK_HIP_CHUNK_BYTES / _UNITS small the
// batch is cut into four chunks and the copy-back thread locks the output chunk by chunk BESIDE the caller's own lock.
static void scn_partial_pins(int reps)
{
  const int n = 256; const size_t ub = 65536;
  Batch B = lzx_batch(0x9191, n, ub, false);
  const size_t csz = B.comp.size();
  std::vector<uint8_t> arena(csz + 8192);
  const int shifts[3] = { 0, 100, 4000 };
  const double lo[3] = { 0.0, 0.25, 0.5 }, hi[3] = { 0.5, 0.75, 1.0 };
  for (int rep = 0; rep < reps; rep++)
    for (int k = 0; k < 3; k++) {
      uint8_t *a = arena.data() + shifts[k];
      memcpy(a, B.comp.data(), csz);
      std::vector<uint8_t> outv(B.out_bytes + 4096 + 64);
      uint8_t *out = outv.data() + shifts[k] % 64;
      const uint8_t *p_in = a + (size_t)(csz * lo[k]);
      uint8_t *p_out = out + (size_t)(B.out_bytes * lo[k]);
      mspack_hip_pin(p_in, (size_t)(csz * (hi[k] - lo[k])));
      mspack_hip_pin(p_out, (size_t)(B.out_bytes * (hi[k] - lo[k])));
      decode_and_compare(B, a, csz, out, B.out_bytes + 64, 0);
      mspack_hip_unpin(p_in); mspack_hip_unpin(p_out);
    }
}
// buffers of every kind of ownership: pageable, locked completely by the caller, out of the library's staging pool
static void scn_ownership(void)
{
  Batch B = lzx_batch(0x5151, 192, 65536, true);
  {
    std::vector<uint8_t> out(B.out_bytes + 64);
    decode_and_compare(B, B.comp.data(), B.comp.size(), out.data(), out.size(), 0);
  }
  {
    uint8_t *in = (uint8_t *) mspack_hip_stage_alloc(B.comp.size()), *out = (uint8_t *) mspack_hip_stage_alloc(B.out_bytes + 64);
    CHECK(in && out);
    memcpy(in, B.comp.data(), B.comp.size());
    decode_and_compare(B, in, B.comp.size(), out, B.out_bytes + 64, 0);
    mspack_hip_stage_free(in); mspack_hip_stage_free(out);
  }
  {
    void *in = nullptr, *out = nullptr;
    CHECK(posix_memalign(&in, 4096, (B.comp.size() + 4095) & ~(size_t) 4095) == 0 && posix_memalign(&out, 4096, (B.out_bytes + 64 + 4095) & ~(size_t) 4095) == 0);
    memcpy(in, B.comp.data(), B.comp.size());
    mspack_hip_pin(in, (B.comp.size() + 4095) & ~(size_t) 4095);
    mspack_hip_pin(out, (B.out_bytes + 64 + 4095) & ~(size_t) 4095);
    decode_and_compare(B, (const uint8_t *) in, B.comp.size(), (uint8_t *) out, B.out_bytes + 64, 0);
    mspack_hip_unpin(in); mspack_hip_unpin(out);
    free(in); free(out);
  }
  mspack_hip_release();
  {
    std::vector<uint8_t> out(B.out_bytes + 64);
    decode_and_compare(B, B.comp.data(), B.comp.size(), out.data(), out.size(), 0);     // (contexts come back after a release)
  }
}
// the sharded entry point (one host thread per shard) and two application threads inside the library at once
static void scn_shards_and_threads(void)
{
  Batch B = lzx_batch(0x7171, 300, 65536, true);
  std::vector<uint8_t> out(B.out_bytes + 64);
  decode_and_compare(B, B.comp.data(), B.comp.size(), out.data(), out.size(), 3);
  Batch C = lzx_batch(0x7272, 200, 65536, false);
  std::vector<uint8_t> out2(C.out_bytes + 64);
  std::thread t1([&]() { for (int i = 0; i < 3; i++) decode_and_compare(B, B.comp.data(), B.comp.size(), out.data(), out.size(), 0); });
  std::thread t2([&]() { for (int i = 0; i < 3; i++) decode_and_compare(C, C.comp.data(), C.comp.size(), out2.data(), out2.size(), 0); });
  t1.join(); t2.join();
}

// ---- the object API: decompressors created and destroyed over and over in one process (tests/test_gpu_hostpath.py::
// test_many_decompressors_one_process): a CHM of 256 reset intervals and a cabinet of LZX folders, from memory ----
struct MemFile { const uint8_t *data; size_t size, pos; std::vector<uint8_t> *sink; };
struct MemSys { struct mspack_system sys; const uint8_t *img; size_t img_size; std::vector<uint8_t> out; };
static struct mspack_file *m_open(struct mspack_system *self, const char *name, int mode) {
  MemSys *M = (MemSys *) self;
  MemFile *f = new MemFile();
  if (mode == MSPACK_SYS_OPEN_READ) { f->data = M->img; f->size = M->img_size; f->pos = 0; f->sink = nullptr; }
  else { f->data = nullptr; f->size = 0; f->pos = 0; f->sink = &M->out; M->out.clear(); }
  (void) name;
  return (struct mspack_file *) f;
}
static void m_close(struct mspack_file *file) { delete (MemFile *) file; }
static int m_read(struct mspack_file *file, void *buf, int bytes) {
  MemFile *f = (MemFile *) file;
  size_t n = f->size - f->pos; if (n > (size_t) bytes) n = (size_t) bytes;
  memcpy(buf, f->data + f->pos, n); f->pos += n;
  return (int) n;
}
static int m_write(struct mspack_file *file, void *buf, int bytes) {
  MemFile *f = (MemFile *) file;
  f->sink->insert(f->sink->end(), (uint8_t *) buf, (uint8_t *) buf + bytes);
  return bytes;
}
static int m_seek(struct mspack_file *file, off_t off, int mode) {
  MemFile *f = (MemFile *) file;
  off_t base = mode == MSPACK_SYS_SEEK_START ? 0 : (mode == MSPACK_SYS_SEEK_CUR ? (off_t) f->pos : (off_t) f->size);
  if (base + off < 0 || (size_t)(base + off) > f->size) return -1;
  f->pos = (size_t)(base + off);
  return 0;
}
static off_t m_tell(struct mspack_file *file) { return (off_t)((MemFile *) file)->pos; }
static void m_msg(struct mspack_file *, const char *fmt, ...) { (void) fmt; }
static void *m_alloc(struct mspack_system *, size_t n) { return malloc(n); }
static void m_free(void *p) { free(p); }
static void m_copy(void *s, void *d, size_t n) { memcpy(d, s, n); }
static void memsys_init(MemSys &M, const std::vector<uint8_t> &img) {
  M.sys.open = m_open; M.sys.close = m_close; M.sys.read = m_read; M.sys.write = m_write; M.sys.seek = m_seek; M.sys.tell = m_tell;
  M.sys.message = m_msg; M.sys.alloc = m_alloc; M.sys.free = m_free; M.sys.copy = m_copy; M.sys.null_ptr = nullptr;
  M.img = img.data(); M.img_size = img.size();
}
static void scn_lifetimes(int reps)
{
  // a CHM: one LZX-21 stream of 256 intervals of two frames
  const size_t N = (size_t) 256 * 65536, nfr = N / 32768;
  std::vector<uint8_t> plain(N), lzx(mspk_lzx_bound(N));
  mspk_gen_plaintext(0xC4, MSPK_TEXT_MIX, plain.data(), N);
  std::vector<uint64_t> foff(nfr + 1);
  const size_t lz = mspk_lzx_encode(plain.data(), N, 21, 2, nullptr, lzx.data(), lzx.size(), foff.data());
  CHECK(lz != 0);
  const int nf = 37;
  std::vector<std::string> names(nf);
  std::vector<mspk_chm_file> files(nf);
  for (int i = 0; i < nf; i++) {
    names[i] = "/doc" + std::to_string(1000 + i) + ".html";
    files[i].name = names[i].c_str(); files[i].offset = (uint64_t) i * (N / nf); files[i].length = N / nf - 17;
  }
  std::vector<uint8_t> chm(mspk_chm_bound(lz, nfr, nf));
  const size_t cs = mspk_chm_write(lzx.data(), lz, foff.data(), nfr, N, 21, 2, files.data(), nf, chm.data(), chm.size());
  CHECK(cs != 0);
  chm.resize(cs);
  // a cabinet: 12 LZX-18 folders of 12 frames, one file each
  const int nfold = 12; const size_t fsz = 12 * 32768;
  std::vector<uint8_t> cplain((size_t) nfold * fsz);
  mspk_gen_plaintext(0xCAB, MSPK_TEXT_MIX, cplain.data(), cplain.size());
  std::vector<std::vector<uint8_t>> fdata(nfold);
  std::vector<std::vector<uint32_t>> bc(nfold), bu(nfold);
  std::vector<mspk_cab_folder> folders(nfold);
  std::vector<mspk_cab_file> cfiles(nfold);
  std::vector<std::string> cnames(nfold);
  for (int i = 0; i < nfold; i++) {
    fdata[i].resize(mspk_lzx_bound(fsz));
    std::vector<uint64_t> fo(fsz / 32768 + 1);
    const size_t z = mspk_lzx_encode(cplain.data() + (size_t) i * fsz, fsz, 18, 0, nullptr, fdata[i].data(), fdata[i].size(), fo.data());
    CHECK(z != 0);
    for (size_t b = 0; b < fsz / 32768; b++) { bc[i].push_back((uint32_t)(fo[b + 1] - fo[b])); bu[i].push_back(32768); }
    folders[i].comp_type = 3 | (18 << 8); folders[i].data = fdata[i].data(); folders[i].block_comp = bc[i].data(); folders[i].block_uncomp = bu[i].data();
    folders[i].n_blocks = (int)(fsz / 32768);
    cnames[i] = "g" + std::to_string(i) + ".bin";
    cfiles[i].name = cnames[i].c_str(); cfiles[i].length = (uint32_t) fsz; cfiles[i].folder_offset = 0; cfiles[i].folder_index = (uint16_t) i;
  }
  std::vector<uint8_t> cab((size_t) nfold * (fsz + 65536));
  const size_t cz = mspk_cab_write(folders.data(), nfold, cfiles.data(), nfold, cab.data(), cab.size());
  CHECK(cz != 0);
  cab.resize(cz);
  // (the library's default system reads and writes real files)
  const std::string tdir = std::string("/tmp/hostcheck_") + std::to_string((long) getpid());
  const std::string f_chm = tdir + ".chm", f_cab = tdir + ".cab", f_out = tdir + ".out";
  { FILE *f = fopen(f_chm.c_str(), "wb"); CHECK(f && fwrite(chm.data(), 1, chm.size(), f) == chm.size()); fclose(f); }
  { FILE *f = fopen(f_cab.c_str(), "wb"); CHECK(f && fwrite(cab.data(), 1, cab.size(), f) == cab.size()); fclose(f); }
  auto slurp = [&](std::vector<uint8_t> &v) {
    FILE *f = fopen(f_out.c_str(), "rb"); CHECK(f);
    v.resize(N); v.resize(fread(v.data(), 1, N, f)); fclose(f);
  };
  for (int it = 0; it < reps; it++) {
    for (int own_alloc = 0; own_alloc < 2; own_alloc++) {
      // (own_alloc 0: the library's default system -- arenas out of the page-locked staging pool; 1: the caller's system and
      // allocator -- arenas from malloc, page-locked around the batch by the driver)
      MemSys M; memsys_init(M, chm);
      {
        struct mschm_decompressor *d = mspack_create_chm_decompressor(own_alloc ? &M.sys : nullptr);
        CHECK(d);
        struct mschmd_header *h = d->open(d, own_alloc ? "mem.chm" : f_chm.c_str());
        CHECK(h);
        int k = 0;
        for (struct mschmd_file *f = h->files; f; f = f->next, k++) {
          if (k % 9 != it % 9) continue;
          CHECK(d->extract(d, f, own_alloc ? "out" : f_out.c_str()) == MSPACK_ERR_OK);
          if (!own_alloc) slurp(M.out);
          const size_t o = (size_t)(k) * (N / nf);
          CHECK(M.out.size() == N / nf - 17 && memcmp(M.out.data(), plain.data() + o, M.out.size()) == 0);
        }
        d->close(d, h);
        mspack_destroy_chm_decompressor(d);
      }
      MemSys C; memsys_init(C, cab);
      {
        struct mscab_decompressor *d = mspack_create_cab_decompressor(own_alloc ? &C.sys : nullptr);
        CHECK(d);
        struct mscabd_cabinet *c = d->open(d, own_alloc ? "mem.cab" : f_cab.c_str());
        CHECK(c);
        int k = 0;
        for (struct mscabd_file *f = c->files; f; f = f->next, k++) {
          if (k % 4 != it % 4) continue;
          CHECK(d->extract(d, f, own_alloc ? "out" : f_out.c_str()) == MSPACK_ERR_OK);
          if (!own_alloc) slurp(C.out);
          CHECK(C.out.size() == fsz && memcmp(C.out.data(), cplain.data() + (size_t) k * fsz, fsz) == 0);
        }
        d->close(d, c);
        mspack_destroy_cab_decompressor(d);
      }
    }
    // heap traffic between lifetimes (blocks that land where the freed arenas were)
    std::vector<std::vector<uint8_t>> junk;
    for (int k = 0; k < 100; k++) junk.emplace_back(1000 + 37 * k);
  }
  remove(f_chm.c_str()); remove(f_cab.c_str()); remove(f_out.c_str());
}

// jobs (mspack_hip.h: mspack_hip_decode_batch_begin): the caller reads a unit's bytes and result the moment _wait_unit has covered it,
// while the library's threads still decode and copy the later chunks -- what TSan is here for; a job ended without a wait; a
// synchronous call of the same thread and one of another thread beside a running job (serialised inside); units waited out of order
static void scn_jobs(int reps)
{
  Batch B = lzx_batch(0x3131, 320, 65536, true);
  Batch C = lzx_batch(0x3232, 100, 65536, false);
  const size_t n = B.units.size(), ub = 65536;
  for (int rep = 0; rep < reps; rep++) {
    const bool staged = rep & 1;                             // (page-locked arenas out of the library's pool / pageable ones)
    std::vector<uint8_t> outv(B.out_bytes + 64 + 4096);
    uint8_t *in = staged ? (uint8_t *) mspack_hip_stage_alloc(B.comp.size()) : B.comp.data();
    uint8_t *out = staged ? (uint8_t *) mspack_hip_stage_alloc(B.out_bytes + 64) : outv.data() + 100 * rep;
    CHECK(in && out);
    if (staged) memcpy(in, B.comp.data(), B.comp.size());
    std::vector<mspack_hip_result> res(n);
    std::vector<mspack_hip_unit> u = B.units;
    mspack_hip_job *job = mspack_hip_decode_batch_begin(u.data(), n, in, B.comp.size(), out, B.out_bytes + 64, res.data());
    CHECK(job);
    std::thread other;
    std::vector<uint8_t> out2(C.out_bytes + 64);
    if (rep % 3 == 1) other = std::thread([&]() { decode_and_compare(C, C.comp.data(), C.comp.size(), out2.data(), out2.size(), 0); });
    if (rep % 3 == 2) {
      // out of order: the last unit first (everything is there then), then the rest
      CHECK(mspack_hip_job_wait_unit(job, n - 1) == 0);
      CHECK(memcmp(out, B.plain.data(), B.out_bytes) == 0);
    }
    for (size_t i = 0; i < n; i++) {
      CHECK(mspack_hip_job_wait_unit(job, i) == 0);
      CHECK(res[i].err == 0 && res[i].out_len == ub);
      CHECK(memcmp(out + i * ub, B.plain.data() + i * ub, ub) == 0);
      if (rep % 3 == 0 && i == n / 2) decode_and_compare(C, C.comp.data(), C.comp.size(), out2.data(), out2.size(), 0);   // (waits for the job's batch)
    }
    CHECK(mspack_hip_job_wait_unit(job, n) != 0);            // (no such unit)
    CHECK(mspack_hip_job_end(job) == 0);
    if (other.joinable()) other.join();
    // ended without a wait: everything is there afterwards
    memset(out, 0, B.out_bytes);
    u = B.units;
    job = mspack_hip_decode_batch_begin(u.data(), n, in, B.comp.size(), out, B.out_bytes + 64, res.data());
    CHECK(job && mspack_hip_job_end(job) == 0);
    CHECK(memcmp(out, B.plain.data(), B.out_bytes) == 0);
    // a batch that fails as a whole (a unit outside the arena): the waiters hear of it, _end returns the code
    u = B.units; u[n - 1].in_off = B.comp.size() + 5;
    job = mspack_hip_decode_batch_begin(u.data(), n, in, B.comp.size(), out, B.out_bytes + 64, res.data());
    CHECK(job);
    CHECK(mspack_hip_job_wait_unit(job, 0) != 0);
    CHECK(mspack_hip_job_end(job) != 0);
    if (staged) { mspack_hip_stage_free(in); mspack_hip_stage_free(out); }
  }
}

int main(int argc, char **argv)
{
  const std::string s = argc > 1 ? argv[1] : "";
  if (s == "partial_pins") scn_partial_pins(argc > 2 ? atoi(argv[2]) : 2);
  else if (s == "ownership") scn_ownership();
  else if (s == "shards_threads") scn_shards_and_threads();
  else if (s == "lifetimes") scn_lifetimes(argc > 2 ? atoi(argv[2]) : 4);
  else if (s == "jobs") scn_jobs(argc > 2 ? atoi(argv[2]) : 6);
  else { fprintf(stderr, "usage: hostcheck partial_pins|ownership|shards_threads|lifetimes [reps]\n"); return 2; }
  mspack_hip_release();
  if (hostcheck_violations()) { fprintf(stderr, "hostcheck: %d violation(s) of the runtime's modelled rules\n", hostcheck_violations()); return 4; }
  printf("HOSTCHECK_OK %s\n", s.c_str());
  return 0;
}
