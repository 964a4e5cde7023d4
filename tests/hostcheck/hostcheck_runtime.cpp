// tests/hostcheck/hostcheck_runtime.cpp -- TEST INFRASTRUCTURE ONLY (never part of the product).
//
// The HOST half of libmspack_amd/csrc/hip/shim.hip -- the chunk planner, the page-lock registry and the cut copies, the copy-back
// thread, the shard threads, the staging pool -- under REAL AddressSanitizer / ThreadSanitizer (none of
// it had ever run under a sanitizer; the wavefront emulator borrows TSan's hooks for lane scheduling and links no runtime).
// shim.hip is compiled for the host with -DMSPACK_HOST_CHECK on top of tests/emu's stand-in <hip/hip_runtime.h>; this file is the
// runtime behind it:
//   * streams are REAL queues with a worker thread each: hipMemcpyAsync / hipMemsetAsync / a launch return at once and run later,
//     events order streams -- so a buffer that is touched before its copy has run, or freed under one, is a race TSan / ASan sees;
//   * no kernel runs: a launch's place in its stream is taken by the CPU stand-in (tests/csrc/batch_standin.c, one oracle call
//     per unit), so results can still be compared with the plaintext;
//   * the page-lock rules the real runtime was found to have are MODELLED and counted as violations:
//       - hipHostRegister of a range that overlaps a live registration         (refused + violation)
//       - hipHostUnregister of an address that is no registration's start       (refused + violation)
//       - a copy whose host side starts inside a registration and ends outside  (hipErrorInvalidValue + violation)
//     `hostcheck_violations()` must be 0 at the end of every scenario: the library never ASKS for any of these;
//   * a copy to or from host memory that is not page-locked blocks its caller until it has run (as the real runtime's does).
#include <hip/hip_runtime.h>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <thread>
#include <vector>
#include "../../include/mspack_hip.h"

extern "C" int mspack_standin_decode_units(const mspack_hip_unit *units, const uint32_t *order, size_t n_units, const void *in, size_t in_bytes,
                                           void *out, size_t out_bytes, mspack_hip_result *results);
extern "C" const char *mspack_standin_last_error(void);

// ---- the lanes' side of the emulator does not exist here ---------------------------------------------------------------
static void no_kernels(const char *what) { fprintf(stderr, "hostcheck: %s called -- no kernel code may run in this build\n", what); abort(); }
extern "C" {
emu_idx emu_thread_idx(void) { no_kernels("threadIdx"); return emu_idx{0, 0, 0}; }
emu_idx emu_block_idx(void) { no_kernels("blockIdx"); return emu_idx{0, 0, 0}; }
emu_idx emu_grid_dim(void) { no_kernels("gridDim"); return emu_idx{0, 0, 0}; }
unsigned emu_readlane(unsigned, unsigned) { no_kernels("readlane"); return 0; }
unsigned emu_readfirstlane(unsigned) { no_kernels("readfirstlane"); return 0; }
unsigned long long emu_ballot(int) { no_kernels("ballot"); return 0; }
unsigned emu_bpermute(unsigned, unsigned) { no_kernels("bpermute"); return 0; }
unsigned emu_dpp(unsigned, unsigned, unsigned, unsigned, unsigned, int) { no_kernels("dpp"); return 0; }
void emu_sleep(void) { no_kernels("s_sleep"); }
unsigned long long emu_clock(void) { return 0; }
}
void emu_launch(dim3, dim3, std::function<void()>) { no_kernels("emu_launch"); }
void emu_test_delay(void) {}

// ---- violations ------------------------------------------------------------------------------------------------------------
static std::atomic<int> g_violations{0};
static void violation(const char *what, const void *p, size_t n) {
  g_violations++;
  fprintf(stderr, "hostcheck VIOLATION: %s (%p, %zu bytes)\n", what, p, n);
}
extern "C" int hostcheck_violations(void) { return g_violations.load(); }

// ---- registrations (hipHostRegister) and the runtime's own page-locked blocks (hipHostMalloc) --------------------------------
struct Reg { uintptr_t a, b; bool own; };
static std::mutex g_reg_mu;
static std::map<uintptr_t, Reg> g_regs;          // by start
static const Reg *reg_holding(uintptr_t p) {     // (g_reg_mu held)
  auto it = g_regs.upper_bound(p);
  if (it == g_regs.begin()) return nullptr;
  --it;
  return p < it->second.b ? &it->second : nullptr;
}
hipError_t hipHostRegister(void *p, size_t n, unsigned) {
  const uintptr_t a = (uintptr_t) p, b = a + n;
  if (!p || !n) return hipErrorInvalidValue;
  std::lock_guard<std::mutex> g(g_reg_mu);
  for (const auto &kv : g_regs)
    if (kv.second.a < b && a < kv.second.b) { violation("hipHostRegister of a range that overlaps a live registration", p, n); return hipErrorInvalidValue; }
  g_regs[a] = Reg{ a, b, false };
  return hipSuccess;
}
hipError_t hipHostUnregister(void *p) {
  std::lock_guard<std::mutex> g(g_reg_mu);
  auto it = g_regs.find((uintptr_t) p);
  if (it == g_regs.end() || it->second.own) { violation("hipHostUnregister of an address that is no registration's start", p, 0); return hipErrorInvalidValue; }
  g_regs.erase(it);
  return hipSuccess;
}
hipError_t hipPointerGetAttributes(hipPointerAttribute_t *a, const void *p) {
  std::lock_guard<std::mutex> g(g_reg_mu);
  a->type = reg_holding((uintptr_t) p) ? hipMemoryTypeHost : hipMemoryTypeUnregistered;
  return hipSuccess;
}
hipError_t emu_hipHostMalloc(void **p, size_t n) {
  void *q = nullptr;
  if (posix_memalign(&q, 4096, n ? n : 1)) return hipErrorOutOfMemory;
  std::lock_guard<std::mutex> g(g_reg_mu);
  g_regs[(uintptr_t) q] = Reg{ (uintptr_t) q, (uintptr_t) q + (n ? n : 1), true };
  *p = q;
  return hipSuccess;
}

// ---- device memory: plain heap blocks (ASan guards their ends) ----------------------------------------------------------------
static std::mutex g_dev_mu;
static std::map<uintptr_t, size_t> g_dev;
hipError_t emu_hipMalloc(void **p, size_t n) {
  void *q = malloc(n ? n : 1);
  if (!q) return hipErrorOutOfMemory;
  memset(q, 0xA5, n);
  std::lock_guard<std::mutex> g(g_dev_mu);
  g_dev[(uintptr_t) q] = n;
  *p = q;
  return hipSuccess;
}
static bool is_device(const void *p) {
  std::lock_guard<std::mutex> g(g_dev_mu);
  auto it = g_dev.upper_bound((uintptr_t) p);
  if (it == g_dev.begin()) return false;
  --it;
  return (uintptr_t) p < it->first + it->second;
}

// ---- streams -----------------------------------------------------------------------------------------------------------------
struct emu_event_ { std::mutex mu; std::condition_variable cv; unsigned long long recorded = 0, done = 0; std::chrono::steady_clock::time_point t; };
struct emu_stream_ {
  std::mutex mu; std::condition_variable cv;
  std::deque<std::function<void()>> q;
  bool busy = false, quit = false;
  std::thread th;
  emu_stream_() { th = std::thread([this]() { run(); }); }
  void run() {
    for (;;) {
      std::function<void()> f;
      {
        std::unique_lock<std::mutex> l(mu);
        cv.wait(l, [&]() { return quit || !q.empty(); });
        if (q.empty()) return;
        f = std::move(q.front()); q.pop_front(); busy = true;
      }
      f();
      { std::lock_guard<std::mutex> l(mu); busy = false; }
      cv.notify_all();
    }
  }
  void push(std::function<void()> f) { { std::lock_guard<std::mutex> l(mu); q.push_back(std::move(f)); } cv.notify_all(); }
  void drain() { std::unique_lock<std::mutex> l(mu); cv.wait(l, [&]() { return q.empty() && !busy; }); }
  ~emu_stream_() { { std::lock_guard<std::mutex> l(mu); quit = true; } cv.notify_all(); if (th.joinable()) th.join(); }
};
static std::mutex g_streams_mu;
static std::vector<emu_stream_ *> g_streams;
static emu_stream_ *null_stream() { static emu_stream_ *s = []() { auto *x = new emu_stream_(); std::lock_guard<std::mutex> g(g_streams_mu); g_streams.push_back(x); return x; }(); return s; }
static emu_stream_ *S(hipStream_t s) { return s ? s : null_stream(); }

hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) {
  *s = new emu_stream_();
  std::lock_guard<std::mutex> g(g_streams_mu);
  g_streams.push_back(*s);
  return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t s) {
  if (!s) return hipErrorInvalidValue;
  s->drain();
  { std::lock_guard<std::mutex> g(g_streams_mu); for (size_t i = 0; i < g_streams.size(); i++) if (g_streams[i] == s) { g_streams.erase(g_streams.begin() + (long) i); break; } }
  delete s;
  return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t s) { S(s)->drain(); return hipSuccess; }
hipError_t hipDeviceSynchronize(void) {
  std::vector<emu_stream_ *> all;
  { std::lock_guard<std::mutex> g(g_streams_mu); all = g_streams; }
  for (emu_stream_ *s : all) s->drain();
  return hipSuccess;
}
hipError_t hipEventCreate(hipEvent_t *e) { *e = new emu_event_(); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
  unsigned long long gen;
  { std::lock_guard<std::mutex> l(e->mu); gen = ++e->recorded; }
  S(s)->push([e, gen]() { { std::lock_guard<std::mutex> l(e->mu); if (e->done < gen) e->done = gen; e->t = std::chrono::steady_clock::now(); } e->cv.notify_all(); });
  return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned) {
  unsigned long long gen;
  { std::lock_guard<std::mutex> l(e->mu); gen = e->recorded; }         // the record that was LAST issued when the wait is issued
  S(s)->push([e, gen]() { std::unique_lock<std::mutex> l(e->mu); e->cv.wait(l, [&]() { return e->done >= gen; }); });
  return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e) {
  std::unique_lock<std::mutex> l(e->mu);
  const unsigned long long gen = e->recorded;
  e->cv.wait(l, [&]() { return e->done >= gen; });
  return hipSuccess;
}
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count(); return hipSuccess; }

// ---- copies ----------------------------------------------------------------------------------------------------------------------
// host side of a copy: 0 = pageable (the call blocks until the copy has run), 1 = inside ONE registration (asynchronous),
// -1 = straddles a registration's boundary (the real runtime refuses it: DESIGN.md 8h)
static int host_side(const void *p, size_t n) {
  const uintptr_t a = (uintptr_t) p, b = a + n;
  std::lock_guard<std::mutex> g(g_reg_mu);
  const Reg *r = reg_holding(a);
  if (r) return b <= r->b ? 1 : -1;
  for (const auto &kv : g_regs) if (kv.second.a < b && a < kv.second.b) return -1;
  return 0;
}
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind k, hipStream_t s) {
  if (!n) return hipSuccess;
  const void *host = k == hipMemcpyHostToDevice ? src : (k == hipMemcpyDeviceToHost ? dst : nullptr);
  if (k == hipMemcpyHostToDevice && !is_device(dst)) violation("H2D copy whose destination is not device memory", dst, n);
  if (k == hipMemcpyDeviceToHost && !is_device(src)) violation("D2H copy whose source is not device memory", src, n);
  int hs = 1;
  if (host) {
    hs = host_side(host, n);
    if (hs < 0) { violation("a copy whose host side straddles a registration's boundary", host, n); return hipErrorInvalidValue; }
  }
  S(s)->push([dst, src, n]() { memmove(dst, src, n); });
  if (hs == 0) S(s)->drain();                                // pageable: the caller is held
  return hipSuccess;
}
hipError_t hipMemcpy(void *dst, const void *src, size_t n, hipMemcpyKind k) {
  const hipError_t e = hipMemcpyAsync(dst, src, n, k, nullptr);
  if (e == hipSuccess) null_stream()->drain();
  return e;
}
hipError_t hipMemsetAsync(void *dst, int v, size_t n, hipStream_t s) {
  if (!is_device(dst)) violation("memset outside device memory", dst, n);
  S(s)->push([dst, v, n]() { memset(dst, v, n); });
  return hipSuccess;
}
hipError_t hipFree(void *p) {
  if (!p) return hipSuccess;
  hipDeviceSynchronize();                                   // (hipFree synchronises the device)
  { std::lock_guard<std::mutex> g(g_dev_mu); if (!g_dev.erase((uintptr_t) p)) { violation("hipFree of an unknown pointer", p, 0); return hipErrorInvalidValue; } }
  free(p);
  return hipSuccess;
}
hipError_t hipHostFree(void *p) {
  if (!p) return hipSuccess;
  { std::lock_guard<std::mutex> g(g_reg_mu); auto it = g_regs.find((uintptr_t) p); if (it == g_regs.end() || !it->second.own) { violation("hipHostFree of an unknown pointer", p, 0); return hipErrorInvalidValue; } g_regs.erase(it); }
  free(p);
  return hipSuccess;
}

// ---- the rest of the host API ------------------------------------------------------------------------------------------------
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : (e == hipErrorOutOfMemory ? "out of memory" : "invalid argument"); }
hipError_t hipGetLastError(void) { return hipSuccess; }
hipError_t hipGetDeviceCount(int *n) { const char *e = getenv("MSPACK_EMU_DEVICES"); *n = e ? atoi(e) : 1; return hipSuccess; }
static thread_local int tls_dev = 0;
hipError_t hipGetDevice(int *d) { *d = tls_dev; return hipSuccess; }
hipError_t hipSetDevice(int d) { tls_dev = d; return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) { p->multiProcessorCount = 4; strcpy(p->name, "hostcheck"); return hipSuccess; }

// ---- a launch: the stand-in decodes the launch's units when the stream gets there ----------------------------------------------------
static std::mutex g_oracle_mu;                               // (the oracle keeps state in statics: one call at a time)
hipError_t hostcheck_launch_kind(unsigned kind, const mspack_hip_unit *d_units, const uint32_t *d_order, size_t n,
                                 const void *d_in, void *d_out, mspack_hip_result *d_results, hipStream_t st)
{
  if (!is_device(d_units) || !is_device(d_in) || !is_device(d_results) || (d_order && !is_device(d_order)))
    violation("a launch with an argument that is not device memory", d_units, n);
  S(st)->push([=]() {
    std::lock_guard<std::mutex> g(g_oracle_mu);
    for (size_t i = 0; i < n; i++) {
      const uint32_t ui = d_order ? d_order[i] : (uint32_t) i;
      if (d_units[ui].kind != kind) continue;
      mspack_hip_unit u = d_units[ui];
      u.flags &= ~(uint32_t) MSPACK_HIP_UF_FRAME_TABLE;                              // (a hint; the oracle is serial)
      const uint32_t one = 0;
      mspack_hip_result r;
      memset(&r, 0, sizeof(r));
      if (mspack_standin_decode_units(&u, &one, 1, d_in, (size_t) -1 >> 1, d_out, (size_t) -1 >> 1, &r) != 0) {
        fprintf(stderr, "hostcheck: stand-in: %s\n", mspack_standin_last_error()); abort();
      }
      d_results[ui] = r;
    }
  });
  return hipSuccess;
}
