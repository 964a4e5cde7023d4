// tests/emu/emu_runtime.cpp -- TEST INFRASTRUCTURE ONLY (see tests/emu/include/hip/hip_runtime.h).
//
// The wavefront emulator behind tests/_build/libmspack_emu.so: runs the real kernel sources of
// libmspack_amd/csrc/hip on the host CPU so that kernel logic can be debugged without a GPU.
//   * one OS thread per resident workgroup (64 threads = one wavefront), blocks handed out in index order;
//   * the 64 lanes of a wavefront are fibers on that thread;
//   * the kernel translation unit is compiled with -fsanitize=thread for its hooks only: an access to LDS (this
//     thread's thread_local statics) or to device memory (emu_hipMalloc ranges) parks the lane; when every live
//     lane is parked the lanes with the deepest stack, then the lowest code address, are released -- convergent
//     code therefore runs in SIMT lock step;
//   * readlane / readfirstlane / ballot / bpermute / DPP are collectives: every live lane must arrive at the same
//     call site, else the emulator aborts with the lanes' positions.
// This file is compiled WITHOUT instrumentation.
#include <stdint.h>
#include <stdio.h>
#include <unistd.h>
#include <stdlib.h>
#include <string.h>
#include <sched.h>
#include <sys/mman.h>
#include <link.h>
#include <dlfcn.h>
#include <time.h>
#include <atomic>
#include <functional>
#include <thread>
#include <vector>
#include <mutex>
#include <chrono>
#include <string>
#include <algorithm>
#include "hip/hip_runtime.h"
#undef threadIdx
#undef blockIdx
#undef gridDim

typedef uint32_t u32;
typedef uint64_t u64;

// ---- fibers ------------------------------------------------------------------------------------------------------
extern "C" void emu_switch(void **save_sp, void *load_sp);
asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size emu_switch,.-emu_switch
)");

enum { L_READY = 0, L_SOFT = 1, L_HARD = 2, L_DONE = 3 };
enum { OP_READLANE = 1, OP_RFL, OP_BALLOT, OP_BPERM, OP_DPP };
#define LANE_STACK (512u << 10)

// ---- static loop table (back edges of the instrumented code, extracted by tests/emu/build_emu.sh) -----------------
// Why: "lowest code address first" alone is wrong across a loop's back edge -- a lane that skipped a divergent
// region at the end of a loop body is back at the loop's top (low address) while the others are still in that region.
// With the loops known, a lane's position is (iteration count of every enclosing loop, outermost first; address):
// the lane that is BEHIND in that order runs first, which is what SIMT reconvergence does for structured code.
struct Loop { u32 head, tail; int parent, level; };
static std::vector<Loop> g_loops;
static uintptr_t g_base = 0;
static std::once_flag g_loops_once;
static void load_loops() {
  Dl_info di; memset(&di, 0, sizeof(di));
  if (!dladdr((void *) &load_loops, &di) || !di.dli_fname) return;
  g_base = (uintptr_t) di.dli_fbase;
  std::string path = std::string(di.dli_fname) + ".loops";
  FILE *f = fopen(path.c_str(), "r");
  if (!f) { fprintf(stderr, "emu: %s missing (tests/emu/build_emu.sh writes it)\n", path.c_str()); abort(); }
  std::vector<std::pair<u32, u32>> raw;
  unsigned h, t;
  while (fscanf(f, "%x %x", &h, &t) == 2) raw.push_back({ h, t });
  fclose(f);
  std::sort(raw.begin(), raw.end(), [](const std::pair<u32, u32> &a, const std::pair<u32, u32> &b) {
    return a.first != b.first ? a.first < b.first : a.second > b.second; });
  std::vector<int> st;
  for (auto &r : raw) {
    if (!g_loops.empty() && g_loops.back().head == r.first) continue;        // same head: the widest was first
    while (!st.empty() && g_loops[st.back()].tail < r.first) st.pop_back();
    Loop L; L.head = r.first; L.tail = r.second; L.parent = st.empty() ? -1 : st.back();
    if (L.parent >= 0 && g_loops[L.parent].tail < L.tail) g_loops[L.parent].tail = L.tail;   // (improper nesting: widen)
    L.level = L.parent < 0 ? 0 : g_loops[L.parent].level + 1;
    g_loops.push_back(L); st.push_back((int) g_loops.size() - 1);
  }
}
static int innermost_loop(u32 pc) {
  int lo = 0, hi = (int) g_loops.size() - 1, k = -1;
  while (lo <= hi) { const int m = (lo + hi) >> 1; if (g_loops[m].head <= pc) { k = m; lo = m + 1; } else hi = m - 1; }
  while (k >= 0 && g_loops[k].tail < pc) k = g_loops[k].parent;
  return k;
}
#define MAX_NEST 24
struct Frame { uintptr_t depth; u32 prev; };
struct Lane {
  void *sp;
  int state;
  uintptr_t pc, depth;
  int op; uintptr_t site;
  std::vector<u32> cnt;             // iteration count of every static loop (current instance)
  std::vector<Frame> frames;        // one per active call depth that has parked
  int chain[MAX_NEST]; u32 cc[MAX_NEST]; int nchain;   // enclosing loops of the park position, outermost first, + counts
};
struct Wave {
  Lane lane[64];
  void *sched_sp;
  u64 alive;
  int cur;                          // lane whose fiber is running (-1: scheduler)
  unsigned block, grid;
  bool slept;
  int gen;                          // completed collectives
  u32 xa[2][64], xb[2][64];         // collective operands, by generation parity
  u64 part[2];                      // the lanes that took part in the collective of that parity
  char *stacks;
  uintptr_t tls_lo, tls_hi;         // this thread's LDS: the module's TLS block
  const std::function<void()> *fn;
  u64 n_soft, n_hard;
};
static thread_local Wave *tls_wave = nullptr;

// ---- device memory ranges ------------------------------------------------------------------------------------------
#define MAX_RANGES 4096
static std::atomic<uintptr_t> g_lo[MAX_RANGES], g_hi[MAX_RANGES];
static std::atomic<int> g_nranges{0};
static std::mutex g_alloc_mu;
static inline bool in_device_memory(uintptr_t a) {
  const int n = g_nranges.load(std::memory_order_acquire);
  for (int i = 0; i < n; i++) if (a >= g_lo[i].load(std::memory_order_relaxed) && a < g_hi[i].load(std::memory_order_relaxed)) return true;
  return false;
}

static int tls_cb(struct dl_phdr_info *info, size_t, void *data) {
  Wave *w = (Wave *) data;
  const uintptr_t me = (uintptr_t) &emu_switch;
  for (int i = 0; i < info->dlpi_phnum; i++) {
    const ElfW(Phdr) *ph = &info->dlpi_phdr[i];
    if (ph->p_type == PT_LOAD && me >= info->dlpi_addr + ph->p_vaddr && me < info->dlpi_addr + ph->p_vaddr + ph->p_memsz) {
      for (int j = 0; j < info->dlpi_phnum; j++)
        if (info->dlpi_phdr[j].p_type == PT_TLS && info->dlpi_tls_data) {
          w->tls_lo = (uintptr_t) info->dlpi_tls_data; w->tls_hi = w->tls_lo + info->dlpi_phdr[j].p_memsz;
        }
      return 1;
    }
  }
  return 0;
}

static void to_sched(Wave *w) {
  const int l = w->cur;
  w->cur = -1;
  emu_switch(&w->lane[l].sp, w->sched_sp);
}

static void lane_main() {
  Wave *w = tls_wave;
  (*w->fn)();
  w = tls_wave;
  w->lane[w->cur].state = L_DONE;
  w->alive &= ~(1ull << w->cur);
  to_sched(w);
  abort();
}

static const bool g_strict_sites = getenv("MSPACK_EMU_LAX_SITES") == nullptr;
static void die_divergent(Wave *w) {
  fprintf(stderr, "emu: block %u: live lanes wait at DIFFERENT collectives (divergent cross-lane operation)\n", w->block);
  for (int l = 0; l < 64; l++)
    if (w->lane[l].state == L_HARD) {
      Dl_info di; memset(&di, 0, sizeof(di)); dladdr((void *) w->lane[l].site, &di);
      fprintf(stderr, "  lane %2d: op %d at +0x%lx\n", l, w->lane[l].op, (unsigned long)(w->lane[l].site - (uintptr_t) di.dli_fbase));
    }
  fprintf(stderr, "  (llvm-symbolizer -e tests/_build/libmspack_emu.so <offsets>)\n");
  abort();
}

static int lane_cmp(const Lane &a, const Lane &b);
static void run_block(Wave *w, unsigned block) {
  w->block = block; w->alive = ~0ull; w->gen = 0; w->cur = -1; w->slept = false;
  for (int l = 0; l < 64; l++) {
    u64 *s = (u64 *)(w->stacks + (size_t)(l + 1) * LANE_STACK);
    *--s = 0; *--s = (u64)(uintptr_t) &lane_main;
    for (int k = 0; k < 6; k++) *--s = 0;
    w->lane[l].sp = s; w->lane[l].state = L_READY;
    w->lane[l].cnt.assign(g_loops.size(), 0u); w->lane[l].frames.clear(); w->lane[l].nchain = 0;
  }
  for (;;) {
    for (int l = 0; l < 64; l++)
      if (w->lane[l].state == L_READY) { w->cur = l; emu_switch(&w->sched_sp, w->lane[l].sp); }
    if (!w->alive) break;
    // every live lane is parked now
    int best = -1;
    for (int l = 0; l < 64; l++)
      if (w->lane[l].state == L_SOFT && (best < 0 || lane_cmp(w->lane[l], w->lane[best]) < 0)) best = l;
    if (best >= 0) {
      for (int l = 0; l < 64; l++)
        if (l != best && w->lane[l].state == L_SOFT && lane_cmp(w->lane[l], w->lane[best]) == 0) w->lane[l].state = L_READY;
      w->lane[best].state = L_READY;
      if (w->slept) { w->slept = false; sched_yield(); }
      continue;
    }
    int op = 0; uintptr_t site = 0; bool first = true;
    for (int l = 0; l < 64; l++)
      if (w->lane[l].state == L_HARD) {
        if (first) { op = w->lane[l].op; site = w->lane[l].site; first = false; }
        else if (w->lane[l].op != op || (w->lane[l].site != site && g_strict_sites)) die_divergent(w);
      }
    u64 part = 0;
    for (int l = 0; l < 64; l++) if (w->lane[l].state == L_HARD) { w->lane[l].state = L_READY; part |= 1ull << l; }
    w->part[w->gen & 1] = part;
    w->gen++;
  }
}

// where is the lane (loop iteration counts of the enclosing loops), updated at every park
static void track(Lane &L, uintptr_t pc_abs, uintptr_t depth) {
  const u32 pc = (u32)(pc_abs - g_base);
  while (!L.frames.empty() && L.frames.back().depth > depth) L.frames.pop_back();      // returned from deeper calls
  if (L.frames.empty() || L.frames.back().depth < depth) { Frame f = { depth, 0u }; L.frames.push_back(f); }
  Frame &F = L.frames.back();
  const u32 prev = F.prev;
  const int in = innermost_loop(pc);
  // a back edge: the innermost loop that holds both the previous and the new position, when the address went down
  if (prev != 0u && pc < prev) {
    int k = in;
    while (k >= 0 && !(g_loops[k].head <= prev && prev <= g_loops[k].tail)) k = g_loops[k].parent;
    if (k >= 0) L.cnt[k]++;
  }
  // loops entered since the previous park start a new instance
  int n = 0;
  for (int k = in; k >= 0; k = g_loops[k].parent) {
    if (prev == 0u || !(g_loops[k].head <= prev && prev <= g_loops[k].tail)) L.cnt[k] = 0;
    n++;
  }
  if (n > MAX_NEST) { fprintf(stderr, "emu: loops nested deeper than %d\n", MAX_NEST); abort(); }
  L.nchain = n;
  for (int k = in, i = n - 1; k >= 0; k = g_loops[k].parent, i--) { L.chain[i] = k; L.cc[i] = L.cnt[k]; }
  F.prev = pc;
}
// < 0: a is behind b (runs first), 0: same position, > 0: a is ahead
static int lane_cmp(const Lane &a, const Lane &b) {
  if (a.depth != b.depth) return a.depth > b.depth ? -1 : 1;                // inside a deeper call: finish it first
  const int n = a.nchain < b.nchain ? a.nchain : b.nchain;
  for (int i = 0; i < n && a.chain[i] == b.chain[i]; i++)
    if (a.cc[i] != b.cc[i]) return a.cc[i] < b.cc[i] ? -1 : 1;
  return a.pc < b.pc ? -1 : (a.pc > b.pc ? 1 : 0);
}

// a lane parks at an access to LDS / device memory
static __attribute__((noinline)) void soft_sync(Wave *w, uintptr_t pc, uintptr_t fa) {
  Lane &L = w->lane[w->cur];
  L.state = L_SOFT; L.pc = pc;
  L.depth = (uintptr_t)(w->stacks + (size_t)(w->cur + 1) * LANE_STACK) - fa;   // (fa: the hook's frame = the caller's stack depth)
  track(L, pc, L.depth);
  w->n_soft++;
  to_sched(w);
}
static inline void access_hook(const void *addr, uintptr_t pc, uintptr_t fa) {
  Wave *w = tls_wave;
  if (!w || w->cur < 0) return;
  const uintptr_t a = (uintptr_t) addr;
  if ((a >= w->tls_lo && a < w->tls_hi) || in_device_memory(a)) soft_sync(w, pc, fa);
}

// a collective: returns the parity of the generation whose operands are complete
static __attribute__((noinline)) int hard_sync(Wave *w, int op, uintptr_t site, uintptr_t fa, u32 a, u32 b) {
  const int l = w->cur, p = w->gen & 1;
  w->xa[p][l] = a; w->xb[p][l] = b;
  Lane &L = w->lane[l];
  L.state = L_HARD; L.op = op; L.site = site;
  track(L, site, (uintptr_t)(w->stacks + (size_t)(l + 1) * LANE_STACK) - fa);
  w->n_hard++;
  to_sched(w);
  return p;
}
static Wave *must_wave() {
  Wave *w = tls_wave;
  if (!w || w->cur < 0) { fprintf(stderr, "emu: device builtin called outside a kernel\n"); abort(); }
  return w;
}

extern "C" {
emu_idx emu_thread_idx(void) { Wave *w = must_wave(); emu_idx r = { (unsigned) w->cur, 0, 0 }; return r; }
emu_idx emu_block_idx(void) { Wave *w = must_wave(); emu_idx r = { w->block, 0, 0 }; return r; }
emu_idx emu_grid_dim(void) { Wave *w = must_wave(); emu_idx r = { w->grid, 1, 1 }; return r; }

unsigned emu_readlane(unsigned v, unsigned l) {
  Wave *w = must_wave();
  const int p = hard_sync(w, OP_READLANE, (uintptr_t) __builtin_return_address(0), (uintptr_t) __builtin_frame_address(0), v, l);
  const unsigned src = w->xb[p][w->cur] & 63u;
  return w->xa[p][src];
}
unsigned emu_readfirstlane(unsigned v) {
  Wave *w = must_wave();
  const int p = hard_sync(w, OP_RFL, (uintptr_t) __builtin_return_address(0), (uintptr_t) __builtin_frame_address(0), v, 0);
  // every lane that deposited was live at the time; the lowest of them is the "first active lane"
  return w->xa[p][__builtin_ctzll(w->part[p])];
}
unsigned long long emu_ballot(int pr) {
  Wave *w = must_wave();
  const int p = hard_sync(w, OP_BALLOT, (uintptr_t) __builtin_return_address(0), (uintptr_t) __builtin_frame_address(0), pr ? 1u : 0u, 0);
  u64 m = 0;
  for (int l = 0; l < 64; l++) if (((w->part[p] >> l) & 1ull) && w->xa[p][l]) m |= 1ull << l;
  return m;
}
unsigned emu_bpermute(unsigned addr, unsigned v) {
  Wave *w = must_wave();
  const int p = hard_sync(w, OP_BPERM, (uintptr_t) __builtin_return_address(0), (uintptr_t) __builtin_frame_address(0), v, addr);
  const unsigned src = (w->xb[p][w->cur] >> 2) & 63u;
  return ((w->part[p] >> src) & 1ull) ? w->xa[p][src] : 0u;
}
unsigned emu_dpp(unsigned old, unsigned src, unsigned ctrl, unsigned row_mask, unsigned bank_mask, int bound_ctrl) {
  Wave *w = must_wave();
  const int p = hard_sync(w, OP_DPP, (uintptr_t) __builtin_return_address(0), (uintptr_t) __builtin_frame_address(0), src, ctrl);
  const unsigned l = (unsigned) w->cur, row = l >> 4, r = l & 15u;
  if (!((row_mask >> row) & 1u) || !((bank_mask >> (r >> 2)) & 1u)) return old;
  const unsigned inval = bound_ctrl ? 0u : old;
  if (ctrl >= 0x111u && ctrl <= 0x11Fu) { const unsigned n = ctrl & 15u; return r >= n ? w->xa[p][l - n] : inval; }   // row_shr:n
  if (ctrl >= 0x101u && ctrl <= 0x10Fu) { const unsigned n = ctrl & 15u; return r + n < 16u ? w->xa[p][l + n] : inval; } // row_shl:n
  if (ctrl == 0x142u) return row >= 1u ? w->xa[p][row * 16u - 1u] : inval;                                              // row_bcast:15
  if (ctrl == 0x143u) return row >= 2u ? w->xa[p][31] : inval;                                                          // row_bcast:31
  fprintf(stderr, "emu: DPP control 0x%x not modelled\n", ctrl); abort();
}
void emu_sleep(void) {
  Wave *w = must_wave();
  w->slept = true;
  soft_sync(w, (uintptr_t) __builtin_return_address(0), (uintptr_t) __builtin_frame_address(0));
}
unsigned long long emu_clock(void) {
  struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
  return (unsigned long long) ts.tv_sec * 1000000000ull + (unsigned long long) ts.tv_nsec;
}

// ---- instrumentation hooks (the ThreadSanitizer ABI, implemented here; no TSan runtime is linked) -------------------
#define PC() ((uintptr_t) __builtin_return_address(0))
#define FA() ((uintptr_t) __builtin_frame_address(0))
void __tsan_init(void) {}
// -fsanitize-coverage=bb,no-prune,trace-pc: called at the top of every basic block -- exact loop bookkeeping
void __sanitizer_cov_trace_pc(void) {
  Wave *w = tls_wave;
  if (!w || w->cur < 0) return;
  track(w->lane[w->cur], PC(), (uintptr_t)(w->stacks + (size_t)(w->cur + 1) * LANE_STACK) - FA());
}
void __tsan_func_entry(void *) {}
void __tsan_func_exit(void) {}
void __tsan_read1(void *a) { access_hook(a, PC(), FA()); }
void __tsan_read2(void *a) { access_hook(a, PC(), FA()); }
void __tsan_read4(void *a) { access_hook(a, PC(), FA()); }
void __tsan_read8(void *a) { access_hook(a, PC(), FA()); }
void __tsan_read16(void *a) { access_hook(a, PC(), FA()); }
void __tsan_write1(void *a) { access_hook(a, PC(), FA()); }
void __tsan_write2(void *a) { access_hook(a, PC(), FA()); }
void __tsan_write4(void *a) { access_hook(a, PC(), FA()); }
void __tsan_write8(void *a) { access_hook(a, PC(), FA()); }
void __tsan_write16(void *a) { access_hook(a, PC(), FA()); }
void __tsan_unaligned_read2(void *a) { access_hook(a, PC(), FA()); }
void __tsan_unaligned_read4(void *a) { access_hook(a, PC(), FA()); }
void __tsan_unaligned_read8(void *a) { access_hook(a, PC(), FA()); }
void __tsan_unaligned_read16(void *a) { access_hook(a, PC(), FA()); }
void __tsan_unaligned_write2(void *a) { access_hook(a, PC(), FA()); }
void __tsan_unaligned_write4(void *a) { access_hook(a, PC(), FA()); }
void __tsan_unaligned_write8(void *a) { access_hook(a, PC(), FA()); }
void __tsan_unaligned_write16(void *a) { access_hook(a, PC(), FA()); }
void __tsan_read_range(void *a, unsigned long) { access_hook(a, PC(), FA()); }
void __tsan_write_range(void *a, unsigned long) { access_hook(a, PC(), FA()); }
void __tsan_vptr_update(void **, void *) {}
void __tsan_vptr_read(void **) {}
void *__tsan_memcpy(void *d, const void *s, unsigned long n) { access_hook(s, PC(), FA()); return memcpy(d, s, n); }
void *__tsan_memmove(void *d, const void *s, unsigned long n) { access_hook(s, PC(), FA()); return memmove(d, s, n); }
void *__tsan_memset(void *d, int v, unsigned long n) { access_hook(d, PC(), FA()); return memset(d, v, n); }

#define ATOMICS(N, T)                                                                                               \
  T __tsan_atomic##N##_load(const volatile T *p, int mo) { access_hook((const void *) p, PC(), FA()); return __atomic_load_n(p, __ATOMIC_SEQ_CST); } \
  void __tsan_atomic##N##_store(volatile T *p, T v, int mo) { access_hook((const void *) p, PC(), FA()); __atomic_store_n(p, v, __ATOMIC_SEQ_CST); } \
  T __tsan_atomic##N##_exchange(volatile T *p, T v, int mo) { access_hook((const void *) p, PC(), FA()); return __atomic_exchange_n(p, v, __ATOMIC_SEQ_CST); } \
  T __tsan_atomic##N##_fetch_add(volatile T *p, T v, int mo) { access_hook((const void *) p, PC(), FA()); return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); } \
  T __tsan_atomic##N##_fetch_sub(volatile T *p, T v, int mo) { access_hook((const void *) p, PC(), FA()); return __atomic_fetch_sub(p, v, __ATOMIC_SEQ_CST); } \
  T __tsan_atomic##N##_fetch_and(volatile T *p, T v, int mo) { access_hook((const void *) p, PC(), FA()); return __atomic_fetch_and(p, v, __ATOMIC_SEQ_CST); } \
  T __tsan_atomic##N##_fetch_or(volatile T *p, T v, int mo) { access_hook((const void *) p, PC(), FA()); return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); } \
  T __tsan_atomic##N##_fetch_xor(volatile T *p, T v, int mo) { access_hook((const void *) p, PC(), FA()); return __atomic_fetch_xor(p, v, __ATOMIC_SEQ_CST); } \
  int __tsan_atomic##N##_compare_exchange_strong(volatile T *p, T *e, T v, int mo, int fmo) { access_hook((const void *) p, PC(), FA()); return __atomic_compare_exchange_n(p, e, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST); } \
  int __tsan_atomic##N##_compare_exchange_weak(volatile T *p, T *e, T v, int mo, int fmo) { access_hook((const void *) p, PC(), FA()); return __atomic_compare_exchange_n(p, e, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST); } \
  T __tsan_atomic##N##_compare_exchange_val(volatile T *p, T e, T v, int mo, int fmo) { access_hook((const void *) p, PC(), FA()); __atomic_compare_exchange_n(p, &e, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST); return e; }
ATOMICS(8, unsigned char)
ATOMICS(16, unsigned short)
ATOMICS(32, unsigned int)
ATOMICS(64, unsigned long long)
void __tsan_atomic_thread_fence(int) { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
void __tsan_atomic_signal_fence(int) {}
} // extern "C"

// ---- launch ------------------------------------------------------------------------------------------------------------
static int env_threads() { const char *e = getenv("MSPACK_EMU_THREADS"); int v = e ? atoi(e) : 16; return v < 1 ? 1 : (v > 256 ? 256 : v); }
static std::atomic<unsigned long long> g_soft{0}, g_hard{0}, g_blocks{0};

void emu_launch(dim3 grid, dim3 block, std::function<void()> fn)
{
  if (block.x != 64u || block.y != 1u || block.z != 1u || grid.y != 1u || grid.z != 1u) {
    fprintf(stderr, "emu: only 1-D grids of 64-thread workgroups are modelled\n"); abort();
  }
  const unsigned n = grid.x;
  if (n == 0) return;
  std::call_once(g_loops_once, load_loops);
  std::atomic<unsigned> next{0};
  unsigned nt = (unsigned) env_threads(); if (nt > n) nt = n;
  auto worker = [&]() {
    Wave *w = new Wave();
    w->stacks = (char *) mmap(nullptr, (size_t) 64 * LANE_STACK, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (w->stacks == (char *) MAP_FAILED) { perror("emu: mmap"); abort(); }
    w->fn = &fn; w->grid = n; w->n_soft = w->n_hard = 0;
    tls_wave = w;                                        // (touches this module's TLS block: it exists from here on)
    w->tls_lo = w->tls_hi = 0;
    dl_iterate_phdr(tls_cb, w);
    for (;;) {
      const unsigned b = next.fetch_add(1);
      if (b >= n) break;
      run_block(w, b);
    }
    g_soft += w->n_soft; g_hard += w->n_hard;
    tls_wave = nullptr;
    munmap(w->stacks, (size_t) 64 * LANE_STACK);
    delete w;
  };
  g_blocks += n;
  if (nt == 1) { std::thread t(worker); t.join(); }       // (always a fresh thread: LDS = its TLS, as on a fresh workgroup)
  else {
    std::vector<std::thread> th;
    for (unsigned i = 0; i < nt; i++) th.emplace_back(worker);
    for (auto &t : th) t.join();
  }
  if (getenv("MSPACK_EMU_STATS"))
    fprintf(stderr, "emu: %llu blocks so far, %llu memory scheduling points, %llu collectives\n",
            (unsigned long long) g_blocks.load(), (unsigned long long) g_soft.load(), (unsigned long long) g_hard.load());
}

// ---- host runtime ------------------------------------------------------------------------------------------------------
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : (e == hipErrorOutOfMemory ? "out of memory" : "error"); }
hipError_t hipGetLastError(void) { return hipSuccess; }
hipError_t hipGetDeviceCount(int *n) { const char *e = getenv("MSPACK_EMU_DEVICES"); *n = e ? atoi(e) : 1; return hipSuccess; }
static thread_local int tls_dev = 0;
hipError_t hipGetDevice(int *d) { *d = tls_dev; return hipSuccess; }
hipError_t hipSetDevice(int d) { tls_dev = d; return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) {
  const char *e = getenv("MSPACK_EMU_CUS");
  p->multiProcessorCount = e ? atoi(e) : 1; strcpy(p->name, "wavefront emulator"); return hipSuccess;
}
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t emu_hipMalloc(void **p, size_t n) {
  void *q = nullptr;
  if (posix_memalign(&q, 4096, n + 8192)) return hipErrorOutOfMemory;
  memset(q, 0xA5, n + 8192);                              // (device memory is not zeroed by hipMalloc)
  std::lock_guard<std::mutex> g(g_alloc_mu);
  int slot = -1;
  const int cnt = g_nranges.load();
  for (int i = 0; i < cnt; i++) if (g_lo[i].load() == 0 && g_hi[i].load() == 0) { slot = i; break; }
  if (slot < 0) { if (cnt >= MAX_RANGES) { free(q); return hipErrorOutOfMemory; } slot = cnt; }
  g_lo[slot].store((uintptr_t) q); g_hi[slot].store((uintptr_t) q + n + 8192);
  if (slot == cnt) g_nranges.store(cnt + 1, std::memory_order_release);
  *p = q;
  return hipSuccess;
}
hipError_t hipFree(void *p) {
  if (!p) return hipSuccess;
  std::lock_guard<std::mutex> g(g_alloc_mu);
  const int cnt = g_nranges.load();
  for (int i = 0; i < cnt; i++) if (g_lo[i].load() == (uintptr_t) p) { g_hi[i].store(0); g_lo[i].store(0); free(p); return hipSuccess; }
  return hipErrorInvalidValue;
}
hipError_t emu_hipHostMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
void emu_test_delay(void) {
  static const int us = getenv("MSPACK_EMU_PUBLISH_DELAY_US") ? atoi(getenv("MSPACK_EMU_PUBLISH_DELAY_US")) : 0;
  if (us > 0) usleep((useconds_t) us);
}
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind, hipStream_t) { memmove(dst, src, n); return hipSuccess; }
hipError_t hipMemcpy(void *dst, const void *src, size_t n, hipMemcpyKind) { memmove(dst, src, n); return hipSuccess; }
hipError_t hipMemsetAsync(void *dst, int v, size_t n, hipStream_t) { memset(dst, v, n); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = (hipStream_t) malloc(8); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
struct emu_event_ { std::chrono::steady_clock::time_point t; };
hipError_t hipEventCreate(hipEvent_t *e) { *e = new emu_event_(); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count(); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
