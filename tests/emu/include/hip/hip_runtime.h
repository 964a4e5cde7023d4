// tests/emu/include/hip/hip_runtime.h -- TEST INFRASTRUCTURE ONLY.
//
// A stand-in for <hip/hip_runtime.h> that lets libmspack_amd/csrc/hip/shim.hip -- the REAL kernel sources,
// unchanged -- be compiled for the host CPU and executed by the wavefront emulator in tests/emu/emu_runtime.cpp.
// Purpose: kernel LOGIC can be debugged in the development container (which has no GPU) before GPU minutes
// are spent on it.  It is not a fallback: the product library (libmspack_hip.so) never contains, loads or
// links any of this; only tests/ build tests/_build/libmspack_emu.so and point MSPACK_HIP_SO at it.
//
// Model: one OS thread per resident workgroup (= one wavefront of 64 lanes); the 64 lanes are fibers.  The
// translation unit is compiled with -fsanitize=thread purely for its instrumentation hooks (no TSan runtime is
// linked): every access to LDS (the thread's `__shared__` = thread_local statics) or to device memory
// (hipMalloc'ed ranges) is a scheduling point, and the emulator always advances the lanes with the deepest call
// stack / lowest code address first -- which executes convergent code in exact SIMT lock step (an instruction
// completes for all lanes before the next one starts).  Cross-lane builtins (readlane, ballot, DPP, bpermute)
// are collectives that every live lane must reach at the same call site; anything else aborts with a
// diagnostic.  What this cannot model: memory visibility between workgroups (caches, fences) and timing.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <functional>

#define MSPACK_WAVE_EMU 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static thread_local
#define __launch_bounds__(...)
#define amdgpu_waves_per_eu(...) unused
#define amdgpu_flat_work_group_size(...) unused
#define __align__(n) __attribute__((aligned(n)))
#define HIP_SYMBOL(x) x

struct uint2 { unsigned x, y; };
struct __attribute__((aligned(16))) uint4 { unsigned x, y, z, w; };
static inline uint2 make_uint2(unsigned x, unsigned y) { uint2 r = { x, y }; return r; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { uint4 r = { x, y, z, w }; return r; }
struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };

// ---- the emulator's entry points (tests/emu/emu_runtime.cpp; compiled WITHOUT instrumentation) ----------------
extern "C" {
struct emu_idx { unsigned x, y, z; };
emu_idx emu_thread_idx(void);
emu_idx emu_block_idx(void);
emu_idx emu_grid_dim(void);
unsigned emu_readlane(unsigned v, unsigned l);
unsigned emu_readfirstlane(unsigned v);
unsigned long long emu_ballot(int p);
unsigned emu_bpermute(unsigned addr, unsigned v);
unsigned emu_dpp(unsigned old, unsigned src, unsigned ctrl, unsigned row_mask, unsigned bank_mask, int bound_ctrl);
void emu_sleep(void);
unsigned long long emu_clock(void);
}
void emu_launch(dim3 grid, dim3 block, std::function<void()> fn);

#define threadIdx (emu_thread_idx())
#define blockIdx (emu_block_idx())
#define gridDim (emu_grid_dim())

// ---- device builtins -----------------------------------------------------------------------------------------
#define __builtin_amdgcn_readlane(v, l) ((int) emu_readlane((unsigned)(v), (unsigned)(l)))
#define __builtin_amdgcn_readfirstlane(v) ((int) emu_readfirstlane((unsigned)(v)))
#define __builtin_amdgcn_ds_bpermute(a, v) ((int) emu_bpermute((unsigned)(a), (unsigned)(v)))
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) ((int) emu_dpp((unsigned)(old), (unsigned)(src), (ctrl), (rm), (bm), (bc)))
#define __ballot(p) emu_ballot((p) ? 1 : 0)
#define __builtin_amdgcn_inverse_ballot_w64(m) ((((unsigned long long)(m)) >> emu_thread_idx().x) & 1ull)
static inline unsigned emu_mbcnt_lo_(unsigned m, unsigned v) { unsigned l = emu_thread_idx().x; return v + (unsigned) __builtin_popcount(m & (l >= 32u ? 0xFFFFFFFFu : ((1u << l) - 1u))); }
static inline unsigned emu_mbcnt_hi_(unsigned m, unsigned v) { unsigned l = emu_thread_idx().x; return v + (unsigned) __builtin_popcount(m & (l <= 32u ? 0u : ((1u << (l - 32u)) - 1u))); }
#define __builtin_amdgcn_mbcnt_lo(m, v) emu_mbcnt_lo_((unsigned)(m), (unsigned)(v))
#define __builtin_amdgcn_mbcnt_hi(m, v) emu_mbcnt_hi_((unsigned)(m), (unsigned)(v))
static inline unsigned emu_alignbyte_(unsigned hi, unsigned lo, unsigned sh) { return (unsigned)(((((unsigned long long) hi) << 32) | lo) >> (8u * (sh & 3u))); }
static inline unsigned emu_alignbit_(unsigned hi, unsigned lo, unsigned sh) { return (unsigned)(((((unsigned long long) hi) << 32) | lo) >> (sh & 31u)); }
#define __builtin_amdgcn_alignbyte(hi, lo, sh) emu_alignbyte_((unsigned)(hi), (unsigned)(lo), (unsigned)(sh))
#define __builtin_amdgcn_alignbit(hi, lo, sh) emu_alignbit_((unsigned)(hi), (unsigned)(lo), (unsigned)(sh))
static inline unsigned emu_perm_(unsigned s0, unsigned s1, unsigned sel) {      // v_perm_b32: bytes 0-3 = s1, 4-7 = s0
  unsigned long long src = (((unsigned long long) s0) << 32) | s1;
  unsigned r = 0;
  for (int i = 0; i < 4; i++) {
    unsigned s = (sel >> (8 * i)) & 0xFFu, b;
    if (s <= 7u) b = (unsigned)(src >> (8u * s)) & 0xFFu;
    else if (s == 12u) b = 0;
    else if (s >= 13u) b = 0xFFu;
    else { fprintf(stderr, "emu: v_perm_b32 selector %u not modelled\n", s); abort(); }
    r |= b << (8 * i);
  }
  return r;
}
#define __builtin_amdgcn_perm(a, b, sel) emu_perm_((unsigned)(a), (unsigned)(b), (unsigned)(sel))
#define __builtin_amdgcn_fence(order, scope) __atomic_thread_fence(__ATOMIC_SEQ_CST)
#define __builtin_amdgcn_wave_barrier() do { } while (0)
#define __builtin_amdgcn_s_sleep(n) emu_sleep()
#define __builtin_amdgcn_s_memtime() emu_clock()
#define __builtin_amdgcn_s_memrealtime() emu_clock()
#define __builtin_amdgcn_rcp(x) (1.0 / (x))
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
#define __builtin_amdgcn_s_setprio(n) do { } while (0)
#define __builtin_amdgcn_is_shared(p) false
#define __builtin_amdgcn_is_private(p) false
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long) v) : 64; }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned) v) : 32; }
static inline unsigned __brev(unsigned v) {
  v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
  v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
  v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4);
  return __builtin_bswap32(v);
}
template <typename T> static inline T atomicAdd(T *p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
template <typename T> static inline T atomicOr(T *p, T v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
template <typename T> static inline T atomicAnd(T *p, T v) { return __atomic_fetch_and(p, v, __ATOMIC_RELAXED); }
template <typename T> static inline T atomicExch(T *p, T v) { return __atomic_exchange_n(p, v, __ATOMIC_RELAXED); }
template <typename T> static inline T atomicCAS(T *p, T cmp, T v) { __atomic_compare_exchange_n(p, &cmp, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED); return cmp; }
template <typename T> static inline T atomicMax(T *p, T v) { T o = __atomic_load_n(p, __ATOMIC_RELAXED); while (o < v && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) { } return o; }
template <typename T> static inline T atomicMin(T *p, T v) { T o = __atomic_load_n(p, __ATOMIC_RELAXED); while (o > v && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) { } return o; }
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
#define __HIP_MEMORY_SCOPE_SYSTEM 5
#define __hip_atomic_load(p, order, scope) __atomic_load_n((p), (order))
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n((p), (v), (order))
#define __hip_atomic_fetch_add(p, v, order, scope) __atomic_fetch_add((p), (v), (order))
#define __hip_atomic_fetch_or(p, v, order, scope) __atomic_fetch_or((p), (v), (order))
#define __hip_atomic_exchange(p, v, order, scope) __atomic_exchange_n((p), (v), (order))
template <typename T> static inline bool emu_cas_(T *p, T *exp, T v, int so, int fo) { return __atomic_compare_exchange_n(p, exp, v, false, so, fo); }
#define __hip_atomic_compare_exchange_strong(p, exp, v, so, fo, scope) emu_cas_((p), (exp), (v), (so), (fo))

// ---- host runtime (synchronous: a "launch" returns when the grid has run) ---------------------------------------
typedef int hipError_t;
#define hipSuccess 0
#define hipErrorInvalidValue 1
#define hipErrorOutOfMemory 2
typedef struct emu_stream_ *hipStream_t;
typedef struct emu_event_ *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
#define hipStreamNonBlocking 1
#define hipHostMallocDefault 0
#define hipHostMallocPortable 1
#define hipEventDisableTiming 2
struct hipDeviceProp_t { int multiProcessorCount; char name[64]; };
const char *hipGetErrorString(hipError_t e);
hipError_t hipGetLastError(void);
hipError_t hipGetDeviceCount(int *n);
hipError_t hipGetDevice(int *d);
hipError_t hipSetDevice(int d);
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int d);
hipError_t hipDeviceSynchronize(void);
template <typename K> static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int *n, K, int, size_t) {
  const char *e = getenv("MSPACK_EMU_OCC"); *n = e ? atoi(e) : 4; return hipSuccess; }
hipError_t emu_hipMalloc(void **p, size_t n);
template <typename T> static inline hipError_t hipMalloc(T **p, size_t n) { return emu_hipMalloc((void **) p, n); }
hipError_t hipFree(void *p);
hipError_t emu_hipHostMalloc(void **p, size_t n);
template <typename T> static inline hipError_t hipHostMalloc(T **p, size_t n, unsigned flags = 0) { (void) flags; return emu_hipHostMalloc((void **) p, n); }
hipError_t hipHostFree(void *p);
/* test hook: a wave pauses (MSPACK_EMU_PUBLISH_DELAY_US microseconds, default 0) right after it has published partial
 * progress, so that the consumer's path for partial progress runs whatever the host's thread timing is */
void emu_test_delay(void);
#define hipHostRegisterDefault 0
#define hipHostRegisterPortable 1
enum hipMemoryType { hipMemoryTypeUnregistered = 0, hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2 };
struct hipPointerAttribute_t { hipMemoryType type; };
#ifdef MSPACK_HOST_CHECK     /* tests/hostcheck: the host half of shim.hip under real sanitizers, against a MODEL of the runtime's rules */
hipError_t hipHostRegister(void *p, size_t n, unsigned flags);
hipError_t hipHostUnregister(void *p);
hipError_t hipPointerGetAttributes(hipPointerAttribute_t *a, const void *p);
#else
static inline hipError_t hipHostRegister(void *, size_t, unsigned) { return hipSuccess; }
static inline hipError_t hipHostUnregister(void *) { return hipSuccess; }
static inline hipError_t hipPointerGetAttributes(hipPointerAttribute_t *a, const void *) { a->type = hipMemoryTypeUnregistered; return hipSuccess; }
#endif
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind k, hipStream_t s = 0);
hipError_t hipMemcpy(void *dst, const void *src, size_t n, hipMemcpyKind k);
hipError_t hipMemsetAsync(void *dst, int v, size_t n, hipStream_t s = 0);
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned flags);
static inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned flags, int) { return hipStreamCreateWithFlags(s, flags); }
static inline hipError_t hipDeviceGetStreamPriorityRange(int *least, int *greatest) { *least = 1; *greatest = -1; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipEventCreate(hipEvent_t *e);
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned flags);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s = 0);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b);
hipError_t hipEventDestroy(hipEvent_t e);
template <typename T> static inline hipError_t hipMemcpyFromSymbol(void *dst, const T &sym, size_t n) { memcpy(dst, &sym, n); return hipSuccess; }
template <typename T> static inline hipError_t hipMemcpyToSymbol(T &sym, const void *src, size_t n) { memcpy(&sym, src, n); return hipSuccess; }
#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) emu_launch(dim3(grid), dim3(block), [=]() { kern(__VA_ARGS__); })
