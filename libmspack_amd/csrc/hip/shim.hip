// shim.hip -- kernels' entry point + the extern "C" ABI declared in include/mspack_hip.h.
// Host code elsewhere in the library is plain C and reaches HIP only through these functions.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <thread>
#include <chrono>
#include <vector>
#include <algorithm>
#include "wave_common.hpp"
#include "spec_queue.hpp"
// lzx_kernel.hpp is compiled twice: plain LZX (CAB, CHM) and LZX DELTA (OAB) -- see its header
namespace lzxn {
#include "lzx_kernel.hpp"
}
#define LZX_DELTA 1
namespace lzxd {
#include "lzx_kernel.hpp"
}
#undef LZX_DELTA
#include "mszip_kernel.hpp"
#include "qtm_kernel.hpp"
#include "lzss_kernel.hpp"

// One wavefront == one workgroup == one unit.  blockIdx -> unit through the optional launch order
// (longest unit first keeps the tail of the batch short).  One kernel per codec (their register
// budgets differ a lot); a block whose unit belongs to another codec exits at once.
__device__ __forceinline__ bool pick_unit(const mspack_hip_unit *units, const u32 *order, u32 n_units,
                                          u32 kind, u32 &ui)
{
  u32 b = blockIdx.x;
  if (b >= n_units) return false;
  ui = rfl(order ? order[b] : b);
  return units[ui].kind == kind;
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4)))
void mspack_decode_lzx(const mspack_hip_unit *units, const u32 *order, u32 n_units,
                       const u8 *in_arena, u8 *out_arena, mspack_hip_result *results,
                       int32_t *frame_meta)
{
  __shared__ lzxn::LzxShared sh;
  u32 ui;
  if (!pick_unit(units, order, n_units, MSPACK_HIP_KIND_LZX, ui)) return;
  const mspack_hip_unit u = units[ui];
  mspack_hip_result *res = &results[ui];
  const u32 lane = threadIdx.x;
  lzxn::lzx_decode_unit(u, in_arena, out_arena, frame_meta, res, &sh);
  // E8 translation, frame by frame, once the unit no longer needs its window (lzxd.c:706-736)
  if (frame_meta) {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    u32 produced = rfl(res->out_len);
    u32 nfr = (produced + LZX_FRAME - 1u) / LZX_FRAME;
    for (u32 f = 0; f < nfr; f++) {
      int32_t fs = (int32_t) rfl((u32) frame_meta[u.frame_base + f]);
      if (fs == 0) continue;
      // the frame size the decoder saw: full frames except the last one of the stream
      u32 fsize = u.out_len - f * LZX_FRAME; if (fsize > LZX_FRAME) fsize = LZX_FRAME;
      lzxn::lzx_e8_frame(out_arena + u.out_off + (size_t) f * LZX_FRAME, fsize,
                         (int32_t)((u32) u.e8_base + f * LZX_FRAME), fs, lane);
    }
  }
}

// LZX DELTA units (OAB blocks): the same decoder compiled with LZX_DELTA (17.4 KiB of LDS: 9 units per CU)
__global__ __launch_bounds__(64)
void mspack_decode_lzxd(const mspack_hip_unit *units, const u32 *order, u32 n_units,
                        const u8 *in_arena, u8 *out_arena, mspack_hip_result *results,
                        int32_t *frame_meta)
{
  __shared__ lzxd::LzxShared sh;
  u32 ui;
  if (!pick_unit(units, order, n_units, MSPACK_HIP_KIND_LZX_DELTA, ui)) return;
  const mspack_hip_unit u = units[ui];
  mspack_hip_result *res = &results[ui];
  const u32 lane = threadIdx.x;
  lzxd::lzx_decode_unit(u, in_arena, out_arena, frame_meta, res, &sh);
  if (frame_meta) {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    u32 produced = rfl(res->out_len);
    u32 nfr = (produced + LZX_FRAME - 1u) / LZX_FRAME;
    for (u32 f = 0; f < nfr; f++) {
      int32_t fs = (int32_t) rfl((u32) frame_meta[u.frame_base + f]);
      if (fs == 0) continue;
      u32 fsize = u.out_len - f * LZX_FRAME; if (fsize > LZX_FRAME) fsize = LZX_FRAME;
      lzxd::lzx_e8_frame(out_arena + u.out_off + (size_t) f * LZX_FRAME, fsize,
                         (int32_t)((u32) u.e8_base + f * LZX_FRAME), fs, lane);
    }
  }
}

__global__ __launch_bounds__(64)
void mspack_decode_mszip(const mspack_hip_unit *units, const u32 *order, u32 n_units,
                         const u8 *in_arena, u8 *out_arena, mspack_hip_result *results)
{
  __shared__ MszipShared sh;
  u32 ui;
  if (!pick_unit(units, order, n_units, MSPACK_HIP_KIND_MSZIP, ui)) return;
  const mspack_hip_unit u = units[ui];
  mszip_decode_unit(u, in_arena, out_arena, &results[ui], &sh);
}

__global__ __launch_bounds__(64)
void mspack_decode_qtm(const mspack_hip_unit *units, const u32 *order, u32 n_units,
                       const u8 *in_arena, u8 *out_arena, mspack_hip_result *results)
{
  __shared__ QtmShared sh;
  u32 ui;
  if (!pick_unit(units, order, n_units, MSPACK_HIP_KIND_QUANTUM, ui)) return;
  const mspack_hip_unit u = units[ui];
  qtm_decode_unit(u, in_arena, out_arena, &results[ui], &sh);
}

__global__ __launch_bounds__(64)
void mspack_decode_lzss(const mspack_hip_unit *units, const u32 *order, u32 n_units,
                        const u8 *in_arena, u8 *out_arena, mspack_hip_result *results)
{
  u32 ui;
  if (!pick_unit(units, order, n_units, MSPACK_HIP_KIND_LZSS, ui)) return;
  const mspack_hip_unit u = units[ui];
  lzss_decode_unit(u, in_arena, out_arena, &results[ui]);
}

__global__ __launch_bounds__(64)
void mspack_decode_kwaj_lzh(const mspack_hip_unit *units, const u32 *order, u32 n_units,
                            const u8 *in_arena, u8 *out_arena, mspack_hip_result *results)
{
  __shared__ LzhShared sh;
  u32 ui;
  if (!pick_unit(units, order, n_units, MSPACK_HIP_KIND_KWAJ_LZH, ui)) return;
  const mspack_hip_unit u = units[ui];
  kwaj_lzh_decode_unit(u, in_arena, out_arena, &results[ui], &sh);
}

// ---------------------------------------------------------------------------------------------------
static thread_local char g_err[256] = "";
static int fail(hipError_t e, const char *what) {
  snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
  return -(int) e;
}
#define CK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(e_, #call); } while (0)

extern "C" {

const char *mspack_hip_version(void) { return "mspack-hip 0.2 (gfx950; LZX/LZX-DELTA/Quantum/MSZIP batch decode)"; }
const char *mspack_hip_last_error(void) { return g_err; }

int mspack_hip_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) { fail(e, "hipGetDeviceCount"); return 0; }
  return n;
}
int mspack_hip_set_device(int device) { CK(hipSetDevice(device)); return 0; }

size_t mspack_hip_frame_scratch_bytes(size_t n_frames_total) { return (n_frames_total + 1) * sizeof(int32_t); }

int mspack_hip_decode_batch_device(const mspack_hip_unit *d_units, const uint32_t *d_order,
                                   size_t n_units, const void *d_in, size_t in_bytes,
                                   void *d_out, size_t out_bytes, mspack_hip_result *d_results,
                                   void *d_frame_scratch, size_t n_frames_total, unsigned kind_mask,
                                   void *stream)
{
  (void) in_bytes; (void) out_bytes; (void) n_frames_total;
  if (n_units == 0) return 0;
  if (kind_mask == 0) kind_mask = 0x7E;     // bit k = units of kind k may be present
  const dim3 grid((unsigned) n_units), block(64);
  hipStream_t st = (hipStream_t) stream;
  if (kind_mask & (1u << MSPACK_HIP_KIND_LZX))
    hipLaunchKernelGGL(mspack_decode_lzx, grid, block, 0, st, d_units, d_order, (u32) n_units,
                       (const u8 *) d_in, (u8 *) d_out, d_results, (int32_t *) d_frame_scratch);
  if (kind_mask & (1u << MSPACK_HIP_KIND_LZX_DELTA))
    hipLaunchKernelGGL(mspack_decode_lzxd, grid, block, 0, st, d_units, d_order, (u32) n_units,
                       (const u8 *) d_in, (u8 *) d_out, d_results, (int32_t *) d_frame_scratch);
  if (kind_mask & (1u << MSPACK_HIP_KIND_MSZIP))
    hipLaunchKernelGGL(mspack_decode_mszip, grid, block, 0, st, d_units, d_order, (u32) n_units,
                       (const u8 *) d_in, (u8 *) d_out, d_results);
  if (kind_mask & (1u << MSPACK_HIP_KIND_QUANTUM))
    hipLaunchKernelGGL(mspack_decode_qtm, grid, block, 0, st, d_units, d_order, (u32) n_units,
                       (const u8 *) d_in, (u8 *) d_out, d_results);
  if (kind_mask & (1u << MSPACK_HIP_KIND_LZSS))
    hipLaunchKernelGGL(mspack_decode_lzss, grid, block, 0, st, d_units, d_order, (u32) n_units,
                       (const u8 *) d_in, (u8 *) d_out, d_results);
  if (kind_mask & (1u << MSPACK_HIP_KIND_KWAJ_LZH))
    hipLaunchKernelGGL(mspack_decode_kwaj_lzh, grid, block, 0, st, d_units, d_order, (u32) n_units,
                       (const u8 *) d_in, (u8 *) d_out, d_results);
  CK(hipGetLastError());
  return 0;
}

double mspack_hip_time_batch_device(const mspack_hip_unit *d_units, const uint32_t *d_order,
                                    size_t n_units, const void *d_in, size_t in_bytes,
                                    void *d_out, size_t out_bytes, mspack_hip_result *d_results,
                                    void *d_frame_scratch, size_t n_frames_total, unsigned kind_mask,
                                    void *stream, int iters)
{
  hipEvent_t e0, e1;
  float ms = 0;
  if (iters < 1) iters = 1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1.0;
  hipEventRecord(e0, (hipStream_t) stream);
  for (int i = 0; i < iters; i++) {
    int rc = mspack_hip_decode_batch_device(d_units, d_order, n_units, d_in, in_bytes, d_out, out_bytes,
                                            d_results, d_frame_scratch, n_frames_total, kind_mask, stream);
    if (rc) { hipEventDestroy(e0); hipEventDestroy(e1); return -1.0; }
  }
  hipEventRecord(e1, (hipStream_t) stream);
  if (hipEventSynchronize(e1) != hipSuccess) { hipEventDestroy(e0); hipEventDestroy(e1); return -1.0; }
  hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0); hipEventDestroy(e1);
  return (double) ms / iters;
}

// frames (incl. the look-ahead slot) a unit needs in the per-frame scratch
static inline size_t unit_frames(const mspack_hip_unit *u) {
  return (u->kind == MSPACK_HIP_KIND_LZX || u->kind == MSPACK_HIP_KIND_LZX_DELTA) ? (size_t) u->out_len / 32768u + 1u : 0u;
}

static int decode_on_current_device(mspack_hip_unit *units, const uint32_t *sel, size_t n_sel,
                                    const void *in, size_t in_bytes, void *out, size_t out_bytes,
                                    mspack_hip_result *results)
{
  // `sel` lists the unit indices this device handles (NULL = all n_sel units, identity).
  // Units keep their arena offsets; only the arenas' touched extents are staged.
  if (n_sel == 0) return 0;
  std::vector<mspack_hip_unit> local(n_sel);
  std::vector<uint32_t> order(n_sel);
  size_t n_frames = 0;
  unsigned kind_mask = 0;
  uint64_t in_lo = ~0ull, in_hi = 0, out_lo = ~0ull, out_hi = 0;
  for (size_t i = 0; i < n_sel; i++) {
    size_t ui = sel ? sel[i] : i;
    local[i] = units[ui];
    local[i].frame_base = (uint32_t) n_frames;
    units[ui].frame_base = (uint32_t) n_frames;
    n_frames += unit_frames(&local[i]);
    kind_mask |= 1u << (local[i].kind & 31u);
    in_lo = std::min<uint64_t>(in_lo, local[i].in_off);
    in_hi = std::max<uint64_t>(in_hi, local[i].in_off + local[i].in_len);
    if (local[i].kind != MSPACK_HIP_KIND_LZX_DELTA) local[i].ref_len = 0;
    // bytes below out_off that belong to the unit: DELTA reference data, the LZSS / LZH window pre-fill
    const uint64_t below = (local[i].kind == MSPACK_HIP_KIND_LZSS || local[i].kind == MSPACK_HIP_KIND_KWAJ_LZH)
                           ? 4096u : local[i].ref_len;
    if (below > local[i].out_off) { snprintf(g_err, sizeof(g_err), "unit's lower region outside arena"); return -1; }
    out_lo = std::min<uint64_t>(out_lo, local[i].out_off - below);
    // MSZIP decodes whole blocks: its region carries 32768 bytes of slack (see mszip_kernel.hpp)
    out_hi = std::max<uint64_t>(out_hi, local[i].out_off + local[i].out_len +
                                        (local[i].kind == MSPACK_HIP_KIND_MSZIP ? 32768u : 0u));
    order[i] = (uint32_t) i;
  }
  if (in_hi > in_bytes || out_hi > out_bytes) { snprintf(g_err, sizeof(g_err), "unit outside arena"); return -1; }
  in_lo &= ~15ull;                                    // keep the units' alignment
  for (size_t i = 0; i < n_sel; i++) { local[i].in_off -= in_lo; local[i].out_off -= out_lo; }
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
    return local[a].in_len + (local[a].out_len >> 2) > local[b].in_len + (local[b].out_len >> 2); });

  size_t in_span = (size_t)(in_hi - in_lo), out_span = (size_t)(out_hi - out_lo);
  // MSPACK_HIP_TRACE=1: phase times of the host-buffer path on stderr
  static const bool trace = getenv("MSPACK_HIP_TRACE") != nullptr;
  auto tnow = []() { return std::chrono::steady_clock::now(); };
  auto tms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return std::chrono::duration<double, std::milli>(b - a).count(); };
  auto t0 = tnow(), t1 = t0, t2 = t0, t3 = t0, t4 = t0;
  void *d_in = nullptr, *d_out = nullptr, *d_units = nullptr, *d_order = nullptr, *d_res = nullptr, *d_fm = nullptr;
  int rc = 0;
  hipError_t e;
#define TRY(call) do { e = (call); if (e != hipSuccess) { rc = fail(e, #call); goto done; } } while (0)
  TRY(hipMalloc(&d_in, in_span + 64));
  TRY(hipMalloc(&d_out, out_span + 64));
  TRY(hipMalloc(&d_units, n_sel * sizeof(mspack_hip_unit)));
  TRY(hipMalloc(&d_order, n_sel * sizeof(uint32_t)));
  TRY(hipMalloc(&d_res, n_sel * sizeof(mspack_hip_result)));
  TRY(hipMalloc(&d_fm, mspack_hip_frame_scratch_bytes(n_frames)));
  t1 = tnow();
  TRY(hipMemcpy(d_in, (const char *) in + in_lo, in_span, hipMemcpyHostToDevice));
  TRY(hipMemset((char *) d_in + in_span, 0, 64));
  TRY(hipMemcpy(d_units, local.data(), n_sel * sizeof(mspack_hip_unit), hipMemcpyHostToDevice));
  TRY(hipMemcpy(d_order, order.data(), n_sel * sizeof(uint32_t), hipMemcpyHostToDevice));
  TRY(hipMemset(d_fm, 0, mspack_hip_frame_scratch_bytes(n_frames)));
  for (size_t i = 0; i < n_sel; i++)                  // LZX DELTA reference data sits below the unit's output
    if (local[i].ref_len)
      TRY(hipMemcpy((char *) d_out + local[i].out_off - local[i].ref_len,
                    (const char *) out + out_lo + local[i].out_off - local[i].ref_len, local[i].ref_len,
                    hipMemcpyHostToDevice));
  t2 = tnow();
  rc = mspack_hip_decode_batch_device((const mspack_hip_unit *) d_units, (const uint32_t *) d_order, n_sel,
                                      d_in, in_span, d_out, out_span, (mspack_hip_result *) d_res, d_fm,
                                      n_frames, kind_mask & 0x7E, nullptr);
  if (rc) goto done;
  TRY(hipDeviceSynchronize());
  t3 = tnow();
  {
    std::vector<mspack_hip_result> r(n_sel);
    TRY(hipMemcpy(r.data(), d_res, n_sel * sizeof(mspack_hip_result), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n_sel; i++) results[sel ? sel[i] : i] = r[i];
    // copy back each unit's produced bytes (units may interleave with other devices' ranges)
    if (!sel) TRY(hipMemcpy((char *) out + out_lo, d_out, out_span, hipMemcpyDeviceToHost));
    else {
      for (size_t i = 0; i < n_sel; i++)
        TRY(hipMemcpy((char *) out + out_lo + local[i].out_off, (char *) d_out + local[i].out_off,
                      local[i].out_len, hipMemcpyDeviceToHost));
    }
  }
  t4 = tnow();
done:
  hipFree(d_in); hipFree(d_out); hipFree(d_units); hipFree(d_order); hipFree(d_res); hipFree(d_fm);
  if (trace && rc == 0)
    fprintf(stderr, "mspack_hip: %zu units: alloc %.2f ms, H2D %.1f MB %.2f ms, kernels %.2f ms, D2H %.1f MB %.2f ms, free %.2f ms\n",
            n_sel, tms(t0, t1), in_span / 1e6, tms(t1, t2), tms(t2, t3), out_span / 1e6, tms(t3, t4), tms(t4, tnow()));
  return rc;
#undef TRY
}

int mspack_hip_decode_batch(mspack_hip_unit *units, size_t n_units, const void *in, size_t in_bytes,
                            void *out, size_t out_bytes, mspack_hip_result *results)
{
  return decode_on_current_device(units, nullptr, n_units, in, in_bytes, out, out_bytes, results);
}

int mspack_hip_decode_batch_multi(mspack_hip_unit *units, size_t n_units, const void *in,
                                  size_t in_bytes, void *out, size_t out_bytes,
                                  mspack_hip_result *results, int n_devices)
{
  int have = mspack_hip_device_count();
  if (n_devices > have) n_devices = have;
  if (n_devices <= 1) return decode_on_current_device(units, nullptr, n_units, in, in_bytes, out, out_bytes, results);
  // deal units longest-first round-robin: static sharding, no inter-device traffic
  std::vector<uint32_t> idx(n_units);
  for (size_t i = 0; i < n_units; i++) idx[i] = (uint32_t) i;
  std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return units[a].in_len > units[b].in_len; });
  std::vector<std::vector<uint32_t>> shard(n_devices);
  for (size_t i = 0; i < n_units; i++) shard[i % n_devices].push_back(idx[i]);
  std::vector<int> rcs(n_devices, 0);
  std::vector<std::thread> th;
  for (int dv = 0; dv < n_devices; dv++) {
    th.emplace_back([&, dv]() {
      if (hipSetDevice(dv) != hipSuccess) { rcs[dv] = -1; return; }
      rcs[dv] = decode_on_current_device(units, shard[dv].data(), shard[dv].size(), in, in_bytes, out, out_bytes, results);
    });
  }
  for (auto &t : th) t.join();
  for (int dv = 0; dv < n_devices; dv++) if (rcs[dv]) return rcs[dv];
  return 0;
}

} // extern "C"
