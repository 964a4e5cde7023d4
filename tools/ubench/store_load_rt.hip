// Microbenchmark: what does "store 64 bytes, then gather bytes written D bytes earlier" cost per step on gfx950?
// (the LZ77 match resolver's inner dependency).  One wave per block; every step's stored value depends on the
// loaded one.  hipcc --offload-arch=gfx950 -O3 store_load_rt.hip -o /tmp/rt && /tmp/rt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(64) void k(unsigned char *buf, size_t per_wave, unsigned dist, int steps, int mode,
                                         unsigned long long *cycles)
{
  unsigned char *b = buf + (size_t) blockIdx.x * per_wave + 65536;
  const unsigned lane = threadIdx.x;
  unsigned v = lane;
  __shared__ unsigned char ring[4096];
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < steps; i++) {
    unsigned char *dst = b + (size_t) i * 64u;
    if (mode == 2) {                                   // LDS ring instead of memory for the source
      ring[(i * 64u + lane) & 4095u] = (unsigned char) v;
      dst[lane] = (unsigned char) v;
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      v = ring[(i * 64u + ((lane * 7u) & 63u) - (dist & 4032u)) & 4095u] + 1u;
    } else if (mode == 1) {                            // scattered byte stores (literals) + coalesced store, then gather
      dst[lane] = (unsigned char) v;
      v = dst[(int)((lane * 7u) & 63u) - (int) dist] + 1u;
    } else {                                           // store dwords (16 lanes), then gather
      if (lane < 16u) ((unsigned *) dst)[lane] = v * 0x01010101u;
      v = dst[(int)((lane * 7u) & 63u) - (int) dist] + 1u;
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cycles[blockIdx.x] = t1 - t0;
  if (v == 0xFFFFFFFFu) buf[0] = 1;
}
int main()
{
  const size_t per_wave = 65536 + 4096 * 64 + 65536;
  const int steps = 4096;
  for (int nw : {1, 4096}) {
    unsigned char *buf; unsigned long long *cyc;
    hipMalloc(&buf, per_wave * nw); hipMalloc(&cyc, 8 * nw);
    hipMemset(buf, 1, per_wave * nw);
    for (int mode : {0, 1, 2}) for (unsigned dist : {64u, 128u, 512u, 2048u, 8192u, 32768u}) {
      hipLaunchKernelGGL(k, dim3(nw), dim3(64), 0, 0, buf, per_wave, dist, steps, mode, cyc);
      hipDeviceSynchronize();
      std::vector<unsigned long long> h(nw);
      hipMemcpy(h.data(), cyc, 8 * nw, hipMemcpyDeviceToHost);
      double s = 0; for (auto x : h) s += (double) x;
      printf("waves %4d mode %d (%s) dist %6u: %.0f cycles/step\n", nw, mode,
             mode == 0 ? "dword store, byte gather" : mode == 1 ? "byte store, byte gather " : "LDS ring source        ", dist, s / nw / steps);
    }
    hipFree(buf); hipFree(cyc);
  }
  return 0;
}
