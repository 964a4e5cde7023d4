// scratch micro-benchmarks: cost (in s_memtime ticks and ns) of dependent instruction patterns for ONE
// wave per SIMD on gfx950.  Not part of the product.   hipcc --offload-arch=gfx950 -O3 ubench.hip -o ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32; typedef unsigned long long u64;
#define REP 256
#define KEEPV(x) asm volatile("" : "+v"(x))
#define KEEPS(x) asm volatile("" : "+s"(x))
__device__ __forceinline__ u32 rdl(u32 v, u32 l) { return (u32) __builtin_amdgcn_readlane((int) v, (int) l); }
__device__ __forceinline__ u32 rfl(u32 v) { return (u32) __builtin_amdgcn_readfirstlane((int) v); }

template <int T>
__global__ __launch_bounds__(64) void k(u32 *g, u64 *ticks, u32 iters, u32 seed)
{
  __shared__ u32 lds[1024];
  u32 lane = threadIdx.x;
  for (u32 i = lane; i < 1024; i += 64) lds[i] = (i * 17u + 5u) & 1023u;
  __syncthreads();
  u32 v = lane + seed, s = rfl(seed), acc = 0;
  u32 *gp = g + blockIdx.x * 4096;
  u64 t0 = __builtin_amdgcn_s_memtime();
  for (u32 it = 0; it < iters; it++) {
    if (T == 0) {            // dependent VALU add chain
#pragma unroll
      for (int r = 0; r < REP; r++) { v = v + 3u; KEEPV(v); }
    } else if (T == 1) {     // dependent SALU chain
#pragma unroll
      for (int r = 0; r < REP; r++) { s = s + 3u; KEEPS(s); }
    } else if (T == 2) {     // VALU -> readlane -> SALU -> VALU round trip
#pragma unroll
      for (int r = 0; r < REP; r++) { u32 q = rdl(v, 5); q += 1u; KEEPS(q); v += q; KEEPV(v); }
    } else if (T == 3) {     // chain walk loop, 8 steps per walk
      u32 vnext = lane + 8u; KEEPV(vnext);
#pragma unroll 1
      for (int r = 0; r < REP / 8; r++) {
        u64 chain = 0; u32 q = 0;
        do { chain |= 1ull << q; q = rdl(vnext, q); } while (q < 64u);
        acc += (u32) __popcll(chain); KEEPS(acc);
      }
    } else if (T == 4) {     // dependent LDS reads
#pragma unroll
      for (int r = 0; r < REP; r++) { v = lds[v & 1023u]; }
    } else if (T == 5) {     // dependent bpermute
#pragma unroll
      for (int r = 0; r < REP; r++) { v = (u32) __builtin_amdgcn_ds_bpermute((int)((v & 63u) << 2), (int) v) + 1u; }
    } else if (T == 6) {     // dependent DPP
#pragma unroll
      for (int r = 0; r < REP; r++) { v += (u32) __builtin_amdgcn_update_dpp(0, (int) v, 0x111, 0xf, 0xf, false); }
    } else if (T == 7) {     // v_cmp -> ballot -> scalar branch (not taken)
#pragma unroll
      for (int r = 0; r < REP; r++) { u64 b = __builtin_amdgcn_ballot_w64(v == 0xFFFFFFF0u + r); if (b) { v += 7u; } v += 1u; KEEPV(v); }
    } else if (T == 8) {     // 64-bit shift chain
      u64 x = ((u64) v << 32) | lane;
#pragma unroll
      for (int r = 0; r < REP; r++) { x = (x << (v & 3u)) | 1ull; asm volatile("" : "+v"(x)); }
      v = (u32)(x >> 32) ^ (u32) x;
    } else if (T == 9) {     // dependent global byte loads (L2/L1-resident 16 KiB region)
#pragma unroll 8
      for (int r = 0; r < REP; r++) { v = ((unsigned char *) gp)[(v * 29u + lane) & 16383u] + v; }
    } else if (T == 10) {    // byte store then dependent byte load of a neighbour lane's byte (RAW through memory)
#pragma unroll 8
      for (int r = 0; r < REP; r++) {
        ((unsigned char *) gp)[(r * 64 + lane) & 16383u] = (unsigned char) v;
        v += ((unsigned char *) gp)[(r * 64 + (lane ^ 1u)) & 16383u];
      }
    } else if (T == 11) {    // taken scalar branch per step (loop of 1 SALU op)
      u32 n = REP; KEEPS(n);
#pragma unroll 1
      while (n) { n--; KEEPS(n); }
    } else if (T == 12) {    // independent VALU ops (4 chains)
      u32 a = v, b = v + 1, c = v + 2, d = v + 3;
#pragma unroll
      for (int r = 0; r < REP / 4; r++) { a += 3u; b += 5u; c += 7u; d += 9u; KEEPV(a); KEEPV(b); KEEPV(c); KEEPV(d); }
      v = a ^ b ^ c ^ d;
    } else if (T == 13) {    // v_cmp -> v_cndmask chain (VCC)
#pragma unroll
      for (int r = 0; r < REP; r++) { v = (v > 100u + r) ? v + 1u : v + 2u; KEEPV(v); }
    } else if (T == 14) {    // LDS write then read (same lane) chain
#pragma unroll
      for (int r = 0; r < REP; r++) { lds[lane] = v; __builtin_amdgcn_wave_barrier(); v = lds[(lane + 1u) & 63u] + 1u; }
    } else if (T == 15) {    // readlane with SGPR index depending on previous readlane (walk without branch)
      u32 q = 0;
#pragma unroll
      for (int r = 0; r < REP; r++) { q = rdl(v, q) & 63u; KEEPS(q); }
      acc += q;
    } else if (T == 16) {    // 4 independent LDS reads then use
#pragma unroll
      for (int r = 0; r < REP / 4; r++) {
        u32 a = lds[v & 1023u], b = lds[(v + 1) & 1023u], c = lds[(v + 2) & 1023u], d = lds[(v + 3) & 1023u];
        v = a + b + c + d;
      }
    }
  }
  u64 t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) ticks[blockIdx.x] = t1 - t0;
  g[blockIdx.x * 4096 + lane] = v + s + acc;
}

template <int T> void run(const char *name, u32 *g, u64 *ticks, int blocks, u32 iters, int steps_per_rep)
{
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<T><<<blocks, 64>>>(g, ticks, 2, 1);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<T><<<blocks, 64>>>(g, ticks, iters, 1);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<u64> h(blocks); hipMemcpy(h.data(), ticks, blocks * 8, hipMemcpyDeviceToHost);
  double avg = 0; for (auto x : h) avg += (double) x; avg /= blocks;
  double steps = (double) iters * REP * steps_per_rep / 1.0;
  printf("%-44s blocks %5d  %8.2f ticks/step  %8.2f ns/step\n", name, blocks, avg / steps, ms * 1e6 / steps);
}

int main(int argc, char **argv)
{
  u32 *g; u64 *ticks; int maxb = 8192;
  hipMalloc(&g, (size_t) maxb * 4096 * 4); hipMemset(g, 1, (size_t) maxb * 4096 * 4); hipMalloc(&ticks, maxb * 8);
  for (int blocks : {1024, 4096}) {
    u32 it = 64;
    run<0>("dependent VALU add", g, ticks, blocks, it, 1);
    run<12>("independent VALU add x4", g, ticks, blocks, it, 1);
    run<1>("dependent SALU add", g, ticks, blocks, it, 1);
    run<2>("readlane->SALU->VALU round trip", g, ticks, blocks, it, 1);
    run<3>("chain walk step (loop)", g, ticks, blocks, it, 1);
    run<15>("readlane(idx from readlane) chain", g, ticks, blocks, it, 1);
    run<4>("dependent LDS read", g, ticks, blocks, it, 1);
    run<16>("4 independent LDS reads (per read)", g, ticks, blocks, it, 1);
    run<14>("LDS write + read neighbour", g, ticks, blocks, it, 1);
    run<5>("dependent ds_bpermute", g, ticks, blocks, it, 1);
    run<6>("dependent DPP add", g, ticks, blocks, it, 1);
    run<7>("v_cmp->ballot->branch(not taken)+add", g, ticks, blocks, it, 1);
    run<13>("v_cmp->v_cndmask chain", g, ticks, blocks, it, 1);
    run<8>("dependent 64-bit shift|or", g, ticks, blocks, it, 1);
    run<11>("taken scalar branch loop step", g, ticks, blocks, it, 1);
    run<9>("dependent global byte load", g, ticks, blocks, 8, 1);
    run<10>("byte store + load neighbour byte", g, ticks, blocks, 8, 1);
  }
  return 0;
}
