// placeholder until the Quantum kernel lands (next commit)
#pragma once
#include "wave_common.hpp"
struct QtmShared { u32 pad[4]; };
__device__ void qtm_decode_unit(const mspack_hip_unit &u, const u8 *in_arena, u8 *out_arena,
                                mspack_hip_result *res, QtmShared *sh) {
  if (threadIdx.x == 0) { res->err = ERR_ARGS; res->flags = 0; res->out_len = 0; res->in_used = 0; }
}
