# scratch (GPU box): the host-buffer entry point (mspack_hip_decode_batch: H2D + kernels + D2H) on the headline batch
import sys, time, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import libmspack_amd as M
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ub = 65536
plain, comp, off, ln = M.corpus_lzx_units(0xBA5E11, 0, n, ub, 21)
units, out_bytes = M.make_units(M.KIND_LZX, off, ln + 4, np.full(n, ub), window_bits=21, reset_frames=2)
for it in range(4):
    t0 = time.perf_counter()
    out, res = M.decode_batch(units, comp, out_bytes)
    dt = time.perf_counter() - t0
    print("decode_batch (host buffers): %.1f ms  %.1f GB/s decompressed  (in %.0f MB, out %.0f MB)%s" %
          (dt * 1e3, n * ub / dt / 1e9, comp.size / 1e6, out_bytes / 1e6, "" if np.array_equal(out[:n * ub], plain) else "  MISMATCH"))
