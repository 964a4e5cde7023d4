"""scratch (GPU box): where a fold task's time goes (mspack_lzx_fold, lzx_fold.hpp) -- the text corpus as ONE LZX-21 folder of N frames through
a -DFOLD_TRACE build (tools/build_variant.sh ftrace -DFOLD_TRACE): per-phase sums over all tasks, in us per frame.
  MSPACK_HIP_SO=build/variants/libmspack_hip_ftrace.so python tools/fold_phases.py [frames] [block_size]"""
import ctypes as C, os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import libmspack_amd as M
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 512
bs = int(sys.argv[2]) if len(sys.argv) > 2 else (4 << 20)
plain = M.gen_plaintext(77, 0, nf * 32768)
lz, fo = M.lzx_encode(plain, 21, 0, M.lzx_opts(block_size=bs))
stream, tab = lz.tobytes(), np.asarray(fo[:-1])
base = (len(stream) + 64 + 15) & ~15
arena = np.zeros(base + 4 * len(tab) + 64, dtype=np.uint8)
arena[:len(stream)] = np.frombuffer(stream, dtype=np.uint8)
arena[base:base + 4 * len(tab)] = np.asarray(tab, dtype=np.uint32).view(np.uint8)
units, out_bytes = M.make_units(M.KIND_LZX, [0], [len(stream)], [plain.size], window_bits=21, reset_frames=0, frame_tabs=[base])
L = M.lib()
ph = (C.c_ulonglong * 16)()
have = hasattr(L, "mspack_hip_debug_fold_phases")
for it in range(3):
    if have: L.mspack_hip_debug_fold_phases(ph)
    t0 = time.perf_counter()
    out, res = M.decode_batch(units, arena, out_bytes)
    dt = time.perf_counter() - t0
    assert res["err"][0] == 0 and np.array_equal(out[:plain.size], plain)
    print("run %d: %.1f ms host wall = %.1f MB/s" % (it, dt * 1e3, plain.size / dt / 1e6))
    if have:
        L.mspack_hip_debug_fold_phases(ph)
        n = max(int(ph[8]), 1)
        names = ["records -> map", "jumps", "own bytes", "wait frame f-2", "early gather + list", "wait frame f-1", "late gather", "publish", "(tasks)", "wait R0-R2"]
        print("  %d fold tasks; us per task: " % n + ", ".join("%s %.1f" % (names[k], ph[k] / 100.0 / n) for k in (0, 9, 1, 2, 3, 4, 5, 6, 7)))
