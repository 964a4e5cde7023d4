"""scratch: a small CHM-style LZX batch with frame tables through the C ABI (meant for the wavefront emulator:
MSPACK_HIP_SO=tests/_build/libmspack_emu.so python tools/emu/pipe_small.py [units] [unit_bytes])"""
import sys, os, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import libmspack_amd as M
n_units = int(sys.argv[1]) if len(sys.argv) > 1 else 6
ub = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
plain, comp, off, ln, tab = M.corpus_lzx_units(0xBA5E11, 0, n_units, ub, 21, frame_tables=True)
units, out_bytes = M.make_units(M.KIND_LZX, off, ln + 4, np.full(n_units, ub), window_bits=21, reset_frames=2, frame_tabs=tab)
t0 = time.time()
out, res = M.decode_batch(units, comp, out_bytes)
print("decode %.1fs" % (time.time() - t0))
print(res)
print("bit-exact:", np.array_equal(out[:n_units * ub], plain), "adopted:", ((res["flags"] & M.F_FRAMES_ADOPTED) != 0).mean())
