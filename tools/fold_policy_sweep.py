"""scratch (GPU box): where do the fold tasks pay?  n folders of f frames each (text, LZX-21 with blocks of 1 MiB; MSZIP with history) through
mspack_hip_decode_batch -- run it with MSPACK_HIP_FOLD=0, 1 and 2 and compare (the rule: shim.hip, lzx_fold_on).
  MSPACK_HIP_FOLD=1 python tools/fold_policy_sweep.py"""
import os, sys, time, zlib
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import libmspack_amd as M
shapes = [(4, 256), (16, 64), (32, 32), (64, 16), (128, 8), (128, 4), (256, 8), (512, 4), (1024, 2)]
pol = os.environ.get("MSPACK_HIP_FOLD", "1")
for kind in (M.KIND_LZX, M.KIND_MSZIP):
    for n, f in shapes:
        ub = f * 32768
        plain = M.gen_plaintext(900 + n, 0, ub)          # (every folder the same text: the launch's shape is what is measured)
        if kind == M.KIND_LZX:
            lz, fo = M.lzx_encode(plain, 21, 0, M.lzx_opts(block_size=1 << 20))
            stream, tab = lz.tobytes(), np.asarray(fo[:-1], dtype=np.uint32)
        else:
            blocks, prev = [], None
            for k in range(0, ub, 32768):
                b = plain[k:k + 32768].tobytes()
                c = zlib.compressobj(6, zlib.DEFLATED, -15, 9, 0, prev) if prev else zlib.compressobj(6, zlib.DEFLATED, -15)
                blocks.append(b"CK" + c.compress(b) + c.flush()); prev = b
            stream, tab = b"".join(blocks), np.cumsum([0] + [len(b) for b in blocks[:-1]]).astype(np.uint32)
        per = (len(stream) + 4 * len(tab) + 64 + 15) & ~15
        arena = np.zeros(per * n + 64, dtype=np.uint8)
        offs, tabs = [], []
        for i in range(n):
            o = i * per
            arena[o:o + len(stream)] = np.frombuffer(stream, dtype=np.uint8)
            t = (o + len(stream) + 3) & ~3
            arena[t:t + 4 * len(tab)] = tab.view(np.uint8)
            offs.append(o); tabs.append(t)
        units, out_bytes = M.make_units(kind, offs, [len(stream)] * n, [ub] * n, window_bits=21 if kind == M.KIND_LZX else 0, reset_frames=0,
                                        frame_tabs=tabs, out_slack=32768 if kind == M.KIND_MSZIP else 0)
        best = 1e9
        for it in range(3):
            t0 = time.perf_counter()
            out, res = M.decode_batch(units, arena, out_bytes)
            best = min(best, time.perf_counter() - t0)
        ok = bool((res["err"] == 0).all()) and all(np.array_equal(out[int(units["out_off"][i]):int(units["out_off"][i]) + ub], plain) for i in (0, n // 2, n - 1))
        print("FOLD=%s %-5s %5d folders x %4d frames (%4d MiB): %8.2f ms host wall = %8.1f MB/s %s" %
              (pol, "LZX" if kind == M.KIND_LZX else "MSZIP", n, f, n * ub >> 20, best * 1e3, n * ub / best / 1e6, "ok" if ok else "WRONG"), flush=True)
