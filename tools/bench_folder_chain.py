"""scratch (GPU box): ONE long folder per codec through mspack_hip_decode_batch -- the per-folder chain.
  python tools/bench_folder_chain.py [blocks]       (run under rocprofv3 --kernel-trace --stats for the split by kernel)
large-files.cab's three folders cut to N CFDATA blocks (Microsoft's encoder: a 64-byte line repeated; LZX blocks of megabytes),
and this build's text corpus as one LZX-21 folder of 512 frames and one MSZIP folder of 2000 blocks."""
import os, sys, time, zlib
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import libmspack_amd as M
import helpers
from test_gpu_large_files import inner_cabinet
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4096


def one(name, kind, stream, tab, total, wb):
    base = (len(stream) + 64 + 15) & ~15
    arena = np.zeros(base + 4 * len(tab) + 64, dtype=np.uint8)
    arena[:len(stream)] = np.frombuffer(stream, dtype=np.uint8)
    arena[base:base + 4 * len(tab)] = np.asarray(tab, dtype=np.uint32).view(np.uint8)
    units, out_bytes = M.make_units(kind, [0], [len(stream)], [total], window_bits=wb, reset_frames=0, frame_tabs=[base],
                                    out_slack=32768 if kind == M.KIND_MSZIP else 0)
    best = 1e9
    for it in range(3):
        t0 = time.perf_counter()
        out, res = M.decode_batch(units, arena, out_bytes)
        best = min(best, time.perf_counter() - t0)
    assert res["err"][0] == 0 and res["out_len"][0] == total, res[0]
    print("%-34s %5d blocks %9d bytes: %8.1f ms host wall = %7.1f MB/s (adopted %d)" %
          (name, len(tab), total, best * 1e3, total / best / 1e6, int(res["flags"][0] & 32 != 0)), flush=True)
    return out[:total]


folders = helpers.cab_folders(helpers.cab_cut_folders(inner_cabinet(), nb))
for nm, f in zip(("large-files MSZIP", "large-files LZX-15", "large-files LZX-21"), folders):
    blocks = [p for p, _u in f["blocks"]]
    total = sum(u for _p, u in f["blocks"])
    m = f["comp_type"] & 0x0F
    tab = np.cumsum([0] + [len(b) for b in blocks[:-1]])
    one(nm, M.KIND_MSZIP if m == 1 else M.KIND_LZX, b"".join(blocks), tab, total, (f["comp_type"] >> 8) & 0x1F)
plain = M.gen_plaintext(77, 0, 512 * 32768)
lz, fo = M.lzx_encode(plain, 21, 0)
o = one("text, LZX-21 (one block per frame)", M.KIND_LZX, lz.tobytes(), np.asarray(fo[:-1]), plain.size, 21)
assert np.array_equal(o, plain)
lz, fo = M.lzx_encode(plain, 21, 0, M.lzx_opts(block_size=4 << 20))
o = one("text, LZX-21 (4 MiB blocks)", M.KIND_LZX, lz.tobytes(), np.asarray(fo[:-1]), plain.size, 21)
assert np.array_equal(o, plain)
plain = M.gen_plaintext(78, 0, 2000 * 32768)
blocks, prev = [], None
for k in range(0, plain.size, 32768):
    b = plain[k:k + 32768].tobytes()
    c = zlib.compressobj(6, zlib.DEFLATED, -15, 9, 0, prev) if prev else zlib.compressobj(6, zlib.DEFLATED, -15)
    blocks.append(b"CK" + c.compress(b) + c.flush()); prev = b
o = one("text, MSZIP (history)", M.KIND_MSZIP, b"".join(blocks), np.cumsum([0] + [len(b) for b in blocks[:-1]]), plain.size, 0)
assert np.array_equal(o, plain)
