"""scratch (analysis build -DLZX_PIPE_TRACE): ONE long LZX folder through mspack_lzx_pipe -- where does a frame's time go when the
unit is a chain of thousands of frames?
MSPACK_HIP_NCHUNKS=1 MSPACK_HIP_SO=build/variants/libmspack_hip_trace.so python tools/pipe_trace_folder.py [large|text] [frames]
  large: the reference's large-files.cab, LZX-21 folder, first N CFDATA blocks (a 64-byte line repeated)
  text : N frames of the bench corpus' text as one CAB-style stream"""
import ctypes, os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import libmspack_amd as M
what = sys.argv[1] if len(sys.argv) > 1 else "large"
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
if what == "large":
    import helpers
    from test_gpu_large_files import inner_cabinet
    f = helpers.cab_folders(helpers.cab_cut_folders(inner_cabinet(), nf))[2]
    blocks = [p for p, _u in f["blocks"]]
    total = sum(u for _p, u in f["blocks"])
    wb = (f["comp_type"] >> 8) & 0x1F
    stream = b"".join(blocks)
    fo = np.cumsum([0] + [len(b) for b in blocks[:-1]]).astype(np.uint32)
    plain = None
else:
    plain = M.gen_plaintext(77, 0, nf * 32768)
    lz, fo = M.lzx_encode(plain, 21, 0)
    stream = lz.tobytes(); total = plain.size; wb = 21
    fo = np.asarray(fo[:-1], dtype=np.uint32)
base = (len(stream) + 64 + 15) & ~15
arena = np.zeros(base + 4 * len(fo) + 64, dtype=np.uint8)
arena[:len(stream)] = np.frombuffer(stream, dtype=np.uint8)
arena[base:base + 4 * len(fo)] = fo.view(np.uint8)
units, out_bytes = M.make_units(M.KIND_LZX, [0], [len(stream)], [total], window_bits=wb, reset_frames=0, frame_tabs=[base])
L = M.lib()
ph = np.zeros(32, dtype=np.uint64)
have = hasattr(L, "mspack_hip_debug_pipe_phases")
if have:
    L.mspack_hip_debug_pipe_phases.argtypes = [ctypes.c_void_p]
import time
for it in range(3):
    if it == 2 and have:
        L.mspack_hip_debug_pipe_phases(ph.ctypes.data)
    t0 = time.perf_counter()
    out, res = M.decode_batch(units, arena, out_bytes)
    dt = time.perf_counter() - t0
    print("call %d: %.1f ms host wall (%.0f MB/s), err %d flags %#x" % (it, dt * 1e3, total / dt / 1e6, res["err"][0], res["flags"][0]))
assert res["err"][0] == 0 and res["out_len"][0] == total
if plain is not None:
    assert np.array_equal(out[:total], plain)
n = len(fo)
if have:
    L.mspack_hip_debug_pipe_phases(ph.ctypes.data)
    names = ["P wait prev header", "P header decode", "P record + publish", "P table builds", "P parse_emit total", "P final publish",
             "  emit: staging", "  emit: sync + count rounds", "  emit: last walk (values, literals, records)"]
    print("parse side, us per frame (%d frames):" % n)
    for k, nm in enumerate(names):
        print("  %-46s %8.1f" % (nm, ph[k] / 100.0 / n))
    print("resolve side, us per frame: front (load, R0-R2, checks) %.1f  push %.1f  resolve / run fill %.1f" %
          (ph[16 + 9] / 100.0 / n, ph[16 + 10] / 100.0 / n, ph[16 + 11] / 100.0 / n))
    T = min(2 * n + 2, 1 << 16)
    a = np.zeros(4 * T, dtype=np.uint64)
    L.mspack_hip_debug_pipe_trace.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    assert L.mspack_hip_debug_pipe_trace(a.ctypes.data, a.size) == 0
    a = a.reshape(T, 4)
    v = a[a[:, 1] != 0]
    t0 = v[:, 0].min()
    st = (v[:, 0] - t0) / 100.0; en = (v[:, 1] - t0) / 100.0
    fr = ((v[:, 2] & 0xFFFFFFFF) >> 1).astype(int)
    wait = (v[:, 3] & ((1 << 40) - 1)) / 100.0
    o = np.argsort(fr)
    st, en, fr, wait = st[o], en[o], fr[o], wait[o]
    print("tasks traced %d; launch span %.0f us = %.1f us per frame; task duration mean %.0f us (waited %.0f of it)" %
          (len(v), en.max(), en.max() / max(1, len(v)), (en - st).mean(), wait.mean()))
    d_end = np.diff(en)
    print("frame-to-frame spacing of task ENDS (the chain's rate): mean %.1f us, median %.1f, p90 %.1f" %
          (d_end.mean(), np.median(d_end), np.percentile(d_end, 90)))
    print("first 12 tasks (frame, start, end, waited):", [(int(f), round(float(s)), round(float(e)), round(float(w))) for f, s, e, w in zip(fr[:12], st[:12], en[:12], wait[:12])])
