"""ctypes mirror of csrc/bench/api_bench.c (BENCH / TEST INFRASTRUCTURE): a container image through
mspack_create_cab/chm_decompressor() -> open() -> extract() of every file, driven by a C in-memory mspack_system, with the
time split.  Loads libmspack_apibench.so, which links libmspack_hip.so: no GPU, no result (the drivers have no CPU path)."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libmspack_apibench.so")


class Stats(C.Structure):
    _fields_ = [("total_s", C.c_double), ("open_s", C.c_double), ("first_extract_s", C.c_double), ("read_s", C.c_double),
                ("write_s", C.c_double), ("lib_plan_ms", C.c_double), ("lib_issue_ms", C.c_double), ("lib_drain_ms", C.c_double),
                ("bytes_out", C.c_ulonglong), ("bytes_read", C.c_ulonglong), ("n_files", C.c_uint), ("n_errors", C.c_uint),
                ("n_messages", C.c_uint), ("lib_calls", C.c_uint), ("first_error", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO):
            from . import build as B
            B.build_all()
        L = C.CDLL(SO)
        for fn in (L.mspk_api_bench_cab, L.mspk_api_bench_chm):
            fn.restype = C.c_int
            fn.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint, C.POINTER(Stats)]
        L.mspk_api_cab_run.restype = C.c_int
        L.mspk_api_cab_run.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t]
        _lib = L
    return _lib


def cab_run(image, order, fix_mszip=0, salvage=0, cap=1 << 26, L=None):
    """the files `order` of a cabinet image extracted with one decompressor -> (rc, [(err, bytes)], message lines)"""
    L = L or lib()
    img = np.frombuffer(bytes(image), dtype=np.uint8)
    n = len(order)
    arr = np.array(order, dtype=np.int32)
    out = np.zeros(cap, dtype=np.uint8)
    offs = np.zeros(max(n, 1), dtype=np.uint64); lens = np.zeros(max(n, 1), dtype=np.uint64); errs = np.zeros(max(n, 1), dtype=np.int32)
    msgs = C.create_string_buffer(1 << 16)
    rc = L.mspk_api_cab_run(img.ctypes.data, img.size, arr.ctypes.data, n, fix_mszip, salvage, out.ctypes.data, cap,
                            offs.ctypes.data, lens.ctypes.data, errs.ctypes.data, msgs, len(msgs))
    res = [(int(errs[i]), out[int(offs[i]):int(offs[i]) + int(lens[i])].tobytes()) for i in range(n)] if rc == 0 else []
    return rc, res, msgs.value.decode("latin1").splitlines()


def run(kind, image, out_cap, max_files=65536):
    """kind 'cab' | 'chm'.  -> (rc, out uint8[bytes_out], offsets uint64[n_files + 1], stats dict)"""
    image = np.frombuffer(bytes(image), dtype=np.uint8) if not isinstance(image, np.ndarray) else np.ascontiguousarray(image, dtype=np.uint8)
    out = np.zeros(int(out_cap) + 64, dtype=np.uint8)
    offs = np.zeros(max_files + 1, dtype=np.uint64)
    st = Stats()
    fn = lib().mspk_api_bench_cab if kind == "cab" else lib().mspk_api_bench_chm
    rc = fn(image.ctypes.data, image.size, out.ctypes.data, int(out_cap), offs.ctypes.data, max_files + 1, C.byref(st))
    d = {k: getattr(st, k) for k, _t in Stats._fields_}
    return rc, out[:st.bytes_out], offs[:min(st.n_files, max_files) + 1], d


def summary(d, reps_note=""):
    """the bench line's through_api object from a stats dict"""
    tot = d["total_s"]
    other = tot - d["open_s"] - d["read_s"] - d["write_s"] - (d["lib_plan_ms"] + d["lib_issue_ms"] + d["lib_drain_ms"]) * 1e-3
    # (a driver whose batch runs as a job -- include/mspack_hip.h -- has the library's phases on a thread of the library's own, beside
    # the caller's sys->write: the phases then add up to MORE than the wall time, and the difference is what ran side by side)
    overlapped = max(0.0, -other)
    other = max(0.0, other)
    return {"MBps": round(d["bytes_out"] / tot / 1e6, 1), "seconds": round(tot, 4), "files": d["n_files"], "errors": d["n_errors"],
            "bytes_out": d["bytes_out"],
            "split_ms": {"open (headers, file list)": round(d["open_s"] * 1e3, 2),
                         "sys->read + seek (the drivers' gather)": round(d["read_s"] * 1e3, 2),
                         "library: plan + buffers": round(d["lib_plan_ms"], 2),
                         "library: H2D + launches": round(d["lib_issue_ms"], 2),
                         "library: kernels + D2H": round(d["lib_drain_ms"], 2),
                         "sys->write": round(d["write_s"] * 1e3, 2),
                         "drivers' own work (block checksums, arenas, slicing)": round(other * 1e3, 2),
                         "minus what ran side by side (a job's batch beside sys->write)": -round(overlapped * 1e3, 2)},
            "batch_calls": d["lib_calls"], "first_extract_ms": round(d["first_extract_s"] * 1e3, 2),
            "what": "mspack_create_*_decompressor() -> open() -> extract() of every file, C in-memory mspack_system "
                    "(libmspack_amd/csrc/bench/api_bench.c)" + reps_note}


# ---- BASELINE configs 2, 3 and 4 as CONTAINERS (synthetic; the writers are csrc/corpus/containers.c) ------------------
def build_config2_cab(M, n=4096, ub=32768, blobs=None, plain=None):
    """ONE cabinet of n folders, each one MSZIP CFDATA block of ub bytes and one file.  -> (image, plaintext)"""
    import zlib
    if plain is None:
        plain = M.gen_plaintext(0xC0FFEE, 0, n * ub)
    folders, files = [], []
    for i in range(n):
        if blobs is not None:
            blob = blobs[i]
        else:
            co = zlib.compressobj(6, zlib.DEFLATED, -15)
            blob = b"CK" + co.compress(plain[i * ub:(i + 1) * ub].tobytes()) + co.flush()
        folders.append((1, [blob], [ub]))
        files.append((b"f%05d.bin" % i, ub, 0, i))
    return M.cab_write(folders, files), plain


def build_config3_chm(M, n=1024, ub=65536, n_files=61, seed=0xBA5E11, threads=None):
    """ONE CHM whose section 1 is n LZX reset intervals of ub bytes (window 2^21, reset every ub / 32 KiB frames): the
    intervals are encoded independently (every one starts from reset state, as a reset interval does) and concatenated --
    the reset table lists every frame.  n_files files of unequal length tile the section.  -> (image, plaintext, [(off, len)])"""
    plain, comp, off, ln, tab = M.corpus_lzx_units(seed, 0, n, ub, 21, n_threads=threads, frame_tables=True)
    fpu = ub // 32768
    parts, frame_off, pos = [], [], 0
    for i in range(n):
        o, l = int(off[i]), int(ln[i])
        t = comp[int(tab[i]):int(tab[i]) + 4 * fpu].view(np.uint32)
        frame_off += [pos + int(x) for x in t]
        parts.append(comp[o:o + l])
        pos += l
    frame_off.append(pos)
    lzx = np.concatenate(parts)
    total = n * ub
    # file boundaries: unequal, not aligned to frames or intervals
    rng = np.random.default_rng(3)
    cuts = np.sort(rng.choice(np.arange(1, total), size=n_files - 1, replace=False)) if n_files > 1 else np.array([], dtype=np.int64)
    bounds = [0] + [int(c) for c in cuts] + [total]
    files = [(b"/doc%04d.html" % k, bounds[k], bounds[k + 1] - bounds[k]) for k in range(n_files)]
    image = M.chm_write(lzx, np.array(frame_off, dtype=np.uint64), total, 21, fpu, files)
    return image, plain, [(f[1], f[2]) for f in files]


def build_config4_cab(M, n=512, frames=32, window_bits=21, blobs_frames=None, plain=None):
    """ONE cabinet of n Quantum folders (comp_type 0x0002 | window << 8) of `frames` CFDATA blocks each, one file per folder.
    blobs_frames: per folder (folder stream incl. the 0xFF trailers, frame payload sizes) as M.qtm_encode returns them.
    -> (image, plaintext)"""
    from concurrent.futures import ThreadPoolExecutor
    ub = frames * 32768
    if plain is None:
        plain = M.gen_plaintext(0xC0FFEE, 0, n * ub)
    if blobs_frames is None:
        with ThreadPoolExecutor(max_workers=os.cpu_count() or 1) as ex:
            blobs_frames = list(ex.map(lambda i: M.qtm_encode(plain[i * ub:(i + 1) * ub], window_bits), range(n)))
    folders, files = [], []
    for i, (st, fs) in enumerate(blobs_frames):
        st = bytes(st)
        blocks, p = [], 0
        for k in range(len(fs)):
            blocks.append(st[p:p + int(fs[k])])           # the payload without the 0xFF trailer (cabd appends it: cabd.c:1416)
            p += int(fs[k]) + 1
        folders.append((2 | (window_bits << 8), blocks, [32768] * len(blocks)))
        files.append((b"q%04d.bin" % i, ub, 0, i))
    return M.cab_write(folders, files), plain
