"""libmspack_amd -- MI355X-native batched LZX / Quantum / MSZIP decompression behind libmspack's API.

This Python package is only a thin ctypes mirror of the C ABI in include/mspack_hip.h (and of the
corpus generators used by tests and bench.py).  The product is the shared library
libmspack_hip.so (hand-written HIP kernels + C host drivers); there is no CPU fallback: if the
library is missing, or a GPU call fails, errors are raised.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
HIP_SO = os.environ.get("MSPACK_HIP_SO", os.path.join(HERE, "libmspack_hip.so"))   # env override: kernel experiments only
CORPUS_SO = os.path.join(HERE, "libmspack_corpus.so")

KIND_MSZIP, KIND_QUANTUM, KIND_LZX, KIND_LZX_DELTA, KIND_LZSS, KIND_KWAJ_LZH = 1, 2, 3, 4, 5, 6
F_E8_APPLIED, F_LOOKAHEAD_READ, F_INTEL_HEADER, F_BLOCK_OPEN, F_FRAMES_ADOPTED = 1, 2, 4, 16, 32
UF_MSZIP_REPAIR = 1
ERR_OK, ERR_ARGS, ERR_OPEN, ERR_READ, ERR_WRITE, ERR_SEEK, ERR_NOMEMORY, ERR_SIGNATURE, \
    ERR_DATAFORMAT, ERR_CHECKSUM, ERR_CRUNCH, ERR_DECRUNCH = range(12)

# struct mspack_hip_unit / mspack_hip_result (include/mspack_hip.h)
UNIT_DTYPE = np.dtype([("in_off", "<u8"), ("out_off", "<u8"), ("in_len", "<u4"), ("out_len", "<u4"),
                       ("frame_base", "<u4"), ("e8_base", "<i4"), ("kind", "u1"), ("window_bits", "u1"),
                       ("reset_frames", "<u2"), ("flags", "<u4"), ("ref_len", "<u4"), ("in_chunk", "<u4")],
                      align=False)
RESULT_DTYPE = np.dtype([("err", "<i4"), ("flags", "<u4"), ("out_len", "<u4"), ("in_used", "<u4"),
                         ("good_len", "<u4"), ("in_next", "<u4")])
assert UNIT_DTYPE.itemsize == 48 and RESULT_DTYPE.itemsize == 24


class MspackHipError(RuntimeError):
    pass


_lib = None
_corpus = None


def lib():
    """Load libmspack_hip.so; fails loudly when it has not been built (no fallback path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(HIP_SO):
            raise MspackHipError("%s not built: run `python -c 'import __graft_entry__ as g; g.build()'`" % HIP_SO)
        L = C.CDLL(HIP_SO)
        vp, sz = C.c_void_p, C.c_size_t
        L.mspack_hip_version.restype = C.c_char_p
        L.mspack_hip_last_error.restype = C.c_char_p
        L.mspack_hip_device_count.restype = C.c_int
        L.mspack_hip_set_device.argtypes = [C.c_int]
        L.mspack_hip_frame_scratch_bytes.restype = sz
        L.mspack_hip_frame_scratch_bytes.argtypes = [sz]
        L.mspack_hip_decode_batch_device.argtypes = [vp, vp, sz, vp, sz, vp, sz, vp, vp, sz, C.c_uint, vp]
        L.mspack_hip_time_batch_device.restype = C.c_double
        L.mspack_hip_time_batch_device.argtypes = [vp, vp, sz, vp, sz, vp, sz, vp, vp, sz, C.c_uint, vp, C.c_int]
        L.mspack_hip_decode_batch.argtypes = [vp, sz, vp, sz, vp, sz, vp]
        L.mspack_hip_decode_batch_multi.argtypes = [vp, sz, vp, sz, vp, sz, vp, C.c_int]
        L.mspack_hip_decode_batch_to_device.argtypes = [vp, sz, vp, sz, vp, sz, vp]
        L.mspack_hip_decode_batch_begin.restype = vp
        L.mspack_hip_decode_batch_begin.argtypes = [vp, sz, vp, sz, vp, sz, vp]
        L.mspack_hip_job_wait_unit.argtypes = [vp, sz]
        L.mspack_hip_job_end.argtypes = [vp]
        L.mspack_hip_release.restype = None
        _lib = L
    return _lib


EXPORTED_SYMBOLS = [
    "mspack_hip_device_count", "mspack_hip_set_device", "mspack_hip_version", "mspack_hip_last_error",
    "mspack_hip_decode_batch_device", "mspack_hip_frame_scratch_bytes", "mspack_hip_decode_batch",
    "mspack_hip_decode_batch_multi", "mspack_hip_time_batch_device",
    "mspack_hip_decode_batch_to_device", "mspack_hip_release",
    "mspack_hip_set_default_devices", "mspack_hip_default_devices", "mspack_hip_set_cache_mb", "mspack_hip_cache_mb",
    "mspack_hip_host_path_stats", "mspack_hip_pin", "mspack_hip_unpin", "mspack_hip_stage_alloc", "mspack_hip_stage_free",
    "mspack_hip_decode_batch_begin", "mspack_hip_job_wait_unit", "mspack_hip_job_end",
]


def _check(rc, what):
    if rc != 0:
        raise MspackHipError("%s failed (%d): %s" % (what, rc, lib().mspack_hip_last_error().decode()))


def frames_of(units):
    """per-unit slots in the work scratch: LZX one per frame + one spare for the look-ahead frame; MSZIP units that
    carry a frame table (and are not in repair / KWAJ mode) one per CFDATA block"""
    lzx = (units["kind"] == KIND_LZX) | (units["kind"] == KIND_LZX_DELTA)
    zipt = (units["kind"] == KIND_MSZIP) & ((units["flags"] & UF_FRAME_TABLE) != 0) & ((units["flags"] & 5) == 0)
    return np.where(lzx, units["out_len"] // 32768 + 1, np.where(zipt, (units["out_len"].astype(np.int64) + 32767) // 32768, 0)).astype(np.int64)


UF_FRAME_TABLE = 8
UF_QTM_MARKS = 64           # Quantum: what requests ending at marked positions hold back (include/mspack_hip.h)
QTM_MARK_FAILS = 0xFFFFFFFF


def make_units(kind, in_offs, in_lens, out_lens, window_bits=0, reset_frames=0, e8_base=0, flags=0,
               out_slack=0, ref_lens=0, frame_tabs=None):
    """Build a unit table; output regions are laid out back to back (16-byte aligned, plus
    `out_slack` bytes each: MSZIP units need 32768 bytes of slack after out_len).  LZX DELTA units
    get `ref_lens` bytes of room for their reference data right below their output."""
    n = len(in_offs)
    u = np.zeros(n, dtype=UNIT_DTYPE)
    u["in_off"] = in_offs
    u["in_len"] = in_lens
    u["out_len"] = out_lens
    u["kind"] = kind
    u["window_bits"] = window_bits
    u["reset_frames"] = reset_frames
    u["e8_base"] = e8_base
    u["flags"] = flags
    u["ref_len"] = ref_lens
    if frame_tabs is not None:               # LZX: arena offsets of the units' frame tables (MSPACK_HIP_UF_FRAME_TABLE)
        u["flags"] |= UF_FRAME_TABLE
        u["in_chunk"] = np.asarray(frame_tabs, dtype=np.uint64) // 4
    rl = (u["ref_len"].astype(np.int64) + 15) & ~15
    rl = np.where((u["kind"] == KIND_LZSS) | (u["kind"] == KIND_KWAJ_LZH), 4096, rl)   # window pre-fill room
    sizes = ((np.asarray(out_lens, dtype=np.int64) + out_slack + 15) & ~15) + rl
    offs = np.zeros(n, dtype=np.int64)
    if n:
        offs[1:] = np.cumsum(sizes)[:-1]
    u["out_off"] = offs + rl
    fr = frames_of(u)
    fb = np.zeros(n, dtype=np.int64)
    if n:
        fb[1:] = np.cumsum(fr)[:-1]
    u["frame_base"] = fb
    return u, int(sizes.sum())


def decode_batch(units, in_arena, out_bytes, n_devices=1, refs=None):
    """Host-buffer batch decode through mspack_hip_decode_batch[_multi].
    refs: per-unit reference data (bytes) of LZX DELTA units, placed right below each unit's output.
    -> (out uint8 array, results structured array)"""
    units = np.ascontiguousarray(units, dtype=UNIT_DTYPE)
    in_arena = np.ascontiguousarray(in_arena, dtype=np.uint8)
    out = np.zeros(max(int(out_bytes), 1), dtype=np.uint8)
    if refs is not None:
        for u, r in zip(units, refs):
            if len(r):
                o = int(u["out_off"])
                out[o - len(r):o] = np.frombuffer(bytes(r), dtype=np.uint8)
    res = np.zeros(len(units), dtype=RESULT_DTYPE)
    L = lib()
    if n_devices > 1:
        rc = L.mspack_hip_decode_batch_multi(units.ctypes.data, len(units), in_arena.ctypes.data, in_arena.size,
                                             out.ctypes.data, out.size, res.ctypes.data, n_devices)
    elif os.environ.get("MSPACK_PY_VIA_JOBS") and len(units):
        # (parity sweeps of the job entry points: the same batch through _begin / _wait_unit / _end, a third of the units waited for
        # one by one -- their results are looked at the moment the wait returns)
        job = L.mspack_hip_decode_batch_begin(units.ctypes.data, len(units), in_arena.ctypes.data, in_arena.size,
                                              out.ctypes.data, out.size, res.ctypes.data)
        if not job:
            raise MspackHipError("mspack_hip_decode_batch_begin returned no job")
        res["err"] = 0x7777
        for i in np.argsort(units["in_off"], kind="stable")[::3]:
            if L.mspack_hip_job_wait_unit(job, int(i)) == 0 and int(res["err"][int(i)]) == 0x7777:
                L.mspack_hip_job_end(job)
                raise MspackHipError("unit %d: the wait returned before the unit's result was written" % int(i))
        rc = L.mspack_hip_job_end(job)
    else:
        rc = L.mspack_hip_decode_batch(units.ctypes.data, len(units), in_arena.ctypes.data, in_arena.size,
                                       out.ctypes.data, out.size, res.ctypes.data)
    _check(rc, "mspack_hip_decode_batch")
    return out, res


# ---- corpus generators (test / bench infrastructure) ---------------------------------------------
class LzxOpts(C.Structure):
    _fields_ = [("block_mode", C.c_int), ("block_size", C.c_int), ("chain_depth", C.c_int),
                ("use_repeats", C.c_int), ("lazy", C.c_int), ("intel_filesize", C.c_int32),
                ("e8_base", C.c_int32), ("delta", C.c_int), ("ref", C.c_void_p), ("ref_len", C.c_size_t)]


def corpus():
    global _corpus
    if _corpus is None:
        if not os.path.exists(CORPUS_SO):
            raise MspackHipError("%s not built" % CORPUS_SO)
        L = C.CDLL(CORPUS_SO)
        vp, sz = C.c_void_p, C.c_size_t
        L.mspk_gen_plaintext.argtypes = [C.c_uint64, C.c_int, vp, sz]
        L.mspk_lzx_encode.restype = sz
        L.mspk_lzx_encode.argtypes = [vp, sz, C.c_int, C.c_int, C.POINTER(LzxOpts), vp, sz, vp]
        L.mspk_lzx_bound.restype = sz
        L.mspk_lzx_bound.argtypes = [sz]
        L.mspk_corpus_lzx_units.restype = sz
        L.mspk_corpus_lzx_units.argtypes = [C.c_uint64, C.c_int, C.c_int, sz, C.c_int, C.POINTER(LzxOpts),
                                            C.c_int, vp, vp, sz, vp, vp]
        L.mspk_corpus_lzx_units_at.restype = sz
        L.mspk_corpus_lzx_units_at.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_int, sz, C.c_int, C.POINTER(LzxOpts),
                                               C.c_int, vp, vp, sz, vp, vp]
        L.mspk_corpus_lzx_units_ft.restype = sz
        L.mspk_corpus_lzx_units_ft.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_int, sz, C.c_int, C.POINTER(LzxOpts),
                                               C.c_int, vp, vp, sz, vp, vp, vp]
        L.mspk_cab_write.restype = sz
        L.mspk_cab_write.argtypes = [vp, C.c_int, vp, C.c_int, vp, sz]
        L.mspk_chm_write.restype = sz
        L.mspk_chm_write.argtypes = [vp, sz, vp, sz, C.c_uint64, C.c_int, C.c_int, vp, C.c_int, vp, sz]
        L.mspk_chm_bound.restype = sz
        L.mspk_chm_bound.argtypes = [sz, sz, C.c_int]
        L.mspk_qtm_encode.restype = sz
        L.mspk_qtm_encode.argtypes = [vp, sz, C.c_int, C.c_int, vp, sz, vp]
        L.mspk_qtm_bound.restype = sz
        L.mspk_qtm_bound.argtypes = [sz]
        _corpus = L
    return _corpus


TEXT_MIX, TEXT_ENGLISH, TEXT_BINARY, TEXT_RECORDS, TEXT_RANDOM, TEXT_REPETITIVE = range(6)


def gen_plaintext(seed, kind, n):
    buf = np.empty(n, dtype=np.uint8)
    corpus().mspk_gen_plaintext(seed, kind, buf.ctypes.data, n)
    return buf


def lzx_opts(mode=0, block_size=0, depth=0, repeats=1, lazy=1, intel_filesize=0, e8_base=0):
    return LzxOpts(mode, block_size, depth, repeats, lazy, intel_filesize, e8_base, 0, None, 0)


def lzxd_encode(data, window_bits, ref=b"", **kw):
    """LZX DELTA stream (one lzxd_init'ed block as OAB files hold them) -> compressed bytes (np.uint8)"""
    o = lzx_opts(**kw)
    refbuf = np.frombuffer(bytes(ref), dtype=np.uint8) if len(ref) else np.zeros(1, dtype=np.uint8)
    o.delta = 1
    o.ref = refbuf.ctypes.data if len(ref) else None
    o.ref_len = len(ref)
    comp, _fo = lzx_encode(data, window_bits, 0, o)
    return comp


def lzx_encode(data, window_bits, reset_frames, opts=None):
    """-> (compressed bytes (np.uint8), frame_off (np.uint64, n_frames+1))"""
    data = np.ascontiguousarray(data, dtype=np.uint8)
    n = data.size
    L = corpus()
    cap = L.mspk_lzx_bound(n)
    dst = np.empty(cap, dtype=np.uint8)
    nfr = (n + 32767) // 32768
    fo = np.zeros(nfr + 1, dtype=np.uint64)
    o = opts if opts is not None else lzx_opts()
    m = L.mspk_lzx_encode(data.ctypes.data, n, window_bits, reset_frames, C.byref(o), dst.ctypes.data, cap,
                          fo.ctypes.data)
    if m == 0:
        raise MspackHipError("mspk_lzx_encode overflow")
    return dst[:m].copy(), fo


def corpus_lzx_units(base_seed, kind, n_units, unit_bytes, window_bits, opts=None, n_threads=None, first_unit=0,
                     frame_tables=False):
    """Batch of independent LZX units (one reset interval each); first_unit: global index of the first
    one when the call makes a shard of a larger list (unit seeds follow the global index).
    -> (plain [n_units*unit_bytes], comp arena, comp_off u64[n], comp_len u32[n])
    frame_tables=True: every unit's frame table (where each 32 KiB frame starts in the compressed stream -- what a
    CHM reset table or a cabinet's CFDATA sizes state) is written into the arena behind the unit, and a fifth
    value tab_off u64[n] (arena offsets) is returned: make_units(..., frame_tabs=tab_off)."""
    L = corpus()
    if n_threads is None:
        n_threads = os.cpu_count() or 1
    plain = np.empty(n_units * unit_bytes, dtype=np.uint8)
    nfr = (unit_bytes + 32767) // 32768
    cap = n_units * (L.mspk_lzx_bound(unit_bytes) + 24 + 4 * nfr)
    comp = np.zeros(cap, dtype=np.uint8)
    off = np.zeros(n_units, dtype=np.uint64)
    ln = np.zeros(n_units, dtype=np.uint32)
    tab = np.zeros(n_units, dtype=np.uint64)
    o = opts if opts is not None else lzx_opts()
    total = L.mspk_corpus_lzx_units_ft(base_seed, first_unit, kind, n_units, unit_bytes, window_bits, C.byref(o),
                                       n_threads, plain.ctypes.data, comp.ctypes.data, cap, off.ctypes.data,
                                       ln.ctypes.data, tab.ctypes.data if frame_tables else None)
    if total == 0:
        raise MspackHipError("mspk_corpus_lzx_units failed")
    if frame_tables:
        return plain, comp[:total + 64].copy(), off, ln, tab
    return plain, comp[:total + 64].copy(), off, ln


def qtm_encode(data, window_bits, chain_depth=0):
    """-> (folder stream as cabd feeds it to qtmd: every frame followed by the 0xFF trailer,
           frame payload sizes u32[n_frames])"""
    data = np.ascontiguousarray(data, dtype=np.uint8)
    n = data.size
    L = corpus()
    cap = L.mspk_qtm_bound(n)
    dst = np.empty(cap, dtype=np.uint8)
    nfr = (n + 32767) // 32768
    fs = np.zeros(max(nfr, 1), dtype=np.uint32)
    m = L.mspk_qtm_encode(data.ctypes.data, n, window_bits, chain_depth, dst.ctypes.data, cap, fs.ctypes.data)
    if m == 0 and n:
        raise MspackHipError("mspk_qtm_encode failed")
    parts, pos = [], 0
    for k in range(nfr):
        parts.append(dst[pos:pos + int(fs[k])].tobytes() + b"\xff")
        pos += int(fs[k])
    return b"".join(parts), fs[:nfr]


class _CabFolder(C.Structure):
    _fields_ = [("comp_type", C.c_int), ("data", C.c_void_p), ("block_comp", C.c_void_p),
                ("block_uncomp", C.c_void_p), ("n_blocks", C.c_int)]


class _CabFile(C.Structure):
    _fields_ = [("name", C.c_char_p), ("length", C.c_uint32), ("folder_offset", C.c_uint32),
                ("folder_index", C.c_uint16)]


class _ChmFile(C.Structure):
    _fields_ = [("name", C.c_char_p), ("offset", C.c_uint64), ("length", C.c_uint64)]


def cab_write(folders, files):
    """folders: list of (comp_type, [block payload bytes], [block uncompressed sizes]);
    files: list of (name bytes, length, folder_offset, folder_index) -> cabinet bytes"""
    keep = []
    fa = (_CabFolder * len(folders))()
    for i, (ct, blocks, usz) in enumerate(folders):
        data = np.frombuffer(b"".join(blocks), dtype=np.uint8).copy() if blocks else np.zeros(1, np.uint8)
        bc = np.array([len(b) for b in blocks], dtype=np.uint32)
        bu = np.array(usz, dtype=np.uint32)
        keep += [data, bc, bu]
        fa[i] = _CabFolder(ct, data.ctypes.data, bc.ctypes.data, bu.ctypes.data, len(blocks))
    fl = (_CabFile * len(files))()
    for i, (name, length, off, fidx) in enumerate(files):
        fl[i] = _CabFile(name, length, off, fidx)
    cap = sum(len(b) + 8 for _ct, bl, _u in folders for b in bl) + 64 * (len(files) + 4) + 4096
    dst = np.zeros(cap, dtype=np.uint8)
    n = corpus().mspk_cab_write(fa, len(folders), fl, len(files), dst.ctypes.data, cap)
    if n == 0:
        raise MspackHipError("mspk_cab_write failed")
    return dst[:n].tobytes()


def chm_write(lzx, frame_off, uncomp_len, window_bits, reset_frames, files):
    """files: list of (name bytes starting with '/', offset, length) in the uncompressed stream"""
    lzx = np.ascontiguousarray(lzx, dtype=np.uint8)
    fo = np.ascontiguousarray(frame_off, dtype=np.uint64)
    n_frames = len(fo) - 1
    fl = (_ChmFile * len(files))()
    for i, (name, off, ln) in enumerate(files):
        fl[i] = _ChmFile(name, off, ln)
    L = corpus()
    cap = L.mspk_chm_bound(lzx.size, n_frames, len(files))
    dst = np.zeros(cap, dtype=np.uint8)
    n = L.mspk_chm_write(lzx.ctypes.data, lzx.size, fo.ctypes.data, n_frames, uncomp_len, window_bits,
                         reset_frames, fl, len(files), dst.ctypes.data, cap)
    if n == 0:
        raise MspackHipError("mspk_chm_write failed")
    return dst[:n].tobytes()
