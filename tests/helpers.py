"""Shared test helpers: ctypes bindings for the oracle (liboracle.so), the compiled reference
(oracle/_ref/librefharness.so, optional) and a tiny CAB folder gatherer used only by tests.

Nothing here is on the product path."""
import ctypes as C
import os
import struct
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
REF_DIR = "/root/reference"

ERR_OK, ERR_ARGS, ERR_OPEN, ERR_READ, ERR_WRITE, ERR_SEEK, ERR_NOMEMORY, ERR_SIGNATURE, \
    ERR_DATAFORMAT, ERR_CHECKSUM, ERR_CRUNCH, ERR_DECRUNCH = range(12)

F_E8_APPLIED, F_LOOKAHEAD_READ, F_INTEL_HEADER, F_BLOCK_OPEN = 1, 2, 4, 16


class OracleResult(C.Structure):
    _fields_ = [("err", C.c_int32), ("flags", C.c_uint32), ("out_len", C.c_uint64),
                ("in_used", C.c_uint64), ("in_next", C.c_uint64)]


def _build_oracle():
    so = os.path.join(ORACLE_DIR, "liboracle.so")
    srcs = [os.path.join(ORACLE_DIR, f) for f in
            ("lzx_oracle.c", "mszip_oracle.c", "qtm_oracle.c", "lzss_oracle.c", "cab_oracle.c", "oracle.h", "oracle_huff.h")]
    if (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


_oracle = None


def oracle():
    global _oracle
    if _oracle is None:
        lib = C.CDLL(_build_oracle())
        lib.oracle_lzx_decode.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_uint64,
                                          C.c_uint64, C.c_int, C.c_int, C.c_int32, C.POINTER(OracleResult)]
        lib.oracle_mszip_decode.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_uint64,
                                            C.c_int, C.POINTER(C.c_uint32), C.c_int, C.POINTER(C.c_int),
                                            C.POINTER(OracleResult)]
        lib.oracle_qtm_decode.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_uint64,
                                          C.c_int, C.POINTER(OracleResult)]
        lib.oracle_huff_accepts.argtypes = [C.c_char_p, C.c_int, C.c_int]
        lib.oracle_lzss_decode.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(OracleResult)]
        lib.oracle_kwaj_lzh_decode.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(OracleResult)]
        lib.oracle_lzxd_decode.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_uint64,
                                           C.c_uint64, C.c_int, C.c_int, C.c_int32, C.c_int, C.c_char_p, C.c_size_t,
                                           C.POINTER(OracleResult)]
        lib.oracle_cab_checksum.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32]
        lib.oracle_cab_checksum.restype = C.c_uint32
        _oracle = lib
    return _oracle


def oracle_cab_checksum(data, seed=0):
    """cabd_checksum (cabd.c:1462-1479) of `data` with seed `seed`"""
    return int(oracle().oracle_cab_checksum(bytes(data), len(data), seed))


def cab_blocks_with_checksums(cab):
    """[(stored checksum, the 4 header bytes cbData|cbUncomp, payload)] of every CFDATA block of a single cabinet"""
    files_off, = struct.unpack_from("<I", cab, 0x10)
    nfolders, _nfiles, flags = struct.unpack_from("<HHH", cab, 0x1A)
    p, fres, dres = 0x24, 0, 0
    if flags & 4:
        hres, fres, dres = struct.unpack_from("<HBB", cab, p)
        p += 4 + hres
    for bit in (1, 2):
        if flags & bit:
            for _ in range(2):
                p = cab.index(b"\0", p) + 1
    out = []
    for _ in range(nfolders):
        doff, nblocks, _ct = struct.unpack_from("<IHH", cab, p)
        p += 8 + fres
        q = doff
        for _b in range(nblocks):
            if q + 8 > len(cab):
                break
            csum, cb, _cu = struct.unpack_from("<IHH", cab, q)
            if q + 8 + dres + cb > len(cab):
                break
            out.append((csum, cab[q + 4:q + 8], cab[q + 8 + dres:q + 8 + dres + cb]))
            q += 8 + dres + cb
    return out


def oracle_lzx(data, out_bytes, window_bits, reset_frames=0, length=None, e8_base=0):
    """-> (err, bytes, OracleResult)"""
    if length is None:
        length = out_bytes
    buf = C.create_string_buffer(max(int(out_bytes), 1))
    res = OracleResult()
    oracle().oracle_lzx_decode(bytes(data), len(data), buf, int(out_bytes), int(out_bytes), int(length),
                               window_bits, reset_frames, e8_base, C.byref(res))
    return res.err, buf.raw[:min(res.out_len, out_bytes)], res


def oracle_lzx_open_resets(cap=256):
    """the reset points (frame indices) at which the LAST oracle_lzx call of this thread found a block still open
    (lzxd.c:423-431: where the reference warns) -> (count, [frames])"""
    fr = (C.c_uint32 * cap)()
    f = oracle().oracle_lzx_open_resets
    f.restype = C.c_uint32
    n = f(fr, cap)
    return n, list(fr[:min(n, cap)])


def lzx_set_bits(buf, byte_off, first_bit, n_bits, value):
    """rewrite n_bits of an LZX stream (bytearray / uint8 array), counted from byte_off in the order the decoder reads them:
    16-bit little-endian words, most significant bit first (readbits.h)"""
    for k in range(n_bits):
        b = first_bit + k
        pos = byte_off + 2 * (b // 16) + (1 if (b % 16) < 8 else 0)
        bit = 7 - (b % 8)
        if (value >> (n_bits - 1 - k)) & 1:
            buf[pos] |= 1 << bit
        else:
            buf[pos] &= ~(1 << bit) & 0xFF


def oracle_lzxd(data, out_bytes, window_bits, ref=b"", reset_frames=0, length=None, e8_base=0):
    """LZX DELTA through the oracle -> (err, bytes, OracleResult)"""
    if length is None:
        length = out_bytes
    buf = C.create_string_buffer(max(int(out_bytes), 1))
    res = OracleResult()
    oracle().oracle_lzxd_decode(bytes(data), len(data), buf, int(out_bytes), int(out_bytes), int(length),
                                window_bits, reset_frames, e8_base, 1, bytes(ref), len(ref), C.byref(res))
    return res.err, buf.raw[:min(res.out_len, out_bytes)], res


def oracle_lzss(data, mode, cap=None):
    cap = cap if cap is not None else len(data) * 9 + 64
    buf = C.create_string_buffer(max(cap, 1))
    res = OracleResult()
    oracle().oracle_lzss_decode(bytes(data), len(data), mode, buf, cap, C.byref(res))
    return res.err, buf.raw[:min(res.out_len, cap)], res


def oracle_kwaj_lzh(data, cap=None):
    cap = cap if cap is not None else len(data) * 40 + 4096
    buf = C.create_string_buffer(max(cap, 1))
    res = OracleResult()
    oracle().oracle_kwaj_lzh_decode(bytes(data), len(data), buf, cap, C.byref(res))
    return res.err, buf.raw[:min(res.out_len, cap)], res


def oracle_mszip(data, out_bytes, repair=0):
    buf = C.create_string_buffer(max(int(out_bytes), 1))
    res = OracleResult()
    lens = (C.c_uint32 * 70000)()
    nb = C.c_int(0)
    oracle().oracle_mszip_decode(bytes(data), len(data), buf, int(out_bytes), int(out_bytes), repair,
                                 lens, 70000, C.byref(nb), C.byref(res))
    return res.err, buf.raw[:min(res.out_len, out_bytes)], res, list(lens[:nb.value])


def oracle_qtm(data, out_bytes, window_bits):
    buf = C.create_string_buffer(max(int(out_bytes), 1))
    res = OracleResult()
    oracle().oracle_qtm_decode(bytes(data), len(data), buf, int(out_bytes), int(out_bytes), window_bits,
                               C.byref(res))
    return res.err, buf.raw[:min(res.out_len, out_bytes)], res


def oracle_qtm_marks(data, out_bytes, window_bits, marks):
    """one oracle decode of the whole request with marks (oracle_qtm_set_marks): -> (err, [what a request ending at each mark holds back])"""
    import numpy as np
    m = np.ascontiguousarray(np.asarray(marks, dtype=np.uint32))
    log = np.zeros(max(len(m), 1), dtype=np.uint32)
    L = oracle()
    L.oracle_qtm_set_marks.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    L.oracle_qtm_set_marks.restype = None
    L.oracle_qtm_set_marks(m.ctypes.data, len(m), log.ctypes.data)
    err, _o, res = oracle_qtm(data, out_bytes, window_bits)
    return err, log[:len(m)].tolist()


# ---- the compiled reference (only where oracle/_ref exists) ------------------------------------
_ref = None


def have_ref():
    return os.path.exists(os.path.join(ORACLE_DIR, "_ref", "librefharness.so"))


def ref():
    global _ref
    if _ref is None:
        lib = C.CDLL(os.path.join(ORACLE_DIR, "_ref", "librefharness.so"))
        sz = C.POINTER(C.c_size_t)
        lib.refh_lzx.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_longlong, C.c_int,
                                 C.c_int, C.c_longlong, sz]
        lib.refh_lzxd.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_longlong, C.c_int, C.c_int,
                                  C.c_longlong, C.c_char_p, C.c_size_t, sz]
        lib.refh_mszip.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_longlong, C.c_int, sz]
        lib.refh_qtm.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_longlong, C.c_int, sz]
        lib.refh_qtm_carry.argtypes = [C.c_char_p, C.c_size_t, C.c_longlong, C.c_int, C.POINTER(C.c_longlong)]
        lib.refh_cab_list.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.c_uint),
                                      C.POINTER(C.c_uint), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                      C.c_char_p, C.c_int]
        lib.refh_cab_extract.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.c_int, C.c_void_p,
                                         C.c_size_t, sz, sz, C.POINTER(C.c_int), C.c_int, C.c_int]
        lib.refh_chm_list.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.c_longlong),
                                      C.POINTER(C.c_longlong), C.POINTER(C.c_int), C.c_char_p, C.c_int]
        lib.refh_chm_extract.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.c_int, C.c_void_p,
                                         C.c_size_t, sz, sz, C.POINTER(C.c_int)]
        lib.refh_cabset.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_int, C.POINTER(C.c_int), C.c_int,
                                    C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(C.c_uint), C.POINTER(C.c_uint),
                                    C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_uint), C.c_char_p, C.c_int,
                                    C.c_void_p, C.c_size_t, sz, sz, C.POINTER(C.c_int)]
        lib.refh_cab_search.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_longlong),
                                        C.POINTER(C.c_int), C.c_char_p, C.c_int]
        lib.refh_chm_find.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                      C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
        lib.refh_oab.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, sz, C.c_int]
        lib.refh_szdd_kwaj.argtypes = [C.c_int, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, sz, C.POINTER(C.c_int),
                                       C.POINTER(C.c_longlong), C.c_char_p, C.POINTER(C.c_int)]
        lib.refh_bench.restype = C.c_double
        lib.refh_bench.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                   C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_ulonglong), C.POINTER(C.c_int)]
        lib.refh_messages.restype = C.c_size_t
        lib.refh_messages.argtypes = [C.c_char_p, C.c_size_t]
        _ref = lib
    return _ref


def ref_messages():
    """what the reference said through sys->message since the last call (formatted lines; ref_cab_extract puts a
    '#extract i' line in front of every extract call)"""
    buf = C.create_string_buffer(1 << 16)
    ref().refh_messages(buf, len(buf))
    return buf.value.decode("latin1").splitlines()


def ref_message_handles():
    """per line of ref_messages() since the last call: 'H' = the reference passed a file handle to sys->message, '-' = NULL, '#' = one of
    the harness's own '#extract i' marks.  Call it right after ref_messages()."""
    buf = C.create_string_buffer(1 << 12)
    ref().refh_message_handles(buf, len(buf))
    return buf.value.decode("latin1")


def ref_lzx(data, out_bytes, window_bits, reset_frames=0, length=None):
    if length is None:
        length = out_bytes
    buf = C.create_string_buffer(max(int(out_bytes), 1))
    w = C.c_size_t(0)
    err = ref().refh_lzx(bytes(data), len(data), buf, int(out_bytes), int(out_bytes), window_bits,
                         reset_frames, int(length), C.byref(w))
    return err, buf.raw[:min(w.value, out_bytes)], w.value


def ref_lzxd(data, out_bytes, window_bits, ref=b"", reset_frames=0, length=None):
    if length is None:
        length = out_bytes
    buf = C.create_string_buffer(max(int(out_bytes), 1))
    w = C.c_size_t(0)
    err = globals()["ref"]().refh_lzxd(bytes(data), len(data), buf, int(out_bytes), int(out_bytes), window_bits,
                                       reset_frames, int(length), bytes(ref), len(ref), C.byref(w))
    return err, buf.raw[:min(w.value, out_bytes)], w.value


def ref_mszip(data, out_bytes, repair=0):
    buf = C.create_string_buffer(max(int(out_bytes), 1))
    w = C.c_size_t(0)
    err = ref().refh_mszip(bytes(data), len(data), buf, int(out_bytes), int(out_bytes), repair, C.byref(w))
    return err, buf.raw[:min(w.value, out_bytes)], w.value


def ref_qtm(data, out_bytes, window_bits):
    buf = C.create_string_buffer(max(int(out_bytes), 1))
    w = C.c_size_t(0)
    err = ref().refh_qtm(bytes(data), len(data), buf, int(out_bytes), int(out_bytes), window_bits, C.byref(w))
    return err, buf.raw[:min(w.value, out_bytes)], w.value


def ref_qtm_carry(data, out_bytes, window_bits):
    """the real qtmd after ONE qtmd_decompress(out_bytes): (err, o_end - o_ptr) -- what it holds back for the next call (qtmd.c:268-276)"""
    c = C.c_longlong(-1)
    err = ref().refh_qtm_carry(bytes(data), len(data), int(out_bytes), window_bits, C.byref(c))
    return err, c.value


def ref_cab_list(cab):
    n = 4096
    lens = (C.c_uint * n)(); offs = (C.c_uint * n)(); cts = (C.c_int * n)(); fids = (C.c_int * n)()
    names = C.create_string_buffer(n * 64)
    k = ref().refh_cab_list(cab, len(cab), n, lens, offs, cts, fids, names, 64)
    if k < 0:
        return -k, []
    out = []
    for i in range(min(k, n)):
        nm = names.raw[i * 64:(i + 1) * 64].split(b"\0")[0]
        out.append(dict(name=nm, length=lens[i], offset=offs[i], comp_type=cts[i], folder=fids[i]))
    return 0, out


def ref_cab_extract(cab, order, cap=1 << 26, fix_mszip=0, salvage=0):
    """-> list of (err, bytes) for the files extracted in `order` with ONE decompressor."""
    n = len(order)
    arr = (C.c_int * n)(*order)
    offs = (C.c_size_t * n)(); lens = (C.c_size_t * n)(); errs = (C.c_int * n)()
    buf = C.create_string_buffer(cap)
    rc = ref().refh_cab_extract(cab, len(cab), arr, n, buf, cap, offs, lens, errs, fix_mszip, salvage)
    if rc:
        return rc, []
    return 0, [(errs[i], buf.raw[offs[i]:offs[i] + min(lens[i], cap - offs[i])]) for i in range(n)]


def ref_cabset(blobs, ops, list_cab, cap=1 << 24, maxfiles=256):
    """Reference: open the cabinets, run the join ops [(0 append | 1 prepend, a, b)], list cab[list_cab] and
    extract every file.  -> (n_files or -err, [op return codes], [file dicts incl. err and data])"""
    n = len(blobs)
    arr = (C.c_char_p * n)(*blobs)
    lens = (C.c_size_t * n)(*[len(b) for b in blobs])
    flat = [x for op in ops for x in op]
    opa = (C.c_int * max(len(flat), 1))(*flat)
    operr = (C.c_int * max(len(ops), 1))()
    ln = (C.c_uint * maxfiles)(); of = (C.c_uint * maxfiles)(); ct = (C.c_int * maxfiles)(); fid = (C.c_int * maxfiles)()
    fb = (C.c_uint * maxfiles)(); names = C.create_string_buffer(maxfiles * 64)
    buf = C.create_string_buffer(cap)
    offs = (C.c_size_t * maxfiles)(); outl = (C.c_size_t * maxfiles)(); errs = (C.c_int * maxfiles)()
    k = ref().refh_cabset(arr, lens, n, opa, len(ops), operr, list_cab, maxfiles, ln, of, ct, fid, fb, names, 64,
                          buf, cap, offs, outl, errs)
    files = []
    for i in range(max(min(k, maxfiles), 0)):
        files.append(dict(name=names.raw[i * 64:(i + 1) * 64].split(b"\0")[0], length=ln[i], offset=of[i],
                          comp_type=ct[i], folder=fid[i], folder_blocks=fb[i], err=errs[i],
                          data=buf.raw[offs[i]:offs[i] + min(outl[i], cap - offs[i])]))
    return k, [operr[i] for i in range(len(ops))], files


def ref_cab_search(blob, searchbuf=0, cap=64):
    offs = (C.c_longlong * cap)(); nf = (C.c_int * cap)(); names = C.create_string_buffer(cap * 64)
    k = ref().refh_cab_search(blob, len(blob), searchbuf, cap, offs, nf, names, 64)
    if k < 0:
        return k
    return [(offs[i], nf[i], names.raw[i * 64:(i + 1) * 64].split(b"\0")[0]) for i in range(min(k, cap))]


def ref_chm_list(chm):
    n = 16384
    lens = (C.c_longlong * n)(); offs = (C.c_longlong * n)(); secs = (C.c_int * n)()
    names = C.create_string_buffer(n * 128)
    k = ref().refh_chm_list(chm, len(chm), n, lens, offs, secs, names, 128)
    if k < 0:
        return -k, []
    out = []
    for i in range(min(k, n)):
        nm = names.raw[i * 128:(i + 1) * 128].split(b"\0")[0]
        out.append(dict(name=nm, length=lens[i], offset=offs[i], section=secs[i]))
    return 0, out


def ref_oab(blob, base=None, cap=1 << 26, decompbuf=0):
    """Reference msoab_decompressor -> (err, bytes written)"""
    buf = C.create_string_buffer(cap)
    w = C.c_size_t(0)
    err = ref().refh_oab(blob, len(blob), base, len(base) if base is not None else 0, buf, cap, C.byref(w), decompbuf)
    return err, buf.raw[:min(w.value, cap)]


def ref_szdd_kwaj(kind, blob, cap=1 << 24):
    """Reference SZDD (kind 0) / KWAJ (kind 1) open + extract
    -> dict(open_err, err, data, comp_type|format, length, filename|missing char)"""
    buf = C.create_string_buffer(cap)
    w = C.c_size_t(0); ct = C.c_int(0); ln = C.c_longlong(0); oe = C.c_int(0)
    fn = C.create_string_buffer(16)
    err = ref().refh_szdd_kwaj(kind, blob, len(blob), buf, cap, C.byref(w), C.byref(ct), C.byref(ln), fn, C.byref(oe))
    return dict(open_err=oe.value, err=err, data=buf.raw[:min(w.value, cap)], comp_type=ct.value, length=ln.value,
                filename=fn.value)


def ref_chm_find(chm, names):
    """Reference fast_open + fast_find for every name -> (open err, [(err, section or -1, offset, length)])"""
    n = len(names)
    blob = b"".join(nm + b"\0" for nm in names)
    errs = (C.c_int * n)(); secs = (C.c_int * n)(); offs = (C.c_longlong * n)(); lens = (C.c_longlong * n)()
    rc = ref().refh_chm_find(chm, len(chm), blob, n, errs, secs, offs, lens)
    if rc:
        return rc, []
    return 0, [(errs[i], secs[i], offs[i], lens[i]) for i in range(n)]


def ref_chm_extract(chm, order, cap=1 << 26):
    n = len(order)
    arr = (C.c_int * n)(*order)
    offs = (C.c_size_t * n)(); lens = (C.c_size_t * n)(); errs = (C.c_int * n)()
    buf = C.create_string_buffer(cap)
    rc = ref().refh_chm_extract(chm, len(chm), arr, n, buf, cap, offs, lens, errs)
    if rc:
        return rc, []
    return 0, [(errs[i], buf.raw[offs[i]:offs[i] + min(lens[i], cap - offs[i])]) for i in range(n)]


# ---- minimal CAB folder gatherer (tests only; format: cab.h:16-67, cabd.c:1362-1479) -------------
def cab_folders(cab):
    """Return [dict(comp_type, blocks=[(payload, cb_uncomp)], files=[(name, offset, length)])].
    Single-cabinet files only; no checksum validation."""
    if cab[:4] != b"MSCF":
        raise ValueError("not a cabinet")
    files_off, = struct.unpack_from("<I", cab, 0x10)
    nfolders, nfiles, flags = struct.unpack_from("<HHH", cab, 0x1A)
    p = 0x24
    hres = fres = dres = 0
    if flags & 4:
        hres, fres, dres = struct.unpack_from("<HBB", cab, p)
        p += 4 + hres
    for bit in (1, 2):
        if flags & bit:
            for _ in range(2):
                p = cab.index(b"\0", p) + 1
    folders = []
    for _ in range(nfolders):
        doff, nblocks, ctype = struct.unpack_from("<IHH", cab, p)
        p += 8 + fres
        blocks = []
        q = doff
        for _b in range(nblocks):
            if q + 8 > len(cab):
                break
            _csum, cb, cu = struct.unpack_from("<IHH", cab, q)
            q += 8 + dres
            blocks.append((cab[q:q + cb], cu))
            q += cb
        folders.append(dict(comp_type=ctype, blocks=blocks, files=[]))
    p = files_off
    for _ in range(nfiles):
        size, foff, fidx = struct.unpack_from("<IIH", cab, p)
        e = cab.index(b"\0", p + 16)
        if fidx < len(folders):
            folders[fidx]["files"].append((cab[p + 16:e], foff, size))
        p = e + 1
    return folders


def folder_stream(folder):
    """Concatenate a folder's CFDATA payloads the way cabd_sys_read feeds the codec
    (Quantum: 0xFF appended after every block, cabd.c:1330-1332)."""
    qtm = (folder["comp_type"] & 0x0F) == 2
    return b"".join(p + (b"\xff" if qtm else b"") for p, _ in folder["blocks"])


def cab_cut_folders(cab, n_blocks):
    """A copy of a single cabinet whose folders hold only their first n_blocks CFDATA blocks: CFFOLDER.cCFData and the
    files' cbFile / uoffFolderStart clipped (cab.h offsets).  Files that start beyond the cut keep a length of 0."""
    import struct
    b = bytearray(cab)
    assert b[:4] == b"MSCF"
    coff_files, = struct.unpack_from("<I", b, 16)
    n_folders, n_files, flags = struct.unpack_from("<HHH", b, 26)
    pos = 36
    folder_resv = 0
    if flags & 4:
        hdr_resv, folder_resv, _data_resv = struct.unpack_from("<HBB", b, 36)
        pos = 40 + hdr_resv
    for _ in range(2):                       # szCabinetPrev/szDiskPrev, szCabinetNext/szDiskNext
        if flags & (1 if _ == 0 else 2):
            for _s in range(2):
                pos = b.index(0, pos) + 1
    limit = n_blocks * 32768
    for i in range(n_folders):
        o = pos + i * (8 + folder_resv)
        nb, = struct.unpack_from("<H", b, o + 4)
        struct.pack_into("<H", b, o + 4, min(nb, n_blocks))
    o = coff_files
    for i in range(n_files):
        length, start = struct.unpack_from("<II", b, o)
        end = min(start + length, limit)
        struct.pack_into("<I", b, o, max(end - min(start, limit), 0))
        o += 16
        o = b.index(0, o) + 1
    return bytes(b)
