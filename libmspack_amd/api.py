"""ctypes mirror of include/mspack.h (the libmspack-compatible object API served by libmspack_hip.so).

Names and argument meaning follow the reference API (mspack_create_cab_decompressor -> open /
extract / close, mspack_create_chm_decompressor -> open / fast_open / fast_find / extract / close),
so tests read like libmspack/test/cabd_test.c.  File I/O goes through the library's default
stdio mspack_system (sys = NULL), or -- mem=True -- through an in-memory mspack_system handed to
mspack_create_*_decompressor(sys) (MemSystem below: the same semantics as the memory system the goldens
were recorded with, oracle/ref_harness.c: reads are clipped at the end, a seek beyond the end fails)."""
import ctypes as C
import os
import tempfile

from . import lib

off_t = C.c_int64


class MscabdFolder(C.Structure):
    pass


class MscabdFile(C.Structure):
    pass


class MscabdCabinet(C.Structure):
    pass


MscabdFolder._fields_ = [("next", C.POINTER(MscabdFolder)), ("comp_type", C.c_int), ("num_blocks", C.c_uint)]
MscabdFile._fields_ = [("next", C.POINTER(MscabdFile)), ("filename", C.c_char_p), ("length", C.c_uint),
                       ("attribs", C.c_int), ("time_h", C.c_char), ("time_m", C.c_char), ("time_s", C.c_char),
                       ("date_d", C.c_char), ("date_m", C.c_char), ("date_y", C.c_int),
                       ("folder", C.POINTER(MscabdFolder)), ("offset", C.c_uint)]
MscabdCabinet._fields_ = [("next", C.POINTER(MscabdCabinet)), ("filename", C.c_char_p), ("base_offset", off_t),
                          ("length", C.c_uint), ("prevcab", C.POINTER(MscabdCabinet)),
                          ("nextcab", C.POINTER(MscabdCabinet)), ("prevname", C.c_char_p), ("nextname", C.c_char_p),
                          ("previnfo", C.c_char_p), ("nextinfo", C.c_char_p), ("files", C.POINTER(MscabdFile)),
                          ("folders", C.POINTER(MscabdFolder)), ("set_id", C.c_ushort), ("set_index", C.c_ushort),
                          ("header_resv", C.c_ushort), ("flags", C.c_int)]


class MscabDecompressor(C.Structure):
    pass


_P = C.POINTER
MscabDecompressor._fields_ = [
    ("open", C.CFUNCTYPE(_P(MscabdCabinet), _P(MscabDecompressor), C.c_char_p)),
    ("close", C.CFUNCTYPE(None, _P(MscabDecompressor), _P(MscabdCabinet))),
    ("search", C.CFUNCTYPE(_P(MscabdCabinet), _P(MscabDecompressor), C.c_char_p)),
    ("append", C.CFUNCTYPE(C.c_int, _P(MscabDecompressor), _P(MscabdCabinet), _P(MscabdCabinet))),
    ("prepend", C.CFUNCTYPE(C.c_int, _P(MscabDecompressor), _P(MscabdCabinet), _P(MscabdCabinet))),
    ("extract", C.CFUNCTYPE(C.c_int, _P(MscabDecompressor), _P(MscabdFile), C.c_char_p)),
    ("set_param", C.CFUNCTYPE(C.c_int, _P(MscabDecompressor), C.c_int, C.c_int)),
    ("last_error", C.CFUNCTYPE(C.c_int, _P(MscabDecompressor))),
]


class MschmdHeader(C.Structure):
    pass


class MschmdFile(C.Structure):
    pass


class MschmdSection(C.Structure):
    _fields_ = [("chm", _P(MschmdHeader)), ("id", C.c_uint)]


class MschmdSecUncompressed(C.Structure):
    _fields_ = [("base", MschmdSection), ("offset", off_t)]


class MschmdSecMscompressed(C.Structure):
    _fields_ = [("base", MschmdSection), ("content", _P(MschmdFile)), ("control", _P(MschmdFile)),
                ("rtable", _P(MschmdFile)), ("spaninfo", _P(MschmdFile))]


MschmdFile._fields_ = [("next", _P(MschmdFile)), ("section", _P(MschmdSection)), ("offset", off_t),
                       ("length", off_t), ("filename", C.c_char_p)]
MschmdHeader._fields_ = [("version", C.c_uint), ("timestamp", C.c_uint), ("language", C.c_uint),
                         ("filename", C.c_char_p), ("length", off_t), ("files", _P(MschmdFile)),
                         ("sysfiles", _P(MschmdFile)), ("sec0", MschmdSecUncompressed),
                         ("sec1", MschmdSecMscompressed), ("dir_offset", off_t), ("num_chunks", C.c_uint),
                         ("chunk_size", C.c_uint), ("density", C.c_uint), ("depth", C.c_uint),
                         ("index_root", C.c_uint), ("first_pmgl", C.c_uint), ("last_pmgl", C.c_uint),
                         ("chunk_cache", C.c_void_p)]


class MschmDecompressor(C.Structure):
    pass


MschmDecompressor._fields_ = [
    ("open", C.CFUNCTYPE(_P(MschmdHeader), _P(MschmDecompressor), C.c_char_p)),
    ("close", C.CFUNCTYPE(None, _P(MschmDecompressor), _P(MschmdHeader))),
    ("extract", C.CFUNCTYPE(C.c_int, _P(MschmDecompressor), _P(MschmdFile), C.c_char_p)),
    ("last_error", C.CFUNCTYPE(C.c_int, _P(MschmDecompressor))),
    ("fast_open", C.CFUNCTYPE(_P(MschmdHeader), _P(MschmDecompressor), C.c_char_p)),
    ("fast_find", C.CFUNCTYPE(C.c_int, _P(MschmDecompressor), _P(MschmdHeader), C.c_char_p, _P(MschmdFile), C.c_int)),
]

MSCABD_PARAM_SEARCHBUF, MSCABD_PARAM_FIXMSZIP, MSCABD_PARAM_DECOMPBUF, MSCABD_PARAM_SALVAGE = 0, 1, 2, 3
MSCABD_PARAM_HIP_DEVICES, MSCABD_PARAM_HIP_CACHE_MB = 100, 101


def _setup(L=None):
    """L: an already loaded library exporting the mspack.h API (tests of the host logic pass their CPU
    stand-in build here); default = libmspack_hip.so"""
    if L is None:
        L = lib()
    L.mspack_create_cab_decompressor.restype = _P(MscabDecompressor)
    L.mspack_create_cab_decompressor.argtypes = [C.c_void_p]
    L.mspack_destroy_cab_decompressor.argtypes = [_P(MscabDecompressor)]
    L.mspack_create_chm_decompressor.restype = _P(MschmDecompressor)
    L.mspack_create_chm_decompressor.argtypes = [C.c_void_p]
    L.mspack_destroy_chm_decompressor.argtypes = [_P(MschmDecompressor)]
    L.mspack_version.argtypes = [C.c_int]
    L.mspack_sys_selftest_internal.argtypes = [C.c_int]
    return L


class MspackSystem(C.Structure):
    """struct mspack_system (mspack.h:285-455): 10 function pointers + a NULL"""
    _fields_ = [("open", C.c_void_p), ("close", C.c_void_p), ("read", C.c_void_p), ("write", C.c_void_p),
                ("seek", C.c_void_p), ("tell", C.c_void_p), ("message", C.c_void_p), ("alloc", C.c_void_p),
                ("free", C.c_void_p), ("copy", C.c_void_p), ("null_ptr", C.c_void_p)]


class MemSystem:
    """An in-memory mspack_system implemented with ctypes callbacks.  "File names" are keys of
    self.files (bytes -> bytes for inputs); files opened for writing collect into self.outputs.
    alloc / free / copy are the library's own defaults; messages are collected in self.messages."""

    def __init__(self, L):
        self.files = {}
        self.outputs = {}
        self._open = {}
        self._next = 1
        dflt = C.POINTER(MspackSystem).in_dll(L, "mspack_default_system").contents
        OPEN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_char_p, C.c_int)
        CLOSE = C.CFUNCTYPE(None, C.c_void_p)
        RW = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int)
        SEEK = C.CFUNCTYPE(C.c_int, C.c_void_p, off_t, C.c_int)
        TELL = C.CFUNCTYPE(off_t, C.c_void_p)

        def m_open(_self, name, mode):
            if mode == 0:
                if name not in self.files:
                    return None
                f = dict(data=self.files[name], pos=0, w=None)
            else:
                self.outputs[name] = bytearray()
                f = dict(data=None, pos=0, w=self.outputs[name])
            h = self._next; self._next += 1
            self._open[h] = f
            return h

        def m_close(h):
            self._open.pop(h, None)

        def m_read(h, buf, n):
            f = self._open.get(h)
            if f is None or f["w"] is not None or n < 0:
                return -1
            d = f["data"]
            n = min(n, len(d) - f["pos"])
            if n > 0:
                C.memmove(buf, d[f["pos"]:f["pos"] + n], n)
                f["pos"] += n
            return max(n, 0)

        def m_write(h, buf, n):
            f = self._open.get(h)
            if f is None or f["w"] is None or n < 0:
                return -1
            f["w"] += C.string_at(buf, n)
            return n

        def m_seek(h, off, mode):
            f = self._open.get(h)
            if f is None or f["data"] is None:
                return -1
            base = (0, f["pos"], len(f["data"]))[mode] if 0 <= mode <= 2 else None
            if base is None or base + off < 0 or base + off > len(f["data"]):
                return -1
            f["pos"] = base + off
            return 0

        def m_tell(h):
            f = self._open.get(h)
            return f["pos"] if f else 0

        def m_message(_h, fmt):           # variadic in C; the extra arguments are simply not looked at
            self.messages.append(fmt)
            self.message_handles.append(bool(_h))      # (the reference passes the cabinet's handle with some lines, NULL with others)

        self.messages = []
        self.message_handles = []
        MSG = C.CFUNCTYPE(None, C.c_void_p, C.c_char_p)
        self._cbs = [OPEN(m_open), CLOSE(m_close), RW(m_read), RW(m_write), SEEK(m_seek), TELL(m_tell), MSG(m_message)]
        vp = [C.cast(cb, C.c_void_p).value for cb in self._cbs]
        self.sys = MspackSystem(vp[0], vp[1], vp[2], vp[3], vp[4], vp[5], vp[6], dflt.alloc, dflt.free,
                                dflt.copy, None)

    def ptr(self):
        return C.addressof(self.sys)


def _walk(ptr):
    while ptr:
        yield ptr
        ptr = ptr.contents.next


class Cab:
    """with Cab(path_or_bytes) as cab: cab.files -> [(name, length, offset, comp_type)];
    cab.extract(i) -> (err, bytes)"""

    def __init__(self, src, fix_mszip=0, salvage=0, mem=False, L=None):
        self.L = _setup(L)
        self._tmp = None
        self.mem = None
        if mem:
            self.mem = MemSystem(self.L)
            self.mem.files[b"mem:in"] = bytes(src)
            self.path = b"mem:in"
        else:
            if isinstance(src, (bytes, bytearray)):
                fd, self._tmp = tempfile.mkstemp(suffix=".cab")
                os.write(fd, src); os.close(fd)
                src = self._tmp
            self.path = os.fsencode(src)
        self.d = self.L.mspack_create_cab_decompressor(self.mem.ptr() if self.mem else None)
        if not self.d:
            raise RuntimeError("mspack_create_cab_decompressor failed")
        self.d.contents.set_param(self.d, MSCABD_PARAM_FIXMSZIP, fix_mszip)
        self.d.contents.set_param(self.d, MSCABD_PARAM_SALVAGE, salvage)
        self.cab = self.d.contents.open(self.d, self.path)
        self.open_error = self.d.contents.last_error(self.d)
        self._files = list(_walk(self.cab.contents.files)) if self.cab else []

    @property
    def files(self):
        return [(f.contents.filename, f.contents.length, f.contents.offset,
                 f.contents.folder.contents.comp_type if f.contents.folder else -1) for f in self._files]

    def extract(self, i):
        if self.mem:
            err = self.d.contents.extract(self.d, self._files[i], b"mem:out")
            return err, bytes(self.mem.outputs.get(b"mem:out", b""))
        fd, out = tempfile.mkstemp(suffix=".out")
        os.close(fd)
        try:
            err = self.d.contents.extract(self.d, self._files[i], os.fsencode(out))
            with open(out, "rb") as fh:
                return err, fh.read()
        finally:
            os.unlink(out)

    def close(self):
        if self.cab:
            self.d.contents.close(self.d, self.cab); self.cab = None
        if self.d:
            self.L.mspack_destroy_cab_decompressor(self.d); self.d = None
        if self._tmp:
            os.unlink(self._tmp); self._tmp = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()



MSCABD_PARAM_SEARCHBUF = 0


class CabSet:
    """Several cabinets on ONE decompressor, joined with append()/prepend() like cabextract does
    (reference cabd.c:870-1064):  with CabSet([p1, p2, ...]) as s: s.append(0, 1); s.files(0); s.extract(f)"""

    def __init__(self, srcs, fix_mszip=0, salvage=0, mem=False, L=None):
        self.L = _setup(L)
        self._tmps = []
        self.paths = []
        self.mem = MemSystem(self.L) if mem else None          # (mem=True: srcs are bytes, kept in an in-memory mspack_system)
        for k, src in enumerate(srcs):
            if self.mem:
                self.mem.files[b"mem:in%d" % k] = bytes(src)
                self.paths.append(b"mem:in%d" % k)
                continue
            if isinstance(src, (bytes, bytearray)):
                fd, tmp = tempfile.mkstemp(suffix=".cab")
                os.write(fd, src); os.close(fd)
                self._tmps.append(tmp)
                src = tmp
            self.paths.append(os.fsencode(src))           # must outlive the cabinets (mspack.h:968-969)
        self.d = self.L.mspack_create_cab_decompressor(self.mem.ptr() if self.mem else None)
        if not self.d:
            raise RuntimeError("mspack_create_cab_decompressor failed")
        self.d.contents.set_param(self.d, MSCABD_PARAM_FIXMSZIP, fix_mszip)
        self.d.contents.set_param(self.d, MSCABD_PARAM_SALVAGE, salvage)
        self.cabs, self.open_errors = [], []
        for p in self.paths:
            c = self.d.contents.open(self.d, p)
            self.cabs.append(c)
            self.open_errors.append(0 if c else self.d.contents.last_error(self.d))      # (last_error() is the LAST call's)

    def _cab(self, i):
        return self.cabs[i] if i is not None and i >= 0 else None

    def append(self, a, b):
        return self.d.contents.append(self.d, self._cab(a), self._cab(b))

    def prepend(self, a, b):
        return self.d.contents.prepend(self.d, self._cab(a), self._cab(b))

    def file_ptrs(self, i):
        return list(_walk(self.cabs[i].contents.files))

    def files(self, i):
        """[(name, length, offset, comp_type, folder ordinal, folder num_blocks)] of cabinet i's list"""
        folders = [C.addressof(f.contents) for f in _walk(self.cabs[i].contents.folders)]
        out = []
        for f in self.file_ptrs(i):
            fo = f.contents.folder
            out.append((f.contents.filename, f.contents.length, f.contents.offset,
                        fo.contents.comp_type if fo else -1,
                        folders.index(C.addressof(fo.contents)) if fo else -1,
                        fo.contents.num_blocks if fo else 0))
        return out

    def extract(self, fptr):
        if self.mem:
            self.mem.outputs.clear()
            err = self.d.contents.extract(self.d, fptr, b"mem:out")
            return err, bytes(self.mem.outputs.get(b"mem:out", b""))
        fd, out = tempfile.mkstemp(suffix=".out")
        os.close(fd)
        try:
            err = self.d.contents.extract(self.d, fptr, os.fsencode(out))
            with open(out, "rb") as fh:
                return err, fh.read()
        finally:
            os.unlink(out)

    def close(self):
        if self.d:
            # joined cabinets are freed with the one they are joined to: close one per chain
            seen = set()
            for c in self.cabs:
                if not c:
                    continue
                chain = set()
                w = c
                while w:
                    chain.add(C.addressof(w.contents)); w = w.contents.prevcab
                w = c
                while w:
                    chain.add(C.addressof(w.contents)); w = w.contents.nextcab
                if chain & seen:
                    continue
                seen |= chain
                self.d.contents.close(self.d, c)
            self.cabs = []
            self.L.mspack_destroy_cab_decompressor(self.d); self.d = None
        for t in self._tmps:
            os.unlink(t)
        self._tmps = []

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def cab_search(src, searchbuf=0):
    """search(): [(base_offset, n_files, first file name)] of the cabinets found inside a file"""
    L = _setup()
    tmp = None
    if isinstance(src, (bytes, bytearray)):
        fd, tmp = tempfile.mkstemp(suffix=".bin")
        os.write(fd, src); os.close(fd)
        src = tmp
    path = os.fsencode(src)
    d = L.mspack_create_cab_decompressor(None)
    try:
        if searchbuf:
            d.contents.set_param(d, MSCABD_PARAM_SEARCHBUF, searchbuf)
        head = d.contents.search(d, path)
        err = d.contents.last_error(d)
        found = []
        for c in _walk(head):
            fl = list(_walk(c.contents.files))
            found.append((c.contents.base_offset, len(fl), fl[0].contents.filename if fl else b""))
        if head:
            d.contents.close(d, head)
        return err, found
    finally:
        L.mspack_destroy_cab_decompressor(d)
        if tmp:
            os.unlink(tmp)


class Chm:
    def __init__(self, src, fast=False, mem=False, L=None):
        self.L = _setup(L)
        self._tmp = None
        self.mem = None
        if mem:
            self.mem = MemSystem(self.L)
            self.mem.files[b"mem:in"] = bytes(src)
            self.path = b"mem:in"
        else:
            if isinstance(src, (bytes, bytearray)):
                fd, self._tmp = tempfile.mkstemp(suffix=".chm")
                os.write(fd, src); os.close(fd)
                src = self._tmp
            self.path = os.fsencode(src)
        self.d = self.L.mspack_create_chm_decompressor(self.mem.ptr() if self.mem else None)
        m = self.d.contents
        self.chm = (m.fast_open if fast else m.open)(self.d, self.path)
        self.open_error = m.last_error(self.d)
        self._files = list(_walk(self.chm.contents.files)) if self.chm else []

    @property
    def files(self):
        return [(f.contents.filename, f.contents.length, f.contents.offset, f.contents.section.contents.id)
                for f in self._files]

    def _extract_ptr(self, fptr):
        if self.mem:
            err = self.d.contents.extract(self.d, fptr, b"mem:out")
            return err, bytes(self.mem.outputs.get(b"mem:out", b""))
        fd, out = tempfile.mkstemp(suffix=".out")
        os.close(fd)
        try:
            err = self.d.contents.extract(self.d, fptr, os.fsencode(out))
            with open(out, "rb") as fh:
                return err, fh.read()
        finally:
            os.unlink(out)

    def extract(self, i):
        return self._extract_ptr(self._files[i])

    def find(self, name):
        f = MschmdFile()
        err = self.d.contents.fast_find(self.d, self.chm, name, C.byref(f), C.sizeof(MschmdFile))
        return err, (f if f.section else None)

    def extract_found(self, f):
        return self._extract_ptr(C.pointer(f))

    def close(self):
        if self.chm:
            self.d.contents.close(self.d, self.chm); self.chm = None
        if self.d:
            self.L.mspack_destroy_chm_decompressor(self.d); self.d = None
        if self._tmp:
            os.unlink(self._tmp); self._tmp = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class MsoabDecompressor(C.Structure):
    pass


MsoabDecompressor._fields_ = [
    ("decompress", C.CFUNCTYPE(C.c_int, _P(MsoabDecompressor), C.c_char_p, C.c_char_p)),
    ("decompress_incremental", C.CFUNCTYPE(C.c_int, _P(MsoabDecompressor), C.c_char_p, C.c_char_p, C.c_char_p)),
    ("set_param", C.CFUNCTYPE(C.c_int, _P(MsoabDecompressor), C.c_int, C.c_int)),
]


def oab_decompress(blob, base=None, decompbuf=0, L=None):
    """mspack_create_oab_decompressor -> decompress / decompress_incremental over temporary files
    -> (err, output bytes).  L: as in _setup (the CPU stand-in build of the host logic: tests)"""
    L = L or lib()
    L.mspack_create_oab_decompressor.restype = _P(MsoabDecompressor)
    L.mspack_create_oab_decompressor.argtypes = [C.c_void_p]
    L.mspack_destroy_oab_decompressor.argtypes = [_P(MsoabDecompressor)]
    tmps = []

    def tmp(data):
        fd, path = tempfile.mkstemp(suffix=".oab")
        os.write(fd, data); os.close(fd)
        tmps.append(path)
        return os.fsencode(path)
    d = L.mspack_create_oab_decompressor(None)
    try:
        if decompbuf:
            d.contents.set_param(d, 0, decompbuf)
        pin, pout = tmp(bytes(blob)), tmp(b"")
        if base is None:
            err = d.contents.decompress(d, pin, pout)
        else:
            err = d.contents.decompress_incremental(d, pin, tmp(bytes(base)), pout)
        with open(os.fsdecode(pout), "rb") as fh:
            return err, fh.read()
    finally:
        L.mspack_destroy_oab_decompressor(d)
        for t in tmps:
            os.unlink(t)


class MsszdddHeader(C.Structure):
    _fields_ = [("format", C.c_int), ("length", off_t), ("missing_char", C.c_char)]


class MsszddDecompressor(C.Structure):
    pass


MsszddDecompressor._fields_ = [
    ("open", C.CFUNCTYPE(_P(MsszdddHeader), _P(MsszddDecompressor), C.c_char_p)),
    ("close", C.CFUNCTYPE(None, _P(MsszddDecompressor), _P(MsszdddHeader))),
    ("extract", C.CFUNCTYPE(C.c_int, _P(MsszddDecompressor), _P(MsszdddHeader), C.c_char_p)),
    ("decompress", C.CFUNCTYPE(C.c_int, _P(MsszddDecompressor), C.c_char_p, C.c_char_p)),
    ("last_error", C.CFUNCTYPE(C.c_int, _P(MsszddDecompressor))),
]


class MskwajdHeader(C.Structure):
    _fields_ = [("comp_type", C.c_ushort), ("data_offset", off_t), ("headers", C.c_int), ("length", off_t),
                ("filename", C.c_char_p), ("extra", C.c_char_p), ("extra_length", C.c_ushort)]


class MskwajDecompressor(C.Structure):
    pass


MskwajDecompressor._fields_ = [
    ("open", C.CFUNCTYPE(_P(MskwajdHeader), _P(MskwajDecompressor), C.c_char_p)),
    ("close", C.CFUNCTYPE(None, _P(MskwajDecompressor), _P(MskwajdHeader))),
    ("extract", C.CFUNCTYPE(C.c_int, _P(MskwajDecompressor), _P(MskwajdHeader), C.c_char_p)),
    ("decompress", C.CFUNCTYPE(C.c_int, _P(MskwajDecompressor), C.c_char_p, C.c_char_p)),
    ("last_error", C.CFUNCTYPE(C.c_int, _P(MskwajDecompressor))),
]


def szdd_kwaj_extract(kind, blob, L=None):
    """kind 0 = SZDD, 1 = KWAJ: open() + extract() over temporary files
    -> dict(open_err, err, data, comp_type (SZDD: format), length, filename (SZDD: the missing character)).
    L: as in _setup (the CPU stand-in build of the host logic: tests)"""
    L = L or lib()
    T = MsszddDecompressor if kind == 0 else MskwajDecompressor
    create = L.mspack_create_szdd_decompressor if kind == 0 else L.mspack_create_kwaj_decompressor
    destroy = L.mspack_destroy_szdd_decompressor if kind == 0 else L.mspack_destroy_kwaj_decompressor
    create.restype = _P(T); create.argtypes = [C.c_void_p]; destroy.argtypes = [_P(T)]
    fd, pin = tempfile.mkstemp(suffix=".in"); os.write(fd, bytes(blob)); os.close(fd)
    fd, pout = tempfile.mkstemp(suffix=".out"); os.close(fd)
    d = create(None)
    try:
        h = d.contents.open(d, os.fsencode(pin))
        if not h:
            e = d.contents.last_error(d)
            return dict(open_err=e, err=e, data=b"", comp_type=-1, length=-1, filename=b"")
        hc = h.contents
        info = dict(open_err=0, comp_type=hc.format if kind == 0 else hc.comp_type, length=hc.length,
                    filename=(hc.missing_char.rstrip(b"\0") if kind == 0 else (hc.filename or b"")))
        info["err"] = d.contents.extract(d, h, os.fsencode(pout))
        d.contents.close(d, h)
        with open(pout, "rb") as fh:
            info["data"] = fh.read()
        return info
    finally:
        destroy(d)
        os.unlink(pin); os.unlink(pout)
