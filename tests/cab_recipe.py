"""A small synthetic cabinet with one folder per method -- MSZIP (with history), LZX, Quantum (window 2^15), stored -- and two files
per folder that tile it ("a<k>.bin" = the first `cut` bytes, "b<k>.bin" = the rest), rebuilt byte for byte from a seed (the corpus
generators are deterministic; goldens keep the cabinet's MD5).  Shared by tests/golden/make_cab_sticky_golden.py, tests/test_cab_sticky.py
and tools/fuzz_drivers_cpu.py."""
import struct
import zlib

import numpy as np

import libmspack_amd as M


def base_cab(seed, cut=None):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(40000, 140000))
    data = M.gen_plaintext(seed, int(rng.integers(0, 4)), n)
    mb, mu, prev = [], [], None
    for k in range(0, n, 32768):
        b = data[k:k + 32768].tobytes()
        c = zlib.compressobj(6, zlib.DEFLATED, -15, 9, 0, prev) if prev else zlib.compressobj(6, zlib.DEFLATED, -15)
        mb.append(b"CK" + c.compress(b) + c.flush()); mu.append(len(b)); prev = b
    wb = int(rng.integers(15, 19))
    lz, fo = M.lzx_encode(data, wb, 0)
    lb = [lz[int(fo[i]):int(fo[i + 1])].tobytes() for i in range(len(fo) - 1)]
    qs, fs = M.qtm_encode(data, 15)
    pos, qb = 0, []
    for s in fs:
        qb.append(bytes(qs[pos:pos + int(s)])); pos += int(s) + 1
    sb = [data[k:k + 32768].tobytes() for k in range(0, n, 32768)]
    folders = [(1, mb, mu), (3 | (wb << 8), lb, mu), (2 | (15 << 8), qb, mu), (0, sb, mu)]
    cut0 = int(rng.integers(1, n - 1))
    cut = cut0 if cut is None else (n // 32768 * 32768 + 100 if cut == "last_window" and n % 32768 > 200 and n > 32768 else cut0)
    files = []
    for fi in range(4):
        files.append((b"a%d.bin" % fi, cut, 0, fi)); files.append((b"b%d.bin" % fi, n - cut, cut, fi))
    return bytearray(M.cab_write(folders, files))


def file_entry_offsets(cab):
    """byte positions of the CFFILE entries (cab.h:60-67)"""
    coff_files, = struct.unpack_from("<I", cab, 16)
    n_files, = struct.unpack_from("<H", cab, 28)
    pos, out = coff_files, []
    for _ in range(n_files):
        out.append(pos)
        pos += 16
        while cab[pos]:
            pos += 1
        pos += 1
    return out


def qtm_cab(seed, wb, cuts, n, kind=0):
    """ONE Quantum folder (window 2^wb) of n bytes, its files cut at `cuts` (ascending positions inside (0, n)): the cabinets of
    tests/golden/cab_qtm_carry.json -- requests then end at chosen places of the token stream (qtmd.c:268-276, 358-374)"""
    data = M.gen_plaintext(seed, kind, n)
    qs, fs = M.qtm_encode(data, wb)
    pos, qb = 0, []
    for s in fs:
        qb.append(bytes(qs[pos:pos + int(s)])); pos += int(s) + 1
    mu = [min(32768, n - k) for k in range(0, n, 32768)]
    edges = [0] + list(cuts) + [n]
    files = [(b"f%02d.bin" % i, edges[i + 1] - edges[i], edges[i], 0) for i in range(len(edges) - 1)]
    return bytearray(M.cab_write([(2 | (wb << 8), qb, mu)], files)), data
