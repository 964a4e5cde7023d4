"""Test-corpus writers for the SZDD / KWAJ formats (the reference has no compressor for them that produces
compressed data): an LZSS encoder (lzssd.c:36-91 read backwards), a KWAJ LZH encoder (kwajd.c:432-563:
five Huffman trees in any of the four length encodings), KWAJ MSZIP blocks, and the SZDD / KWAJ headers
(szddd.c:140-170, kwajd.c:155-250).  Everything here is validated against the real reference in
tests/test_oracle_vs_ref.py before the GPU tests rely on it."""
import heapq
import struct
import zlib

import numpy as np

import libmspack_amd as M


# ---- LZSS ----------------------------------------------------------------------------------------------
def lzss_encode(data, mode=0, max_chain=64):
    """greedy LZSS: control bytes LSB first (MSHELP: inverted), 1 = literal, 0 = (ring position, len 3..18)"""
    start = 4096 - (18 if mode == 2 else 16)
    inv = 0xFF if mode == 1 else 0
    hist = bytearray(b" " * 4096) + bytearray(data)       # linear history with the space pre-fill
    base = 4096
    out = bytearray()
    n = len(data)
    table = {}
    i = 0
    items, ctrl = [], 0
    cnt = 0

    def flush():
        nonlocal items, ctrl, cnt
        out.append(ctrl ^ inv)
        for it in items:
            out.extend(it)
        items, ctrl, cnt = [], 0, 0
    while i < n:
        best_len, best_d = 0, 0
        if i + 3 <= n:
            key = bytes(data[i:i + 3])
            for j in reversed(table.get(key, [])[-max_chain:]):
                d = i - j
                if d > 4095:
                    break
                l = 0
                while l < 18 and i + l < n and hist[base + j + l] == data[i + l]:
                    l += 1
                if l > best_len:
                    best_len, best_d = l, d
            # the pre-fill: runs of spaces match ring history before the stream
        if best_len >= 3:
            ring = (start + i) & 4095
            mpos = (ring - best_d) & 4095
            items.append(bytes([mpos & 0xFF, ((mpos >> 4) & 0xF0) | (best_len - 3)]))
            step = best_len
        else:
            ctrl |= 1 << cnt
            items.append(bytes([data[i]]))
            step = 1
        for k in range(step):
            if i + k + 3 <= n:
                table.setdefault(bytes(data[i + k:i + k + 3]), []).append(i + k)
        i += step
        cnt += 1
        if cnt == 8:
            flush()
    if cnt:
        flush()
    return bytes(out)


# ---- KWAJ LZH ------------------------------------------------------------------------------------------
class BitW:
    def __init__(self):
        self.buf = bytearray(); self.acc = 0; self.n = 0

    def put(self, v, nb):
        for k in range(nb - 1, -1, -1):
            self.acc = (self.acc << 1) | ((v >> k) & 1); self.n += 1
            if self.n == 8:
                self.buf.append(self.acc); self.acc = 0; self.n = 0

    def done(self):
        if self.n:
            self.buf.append(self.acc << (8 - self.n)); self.acc = 0; self.n = 0
        return bytes(self.buf)


def huff_lengths(freq, maxlen=15):
    """code lengths of a COMPLETE prefix code over the symbols with freq > 0 (at least two of them)"""
    syms = [s for s, f in enumerate(freq) if f > 0]
    lens = [0] * len(freq)
    if len(syms) < 2:
        for s in range(len(freq)):
            if s not in syms and len(syms) < 2:
                syms.append(s)
    h = [(max(freq[s], 1), s, (s,)) for s in syms]
    heapq.heapify(h)
    depth = {s: 0 for s in syms}
    while len(h) > 1:
        f1, _, g1 = heapq.heappop(h); f2, _, g2 = heapq.heappop(h)
        for s in g1 + g2:
            depth[s] += 1
        heapq.heappush(h, (f1 + f2, min(g1 + g2), g1 + g2))
    if max(depth.values()) > maxlen:
        return None
    for s, d in depth.items():
        lens[s] = d
    return lens


def canon_codes(lens):
    code, codes = 0, {}
    for l in range(1, 17):
        for s, x in enumerate(lens):
            if x == l:
                codes[s] = (code, l); code += 1
        code <<= 1
    return codes


def put_lens(w, typ, lens):
    n = len(lens)
    if typ == 0:
        return
    if typ == 3:
        for x in lens:
            w.put(x, 4)
    elif typ == 1:
        c = lens[0]; w.put(c, 4)
        for x in lens[1:]:
            if x == c:
                w.put(0, 1)
            elif x == c + 1:
                w.put(2, 2); c = x
            else:
                w.put(3, 2); w.put(x, 4); c = x
    elif typ == 2:
        c = lens[0]; w.put(c, 4)
        for x in lens[1:]:
            if -1 <= x - c <= 1:
                w.put(x - c + 1, 2); c = x
            else:
                w.put(3, 2); w.put(x, 4); c = x


def lzh_encode(data, types=(3, 3, 3, 3, 3), max_chain=32):
    """KWAJ method 3.  Tokens: literal runs of 1..32 and matches of 3..17 at distance 1..4095."""
    n = len(data)
    toks = []                                             # ("L", bytes) | ("M", len, dist)
    table, i, lits = {}, 0, bytearray()
    while i < n:
        best_len, best_d = 0, 0
        if i + 3 <= n:
            for j in reversed(table.get(bytes(data[i:i + 3]), [])[-max_chain:]):
                d = i - j
                if d > 4095:
                    break
                l = 0
                while l < 17 and i + l < n and data[j + l] == data[i + l]:
                    l += 1
                if l > best_len:
                    best_len, best_d = l, d
        step = best_len if best_len >= 3 else 1
        if best_len >= 3:
            if lits:
                toks.append(("L", bytes(lits))); lits = bytearray()
            toks.append(("M", best_len, best_d))
        else:
            lits.append(data[i])
            if len(lits) == 32:
                toks.append(("L", bytes(lits))); lits = bytearray()
        for k in range(step):
            if i + k + 3 <= n:
                table.setdefault(bytes(data[i + k:i + k + 3]), []).append(i + k)
        i += step
    if lits:
        toks.append(("L", bytes(lits)))
    # symbol statistics: MATCHLEN1 after a match or a full run, MATCHLEN2 after a short literal run
    f = [[0] * 16, [0] * 16, [0] * 32, [0] * 64, [0] * 256]
    lit_run = 0
    for t in toks:
        tab = 1 if lit_run else 0
        if t[0] == "M":
            f[tab][t[1] - 2] += 1; f[3][t[2] >> 6] += 1; lit_run = 0
        else:
            f[tab][0] += 1; f[2][len(t[1]) - 1] += 1
            for b in t[1]:
                f[4][b] += 1
            lit_run = 0 if len(t[1]) == 32 else 1
    lens, types = [], list(types)
    for k in range(5):
        if types[k] == 0:
            nb = {16: 4, 32: 5, 64: 6, 256: 8}[len(f[k])]
            lens.append([nb] * len(f[k]))
        else:
            l = huff_lengths(f[k])
            if l is None:
                types[k] = 0; nb = {16: 4, 32: 5, 64: 6, 256: 8}[len(f[k])]; l = [nb] * len(f[k])
            lens.append(l)
    codes = [canon_codes(l) for l in lens]
    w = BitW()
    for k in range(5):
        w.put(types[k], 4)
    w.put(0, 4)                                           # the sixth type field (byte alignment)
    for k in range(5):
        put_lens(w, types[k], lens[k])
    lit_run = 0
    for t in toks:
        tab = 1 if lit_run else 0
        if t[0] == "M":
            w.put(*codes[tab][t[1] - 2]); w.put(*codes[3][t[2] >> 6]); w.put(t[2] & 63, 6); lit_run = 0
        else:
            w.put(*codes[tab][0]); w.put(*codes[2][len(t[1]) - 1])
            for b in t[1]:
                w.put(*codes[4][b])
            lit_run = 0 if len(t[1]) == 32 else 1
    return w.done()


# ---- containers ----------------------------------------------------------------------------------------
def szdd_file(data, qbasic=False, missing=b"x"):
    if qbasic:
        return b"SZ \x88\xF0\x27\x33\xD1" + struct.pack("<I", len(data)) + lzss_encode(data, 2)
    return b"SZDD\x88\xF0\x27\x33" + b"A" + missing + struct.pack("<I", len(data)) + lzss_encode(data, 0)


def kwaj_mszip(data, block=32768):
    out = bytearray()
    for p in range(0, len(data), block):
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        blk = b"CK" + co.compress(data[p:p + block]) + co.flush()
        out += struct.pack("<H", len(blk)) + blk
    return bytes(out) + b"\0\0"


def kwaj_file(data, method, name=None, ext=None, length=True, extra=None, lzh_types=(3, 3, 3, 3, 3)):
    flags, opt = 0, b""
    if length:
        flags |= 1; opt += struct.pack("<I", len(data))
    if name is not None:
        flags |= 8; opt += name + b"\0"
    if ext is not None:
        flags |= 0x10; opt += ext + b"\0"
    if extra is not None:
        flags |= 0x20; opt += struct.pack("<H", len(extra)) + extra
    if method == 0:
        payload = data
    elif method == 1:
        payload = bytes(b ^ 0xFF for b in data)
    elif method == 2:
        payload = lzss_encode(data, 2)
    elif method == 3:
        payload = lzh_encode(data, lzh_types)
    elif method == 4:
        payload = kwaj_mszip(data)
    else:
        payload = data
    off = 14 + len(opt)
    return b"KWAJ\x88\xF0\x27\xD1" + struct.pack("<HHH", method, off, flags) + opt + payload


def texts():
    t0 = M.gen_plaintext(21, 0, 60000).tobytes()
    t1 = M.gen_plaintext(22, 2, 30000).tobytes()
    t2 = (b"the quick brown fox jumps over the lazy dog. " * 400)[:17000]
    t3 = b" " * 5000 + b"spaces before the stream match the window pre-fill" + b" " * 3000
    return [t0, t1, t2, t3, b"x", b""]
