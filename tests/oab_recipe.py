"""OAB (".LZX") container writer for tests -- the layout oabd.c reads (oab.h: headers, block headers) -- on
top of our LZX DELTA encoder, and the deterministic cases shared by tests/golden/make_oab_golden.py
(reference side) and tests/test_oab.py (our side)."""
import struct
import zlib

import numpy as np

import libmspack_amd as M


def oab_crc(data):
    """reflected CRC-32 from 0xffffffff without the final inversion (crc32.h, oabd.c:88-100)"""
    return (~zlib.crc32(data)) & 0xFFFFFFFF


def window_bits_for(size):
    wb = 17
    while wb < 25 and (1 << wb) < size:
        wb += 1
    return wb


def oab_full(data, block, stored_every=0, **kw):
    """full file: version 3.1; blocks of `block` bytes, every stored_every-th one uncompressed"""
    out = [struct.pack("<IIII", 3, 1, block, len(data))]
    for k, p in enumerate(range(0, len(data), block)):
        chunk = data[p:p + block]
        if stored_every and (k % stored_every) == stored_every - 1:
            out.append(struct.pack("<IIII", 0, len(chunk), len(chunk), 0) + chunk)
        else:
            comp = M.lzxd_encode(np.frombuffer(chunk, dtype=np.uint8), window_bits_for(len(chunk)), **kw).tobytes()
            comp += b"\0" * ((-len(comp)) % 4 + 4)                      # trailing padding the reader skips
            out.append(struct.pack("<IIII", 1, len(comp), len(chunk), oab_crc(chunk)) + comp)
    return b"".join(out)


def oab_patch(base, data, block, **kw):
    """incremental patch: version 3.2; block k turns base[k*block:(k+1)*block] into data[...]"""
    nblk = max((len(data) + block - 1) // block, 1)
    sblk = (len(base) + nblk - 1) // nblk
    out = [struct.pack("<IIIIIII", 3, 2, max(block, sblk), len(base), len(data), oab_crc(base), oab_crc(data))]
    for k in range(nblk):
        chunk = data[k * block:(k + 1) * block]
        src = base[k * sblk:(k + 1) * sblk]
        wb = window_bits_for(((len(src) + 32767) & ~32767) + len(chunk))
        comp = M.lzxd_encode(np.frombuffer(chunk, dtype=np.uint8), wb, src, **kw).tobytes()
        comp += b"\0" * ((-len(comp)) % 4 + 4)
        out.append(struct.pack("<IIII", len(comp), len(chunk), len(src), oab_crc(chunk)) + comp)
    return b"".join(out)


def cases():
    """-> list of (name, blob, base or None, expected plaintext)"""
    out = []
    d1 = M.gen_plaintext(11, 0, 700000).tobytes()
    out.append(("full_256k_blocks", oab_full(d1, 262144), None, d1))
    out.append(("full_mixed_stored", oab_full(d1[:300000], 65536, stored_every=3, mode=4, block_size=20000), None, d1[:300000]))
    d2 = bytearray(M.gen_plaintext(12, 2, 400000).tobytes())
    d2[100000:160000] = b"\x41" * 60000                                   # extended match lengths
    d2 = bytes(d2)
    out.append(("full_long_runs", oab_full(d2, 200000), None, d2))
    out.append(("full_tiny", oab_full(b"hello, offline address book\n" * 3, 4096), None, b"hello, offline address book\n" * 3))
    base = M.gen_plaintext(13, 0, 500000).tobytes()
    new = bytearray(base)
    rng = np.random.RandomState(5)
    for _ in range(40):                                                   # scattered edits + an insertion
        k = int(rng.randint(0, len(new) - 100)); new[k:k + int(rng.randint(1, 60))] = bytes(rng.randint(0, 256, 20, dtype=np.uint8))
    new[250000:250000] = M.gen_plaintext(14, 1, 5000).tobytes()
    new = bytes(new)
    out.append(("patch_128k_blocks", oab_patch(base, new, 131072), base, new))
    out.append(("patch_one_block", oab_patch(base[:200000], new[:180000], 262144, mode=2), base[:200000], new[:180000]))
    return out


def damaged(blob, rng, n):
    from test_gpu_fuzz import mutations
    return mutations(blob, rng, n)
