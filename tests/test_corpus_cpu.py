"""CPU: the corpus generators are deterministic and self-consistent (encoder -> oracle round trip),
and the container writers produce what the test-side gatherer parses."""
import hashlib
import zlib

import numpy as np

import libmspack_amd as M
from helpers import cab_folders, folder_stream, oracle_lzx, oracle_qtm


def test_plaintext_is_deterministic(built):
    for kind in range(6):
        a = M.gen_plaintext(42, kind, 10000)
        b = M.gen_plaintext(42, kind, 10000)
        c = M.gen_plaintext(43, kind, 10000)
        assert np.array_equal(a, b) and not np.array_equal(a, c)
    assert hashlib.md5(M.gen_plaintext(1, 0, 4096).tobytes()).hexdigest() == \
        hashlib.md5(M.gen_plaintext(1, 0, 8192)[:4096].tobytes()).hexdigest()


def test_lzx_roundtrip_via_oracle(built):
    d = M.gen_plaintext(3, 0, 100000)
    for wb, reset, kw in [(21, 2, {}), (15, 0, dict(mode=4, block_size=7777)), (18, 1, dict(intel_filesize=77777))]:
        comp, fo = M.lzx_encode(d, wb, reset, M.lzx_opts(**kw))
        assert int(fo[-1]) == comp.size and all(int(fo[i]) <= int(fo[i + 1]) for i in range(len(fo) - 1))
        e, o, _r = oracle_lzx(comp.tobytes() + b"\0" * 8, d.size, wb, reset)
        assert e == 0 and o == d.tobytes()


def test_lzx_units_are_independent(built):
    n, ub = 8, 65536
    plain, comp, off, ln = M.corpus_lzx_units(0xABC, 0, n, ub, 21, n_threads=2)
    for i in range(n):
        s = comp[int(off[i]):int(off[i]) + int(ln[i]) + 4].tobytes()
        e, o, _r = oracle_lzx(s, ub, 21, 2)
        assert e == 0 and o == plain[i * ub:(i + 1) * ub].tobytes()


def test_qtm_roundtrip_via_oracle(built):
    d = M.gen_plaintext(4, 0, 70000)
    for wb in (10, 16, 21):
        s, fs = M.qtm_encode(d, wb)
        assert len(fs) == 3 and len(s) == int(fs.sum()) + 3
        e, o, _r = oracle_qtm(s, d.size, wb)
        assert e == 0 and o == d.tobytes()


def test_cab_writer_parses(built):
    d = M.gen_plaintext(6, 0, 50000)
    blk = [b"CK" + zlib.compress(d[i:i + 32768].tobytes())[2:-4] for i in range(0, d.size, 32768)]
    cab = M.cab_write([(1, blk, [32768, d.size - 32768])], [(b"x.txt", d.size, 0, 0)])
    f = cab_folders(cab)
    assert len(f) == 1 and f[0]["comp_type"] == 1 and f[0]["files"] == [(b"x.txt", 0, d.size)]
    assert folder_stream(f[0]) == b"".join(blk)
