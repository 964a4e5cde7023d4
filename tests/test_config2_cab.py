"""BASELINE config 2 as a container: ONE cabinet with 4096 folders of one MSZIP CFDATA block (32 KiB) each,
through mspack_create_cab_decompressor -> open -> extract (cabd.c:363 allows 65535 folders).
  * CPU (development container): the REAL reference extracts every file of this cabinet to the plaintext --
    the corpus itself is pinned;
  * GPU: the same cabinet through include/mspack.h on libmspack_hip.so; the first extract() decodes all 4096
    folders in one batch, every file must equal the plaintext (and so the reference)."""
import zlib

import numpy as np
import pytest

import libmspack_amd as M
from libmspack_amd import api
import helpers

N, UB = 4096, 32768


def build_cab():
    plain = M.gen_plaintext(0xC0FFEE, 0, N * UB)
    folders, files = [], []
    for i in range(N):
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        blob = b"CK" + co.compress(plain[i * UB:(i + 1) * UB].tobytes()) + co.flush()
        folders.append((1, [blob], [UB]))
        files.append((b"f%04d.bin" % i, UB, 0, i))
    return plain, M.cab_write(folders, files)


@pytest.mark.skipif(not helpers.have_ref(), reason="oracle/_ref (the compiled reference) is not present")
def test_config2_cab_reference_cpu(built):
    plain, cab = build_cab()
    e, lst = helpers.ref_cab_list(cab)
    assert e == 0 and len(lst) == N
    order = list(range(0, N, 7)) + [N - 1, 0]
    rc, outs = helpers.ref_cab_extract(cab, order, cap=len(order) * UB + 4096)
    assert rc == 0
    for i, (err, data) in zip(order, outs):
        assert err == 0 and data == plain[i * UB:(i + 1) * UB].tobytes(), i


@pytest.mark.gpu
def test_config2_cab_gpu(built):
    plain, cab = build_cab()
    with api.Cab(cab, mem=True) as c:
        assert c.open_error == 0 and len(c.files) == N
        for i in list(range(N - 1, -1, -1))[:64] + list(range(N)):        # some backwards, then all in order
            err, data = c.extract(i)
            assert err == 0 and data == plain[i * UB:(i + 1) * UB].tobytes(), i
