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


def test_cab_gather_grows_its_arena_cpu(built, hostlogic):
    """cabd.c gathers the CFDATA payloads of all folders straight into ONE growing input arena whose first size is a guess from the
    cabinet header's length field (round 4).  A header that understates the length (the reference never checks the field
    against the file) makes the arena grow several times WHILE blocks are being read into it: the folders, their frame tables
    and the block that is being read must all survive every move.  Host logic on the CPU stand-in for the batch ABI; expected
    bytes: the plaintext (which the real reference also extracts from this cabinet, where it is built)."""
    import struct
    n, fb = 24, 5                                              # 24 MSZIP folders of 5 blocks: ~1.4 MiB of payload
    ub = fb * UB
    plain = M.gen_plaintext(0xFEED, 2, n * ub)                 # (the binary family: compresses badly, large payloads)
    folders, files = [], []
    for i in range(n):
        blocks, d = [], None
        for b in range(fb):
            co = zlib.compressobj(6, zlib.DEFLATED, -15) if d is None else zlib.compressobj(6, zlib.DEFLATED, -15, zdict=d)
            d = plain[i * ub + b * UB:i * ub + (b + 1) * UB].tobytes()
            blocks.append(b"CK" + co.compress(d) + co.flush())
        folders.append((1, blocks, [UB] * fb))
        files.append((b"g%02d.bin" % i, ub, 0, i))
    cab = bytearray(M.cab_write(folders, files))
    assert len(cab) > (1 << 20)
    struct.pack_into("<I", cab, 8, 64)                         # cbCabinet: "this cabinet is 64 bytes long"
    if helpers.have_ref():
        rc, outs = helpers.ref_cab_extract(bytes(cab), [0, n - 1], cap=2 * ub + 4096)
        assert rc == 0 and [e for e, _d in outs] == [0, 0] and outs[1][1] == plain[(n - 1) * ub:].tobytes()
    with api.Cab(bytes(cab), mem=True, L=hostlogic) as c:
        assert c.open_error == 0 and len(c.files) == n
        for i in [n - 1, 0] + list(range(n)):
            err, data = c.extract(i)
            assert err == 0 and data == plain[i * ub:(i + 1) * ub].tobytes(), i


def _cab_job_scenario(hostlogic):
    """The cabinet driver's batch as a job (include/mspack_hip.h: mspack_hip_decode_batch_begin; csrc/host/cabd.c: struct cab_batch,
    folder_settle): extract() waits for ITS folder's unit and its blocks' checksum units, every folder is settled once, and a
    cabinet that is closed while most of its folders have not been asked for gives everything back (the stand-in's job is lazy:
    such folders are only decoded by the _end() that close() brings).  Jobs on and off: the same bytes, in several orders -- one
    of the folders with a CFDATA block whose checksum is wrong, which the reference refuses with MSPACK_ERR_CHECKSUM (cabd.c:1411-1417):
    found by the checksum unit on the device path, the folder is then gathered again the reference's way."""
    import ctypes
    import os
    import struct
    if hostlogic is not None:
        hostlogic.mspack_standin_jobs_begun.restype = ctypes.c_ulong
        hostlogic.mspack_standin_job_waits.restype = ctypes.c_ulong
    begun = (lambda: hostlogic.mspack_standin_jobs_begun()) if hostlogic is not None else (lambda: 0)
    n, fb = 12, 3
    ub = fb * UB
    plain = M.gen_plaintext(0xF00D, 0, n * ub)
    folders, files = [], []
    for i in range(n):
        blocks, d = [], None
        for b in range(fb):
            co = zlib.compressobj(6, zlib.DEFLATED, -15) if d is None else zlib.compressobj(6, zlib.DEFLATED, -15, zdict=d)
            d = plain[i * ub + b * UB:i * ub + (b + 1) * UB].tobytes()
            blocks.append(b"CK" + co.compress(d) + co.flush())
        folders.append((1, blocks, [UB] * fb))
        files.append((b"j%02d.bin" % i, ub, 0, i))
    good = M.cab_write(folders, files)
    # damage one payload byte of folder 5's second block (its stored checksum no longer fits)
    bad = bytearray(good)
    hdr_folders = 36                                           # CFHEADER without reserve fields
    coff5 = struct.unpack_from("<I", bad, hdr_folders + 8 * 5)[0]
    first_len = struct.unpack_from("<H", bad, coff5 + 4)[0]
    second = coff5 + 8 + first_len
    bad[second + 8 + 10] ^= 0x40
    orders = [list(range(n)), [n - 1, 0, 5, 6], [5], [3]]
    want = None
    if helpers.have_ref():
        rc, outs = helpers.ref_cab_extract(bytes(bad), list(range(n)), cap=n * ub + 4096)
        assert rc == 0
        want = [(e, d) for e, d in outs]
        assert want[5][0] != 0 and all(e == 0 for k, (e, _d) in enumerate(want) if k != 5)
    for image, tag in ((good, "good"), (bytes(bad), "bad")):
        got = {}
        for jobs in ("1", "0"):
            os.environ["MSPACK_HIP_JOBS"] = jobs
            try:
                b0 = begun()
                for oi, order in enumerate(orders):
                    with api.Cab(image, mem=True, L=hostlogic) as c:
                        assert c.open_error == 0 and len(c.files) == n
                        for i in order:
                            err, data = c.extract(i)
                            got.setdefault((oi, i), []).append((err, bytes(data)))
                            if tag == "good" or i != 5:
                                assert err == 0 and data == plain[i * ub:(i + 1) * ub].tobytes(), (tag, jobs, oi, i, err)
                            elif want is not None:
                                assert err == want[5][0] and bytes(data) == want[5][1], (tag, jobs, oi, err, want[5][0], len(data), len(want[5][1]))
                b1 = begun()
                assert hostlogic is None or (b1 > b0) == (jobs == "1"), (tag, jobs, b0, b1)
            finally:
                os.environ.pop("MSPACK_HIP_JOBS", None)
        for key, pair in got.items():
            assert pair[0] == pair[1], (tag, key, pair[0][0], pair[1][0])


def test_cab_batch_runs_as_a_job_cpu(built, hostlogic):
    _cab_job_scenario(hostlogic)


@pytest.mark.gpu
def test_cab_batch_runs_as_a_job_gpu(built, tmp_path):
    """the same scenario on the real library (MSPACK_HIP_JOBS is read once per process there: the jobs-off leg runs the drivers'
    job path too -- what differs from the CPU leg is WHO decodes; the bytes must be the reference's either way)"""
    _cab_job_scenario(None)
