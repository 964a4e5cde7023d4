"""The fold tasks (mspack_lzx_fold / mspack_mszip_fold: lzx_fold.hpp, fold_common.hpp, mszip_kernel.hpp zip_fold_block) -- a folder's chain
of LZ77 copies as one gather pass per frame.  Which launches take them is a rule of the library (few units of many frames); MSPACK_HIP_FOLD=2
makes every launch that can take them do so, =0 none: the existing frame / block parity tests -- right and wrong tables, damaged and cut
streams, blocks that span frames, Microsoft's blocks of megabytes, runs of every period -- must not be able to tell.  (Own processes:
the switch is read when the library loads.)  And the shapes the rule is for, with the shipped default: ONE folder of ordinary data."""
import os
import subprocess
import sys
import zlib

import numpy as np
import pytest

import libmspack_amd as M
from helpers import oracle_lzx, oracle_mszip

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("policy", ["2", "0"], ids=["always", "never"])
def test_frame_and_block_parity_with_the_fold_tasks_forced_on_and_off(built, policy):
    ids = ["tests/test_gpu_lzx_frames.py", "tests/test_gpu_mszip_blocks.py", "tests/test_gpu_runs.py", "tests/test_gpu_lzx_log.py",
           "tests/test_gpu_mszip.py::test_mszip_request_that_ends_inside_a_block"]
    env = dict(os.environ, MSPACK_HIP_FOLD=policy)
    p = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider", "-k",
                        "not launch_paths and not headline_batch"] + ids, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1700)
    assert p.returncode == 0, p.stdout.decode()[-3000:]


def one_folder(kind, stream, tab, total, wb):
    base = (len(stream) + 64 + 15) & ~15
    arena = np.zeros(base + 4 * len(tab) + 64, dtype=np.uint8)
    arena[:len(stream)] = np.frombuffer(stream, dtype=np.uint8)
    arena[base:base + 4 * len(tab)] = np.asarray(tab, dtype=np.uint32).view(np.uint8)
    units, out_bytes = M.make_units(kind, [0], [len(stream)], [total], window_bits=wb, reset_frames=0, frame_tabs=[base],
                                    out_slack=32768 if kind == M.KIND_MSZIP else 0)
    out, res = M.decode_batch(units, arena, out_bytes)
    return out[:total], res[0]


@pytest.mark.parametrize("text", [M.TEXT_MIX, M.TEXT_ENGLISH, M.TEXT_REPETITIVE, M.TEXT_RANDOM], ids=["mix", "english", "repetitive", "random"])
def test_one_folder_of_ordinary_data(built, text):
    """ONE cabinet folder of 96 frames: LZX-21 with one block per frame and with blocks of 1 MiB, LZX-16 (sources beyond the window's
    wrap), MSZIP with history -- the launch shape the fold tasks are for (shipped rule), every byte against the plaintext and the
    result words against the oracle."""
    n = 96 * 32768 + 4321
    plain = M.gen_plaintext(400 + text, text, n)
    for wb, bs in ((21, 0), (21, 1 << 20), (16, 0)):
        lz, fo = M.lzx_encode(plain, wb, 0, M.lzx_opts(block_size=bs) if bs else None)
        out, r = one_folder(M.KIND_LZX, lz.tobytes(), np.asarray(fo[:-1]), n, wb)
        e, o, orc = oracle_lzx(lz.tobytes(), n, wb, 0)
        assert r["err"] == e == 0 and r["out_len"] == n and r["in_next"] == orc.in_next, (wb, bs, r)
        assert np.array_equal(out, plain), (wb, bs)
        assert r["flags"] & M.F_FRAMES_ADOPTED, (wb, bs)
    blocks, prev = [], None
    for k in range(0, n, 32768):
        b = plain[k:k + 32768].tobytes()
        c = zlib.compressobj(6, zlib.DEFLATED, -15, 9, 0, prev) if prev else zlib.compressobj(6, zlib.DEFLATED, -15)
        blocks.append(b"CK" + c.compress(b) + c.flush()); prev = b
    s = b"".join(blocks)
    out, r = one_folder(M.KIND_MSZIP, s, np.cumsum([0] + [len(b) for b in blocks[:-1]]), n, 0)
    e, o, orc, _ = oracle_mszip(s, n)
    assert r["err"] == e == 0 and r["out_len"] == n and np.array_equal(out, plain)
    assert text == M.TEXT_RANDOM or r["flags"] & M.F_FRAMES_ADOPTED       # (zlib STORES incompressible blocks: those are the serial path's)


def test_one_folder_with_damage_in_the_middle(built):
    """a 64-frame LZX folder with a flipped bit in frame 40 and one cut short, an MSZIP folder with a bad block in the middle: the chain
    of folded frames ends there and the serial path reports what the reference reports (error code, byte count, the bytes below)."""
    n = 64 * 32768
    plain = M.gen_plaintext(431, M.TEXT_MIX, n)
    lz, fo = M.lzx_encode(plain, 21, 0, M.lzx_opts(block_size=1 << 19))
    tab = np.asarray(fo[:-1])
    for what in ("flip", "cut"):
        s = bytearray(lz.tobytes())
        if what == "flip":
            s[int(fo[40]) + 777] ^= 0x04
        else:
            s = s[:int(fo[50]) + 100]
        s = bytes(s)
        out, r = one_folder(M.KIND_LZX, s, tab, n, 21)
        e, o, orc = oracle_lzx(s, n, 21, 0)
        assert r["err"] == e and r["out_len"] == orc.out_len, (what, r, e, orc.out_len)
        assert out[:orc.out_len].tobytes() == o[:orc.out_len], what
    blocks, prev = [], None
    for k in range(0, n, 32768):
        b = plain[k:k + 32768].tobytes()
        c = zlib.compressobj(6, zlib.DEFLATED, -15, 9, 0, prev) if prev else zlib.compressobj(6, zlib.DEFLATED, -15)
        blocks.append(bytearray(b"CK" + c.compress(b) + c.flush())); prev = b
    blocks[33][len(blocks[33]) // 2] ^= 0x20
    s = b"".join(bytes(b) for b in blocks)
    out, r = one_folder(M.KIND_MSZIP, s, np.cumsum([0] + [len(b) for b in blocks[:-1]]), n, 0)
    e, o, orc, _ = oracle_mszip(s, n)
    assert r["err"] == e and r["out_len"] == orc.out_len, (r, e, orc.out_len)
    assert out[:orc.out_len].tobytes() == o[:orc.out_len]
