"""The real kernel sources on the wavefront emulator (tests/emu/): kernel LOGIC checked without a GPU.

tests/emu/build_emu.sh compiles libmspack_amd/csrc/hip/shim.hip -- unchanged -- for the host CPU (64 fibers per wavefront,
SIMT lock step through instrumentation hooks, cross-lane builtins as collectives) into
tests/_build/libmspack_emu.so.  A fresh process (the ctypes mirror reads MSPACK_HIP_SO when it is imported) decodes small
batches through the C ABI and compares them with the oracle: LZX units with frame tables through mspack_lzx_pipe (parse
tasks, commit tasks, hand-offs between concurrently running workgroups, the resuming unit kernel), a damaged one and one
with a wrong table, an MSZIP folder and a Quantum folder.  Test infrastructure only: the product library never contains or
loads the emulator, and nothing here is a fallback for it."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
SO = os.path.join(ROOT, "tests", "_build", "libmspack_emu.so")

WORKER = r'''
import sys, zlib
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
import numpy as np
import libmspack_amd as M
import test_gpu_lzx_frames as T
assert "emu" in M.HIP_SO
# ---- LZX with frame tables -> mspack_lzx_pipe: right table, wrong table, damaged and cut streams ----
data = M.gen_plaintext(17, M.TEXT_MIX, 4 * 32768 + 1234)
comp, fo = M.lzx_encode(data, 21, 2)
fo = fo.astype(np.int64)[:-1]
c = comp.tobytes()
bad = bytearray(c); bad[int(fo[1]) + 900] ^= 0x10
streams = [c, c, bytes(bad), c[:int(fo[2]) + 1], c[int(fo[2]):]]
params = [(data.size, 21, 2, 0)] * 4 + [(data.size - 65536, 21, 2, 65536)]
tabs = [fo, fo + 2, fo, fo, fo[2:] - fo[2]]
units, out, res = T.run(streams, params, tabs)
T.check(streams, params, units, out, res, compare_bytes=False)
for i in (0, 1):
    assert res["err"][i] == 0 or res["out_len"][i] == data.size
    assert np.array_equal(out[units["out_off"][i]:units["out_off"][i] + data.size], data), i
assert res["flags"][0] & T.ADOPTED and res["flags"][4] & T.ADOPTED
# ---- (ADVICE round 5) a frame that holds a block's END and the next block's header (lzx_pipe_parse_tail: blocks of 40 000 bytes),
# and run fills whose period is <= 64 and does not divide 64 (spq_fill_run: 24- and 7-byte periods, runs of >= 512 bytes) ----
d2 = M.gen_plaintext(31, M.TEXT_MIX, 3 * 32768 + 99)
c2, f2 = M.lzx_encode(d2, 21, 0, M.lzx_opts(block_size=40000))
rep = (np.frombuffer(bytes(range(24)) * 700, dtype=np.uint8).tolist() + M.gen_plaintext(32, M.TEXT_MIX, 3000).tolist() +
       list(b"abcdefg" * 1500) + M.gen_plaintext(33, M.TEXT_MIX, 2 * 32768).tolist())
d3 = np.asarray(rep[:2 * 32768 + 500], dtype=np.uint8)
c3, f3 = M.lzx_encode(d3, 21, 0)
streams = [c2.tobytes(), c3.tobytes()]
params = [(d2.size, 21, 0, 0), (d3.size, 21, 0, 0)]
units, out, res = T.run(streams, params, [f2.astype(np.int64)[:-1], f3.astype(np.int64)[:-1]])
T.check(streams, params, units, out, res, compare_bytes=False)
for i, dd in enumerate((d2, d3)):
    assert res["err"][i] == 0 and np.array_equal(out[units["out_off"][i]:units["out_off"][i] + dd.size], dd), i
    assert res["flags"][i] & T.ADOPTED, i
# ---- MSZIP: one folder of two blocks with history; Quantum: one folder ----
d = M.gen_plaintext(9, 0, 50000).tobytes()
co = zlib.compressobj(6, zlib.DEFLATED, -15); b0 = b"CK" + co.compress(d[:32768]) + co.flush()
co = zlib.compressobj(6, zlib.DEFLATED, -15, zdict=d[:32768]); b1 = b"CK" + co.compress(d[32768:]) + co.flush()
zs = b0 + b1
qs = bytes(M.qtm_encode(np.frombuffer(d[:40000], dtype=np.uint8), 16)[0])
qo = (len(zs) + 15) & ~15
arena = np.zeros(qo + len(qs) + 96, dtype=np.uint8)
arena[:len(zs)] = np.frombuffer(zs, dtype=np.uint8); arena[qo:qo + len(qs)] = np.frombuffer(qs, dtype=np.uint8)
units, out_bytes = M.make_units([M.KIND_MSZIP, M.KIND_QUANTUM], [0, qo], [len(zs), len(qs)], [len(d), 40000], window_bits=[0, 16], out_slack=32768)
out, res = M.decode_batch(units, arena, out_bytes)
assert res["err"][0] == 0 and out[units["out_off"][0]:units["out_off"][0] + len(d)].tobytes() == d, res[0]
assert res["err"][1] == 0 and out[units["out_off"][1]:units["out_off"][1] + 40000].tobytes() == d[:40000], res[1]
print("EMU_OK")
'''


@pytest.mark.skipif(not os.path.exists(CLANG), reason="the emulator build needs ROCm's clang++")
def test_kernels_on_the_wavefront_emulator(built, tmp_path):
    srcs = [os.path.join(ROOT, "tests", "emu", f) for f in ("emu_runtime.cpp", "build_emu.sh", "include/hip/hip_runtime.h")]
    hip = os.path.join(ROOT, "libmspack_amd", "csrc", "hip")
    srcs += [os.path.join(hip, f) for f in os.listdir(hip) if f.endswith((".hpp", ".hip"))]
    if not os.path.exists(SO) or any(os.path.getmtime(s) > os.path.getmtime(SO) for s in srcs):
        subprocess.check_call(["bash", os.path.join(ROOT, "tests", "emu", "build_emu.sh")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    script = tmp_path / "w.py"
    script.write_text(WORKER % (ROOT, ROOT))
    # (parse waves pause after publishing partial progress: the commit tasks' path for frames still being parsed runs)
    env = dict(os.environ, MSPACK_HIP_SO=SO, MSPACK_EMU_PUBLISH_DELAY_US="3000")
    p = subprocess.run([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1500)
    assert p.returncode == 0 and b"EMU_OK" in p.stdout, p.stdout.decode()[-3000:]


@pytest.mark.skipif(not os.path.exists(CLANG), reason="the emulator build needs ROCm's clang++")
def test_some_gpu_parity_tests_on_the_emulator(built):
    """a few of the -m gpu parity tests, unchanged, against the emulator build (fresh pytest process: the ctypes mirror
    reads MSPACK_HIP_SO when it is imported): MSZIP folders with block tables through the deflate lane parser, LZX units
    with right and wrong frame tables and other token statistics through mspack_lzx_pipe.  (test_gpu_lzx.py,
    test_gpu_kat.py, test_gpu_lzx_frames.py and test_gpu_mszip_blocks.py pass on the emulator too, their 4096-unit batches
    aside -- tens of minutes; these are the ones that finish in a minute.)"""
    assert os.path.exists(SO), "built by test_kernels_on_the_wavefront_emulator"
    ids = ["tests/test_gpu_mszip_blocks.py::test_folders_with_tables",
           "tests/test_gpu_mszip_blocks.py::test_wrong_tables_and_odd_folders",
           "tests/test_gpu_lzx_frames.py::test_wrong_tables_cost_time_not_correctness",
           "tests/test_gpu_lzx_frames.py::test_frames_other_plaintexts",
           "tests/test_gpu_lzx_log.py",            # (the reset log of units whose blocks outlive their frames: serial and with tables)
           "tests/test_gpu_hostpath.py::test_xorsum_units_vs_oracle",               # (round 5: the CFDATA checksum kernel)
           "tests/test_gpu_lzx_frames.py::test_real_cabinet_blocks_of_megabytes"]   # (round 5: frames inside multi-frame blocks)
    env = dict(os.environ, MSPACK_HIP_SO=SO, MSPACK_EMU_PUBLISH_DELAY_US="500")
    p = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"] + ids, cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1700)
    assert p.returncode == 0, p.stdout.decode()[-3000:]


FOLD_WORKER = r'''
import sys, zlib
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
import numpy as np
import libmspack_amd as M
import test_gpu_lzx_frames as T
import test_gpu_fold as F
from helpers import oracle_mszip
assert "emu" in M.HIP_SO
# LZX: a CAB-style and a CHM-style unit of four frames and a tail, a damaged one, one with a wrong table -- all through mspack_lzx_fold
data = M.gen_plaintext(23, M.TEXT_MIX, 4 * 32768 + 777)
streams, params, tabs = [], [], []
for rf in (0, 2):
    comp, fo = M.lzx_encode(data, 21, rf)
    fo = fo.astype(np.int64)[:-1]
    streams.append(comp.tobytes()); params.append((data.size, 21, rf, 0)); tabs.append(fo)
bad = bytearray(streams[0]); bad[int(tabs[0][2]) + 300] ^= 0x08
streams += [bytes(bad), streams[0]]; params += [params[0], params[0]]; tabs += [tabs[0], tabs[0] + 2]
units, out, res = T.run(streams, params, tabs)
T.check(streams, params, units, out, res, compare_bytes=False)
for i in (0, 1, 3):
    assert res["err"][i] == 0 and np.array_equal(out[units["out_off"][i]:units["out_off"][i] + data.size], data), i
# MSZIP: a folder of four blocks with history, one of them damaged in a second copy -- through mspack_mszip_fold
n = 3 * 32768 + 5000
plain = M.gen_plaintext(29, M.TEXT_MIX, n)
blocks, prev = [], None
for k in range(0, n, 32768):
    b = plain[k:k + 32768].tobytes()
    c = zlib.compressobj(6, zlib.DEFLATED, -15, 9, 0, prev) if prev else zlib.compressobj(6, zlib.DEFLATED, -15)
    blocks.append(bytearray(b"CK" + c.compress(b) + c.flush())); prev = b
tab = np.cumsum([0] + [len(b) for b in blocks[:-1]])
s = b"".join(bytes(b) for b in blocks)
o, r = F.one_folder(M.KIND_MSZIP, s, tab, n, 0)
assert r["err"] == 0 and np.array_equal(o, plain)
blocks[2][len(blocks[2]) // 2] ^= 0x20
s = b"".join(bytes(b) for b in blocks)
o, r = F.one_folder(M.KIND_MSZIP, s, tab, n, 0)
e, oo, orc, _ = oracle_mszip(s, n)
assert r["err"] == e and r["out_len"] == orc.out_len and o[:orc.out_len].tobytes() == oo[:orc.out_len]
print("EMU_FOLD_OK")
'''


@pytest.mark.skipif(not os.path.exists(CLANG), reason="the emulator build needs ROCm's clang++")
def test_fold_tasks_on_the_emulator(built, tmp_path):
    """mspack_lzx_fold / mspack_mszip_fold (lzx_fold.hpp, fold_common.hpp) on the wavefront emulator: the records -> source map, R0-R2
    with placeholders and their chain, pointer jumping, the chain's three gather steps, the hand-over to the serial path where a
    frame is damaged -- with the tasks forced on (MSPACK_HIP_FOLD=2; the emulator runs a task on one wave)."""
    assert os.path.exists(SO), "built by test_kernels_on_the_wavefront_emulator"
    script = tmp_path / "w.py"
    script.write_text(FOLD_WORKER % (ROOT, ROOT))
    env = dict(os.environ, MSPACK_HIP_SO=SO, MSPACK_HIP_FOLD="2")
    p = subprocess.run([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1500)
    assert p.returncode == 0 and b"EMU_FOLD_OK" in p.stdout, p.stdout.decode()[-3000:]


STREAM_WORKER = r'''
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
import numpy as np
import libmspack_amd as M
import test_gpu_lzx_frames as T
assert "emu" in M.HIP_SO
# three units of three frames each (a uniform launch: 18 tickets for 32 waves), one of them with blocks that span frames, one damaged
data = [M.gen_plaintext(40 + i, M.TEXT_MIX, 3 * 32768) for i in range(3)]
streams, params, tabs = [], [], []
for i, d in enumerate(data):
    comp, fo = M.lzx_encode(d, 21, 0, M.lzx_opts(block_size=50000) if i == 2 else None)
    streams.append(comp.tobytes()); params.append((d.size, 21, 0, 0)); tabs.append(fo.astype(np.int64)[:-1])
units, out, res = T.run(streams, params, tabs)
T.check(streams, params, units, out, res, compare_bytes=False)
for i, d in enumerate(data):
    assert res["err"][i] == 0 and np.array_equal(out[units["out_off"][i]:units["out_off"][i] + d.size], d), i
    assert res["flags"][i] & T.ADOPTED, i
bad = bytearray(streams[1]); bad[int(tabs[1][1]) + 4000] ^= 0x40
streams[1] = bytes(bad)
units, out, res = T.run(streams, params, tabs)
T.check(streams, params, units, out, res, compare_bytes=False)
print("EMU_STREAM_OK")
'''


@pytest.mark.skipif(not os.path.exists(CLANG), reason="the emulator build needs ROCm's clang++")
def test_resolve_tasks_that_take_frames_up_while_they_are_parsed(built, tmp_path):
    """lzx_pipe_resolve_stream (round 6) on the wavefront emulator: a launch with a wave for every ticket (MSPACK_EMU_CUS=8: 32 waves),
    parse waves that pause behind every published pass -- the resolve tasks must be seen working on partial frames (the trace line), and
    results, flags and bytes must be the oracle's, with a damaged frame in the batch too."""
    assert os.path.exists(SO), "built by test_kernels_on_the_wavefront_emulator"
    script = tmp_path / "w.py"
    script.write_text(STREAM_WORKER % (ROOT, ROOT))
    env = dict(os.environ, MSPACK_HIP_SO=SO, MSPACK_EMU_CUS="8", MSPACK_EMU_PUBLISH_DELAY_US="3000", MSPACK_EMU_STREAM_TRACE="1")
    p = subprocess.run([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1500)
    assert p.returncode == 0 and b"EMU_STREAM_OK" in p.stdout, p.stdout.decode()[-3000:]
    assert b"records in while the frame is parsed" in p.stdout, p.stdout.decode()[-2000:]
