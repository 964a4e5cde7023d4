"""The host-buffer entry points of include/mspack_hip.h on the GPU: the chunked multi-stream pipeline
(mspack_hip_decode_batch), host input -> device output (mspack_hip_decode_batch_to_device) and the sharded
multi-device path (mspack_hip_decode_batch_multi; MSPACK_HIP_FORCE_SHARDS cuts the batch into shards even on
a one-GPU box, so the per-shard staging, the contiguous partition and the result scatter all execute).
Mixed batches (LZX + MSZIP + Quantum units in one call) exercise the per-codec compact launch lists; a batch
whose output regions are laid out in another order than its inputs exercises the per-unit copy-back.
Everything is compared with the CPU oracle and the plaintext."""
import os
import subprocess
import sys
import zlib

import numpy as np
import pytest

import libmspack_amd as M
from helpers import oracle_lzx, oracle_mszip, oracle_qtm, oracle_qtm_marks

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class DevBuf:
    """a device buffer through the HIP runtime the library itself is linked with (torch brings its own copy of the
    runtime, which cannot initialise in a process where /opt/rocm's already has)"""

    def __init__(self, nbytes):
        import ctypes as C
        M.lib()
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.hip.hipFree.argtypes = [C.c_void_p]
        self.n = nbytes
        p = C.c_void_p()
        assert self.hip.hipMalloc(C.byref(p), nbytes) == 0
        self.ptr = p.value
        assert self.hip.hipMemset(self.ptr, 0, nbytes) == 0

    def to_host(self):
        out = np.empty(self.n, dtype=np.uint8)
        assert self.hip.hipMemcpy(out.ctypes.data, self.ptr, self.n, 2) == 0          # hipMemcpyDeviceToHost
        return out

    def free(self):
        self.hip.hipFree(self.ptr)


def mixed_batch(n_each=40, seed=11):
    """-> (units, arena, out_bytes, expect[list of (kind, stream bytes, out_len, wb, plain)])"""
    rng = np.random.default_rng(seed)
    items = []
    for i in range(n_each):
        d = M.gen_plaintext(seed * 1000 + i, int(rng.integers(0, 4)), 65536)
        lz, _fo = M.lzx_encode(d, 21, 2)
        items.append((M.KIND_LZX, lz.tobytes() + b"\0" * 4, d.size, 21, 2, d))
        d2 = M.gen_plaintext(seed * 2000 + i, 0, 32768 * int(rng.integers(1, 4)))
        blocks, prev = [], None
        for k in range(0, d2.size, 32768):
            b = d2[k:k + 32768].tobytes()
            c = zlib.compressobj(6, zlib.DEFLATED, -15, 9, 0, prev) if prev else zlib.compressobj(6, zlib.DEFLATED, -15)
            blocks.append(b"CK" + c.compress(b) + c.flush()); prev = b
        items.append((M.KIND_MSZIP, b"".join(blocks), d2.size, 0, 0, d2))
        if i % 4 == 0:
            d3 = M.gen_plaintext(seed * 3000 + i, 0, 40000)
            qs, _fs = M.qtm_encode(d3, 17)
            items.append((M.KIND_QUANTUM, bytes(qs), d3.size, 17, 0, d3))
    order = rng.permutation(len(items))
    items = [items[i] for i in order]
    offs, pos = [], 0
    for it in items:
        pos = (pos + 15) & ~15
        offs.append(pos); pos += len(it[1])
    # the Quantum units carry marks (MSPACK_HIP_UF_QTM_MARKS, round 6): their tables lie behind all the streams -- another chunk's
    # part of the arena for most of them
    marks = {}
    for i, it in enumerate(items):
        if it[0] == M.KIND_QUANTUM:
            pos = (pos + 3) & ~3
            marks[i] = (pos, np.unique(rng.integers(1, it[2], 40)).astype(np.uint32)); pos += 4 * marks[i][1].size
    arena = np.zeros(pos + 64, dtype=np.uint8)
    for it, o in zip(items, offs):
        arena[o:o + len(it[1])] = np.frombuffer(it[1], dtype=np.uint8)
    for o, m in marks.values():
        arena[o:o + 4 * m.size] = m.view(np.uint8)
    kinds = np.array([it[0] for it in items], dtype=np.uint8)
    units, out_bytes = M.make_units(kinds, offs, [len(it[1]) for it in items], [it[2] for it in items],
                                    window_bits=[it[3] for it in items], reset_frames=[it[4] for it in items],
                                    out_slack=32768)
    for i, (o, m) in marks.items():
        units["flags"][i] |= M.UF_QTM_MARKS; units["in_chunk"][i] = o // 4; units["ref_len"][i] = m.size
    ARENA_OF[id(units)] = arena
    return units, arena, out_bytes, items


ARENA_OF = {}


def check(units, out, res, items):
    for i, (kind, stream, olen, wb, rf, plain) in enumerate(items):
        assert res["err"][i] == 0 and res["out_len"][i] == olen, (i, kind, res[i])
        o = int(units["out_off"][i])
        assert np.array_equal(out[o:o + olen], plain), (i, kind)
        if kind == M.KIND_QUANTUM and units["flags"][i] & M.UF_QTM_MARKS:
            # the unit's log behind its output (mspack_hip.h) against one oracle decode with the same marks -- read back out of the
            # arena the units point into
            m = int(units["ref_len"][i]); lo = o + ((olen + 15) & ~15)
            e, want = oracle_qtm_marks(stream, olen, wb, ARENA_OF[id(units)][int(units["in_chunk"][i]) * 4:][:4 * m].view(np.uint32))
            assert e == 0 and out[lo:lo + 4 * m].view(np.uint32).tolist() == want, (i, "marks")
    # the oracle on a sample of every kind
    seen = set()
    for i, (kind, stream, olen, wb, rf, plain) in enumerate(items):
        if kind in seen:
            continue
        seen.add(kind)
        if kind == M.KIND_LZX:
            e, o, r = oracle_lzx(stream, olen, wb, rf)
        elif kind == M.KIND_MSZIP:
            e, o, r, _ = oracle_mszip(stream, olen)
        else:
            e, o, r = oracle_qtm(stream, olen, wb)
        assert e == 0 and o == plain.tobytes()


def test_mixed_batch_pipeline(built):
    units, arena, out_bytes, items = mixed_batch()
    out, res = M.decode_batch(units, arena, out_bytes)
    check(units, out, res, items)


def test_mixed_batch_interleaved_outputs(built):
    """output regions in reverse order of the inputs: copy-back falls back to one copy per unit"""
    units, arena, out_bytes, items = mixed_batch(n_each=12, seed=5)
    rev = units["out_off"].copy()
    sizes = np.diff(np.concatenate([units["out_off"], [out_bytes]])).astype(np.int64)
    pos = 0
    for i in range(len(units) - 1, -1, -1):
        rev[i] = pos; pos += int(sizes[i])
    units["out_off"] = rev
    out, res = M.decode_batch(units, arena, out_bytes)
    check(units, out, res, items)


def test_to_device(built):
    units, arena, out_bytes, items = mixed_batch(n_each=16, seed=7)
    d_out = DevBuf(out_bytes + 64)
    res = np.zeros(len(units), dtype=M.RESULT_DTYPE)
    u = np.ascontiguousarray(units)
    rc = M.lib().mspack_hip_decode_batch_to_device(u.ctypes.data, len(u), arena.ctypes.data, arena.size, d_out.ptr,
                                                   out_bytes + 64, res.ctypes.data)
    assert rc == 0, M.lib().mspack_hip_last_error()
    check(units, d_out.to_host(), res, items)
    d_out.free()


WORKER = r'''
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np
import libmspack_amd as M
import test_gpu_hostpath as T
units, arena, out_bytes, items = T.mixed_batch(n_each=48, seed=int(sys.argv[1]))
out, res = M.decode_batch(units, arena, out_bytes, n_devices=max(2, M.lib().mspack_hip_device_count()))
T.check(units, out, res, items)
# a failing shard reports through the CALLER's last_error (the workers are other threads)
bad = units.copy(); bad["in_off"][3] = arena.size + 1000
try:
    M.decode_batch(bad, arena, out_bytes, n_devices=2)
    raise SystemExit("no error for a unit outside the arena")
except M.MspackHipError as e:
    assert "outside arena" in str(e), str(e)
print("SHARDS_OK")
'''


@pytest.mark.parametrize("shards", [2, 3, 7])
def test_multi_sharded_path(built, shards, tmp_path):
    """mspack_hip_decode_batch_multi through its sharded code path (sel != NULL), in a fresh process so that the
    environment switch is seen; on a multi-GPU box the shards really go to different devices."""
    script = tmp_path / "w.py"
    script.write_text(WORKER % (ROOT, ROOT))
    env = dict(os.environ, MSPACK_HIP_FORCE_SHARDS=str(shards))
    p = subprocess.run([sys.executable, str(script), str(shards)], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=900)
    assert p.returncode == 0 and b"SHARDS_OK" in p.stdout, p.stdout.decode()[-3000:]


CHUNK_WORKER = r'''
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np
import libmspack_amd as M
import test_gpu_hostpath as T
units, arena, out_bytes, items = T.mixed_batch(n_each=int(sys.argv[1]), seed=21)
for _ in range(2):                                   # (the second call reuses streams, events and arenas)
    out, res = M.decode_batch(units, arena, out_bytes)
    T.check(units, out, res, items)
d_out = T.DevBuf(out_bytes + 64)
res = np.zeros(len(units), dtype=M.RESULT_DTYPE)
u = np.ascontiguousarray(units)
rc = M.lib().mspack_hip_decode_batch_to_device(u.ctypes.data, len(u), arena.ctypes.data, arena.size, d_out.ptr, out_bytes + 64, res.ctypes.data)
assert rc == 0
T.check(units, d_out.to_host(), res, items)
# an output buffer the caller has pinned itself (hipHostMalloc): registering it is refused, the copies back are DMA anyway
import ctypes as C
hip = d_out.hip
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipHostFree.argtypes = [C.c_void_p]
hp = C.c_void_p()
assert hip.hipHostMalloc(C.byref(hp), out_bytes + 64, 0) == 0
pinned = np.frombuffer((C.c_ubyte * (out_bytes + 64)).from_address(hp.value), dtype=np.uint8)
pinned[:] = 0
res = np.zeros(len(units), dtype=M.RESULT_DTYPE)
rc = M.lib().mspack_hip_decode_batch(u.ctypes.data, len(u), arena.ctypes.data, arena.size, hp.value, out_bytes + 64, res.ctypes.data)
assert rc == 0
T.check(units, pinned, res, items)
del pinned
hip.hipHostFree(hp)
print("CHUNKS_OK")
'''


@pytest.mark.parametrize("nchunks,pin", [(2, "1"), (5, "1"), (8, "1"), (4, "0")])
def test_chunked_pipeline_small_chunks(built, nchunks, pin, tmp_path):
    """the copy-in / compute / copy-out streams with many small chunks of a mixed batch (thresholds lowered through
    the environment: every chunk holds LZX, MSZIP and Quantum units, concurrent LZX launches use their own control
    words and slot ranges); output into pageable memory that the call page-locks chunk by chunk, into pageable memory it
    must leave alone, into a buffer the caller pinned itself, and into device memory"""
    script = tmp_path / "w.py"
    script.write_text(CHUNK_WORKER % (ROOT, ROOT))
    env = dict(os.environ, MSPACK_HIP_NCHUNKS=str(nchunks), MSPACK_HIP_CHUNK_BYTES="4096", MSPACK_HIP_CHUNK_UNITS="4", MSPACK_HIP_TRACE="1",
               MSPACK_HIP_PIN_OUT=pin)            # "0": the caller's buffer is never page-locked (plain pageable copies back)
    p = subprocess.run([sys.executable, str(script), "24"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert p.returncode == 0 and b"CHUNKS_OK" in p.stdout, p.stdout.decode()[-3000:]
    assert (b"in %d chunks" % nchunks) in p.stdout, p.stdout.decode()[-3000:]


def test_headline_batch_host_entry_points(built):
    """the 4096-interval headline batch through both host entry points (what bench.py's host_inclusive times)"""
    n, ub = 4096, 65536
    plain, comp, off, ln = M.corpus_lzx_units(0xBA5E11, 0, n, ub, 21)
    units, out_bytes = M.make_units(M.KIND_LZX, off, ln + 4, np.full(n, ub), window_bits=21, reset_frames=2)
    out, res = M.decode_batch(units, comp, out_bytes)
    assert (res["err"] == 0).all() and np.array_equal(out[:n * ub], plain)
    d_out = DevBuf(out_bytes + 64)
    res2 = np.zeros(n, dtype=M.RESULT_DTYPE)
    u = np.ascontiguousarray(units)
    rc = M.lib().mspack_hip_decode_batch_to_device(u.ctypes.data, n, comp.ctypes.data, comp.size, d_out.ptr,
                                                   out_bytes + 64, res2.ctypes.data)
    assert rc == 0 and (res2["err"] == 0).all() and np.array_equal(d_out.to_host()[:n * ub], plain)
    d_out.free()
    M.lib().mspack_hip_release()
    out, res = M.decode_batch(units[:64], comp, out_bytes)                 # contexts come back after a release
    assert (res["err"] == 0).all() and np.array_equal(out[:64 * ub], plain[:64 * ub])


BIG_WORKER = r'''
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
import numpy as np
import libmspack_amd as M
from helpers import oracle_lzx
mode = sys.argv[1]
rng = np.random.default_rng(5)
if mode == "shard8":
    # BASELINE config 5's per-GPU shape and beyond: 16 384 CHM-style intervals of mixed length (1..3 frames), every one
    # with its frame table, cut into 8 shards (mspack_hip_decode_batch_multi; MSPACK_HIP_FORCE_SHARDS=8 on one device)
    parts = []
    for ub, n in ((65536, 8192), (32768, 4096), (98304, 4096)):
        plain, comp, off, ln, tab = M.corpus_lzx_units(0xC0F165 + ub, 0, n, ub, 21, frame_tables=True)
        parts.append((ub, n, plain, comp, off, ln, tab))
    arena = np.concatenate([p[3] for p in parts] + [np.zeros(64, np.uint8)])
    base = np.cumsum([0] + [p[3].size for p in parts])
    offs = np.concatenate([p[4].astype(np.int64) + base[i] for i, p in enumerate(parts)])
    lens = np.concatenate([p[5].astype(np.int64) + 4 for p in parts])
    tabs = np.concatenate([p[6].astype(np.int64) + base[i] for i, p in enumerate(parts)])
    outl = np.concatenate([np.full(p[1], p[0]) for p in parts])
    rf = np.concatenate([np.full(p[1], p[0] // 32768) for p in parts])
    perm = rng.permutation(len(offs))                 # (the unit table in any order; outputs follow the table)
    units, out_bytes = M.make_units(M.KIND_LZX, offs[perm], lens[perm], outl[perm], window_bits=21, reset_frames=rf[perm], frame_tabs=tabs[perm])
    out, res = M.decode_batch(units, arena, out_bytes, n_devices=8)
    assert (res["err"] == 0).all() and (res["out_len"] == outl[perm]).all()
    plain_all = np.concatenate([p[2] for p in parts]); pbase = np.cumsum([0] + [p[0] * p[1] for p in parts])
    src = np.concatenate([pbase[i] + np.arange(p[1], dtype=np.int64) * p[0] for i, p in enumerate(parts)])[perm]
    oo = units["out_off"].astype(np.int64)
    for k in range(len(perm)):
        assert np.array_equal(out[oo[k]:oo[k] + outl[perm][k]], plain_all[src[k]:src[k] + outl[perm][k]]), k
    assert ((res["flags"] & M.F_FRAMES_ADOPTED) != 0).all()
elif mode == "n8192":
    # one launch of 8192 intervals (config 5: 65 536 intervals over 8 GPUs): every byte against the plaintext, a sample
    # of units against the oracle's flags / in_next
    n, ub = 8192, 65536
    plain, comp, off, ln, tab = M.corpus_lzx_units(0xC0F165, 0, n, ub, 21, frame_tables=True)
    units, out_bytes = M.make_units(M.KIND_LZX, off, ln + 4, np.full(n, ub), window_bits=21, reset_frames=2, frame_tabs=tab)
    out, res = M.decode_batch(units, comp, out_bytes)
    assert (res["err"] == 0).all() and (res["out_len"] == ub).all() and np.array_equal(out[:n * ub], plain)
    for i in (0, 1, 4095, 4096, n - 1):
        e, o, r = oracle_lzx(comp[int(off[i]):int(off[i]) + int(ln[i]) + 4].tobytes(), ub, 21, 2)
        assert e == 0 and r.in_next == res["in_next"][i] and (int(res["flags"][i]) & ~M.F_FRAMES_ADOPTED) == r.flags
elif mode == "full65536":
    # BASELINE config 5 AT FULL SIZE on one device: the global list of 65 536 intervals (bench.py --scaling strong: seed
    # 0xC0F165, unit seeds follow the global index), cut into 8 contiguous shards by mspack_hip_decode_batch_multi
    # (MSPACK_HIP_FORCE_SHARDS=8) -- 4 GiB of output.  Every byte against the plaintext, two units of every shard against
    # the oracle's flags / in_next.
    n_sh, per, ub = 8, 8192, 65536
    parts = [M.corpus_lzx_units(0xC0F165, 0, per, ub, 21, first_unit=s * per, frame_tables=True) for s in range(n_sh)]
    sizes = [p[1].size for p in parts]
    base = np.cumsum([0] + sizes)
    arena = np.zeros(int(base[-1]) + 64, dtype=np.uint8)
    for s, p in enumerate(parts):
        arena[base[s]:base[s] + sizes[s]] = p[1]
    offs = np.concatenate([p[2].astype(np.int64) + base[s] for s, p in enumerate(parts)])
    lens = np.concatenate([p[3].astype(np.int64) + 4 for p in parts])
    tabs = np.concatenate([p[4].astype(np.int64) + base[s] for s, p in enumerate(parts)])
    n = n_sh * per
    units, out_bytes = M.make_units(M.KIND_LZX, offs, lens, np.full(n, ub), window_bits=21, reset_frames=2, frame_tabs=tabs)
    out, res = M.decode_batch(units, arena, out_bytes, n_devices=8)
    assert (res["err"] == 0).all() and (res["out_len"] == ub).all()
    assert ((res["flags"] & M.F_FRAMES_ADOPTED) != 0).all()
    for s, p in enumerate(parts):
        assert np.array_equal(out[s * per * ub:(s + 1) * per * ub], p[0]), s
        for i in (s * per, s * per + per - 1):
            e, o, r = oracle_lzx(arena[int(offs[i]):int(offs[i]) + int(lens[i])].tobytes(), ub, 21, 2)
            assert e == 0 and r.in_next == res["in_next"][i] and (int(res["flags"][i]) & ~M.F_FRAMES_ADOPTED) == r.flags, i
print("BIG_OK")
'''


@pytest.mark.parametrize("mode,env", [("n8192", {}), ("shard8", {"MSPACK_HIP_FORCE_SHARDS": "8"}),
                                      ("full65536", {"MSPACK_HIP_FORCE_SHARDS": "8"})])
def test_config5_shapes(built, mode, env, tmp_path):
    """BASELINE config 5's per-GPU shard (8192 intervals in one launch), the sharded entry point over 16 384 mixed
    units whose outputs do NOT ascend with their inputs (the shards then copy back unit by unit), and config 5 at full
    size: all 65 536 intervals through the sharded entry point with 8 forced shards on the one device."""
    script = tmp_path / "w.py"
    script.write_text(BIG_WORKER % (ROOT, ROOT))
    p = subprocess.run([sys.executable, str(script), mode], env=dict(os.environ, **env), stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=1500)
    assert p.returncode == 0 and b"BIG_OK" in p.stdout, p.stdout.decode()[-3000:]


LIFETIMES_WORKER = r'''
import hashlib, json, os, sys, zlib
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np
import libmspack_amd as M
from libmspack_amd import api
import chm_extract_recipe as R
V = [v for v in json.load(open(os.path.join(%r, "tests", "golden", "chm_extract.json"))) if v["tag"].startswith("config3")][0]
chm, _d, files = R.build(V["case"])
assert hashlib.md5(chm).hexdigest() == V["chm_md5"]
want = {idx: exp for idx, exp in zip(V["runs"][1]["order"], V["runs"][1]["results"])}       # what the REAL chmd answered
# a cabinet whose gather arena is > 4 MiB (page-locked by the driver): 24 MSZIP folders of 8 blocks of incompressible bytes
UB, fb, n = 32768, 8, 24
plain = np.random.default_rng(9).integers(0, 256, n * fb * UB, dtype=np.uint8)
folders, cfiles = [], []
for i in range(n):
    blocks = []
    for b in range(fb):
        co = zlib.compressobj(1, zlib.DEFLATED, -15)
        blocks.append(b"CK" + co.compress(plain[(i * fb + b) * UB:(i * fb + b + 1) * UB].tobytes()) + co.flush())
    folders.append((1, blocks, [UB] * fb))
    cfiles.append((b"g%%02d.bin" %% i, fb * UB, 0, i))
cab = M.cab_write(folders, cfiles)
assert len(cab) > (5 << 20)
last = M.lib().mspack_hip_last_error
for it in range(int(sys.argv[1])):
    with api.Chm(chm, mem=True) as c:
        for idx in (len(files) - 1, 0, len(files) // 2):                 # (the first call decodes the whole 1024-interval batch)
            err, data = c.extract(idx)
            assert err == want[idx]["err"] and hashlib.md5(data).hexdigest() == want[idx]["md5"], \
                ("chm lifetime", it, idx, err, c.mem.messages, last())
    with api.Cab(cab, mem=True) as c:
        for i in (n - 1, 0, it %% n):
            err, data = c.extract(i)
            assert err == 0 and data == plain[i * fb * UB:(i + 1) * fb * UB].tobytes(), ("cab lifetime", it, i, err, c.mem.messages, last())
    # heap traffic between lifetimes: blocks that land where the freed arenas were (what made round 4's fault intermittent)
    junk = [bytes(1000 + 37 * k) for k in range(200)]
    del junk
print("LIFETIMES_OK")
'''


def test_many_decompressors_one_process(built, tmp_path):
    """VERDICT round 4, item 1.  25 x (create -> open -> extract -> close -> destroy) of the config-3 CHM (24 MB arena) and of a
    cabinet with a 6 MB gather arena in ONE process, against what the real chmd answered (tests/golden/chm_extract.json) and
    the cabinet's plaintext.  Round 4 failed this from the second lifetime on, intermittently: the arena's page-lock was
    rounded outward and took in the allocator's neighbouring blocks (shim.hip: PinRange)."""
    script = tmp_path / "w.py"
    script.write_text(LIFETIMES_WORKER % (ROOT, ROOT, ROOT))
    p = subprocess.run([sys.executable, str(script), "25"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert p.returncode == 0 and b"LIFETIMES_OK" in p.stdout, p.stdout.decode()[-3000:]


def test_copies_are_cut_at_pin_boundaries(built):
    """The deterministic form of the same fault: a caller page-locks PART of its input arena and of its output buffer
    (mspack_hip_pin) -- every copy the entry points make then starts inside a locked range and ends behind it, or the other
    way round.  The runtime refuses such a copy (hipErrorInvalidValue); the entry points cut theirs at the boundaries.
    Runs on the hardware (256 units); the same scenario at the hardware's size runs in the CPU suite under real ASan / TSan against a
    model of the runtime's page-lock rules (tests/test_hostcheck.py) -- which replaced the emulator run of rounds 5 / 6 there (the
    emulator still takes it, smaller, when asked: MSPACK_HIP_SO=tests/_build/libmspack_emu.so).  History: in round 5 this test aborted the process ONCE in five whole-suite runs on the hardware and was
    moved off it; round 6 put it back (what was found, what was not; tests/conftest.py now keeps the native
    backtrace and the runtime's last words of anything that aborts)."""
    n, ub = (24 if "emu" in os.path.basename(M.HIP_SO) else 256), 65536     # (the emulator decodes ~1 MB/s: the cuts are what is tested, not the kernels)
    plain, comp, off, ln = M.corpus_lzx_units(0x9191, 0, n, ub, 21)
    units, out_bytes = M.make_units(M.KIND_LZX, off, ln + 4, np.full(n, ub), window_bits=21, reset_frames=2)
    L = M.lib()
    import ctypes as C
    L.mspack_hip_pin.argtypes = [C.c_void_p, C.c_size_t]
    L.mspack_hip_unpin.argtypes = [C.c_void_p]
    arena = np.zeros(comp.size + 8192, dtype=np.uint8)
    for shift, lo_frac, hi_frac in ((0, 0.0, 0.5), (100, 0.25, 0.75), (4000, 0.5, 1.0)):
        a = arena[shift:shift + comp.size]
        a[:] = comp
        out = np.zeros(out_bytes + 4096 + 64, dtype=np.uint8)[shift % 64:]
        p_in = a.ctypes.data + int(comp.size * lo_frac)
        p_out = out.ctypes.data + int(out_bytes * lo_frac)
        r_in = L.mspack_hip_pin(p_in, int(comp.size * (hi_frac - lo_frac)))
        r_out = L.mspack_hip_pin(p_out, int(out_bytes * (hi_frac - lo_frac)))
        try:
            res = np.zeros(n, dtype=M.RESULT_DTYPE)
            u = np.ascontiguousarray(units)
            rc = L.mspack_hip_decode_batch(u.ctypes.data, n, a.ctypes.data, a.size, out.ctypes.data, out_bytes + 64, res.ctypes.data)
            assert rc == 0, (shift, r_in, r_out, L.mspack_hip_last_error())
            assert (res["err"] == 0).all() and np.array_equal(out[:n * ub], plain), shift
        finally:
            L.mspack_hip_unpin(p_in)
            L.mspack_hip_unpin(p_out)


def test_xorsum_units_vs_oracle(built):
    """MSPACK_HIP_KIND_XORSUM (cabd_checksum, cabd.c:1462-1479) against the oracle's restatement: payloads at every byte alignment
    and of every length residue (the odd big-endian tail), empty and maximal blocks, alone and riding along a batch that decodes
    (checksum units own no output; the decode units' bytes must be untouched by them)."""
    from helpers import oracle_cab_checksum
    rng = np.random.default_rng(77)
    lens = [0, 1, 2, 3, 4, 5, 7, 8, 255, 256, 257, 4095, 32768 + 6144] + [int(x) for x in rng.integers(1, 40000, 300)]
    arena = rng.integers(0, 256, sum(lens) + 4 * len(lens) + 128, dtype=np.uint8)
    offs, pos = [], 0
    for i, ln in enumerate(lens):
        pos += i % 4                                   # (every alignment)
        offs.append(pos); pos += ln
    u = np.zeros(len(lens), dtype=M.UNIT_DTYPE)
    u["kind"] = 7; u["in_off"] = offs; u["in_len"] = lens
    out, res = M.decode_batch(u, arena, 64)
    for i, (o, ln) in enumerate(zip(offs, lens)):
        assert res["err"][i] == 0 and int(res["in_next"][i]) == oracle_cab_checksum(arena[o:o + ln].tobytes()), (i, o, ln)
    # the same units mixed into a decoding batch, unit table shuffled
    units, marena, out_bytes, items = mixed_batch(n_each=8, seed=3)
    big = np.concatenate([marena, arena])
    u2 = u.copy(); u2["in_off"] += marena.size
    allu = np.concatenate([units, u2])
    perm = rng.permutation(len(allu))
    out, res = M.decode_batch(allu[perm], big, out_bytes)
    inv = np.argsort(perm)
    check(units, out, res[inv][:len(units)], items)
    for i, (o, ln) in enumerate(zip(offs, lens)):
        r = res[inv][len(units) + i]
        assert r["err"] == 0 and int(r["in_next"]) == oracle_cab_checksum(arena[o:o + ln].tobytes()), (i, o, ln)


def test_staging_pool(built):
    """mspack_hip_stage_alloc / _free: page-locked blocks the library keeps and hands out again (the drivers' arenas of a MiB and
    more, when the caller's mspack_system allocates with the library's default allocator); a batch decodes out of and into them;
    a request beyond MSPACK_HIP_PINNED_MB gets NULL (the caller then allocates the ordinary way)."""
    import ctypes as C
    L = M.lib()
    L.mspack_hip_stage_alloc.restype = C.c_void_p
    L.mspack_hip_stage_alloc.argtypes = [C.c_size_t]
    L.mspack_hip_stage_free.argtypes = [C.c_void_p]
    n, ub = 128, 65536
    plain, comp, off, ln = M.corpus_lzx_units(0x5151, 0, n, ub, 21)
    units, out_bytes = M.make_units(M.KIND_LZX, off, ln + 4, np.full(n, ub), window_bits=21, reset_frames=2)
    p_in = L.mspack_hip_stage_alloc(comp.size + 64)
    p_out = L.mspack_hip_stage_alloc(out_bytes + 64)
    assert p_in and p_out and p_in % 4096 == 0 and p_out % 4096 == 0
    a = np.frombuffer((C.c_ubyte * (comp.size + 64)).from_address(p_in), dtype=np.uint8)
    o = np.frombuffer((C.c_ubyte * (out_bytes + 64)).from_address(p_out), dtype=np.uint8)
    a[:comp.size] = comp; a[comp.size:] = 0; o[:] = 0
    res = np.zeros(n, dtype=M.RESULT_DTYPE)
    u = np.ascontiguousarray(units)
    rc = L.mspack_hip_decode_batch(u.ctypes.data, n, p_in, comp.size + 64, p_out, out_bytes + 64, res.ctypes.data)
    assert rc == 0 and (res["err"] == 0).all() and np.array_equal(o[:n * ub], plain)
    del a, o
    L.mspack_hip_stage_free(p_in); L.mspack_hip_stage_free(p_out)
    again = L.mspack_hip_stage_alloc(comp.size + 64)              # the same block comes back
    assert again in (p_in, p_out)
    L.mspack_hip_stage_free(again)
    assert not L.mspack_hip_stage_alloc(1 << 44)                   # beyond any budget
    M.lib().mspack_hip_release()


JOB_WORKER = r"""
import sys, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np
import libmspack_amd as M
import test_gpu_hostpath as T
L = M.lib()
units, arena, out_bytes, items = T.mixed_batch(n_each=int(sys.argv[1]), seed=11)
n = len(units)
# the units in ARENA order are what the chunks are cut from: wait for them in that order, reading each one's bytes and result the
# moment its wait returns -- the later chunks are still being decoded and copied back then
by_arena = np.argsort(units["in_off"], kind="stable")
for rep in range(3):
    out = np.full(out_bytes + 64, 0xAB, dtype=np.uint8)
    res = np.zeros(n, dtype=M.RESULT_DTYPE); res["err"] = 77
    u = units.copy(); T.ARENA_OF[id(u)] = arena
    job = L.mspack_hip_decode_batch_begin(u.ctypes.data, n, arena.ctypes.data, arena.size, out.ctypes.data, out_bytes + 64, res.ctypes.data)
    assert job, "no job"
    early = 0
    for k, i in enumerate(by_arena if rep != 1 else by_arena[::-1]):
        i = int(i)
        assert L.mspack_hip_job_wait_unit(job, i) == 0, L.mspack_hip_last_error()
        kind, stream, olen, wb, rf, plain = items[i]
        assert res["err"][i] == 0 and res["out_len"][i] == olen, (rep, i, res[i])
        o = int(u["out_off"][i])
        assert np.array_equal(out[o:o + olen], plain), (rep, i)
        if rep == 0 and k < n // 4 and (res["err"][by_arena[-1]] == 77):
            early += 1                                    # (the last unit's result had not been written yet: a hand-over ahead of the end)
    assert L.mspack_hip_job_wait_unit(job, n) != 0
    assert L.mspack_hip_job_end(job) == 0
    T.check(u, out, res, items)
    print("rep", rep, "units handed over before the batch had ended:", early)
# a job that is ended without a wait; a synchronous call right behind a begin (waits for the job's batch)
out = np.zeros(out_bytes + 64, dtype=np.uint8); res = np.zeros(n, dtype=M.RESULT_DTYPE); u = units.copy(); T.ARENA_OF[id(u)] = arena
job = L.mspack_hip_decode_batch_begin(u.ctypes.data, n, arena.ctypes.data, arena.size, out.ctypes.data, out_bytes + 64, res.ctypes.data)
out2, res2 = M.decode_batch(units, arena, out_bytes)
assert job and L.mspack_hip_job_end(job) == 0
T.check(u, out, res, items); T.check(units, out2, res2, items)
# a batch that fails as a whole
u = units.copy(); u["in_off"][0] = arena.size + 9
job = L.mspack_hip_decode_batch_begin(u.ctypes.data, n, arena.ctypes.data, arena.size, out.ctypes.data, out_bytes + 64, res.ctypes.data)
assert job and L.mspack_hip_job_wait_unit(job, 1) != 0 and L.mspack_hip_job_end(job) != 0
print("JOBS_OK")
"""


@pytest.mark.gpu
@pytest.mark.parametrize("nchunks", [1, 4, 8])
def test_jobs_hand_over_chunk_by_chunk(built, nchunks, tmp_path):
    """mspack_hip_decode_batch_begin / _job_wait_unit / _job_end (include/mspack_hip.h): a mixed batch cut into 1, 4 and 8 chunks; every
    unit's bytes and result are read the moment its wait returns, in arena order and against it; same bytes as the synchronous call"""
    script = tmp_path / "j.py"
    script.write_text(JOB_WORKER % (ROOT, ROOT))
    env = dict(os.environ, MSPACK_HIP_NCHUNKS=str(nchunks), MSPACK_HIP_CHUNK_BYTES="4096", MSPACK_HIP_CHUNK_UNITS="4", MSPACK_HIP_TRACE="1")
    p = subprocess.run([sys.executable, str(script), "24"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert p.returncode == 0 and b"JOBS_OK" in p.stdout, p.stdout.decode()[-3000:]
    assert (b"in %d chunks" % nchunks) in p.stdout, p.stdout.decode()[-3000:]
