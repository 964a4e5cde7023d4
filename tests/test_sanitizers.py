"""ASan + UBSan over the C the product ships on the host side and over the oracle (the reference's own CI runs its
tests under sanitizers; here the HIP kernels get the wavefront emulator + TSan hooks, tests/emu, and THIS covers the plain C).
tests/_build/libhostlogic_asan.so = libmspack_amd/csrc/host/*.c + the CPU stand-in for the batch ABI + the oracle, compiled with
-fsanitize=address,undefined; a worker process (the sanitizer runtime has to be loaded before python's allocator runs: LD_PRELOAD)
replays driver goldens that walk the drivers' hairy paths -- cabinets with damaged blocks and moved file offsets (several call
orders, salvage on and off: sticky errors, checksum units and the re-gather behind a bad one), the split cabinet sets (blocks
reassembled across cabinets, chains that run out of cabinets), CHMs with lying headers and damaged content, the CHM directory
fixtures (the reference's own fuzz finds), OAB files and patches and SZDD / KWAJ files with their damaged copies.  Any report fails the
test.  CPU only."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tests", "_build", "libhostlogic_asan.so")

WORKER = r'''
import ctypes, json, os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
L = ctypes.CDLL(%(so)r)
import test_cab_sticky as S, test_chm_extract as X, test_cabsets as CS, test_chmdir as D, test_oab as O, test_szdd_kwaj as Z
from libmspack_amd import api
n = 0
for v in S.GOLD:
    S.replay(v, L=L); n += 1
for v in [v for v in X.VECS if v["case"]["n_bytes"] <= (1 << 20)]:
    X.replay(v, L=L); n += 1
for sc in CS.CODED:
    assert CS._extract_all(sc, L=L) == [(f["name"], f["err"], f["out_len"], f["md5"]) for f in sc["files"]]; n += 1
for fx in D.G["fixtures"]:
    path = os.path.join(D.HERE, "golden", "chmdir", fx["file"])
    with api.Chm(path, L=L) as c:
        assert c.open_error == fx["open_err"]
        _ = c.files
    with api.Chm(path, fast=True, L=L) as c:
        if not c.open_error:
            D._finds(c, [q[0].encode("latin-1") for q in fx["finds"]])
    n += 1
for name in sorted(O.G):                       # OAB files and patches, damaged and truncated (oabd.c)
    O.check_case(name, L=L); n += 1
for name in Z.CPU_FILES:                       # SZDD / KWAJ files, damaged (szdd_kwaj.c)
    Z.check_file(name, L=L); n += 1
print("SANITIZED_OK", n)
'''


def _build():
    srcs = sorted(glob.glob(os.path.join(ROOT, "libmspack_amd", "csrc", "host", "*.c"))) + \
        [os.path.join(ROOT, "tests", "csrc", "batch_standin.c")] + sorted(glob.glob(os.path.join(ROOT, "oracle", "*_oracle.c")))
    deps = srcs + glob.glob(os.path.join(ROOT, "include", "*.h")) + glob.glob(os.path.join(ROOT, "oracle", "*.h"))
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    if not os.path.exists(SO) or any(os.path.getmtime(s) > os.path.getmtime(SO) for s in deps):
        subprocess.check_call(["gcc", "-O1", "-g", "-fPIC", "-shared", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                               "-fno-sanitize-recover=undefined", "-Wall", "-Wno-unused-function",
                               "-I", os.path.join(ROOT, "include"), "-o", SO] + srcs + ["-lpthread"])


def test_host_drivers_and_oracle_under_asan_ubsan(built, tmp_path):
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], stdout=subprocess.PIPE).stdout.decode().strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("this gcc has no AddressSanitizer runtime")
    _build()
    script = tmp_path / "w.py"
    script.write_text(WORKER % dict(root=ROOT, so=SO))
    env = dict(os.environ, LD_PRELOAD=os.path.realpath(asan),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=97:allocator_may_return_null=1", UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1500)
    out = p.stdout.decode(errors="replace")
    assert p.returncode == 0 and "SANITIZED_OK" in out, out[-4000:]
    assert "AddressSanitizer" not in out and "runtime error" not in out, out[-4000:]
