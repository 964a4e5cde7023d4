"""The reference's OWN regression programs (libmspack/test/cabd_test.c, chmd_test.c and kwajd_test.c), compiled unchanged
against include/mspack.h and linked with libmspack_hip.so (`make -C oracle reftests`, development container;
the binaries travel to the GPU box in oracle/_ref/), run against the reference's own data files
(tests/golden/ref_fixtures).  Every TEST() of the suites must pass -- the drop-in claim, checked by the
reference's tests."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(name):
    exe = os.path.join(ROOT, "oracle", "_ref", name)
    # never skip: a missing binary on the GPU box would silently drop the strongest boundary evidence
    assert os.path.exists(exe), "%s missing: run `make -C oracle reftests` in the development container " \
        "(build() does) so that oracle/_ref/ travels to the GPU box" % name
    p = subprocess.run([exe], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    out = p.stdout.decode("latin-1")
    ok = out.count(" SUCCESS ")
    assert " FAILED " not in out, [l for l in out.splitlines() if " FAILED " in l][:5]
    assert p.returncode == 0, (p.returncode, out[-2000:], p.stderr.decode("latin-1")[-2000:])
    return ok, out


def test_reference_cabd_test_suite(built):
    ok, out = _run("cabd_test_hip")
    assert ok >= 400 and "ALL %d TESTS PASSED" % ok in out        # cabd_test.c prints its own tally


def test_reference_chmd_test_suite(built):
    ok, out = _run("chmd_test_hip")
    assert ok >= 150 and "ALL %d TESTS PASSED" % ok in out


def test_reference_kwajd_test_suite(built):
    ok, out = _run("kwajd_test_hip")
    assert ok >= 100 and "ALL %d TESTS PASSED" % ok in out
