"""Copies the DATA files of the reference's regression suites (libmspack/test/test_files/{cabd,chmd,kwajd}: small
cabinets, CHMs and KWAJ headers, incl. the must-fail CVE files) to tests/golden/ref_fixtures/, where the reference's own
test programs -- built against our library by `make -C oracle reftests` -- look for them.  Development
container only.  Scripts that generated those files upstream (*.pl) are not copied."""
import os
import shutil

SRC = "/root/reference/libmspack/test/test_files"
HERE = os.path.dirname(os.path.abspath(__file__))
for sub, exts in (("cabd", (".cab",)), ("chmd", (".chm", ".xor")), ("kwajd", (".kwj",))):
    dst = os.path.join(HERE, "ref_fixtures", sub)
    os.makedirs(dst, exist_ok=True)
    n = 0
    for f in sorted(os.listdir(os.path.join(SRC, sub))):
        if f.endswith(exts):
            shutil.copyfile(os.path.join(SRC, sub, f), os.path.join(dst, f)); os.chmod(os.path.join(dst, f), 0o644); n += 1
    print(sub, n, "files")
