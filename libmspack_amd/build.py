"""Build the native pieces in-tree (the .so files travel to the GPU box with the repo snapshot).

  libmspack_amd/libmspack_hip.so     hipcc --offload-arch=gfx950: kernels + C ABI (+ C host drivers)
  libmspack_amd/libmspack_corpus.so  gcc: synthetic corpus generators (test/bench infrastructure)
  libmspack_amd/libmspack_apibench.so gcc: the object API driven by a C in-memory mspack_system (bench/test infrastructure)
  oracle/liboracle.so                gcc: CPU restatement (test infrastructure)
  oracle/_ref/*.so                   gcc on /root/reference sources, only where they exist
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
HIP_SO = os.path.join(HERE, "libmspack_hip.so")
CORPUS_SO = os.path.join(HERE, "libmspack_corpus.so")


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _walk(d, exts):
    out = []
    for base, _dirs, files in os.walk(d):
        out += [os.path.join(base, f) for f in files if f.endswith(exts)]
    return sorted(out)


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def hipcc():
    for c in ("/opt/rocm/bin/hipcc", "hipcc"):
        if os.path.exists(c) or c == "hipcc":
            return c


def build_hip(force=False):
    hip_dir = os.path.join(CSRC, "hip")
    host_dir = os.path.join(CSRC, "host")
    srcs = _walk(hip_dir, (".hip", ".hpp")) + _walk(host_dir, (".c", ".h")) + \
        _walk(os.path.join(ROOT, "include"), (".h",)) + [os.path.join(CSRC, "exports.map")]
    if not (force or _newer(HIP_SO, srcs)):
        return HIP_SO
    objs = []
    for c in _walk(host_dir, (".c",)):
        o = c[:-2] + ".o"
        _run(["gcc", "-O2", "-fPIC", "-Wall", "-I", os.path.join(ROOT, "include"), "-c", c, "-o", o])
        objs.append(o)
    shim_o = os.path.join(hip_dir, "shim.o")
    _run([hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value",
          "-I", os.path.join(ROOT, "include"), "-c", os.path.join(hip_dir, "shim.hip"), "-o", shim_o])
    _run([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", HIP_SO, shim_o] + objs +
         ["-lpthread", "-Wl,--version-script=" + os.path.join(CSRC, "exports.map")])
    return HIP_SO


def build_corpus(force=False):
    d = os.path.join(CSRC, "corpus")
    srcs = _walk(d, (".c", ".h"))
    if force or _newer(CORPUS_SO, srcs):
        _run(["gcc", "-O2", "-fPIC", "-shared", "-Wall", "-o", CORPUS_SO] + _walk(d, (".c",)) + ["-lpthread", "-lm"])
    return CORPUS_SO


APIBENCH_SO = os.path.join(HERE, "libmspack_apibench.so")


def build_apibench(force=False):
    """csrc/bench/api_bench.c: the object API timed with a C in-memory mspack_system (bench / test infrastructure)"""
    src = os.path.join(CSRC, "bench", "api_bench.c")
    if force or _newer(APIBENCH_SO, [src, HIP_SO, os.path.join(ROOT, "include", "mspack.h"), os.path.join(ROOT, "include", "mspack_hip.h")]):
        _run(["gcc", "-O2", "-fPIC", "-shared", "-Wall", "-I", os.path.join(ROOT, "include"), "-o", APIBENCH_SO, src,
              "-L", HERE, "-l:libmspack_hip.so", "-Wl,-rpath,$ORIGIN"])
    return APIBENCH_SO


def build_oracle():
    _run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"])
    if os.path.isdir("/root/reference/libmspack/mspack"):
        _run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])


def build_reftests():
    """the reference's own cabd_test / chmd_test programs against our header + library (checker only)"""
    if os.path.isdir("/root/reference/libmspack/test") and os.path.exists(HIP_SO):
        _run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "reftests"])


def build_all(force=False):
    build_corpus(force)
    build_oracle()
    build_hip(force)
    build_apibench(force)
    build_reftests()


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
