"""scratch: the C drivers' own share of an extract-everything run, timed on the CPU with a batch ABI that decodes nothing
(tools/csrc/null_batch.c).  python tools/hostpath_cpu_profile.py [config2|config4] [reps]"""
import ctypes as C, glob, os, subprocess, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import libmspack_amd as M
from libmspack_amd import apibench
which = sys.argv[1] if len(sys.argv) > 1 else "config2"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
so = os.path.join(R, "build", "variants", "libapibench_null.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
srcs = [os.path.join(R, "libmspack_amd/csrc/bench/api_bench.c")] + sorted(glob.glob(os.path.join(R, "libmspack_amd/csrc/host/*.c"))) + \
       [os.path.join(R, "tools/csrc/null_batch.c")]
subprocess.check_call(["gcc", "-O2", "-g", "-fPIC", "-shared", "-Wall", "-I", os.path.join(R, "include"), "-o", so] + srcs + ["-lpthread"] +
                      os.environ.get("XCFLAGS", "").split())
L = C.CDLL(so)
if which == "config2":
    n = 4096
    cache = "/tmp/config2_cab.npy"
    if os.path.exists(cache): image = np.load(cache)
    else:
        image, plain = apibench.build_config2_cab(M, n); image = np.frombuffer(bytes(image), dtype=np.uint8); np.save(cache, image)
    cap = n * 32768
else:
    n = 512
    cache = "/tmp/config4_cab.npy"
    if os.path.exists(cache): image = np.load(cache)
    else:
        image, plain = apibench.build_config4_cab(M, n); image = np.frombuffer(bytes(image), dtype=np.uint8); np.save(cache, image)
    cap = n * 32 * 32768
fn = L.mspk_api_bench_cab
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint, C.POINTER(apibench.Stats)]
out = np.zeros(cap + 64, dtype=np.uint8); offs = np.zeros(70000, dtype=np.uint64)
for r in range(reps):
    st = apibench.Stats()
    rc = fn(image.ctypes.data, image.size, out.ctypes.data, cap, offs.ctypes.data, 70000, C.byref(st))
    d = {k: getattr(st, k) for k, _t in apibench.Stats._fields_}
    s = apibench.summary(d)
    print(rc, s["seconds"], s["split_ms"])
