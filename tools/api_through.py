"""bench.py's through_api for chosen configs: python tools/api_through.py [2 3 4] -- the object API from C on the container
(libmspack_amd/csrc/bench/api_bench.c), best of 3, with the time split.  MSPACK_ARENA_HUGEPAGES=0 for the comparison."""
import json, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import libmspack_amd as M
from libmspack_amd import apibench as A
which = [int(x) for x in sys.argv[1:]] or [2, 3]
def measure(kind, image, plain, reps=3):
    runs = []
    for k in range(reps + 1):
        rc, o, offs, d = A.run(kind, image, plain.size)
        assert rc == 0 and d["n_errors"] == 0
        if k: runs.append(d)
    assert o.size == plain.size
    return A.summary(min(runs, key=lambda d: d["total_s"]))
for c in which:
    if c == 2:
        img, plain = A.build_config2_cab(M); s = measure("cab", img, plain)
    elif c == 3:
        img, plain, _sl = A.build_config3_chm(M); s = measure("chm", img, plain)
    else:
        img, plain = A.build_config4_cab(M); s = measure("cab", img, plain)
    print("config %d hugepages=%s: %.1f MB/s (first extract() %.2f ms of %.2f)  %s" % (c, os.environ.get("MSPACK_ARENA_HUGEPAGES", "1"), s["MBps"], s["first_extract_ms"], s["seconds"] * 1e3, json.dumps(s["split_ms"])))
