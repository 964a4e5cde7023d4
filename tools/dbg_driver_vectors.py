import sys, json, base64, hashlib
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import libmspack_amd as M
from libmspack_amd import api
VECS = json.load(open('tests/golden/driver_cabs.json'))
BASES = {v["tag"]: base64.b64decode(v["cab_b64"]) for v in VECS if "cab_b64" in v}
def cab_bytes(v):
    if "cab_b64" in v: return BASES[v["tag"]]
    b = bytearray(BASES[v["base"]]); m = v["mutation"]
    if "flip" in m: b[m["flip"][0]] ^= 1 << m["flip"][1]
    if "cut" in m: b = b[:m["cut"]]
    return bytes(b)
tags = sys.argv[1:]
for v in VECS:
    if v["tag"] not in tags: continue
    cab = cab_bytes(v); p = v["params"]
    print("==", v["tag"], p, "len", len(cab))
    for run in v["runs"]:
        with api.Cab(cab, fix_mszip=p.get("fix_mszip", 0), salvage=p.get("salvage", 0)) as c:
            row = []
            for idx, exp in zip(run["order"], run["results"]):
                err, data = c.extract(idx)
                ok = (err == exp["err"]) and (err != 0 or hashlib.md5(data).hexdigest() == exp["md5"])
                row.append("%d:%s got(e%d,n%d) exp(e%d,n%d)" % (idx, "ok" if ok else "BAD", err, len(data), exp["err"], exp["n"]))
            print("  order", run["order"]); print("   ", "\n    ".join(row))
