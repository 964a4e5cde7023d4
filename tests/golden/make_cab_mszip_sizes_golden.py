"""Makes tests/golden/cab_mszip_sizes.json (development container, oracle/_ref): what the REAL cabd answers for cabinets whose CFDATA
headers lie about a block's uncompressed size.  mszipd never reads that field -- a block is as long as its deflate stream
(mszipd.c:377-460) -- so a file that reaches beyond the headers' sum still gets its bytes when the last block really holds them
(found by tools/fuzz_drivers_cpu.py, seed 3 case 94).  Cabinets: one MSZIP folder (zlib level 6, history), two files
that tile the REAL length; the last block's field short by `short` bytes, or long by `long`.
    python tests/golden/make_cab_mszip_sizes_golden.py"""
import hashlib
import json
import os
import sys
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import helpers
import libmspack_amd as M

CASES = [dict(seed=81, n=70000, cut=1234, delta=-1), dict(seed=82, n=70000, cut=40000, delta=-1000), dict(seed=83, n=40000, cut=33000, delta=-7000),
         dict(seed=84, n=70000, cut=1234, delta=+500), dict(seed=85, n=32768 * 2, cut=32768, delta=-32767), dict(seed=86, n=70000, cut=69999, delta=-3)]


def build(c):
    data = M.gen_plaintext(c["seed"], 0, c["n"])
    mb, mu, prev = [], [], None
    for k in range(0, c["n"], 32768):
        b = data[k:k + 32768].tobytes()
        z = zlib.compressobj(6, zlib.DEFLATED, -15, 9, 0, prev) if prev else zlib.compressobj(6, zlib.DEFLATED, -15)
        mb.append(b"CK" + z.compress(b) + z.flush()); mu.append(len(b)); prev = b
    mu[-1] += c["delta"]                                   # the last CFDATA header lies
    files = [(b"a.bin", c["cut"], 0, 0), (b"b.bin", c["n"] - c["cut"], c["cut"], 0)]
    return bytes(M.cab_write([(1, mb, mu)], files)), data


def main():
    assert helpers.have_ref()
    helpers.ref().refh_zero_alloc(1)
    gold = []
    for c in CASES:
        cab, data = build(c)
        g = dict(case=c, cab_md5=hashlib.md5(cab).hexdigest(), runs=[])
        for salvage in (0, 1):
            for order in ([0, 1], [1, 0], [1]):
                rc, res = helpers.ref_cab_extract(cab, order, cap=len(order) * 160000 + 4096, salvage=salvage)
                assert rc == 0
                g["runs"].append(dict(salvage=salvage, order=order, results=[dict(err=e, n=len(d), md5=hashlib.md5(d).hexdigest()) for e, d in res]))
                print(c, salvage, order, [(e, len(d)) for e, d in res])
        gold.append(g)
    json.dump(gold, open(os.path.join(HERE, "cab_mszip_sizes.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
