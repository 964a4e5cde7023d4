"""Golden vectors for tests/test_gpu_messages.py (run in the development container: needs oracle/_ref).
A cabinet with one MSZIP folder of 12 CFDATA blocks (the window carries over from block to block) and four files, damaged
copies of it, extracted by the REAL libmspack cabd with MSCABD_PARAM_FIXMSZIP in several orders: per extract() call the error
code, the MD5 of what was written and the lines the library said through sys->message -- "MSZIP error, %u bytes of data
lost." per repaired block (mszipd.c:427), said again when the decompressor starts the folder over.
  python tests/golden/make_mszip_messages_golden.py -> tests/golden/mszip_messages.json"""
import base64, hashlib, json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import numpy as np
import libmspack_amd as M
import helpers as H
from test_gpu_mszip_blocks import folder_blocks

NB = 12
data = M.gen_plaintext(0xFEED, 0, NB * 32768 - 5000).tobytes()
stream, offs = folder_blocks(data, 6, history=True)
offs = list(offs) + [len(stream)]
blocks = [stream[offs[i]:offs[i + 1]] for i in range(NB)]
usz = [min(32768, len(data) - i * 32768) for i in range(NB)]
cuts = [0, 40000, 150000, 290000, len(data)]
files = [(b"part%d.bin" % k, cuts[k + 1] - cuts[k], cuts[k], 0) for k in range(4)]
cab = M.cab_write([(1, blocks, usz)], files)
# where the blocks' payloads sit in the image: a CFDATA header is 8 bytes, the folder's first block follows the file entries
pos = cab.index(blocks[0][:16]) - 8
starts = []
for b in blocks:
    starts.append(pos + 8); pos += 8 + len(b)

def damaged(spec):
    b = bytearray(cab)
    for blk, rel, bit in spec:
        b[starts[blk] + rel] ^= 1 << bit
    return bytes(b)

cases = []
for tag, spec in (("clean", []), ("block3", [(3, 700, 2)]), ("blocks_2_7", [(2, 300, 5), (7, 1500, 0)]),
                  ("first_and_last", [(0, 50, 1), (11, 200, 7)]), ("ck_signature_5", [(5, 0, 3)])):
    img = damaged(spec)
    runs = []
    for order in ([0, 1, 2, 3], [3, 0], [2, 2, 1], [1, 3, 2]):
        H.ref_messages()
        rc, got = H.ref_cab_extract(img, order, cap=len(data) * len(order) + 4096, fix_mszip=1)
        assert rc == 0, (tag, rc)
        msgs = H.ref_messages()
        runs.append({"order": order, "results": [{"err": e, "n": len(d), "md5": hashlib.md5(d).hexdigest()} for e, d in got], "messages": msgs})
    cases.append({"tag": tag, "mutations": [list(x) for x in spec], "runs": runs})
    print(tag, [r["messages"] for r in runs][0])
doc = {"made_by": "tests/golden/make_mszip_messages_golden.py (libmspack cabd from oracle/_ref, MSCABD_PARAM_FIXMSZIP = 1)",
       "cab_b64": base64.b64encode(cab).decode(), "block_payload_offsets": starts, "cases": cases}
json.dump(doc, open(os.path.join(HERE, "mszip_messages.json"), "w"), indent=1)
