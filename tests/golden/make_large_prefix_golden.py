"""Golden vectors for tests/test_gpu_large_files.py::test_large_files_prefix (run in the development container, where the
reference builds: oracle/_ref).  cabextract's large-files.cab (three 2 GiB folders: MSZIP, LZX-15, LZX-21) is decoded from the
KAT stream in kat_folders.json, its folders are cut to their first N CFDATA blocks (CFFOLDER.cCFData and CFFILE.cbFile patched
-- helpers.cab_cut_folders), and the REAL libmspack cabd extracts the three files: their MD5s are the expected values.
  python tests/golden/make_large_prefix_golden.py  ->  tests/golden/large_prefix.json"""
import base64, hashlib, json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import helpers as H

N_BLOCKS = 4096
k = [v for v in json.load(open(os.path.join(HERE, "kat_folders.json"))) if v["source"].endswith("large-files-cab.cab")][0]
err, inner, _ = H.ref_lzx(base64.b64decode(k["stream_b64"]), k["out_len"], k["window_bits"])
assert err == 0 and hashlib.md5(inner).hexdigest() == k["ref_md5"]
cut = H.cab_cut_folders(inner, N_BLOCKS)
rc, got = H.ref_cab_extract(cut, [0, 1, 2], cap=3 * N_BLOCKS * 32768 + 4096)
assert rc == 0 and [e for e, _ in got] == [0, 0, 0], (rc, [e for e, _ in got])
outs = [o for _, o in got]
assert all(len(o) == N_BLOCKS * 32768 for o in outs)
doc = {"source": "cabextract/test/cabs/large-files-cab.cab -> large-files.cab, every folder cut to its first %d CFDATA blocks" % N_BLOCKS,
       "made_by": "tests/golden/make_large_prefix_golden.py (libmspack cabd from oracle/_ref)",
       "n_blocks": N_BLOCKS, "length": N_BLOCKS * 32768,
       "files": [{"index": i, "md5": hashlib.md5(o).hexdigest(), "bytes": len(o)} for i, o in enumerate(outs)]}
json.dump(doc, open(os.path.join(HERE, "large_prefix.json"), "w"), indent=1)
print(doc)
