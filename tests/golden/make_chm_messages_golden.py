"""Makes tests/golden/chm_messages.json: what the REAL libmspack (oracle/_ref, development container) answers AND says through
sys->message when CHMs whose LZX stream has a block still open at a reset point are extracted -- lzxd.c:423-431,
"WARNING; invalid reset interval detected during LZX decompression", once per lzxd_decompress call that meets such a reset.
The CHMs are recipes (tests/chm_extract_recipe.py): a block header's 24-bit length field is raised by a few bytes, so the
block outlives its frame, the decoder warns at the next reset point, starts over there and goes on decoding correctly.
    python tests/golden/make_chm_messages_golden.py"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import helpers
import chm_extract_recipe as R

F = R.FRAME


def cases():
    out = []
    # every frame its own reset interval; the blocks of frames 1 and 3 claim 100 / 7 bytes too many.  A frame that starts
    # an interval begins with the 1-bit Intel header (lzxd.c:446-453), then 3 bits of block type, then the 24-bit length
    n = 6 * F
    out.append(dict(tag="interval1_two_open_blocks", seed=901, text=0, n_bytes=n, window_bits=16, reset_frames=1,
                    files=R.spread_files(n, 5, 11, F, pinned=(2 * F,)),
                    mutations=[["lzx_bits", 1, 4, 24, F + 100], ["lzx_bits", 3, 4, 24, F + 7]],
                    orders=[[0, 1, 2, 3, 4, 5], [5, 4, 3, 2, 1, 0], [3], [1, 4, 2], [2, 2, 5]]))
    # intervals of two frames: the second frame of interval 1 (frame 3: no Intel header bit in front) runs 50 bytes over
    n = 8 * F
    out.append(dict(tag="interval2_open_block", seed=902, text=1, n_bytes=n, window_bits=17, reset_frames=2,
                    files=R.spread_files(n, 6, 12, 2 * F, pinned=(4 * F,)),
                    mutations=[["lzx_bits", 3, 3, 24, F + 50]],
                    orders=[[0, 1, 2, 3, 4, 5, 6], [6, 0, 3], [4, 5], [2]]))
    # the same stream, undamaged: nothing is said
    out.append(dict(tag="interval2_clean", seed=902, text=1, n_bytes=n, window_bits=17, reset_frames=2,
                    files=R.spread_files(n, 6, 12, 2 * F, pinned=(4 * F,)), mutations=[],
                    orders=[[0, 1, 2, 3, 4, 5, 6], [6, 0, 3]]))
    # the LAST frame's block stays open: the warning comes from the look-ahead frame behind the end of the stream
    n = 4 * F
    out.append(dict(tag="interval1_last_block_open", seed=903, text=0, n_bytes=n, window_bits=16, reset_frames=1,
                    files=R.spread_files(n, 3, 13, F), mutations=[["lzx_bits", 3, 4, 24, F + 9]],
                    orders=[[0, 1, 2], [2], [2, 0]]))
    return out


def main():
    assert helpers.have_ref(), "needs oracle/_ref (the reference built in the development container)"
    gold = []
    for case in cases():
        orders = case.pop("orders")
        chm, _d, files = R.build(case)
        v = dict(tag=case["tag"], case=case, chm_md5=hashlib.md5(chm).hexdigest(), runs=[])
        for order in orders:
            helpers.ref_messages()
            rc, res = helpers.ref_chm_extract(chm, order)
            assert rc == 0, (case["tag"], rc)
            lines = helpers.ref_messages()
            v["runs"].append(dict(order=order, messages=lines,
                                  results=[dict(err=e, n=len(b), md5=hashlib.md5(b).hexdigest()) for e, b in res]))
            print(case["tag"], order, [e for e, _b in res], [l for l in lines if not l.startswith("#")])
        gold.append(v)
    json.dump(gold, open(os.path.join(HERE, "chm_messages.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
