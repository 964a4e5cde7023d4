"""Makes tests/golden/cab_sticky.json (development container, oracle/_ref): what the REAL cabd answers when extract() calls follow a
FAILED call in the same folder.  The reference keeps one decompressor alive while the files' offsets ascend (cabd.c:1136-1175); after
a failure its codec repeats its error for every further call and writes nothing, until a file's offset lies below what it has written
so far and the folder is started over.  A four-folder cabinet (MSZIP, LZX, Quantum, stored; tests/cab_recipe.py) whose
file table is edited: one file's offset is moved far beyond its folder -- in salvage mode the skip to it "succeeds" with nothing
written (out of blocks reads as MSPACK_ERR_OK, cabd.c:1339-1345) and leaves the decompressor in its error state.
    python tests/golden/make_cab_sticky_golden.py"""
import hashlib
import json
import os
import struct
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import helpers
import cab_recipe as F


def main():
    assert helpers.have_ref()
    helpers.ref().refh_zero_alloc(1)
    gold = []
    # the file whose offset is damaged: Quantum / MSZIP / LZX folder; and the Quantum folder again with its second file starting in the
    # folder's LAST window (qtmd writes when its 32 KiB window wraps: that file lies above everything the failed call wrote)
    for seed, victim, cut in ((7000, 4, None), (7025, 0, None), (7050, 2, None), (7075, 4, "last_window")):
        cab = F.base_cab(seed, cut)
        ents = F.file_entry_offsets(cab)
        struct.pack_into("<I", cab, ents[victim] + 4, 4521984)     # uoffFolderStart
        cab = bytes(cab)
        v = dict(seed=seed, victim=victim, cut=cut, cab_md5=hashlib.md5(cab).hexdigest(), runs=[])
        for salvage in (1, 0):
            for order in ([victim, victim + 1], [victim + 1, victim, victim + 1], [victim, victim, victim + 1], [victim + 1, victim],
                          [victim, 7, victim + 1], list(range(8)), list(range(7, -1, -1))):
                rc, res = helpers.ref_cab_extract(cab, order, cap=len(order) * 160000 + 4096, salvage=salvage)
                assert rc == 0
                v["runs"].append(dict(salvage=salvage, order=order,
                                      results=[dict(err=e, n=len(b), md5=hashlib.md5(b).hexdigest()) for e, b in res]))
                print(seed, salvage, order, [(e, len(b)) for e, b in res])
        gold.append(v)
    # a damaged CFDATA block in the MIDDLE of a folder (bad checksum): the file behind it fails after its skip succeeded, the
    # decompressor has then written up to that file's offset (+ what it flushed), and the file in FRONT of it starts the folder over
    for seed, folder in ((7100, 2), (7125, 1), (7150, 0), (7214, 2), (7225, 1), (7230, 0)):   # Quantum, LZX, MSZIP; the last three: the first file lies in front of the damage
        cab = F.base_cab(seed)
        coff, = struct.unpack_from("<I", cab, 36 + 8 * folder)
        cb0, = struct.unpack_from("<H", cab, coff + 4)
        flip = coff + 8 + cb0 + 8 + 20                                    # byte 20 of the folder's second block
        cab[flip] ^= 0x10
        cab = bytes(cab)
        a, b = 2 * folder, 2 * folder + 1
        v = dict(seed=seed, victim=None, cut=None, flip=flip, cab_md5=hashlib.md5(cab).hexdigest(), runs=[])
        for salvage in (0, 1):
            for order in ([b, a], [a, b], [b, b, a], [b, a, b], list(range(8)), list(range(7, -1, -1))):
                rc, res = helpers.ref_cab_extract(cab, order, cap=len(order) * 160000 + 4096, salvage=salvage)
                assert rc == 0
                v["runs"].append(dict(salvage=salvage, order=order,
                                      results=[dict(err=e, n=len(bb), md5=hashlib.md5(bb).hexdigest()) for e, bb in res]))
                print(seed, salvage, order, [(e, len(bb)) for e, bb in res])
        gold.append(v)
    # a flipped bit early in the Quantum folder's FIRST block (fuzz seed 4, case 5): the stream runs on, wrong, to the end of the frame
    # and fails at its trailer -- after the window wrapped and was written.  The failed call has written 32768 bytes although no request
    # longer than 32767 succeeds; a file at offset 0 asked for next starts the folder over.
    for seed, flip, mask in ((4000, 103112, 0x20),):
        cab = F.base_cab(seed)
        cab[flip] ^= mask
        cab = bytes(cab)
        v = dict(seed=seed, victim=None, cut=None, flip=flip, flip_mask=mask, cab_md5=hashlib.md5(cab).hexdigest(), runs=[])
        for salvage in (0, 1):
            for order in ([5, 4], [4, 5], [5, 5, 4], [5, 4, 5], list(range(8)), list(range(7, -1, -1))):
                rc, res = helpers.ref_cab_extract(cab, order, cap=len(order) * 160000 + 4096, salvage=salvage)
                assert rc == 0
                v["runs"].append(dict(salvage=salvage, order=order,
                                      results=[dict(err=e, n=len(bb), md5=hashlib.md5(bb).hexdigest()) for e, bb in res]))
                print(seed, salvage, order, [(e, len(bb)) for e, bb in res])
        gold.append(v)
    json.dump(gold, open(os.path.join(HERE, "cab_sticky.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
