"""MSPACK_HIP_UF_LZX_LOG at the batch ABI: an LZX unit with a reset interval reports the reset points that found a block still
open (lzxd.c:423-431 -- where the reference says "WARNING; invalid reset interval detected during LZX decompression") in a
log behind its output, and decodes through them as the reference does.  Units: streams whose block headers were made to
claim a few bytes too many (the block outlives its frame), decoded serially and with a frame table (frame-parallel tasks up
to the odd frame, the serial path from there); expected: the oracle's bytes, result and list of frames."""
import numpy as np
import pytest

import libmspack_amd as M
import helpers

pytestmark = pytest.mark.gpu
F = 32768
UF_LZX_LOG = 32


def streams():
    out = []
    for seed, wb, rf, nfr, patches in ((1, 16, 1, 6, {1: 100, 3: 7}), (2, 17, 2, 8, {3: 50}), (3, 16, 2, 6, {}),
                                       (4, 18, 1, 5, {0: 1, 1: 2, 2: 3, 3: 4}), (5, 16, 3, 9, {2: 11, 8: 5})):
        d = M.gen_plaintext(4000 + seed, seed & 1, nfr * F)
        lz, fo = M.lzx_encode(d, wb, rf)
        lz = bytearray(lz.tobytes())
        for fr, extra in patches.items():
            helpers.lzx_set_bits(lz, int(fo[fr]), 4 if fr % rf == 0 else 3, 24, F + extra)   # (1 Intel-header bit at an interval's start)
        out.append((bytes(lz), fo, wb, rf, nfr, sorted((fr // rf + 1) * rf for fr in patches)))
    return out


@pytest.mark.parametrize("tables", [False, True], ids=["serial", "frame_tables"])
def test_reset_log(built, tables):
    S = streams()
    cap = 4
    n = len(S)
    units = np.zeros(n, dtype=M.UNIT_DTYPE)
    parts, pos, opos, fbase = [], 0, 0, 0
    for i, (lz, fo, wb, rf, nfr, _exp) in enumerate(S):
        units[i]["in_off"] = pos; units[i]["in_len"] = len(lz)
        parts.append(np.frombuffer(lz, dtype=np.uint8)); pos += len(lz)
        pad = (-pos) % 64 + 64
        parts.append(np.zeros(pad, dtype=np.uint8)); pos += pad
        if tables:
            units[i]["in_chunk"] = pos // 4
            parts.append(np.asarray(fo[:nfr], dtype=np.uint32).view(np.uint8)); pos += 4 * nfr
            pad = (-pos) % 16
            parts.append(np.zeros(pad, dtype=np.uint8)); pos += pad
        units[i]["kind"] = M.KIND_LZX; units[i]["window_bits"] = wb; units[i]["reset_frames"] = rf
        units[i]["out_off"] = opos; units[i]["out_len"] = nfr * F
        units[i]["flags"] = UF_LZX_LOG | (M.UF_FRAME_TABLE if tables else 0)
        units[i]["ref_len"] = cap
        units[i]["frame_base"] = fbase; fbase += nfr + 1
        opos += ((nfr * F + 32768 + 15) & ~15) + ((4 + 4 * cap + 15) & ~15)
    arena = np.concatenate(parts + [np.zeros(64, dtype=np.uint8)])
    out, res = M.decode_batch(units, arena, opos)
    for i, (lz, fo, wb, rf, nfr, exp) in enumerate(S):
        err, want, ores = helpers.oracle_lzx(lz, nfr * F, wb, rf)
        cnt, frames = helpers.oracle_lzx_open_resets()
        assert frames == exp, (i, frames, exp)              # (the streams are what they were made to be)
        o = int(units[i]["out_off"])
        # (all bytes come out; the look-ahead behind the stream's last frame may fail with a read error, as in the reference)
        assert res["err"][i] == err and res["out_len"][i] == ores.out_len == nfr * F and out[o:o + nfr * F].tobytes() == want, i
        lg = out[o + ((nfr * F + 32768 + 15) & ~15):][:4 + 4 * cap].view(np.uint32)
        assert int(lg[0]) == cnt, (i, int(lg[0]), cnt)
        assert lg[1:1 + min(cnt, cap)].tolist() == frames[:cap], (i, lg.tolist(), frames)
