#!/usr/bin/env python
"""Side measurement: MSZIP folders whose blocks reference the previous block (real cabinets do: the 32 KiB
window is not cleared between CFDATA blocks, mszipd.c:267-268) vs independent blocks."""
import sys, zlib
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import libmspack_amd as M
from test_gpu_mszip import folder
from test_gpu_mszip_blocks import folder_blocks

import torch
n, fb = 512, 8                      # folders, blocks per folder
ub = fb * 32768
plain = M.gen_plaintext(0xC0FFEE, 0, n * ub)
if len(sys.argv) > 2:
    n, fb = int(sys.argv[1]), int(sys.argv[2]); ub = fb * 32768
    plain = M.gen_plaintext(0xC0FFEE, 0, n * ub)
for hist, tables in ((False, False), (True, False), (True, True)):
    parts, offs, lens, tabs, pos = [], [], [], [], 0
    for i in range(n):
        blob, boffs = folder_blocks(plain[i * ub:(i + 1) * ub].tobytes(), 6, history=hist)
        tab = np.array(boffs, dtype=np.uint32).tobytes()
        pad = (-len(blob)) % 16
        offs.append(pos); lens.append(len(blob)); tabs.append(pos + len(blob) + pad)
        parts.append(blob + b"\0" * pad + tab + b"\0" * ((-len(tab)) % 16)); pos += len(parts[-1])
    comp = np.frombuffer(b"".join(parts) + b"\0" * 64, dtype=np.uint8)
    off = np.array(offs, dtype=np.uint64); ln = np.array(lens, dtype=np.uint32)
    units, out_bytes = M.make_units(M.KIND_MSZIP, off, ln, np.full(n, ub), out_slack=32768,
                                    frame_tabs=np.array(tabs, dtype=np.uint64) if tables else None)
    order = np.argsort(-(ln.astype(np.int64)), kind="stable").astype(np.uint32)
    dev = torch.device("cuda", 0)
    d_in = torch.from_numpy(comp.copy()).to(dev); d_units = torch.from_numpy(units.view(np.uint8)).to(dev)
    d_order = torch.from_numpy(order.view(np.uint8)).to(dev)
    d_out = torch.zeros(out_bytes + 64, dtype=torch.uint8, device=dev)
    d_res = torch.zeros(n * M.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    L = M.lib()
    nfr = int(M.frames_of(units).sum())
    d_fm = torch.zeros(max(64, L.mspack_hip_frame_scratch_bytes(nfr)), dtype=torch.uint8, device=dev)
    ms = L.mspack_hip_time_batch_device(d_units.data_ptr(), d_order.data_ptr(), n, d_in.data_ptr(), comp.size - 64,
                                        d_out.data_ptr(), out_bytes, d_res.data_ptr(), d_fm.data_ptr(), nfr,
                                        (1 << M.KIND_MSZIP) | (0x80000000 if tables else 0),
                                        torch.cuda.current_stream().cuda_stream, 5 if n * fb < 20000 else 2)
    torch.cuda.synchronize()
    res = d_res.cpu().numpy().view(M.RESULT_DTYPE); out = d_out.cpu().numpy(); oo = units["out_off"].astype(np.int64)
    ok = bool((res["err"] == 0).all()) and all(np.array_equal(out[oo[i]:oo[i] + ub], plain[i * ub:(i + 1) * ub]) for i in range(n))
    print({"history": hist, "block_parse": tables, "adopted": float(((res["flags"] & 32) != 0).mean()), "folders": n, "blocks_per_folder": fb, "ratio": round(float(ln.sum()) / (n * ub), 3),
           "kernel_ms": round(ms, 3), "MBps": round(n * ub / ms / 1e3, 1), "bit_exact": ok})
