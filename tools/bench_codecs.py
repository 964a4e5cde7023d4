#!/usr/bin/env python
"""Side measurements for BASELINE.json's configs 2 and 4 (not the headline metric; bench.py is that):
  mszip : 4096 independent MSZIP CFDATA blocks of 32 KiB ('CK' + raw deflate, zlib level 6) on one GPU
  qtm   : 512 independent Quantum folders (window 2^18, 4 frames = 128 KiB each) on one GPU
Inputs resident in HBM, kernel time from HIP events (mspack_hip_time_batch_device), every byte verified.
Usage: python tools/bench_codecs.py [mszip|qtm] [--units N]"""
import argparse
import sys
import time
import zlib

import numpy as np

sys.path.insert(0, ".")
import libmspack_amd as M  # noqa: E402


def corpus_mszip(n, ub=32768):
    plain = M.gen_plaintext(0xC0FFEE, 0, n * ub)
    parts, offs, lens, pos = [], [], [], 0
    for i in range(n):
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        blob = b"CK" + co.compress(plain[i * ub:(i + 1) * ub].tobytes()) + co.flush()
        pad = (-len(blob)) % 16
        offs.append(pos); lens.append(len(blob)); parts.append(blob + b"\0" * pad); pos += len(blob) + pad
    comp = np.frombuffer(b"".join(parts) + b"\0" * 64, dtype=np.uint8)
    return plain, comp, np.array(offs, dtype=np.uint64), np.array(lens, dtype=np.uint32), ub


def corpus_qtm(n, ub=131072):
    plain = M.gen_plaintext(0xC0FFEE, 0, n * ub)
    parts, offs, lens, pos = [], [], [], 0
    for i in range(n):
        st, _fs = M.qtm_encode(plain[i * ub:(i + 1) * ub], 18)
        blob = bytes(st)
        pad = (-len(blob)) % 16
        offs.append(pos); lens.append(len(blob)); parts.append(blob + b"\0" * pad); pos += len(blob) + pad
    comp = np.frombuffer(b"".join(parts) + b"\0" * 64, dtype=np.uint8)
    return plain, comp, np.array(offs, dtype=np.uint64), np.array(lens, dtype=np.uint32), ub


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("codec", choices=["mszip", "qtm"])
    ap.add_argument("--units", type=int, default=0)
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    import torch
    n = a.units or (4096 if a.codec == "mszip" else 512)
    t0 = time.perf_counter()
    if a.codec == "mszip":
        plain, comp, off, ln, ub = corpus_mszip(n)
        units, out_bytes = M.make_units(M.KIND_MSZIP, off, ln, np.full(n, ub), out_slack=32768)
        kind = M.KIND_MSZIP
    else:
        plain, comp, off, ln, ub = corpus_qtm(n)
        units, out_bytes = M.make_units(M.KIND_QUANTUM, off, ln, np.full(n, ub), window_bits=18)
        kind = M.KIND_QUANTUM
    gen = time.perf_counter() - t0
    order = np.argsort(-(ln.astype(np.int64)), kind="stable").astype(np.uint32)
    dev = torch.device("cuda", 0)
    d_in = torch.from_numpy(comp.copy()).to(dev)
    d_units = torch.from_numpy(units.view(np.uint8)).to(dev)
    d_order = torch.from_numpy(order.view(np.uint8)).to(dev)
    d_out = torch.zeros(out_bytes + 64, dtype=torch.uint8, device=dev)
    d_res = torch.zeros(n * M.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    d_fm = torch.zeros(max(M.lib().mspack_hip_frame_scratch_bytes(int(M.frames_of(units).sum())), 64), dtype=torch.uint8, device=dev)
    L = M.lib()
    stream = torch.cuda.current_stream().cuda_stream
    ms = L.mspack_hip_time_batch_device(d_units.data_ptr(), d_order.data_ptr(), n, d_in.data_ptr(), comp.size - 64,
                                        d_out.data_ptr(), out_bytes, d_res.data_ptr(), d_fm.data_ptr(),
                                        int(M.frames_of(units).sum()), 1 << kind, stream, a.iters)
    torch.cuda.synchronize()
    res = d_res.cpu().numpy().view(M.RESULT_DTYPE)
    out = d_out.cpu().numpy()
    oo = units["out_off"].astype(np.int64)
    ok = bool((res["err"] == 0).all()) and all(
        np.array_equal(out[oo[i]:oo[i] + ub], plain[i * ub:(i + 1) * ub]) for i in range(n))
    print({"codec": a.codec, "units": n, "unit_bytes": ub, "ratio": round(float(ln.sum()) / (n * ub), 3),
           "kernel_ms": round(ms, 3), "decompressed_MBps": round(n * ub / ms / 1e3, 1), "bit_exact": ok,
           "corpus_gen_s": round(gen, 1)})


if __name__ == "__main__":
    main()
