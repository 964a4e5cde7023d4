#!/usr/bin/env python3
"""bench.py -- headline metric of BASELINE.json: decompressed MB/s, LZX 21-bit window,
4096-interval batch (per GPU; weak scaling over --gpus), inputs resident in HBM.

A "step" = one pass of the hot path (mspack_hip_decode_batch_device) over the whole batch of
independent CHM-style reset intervals (64 KiB each = reset interval of 2 frames).
Prints ONE JSON line on rank 0.  Multi-GPU: one process per GPU, no data-path collective (units are
independent); torch.distributed (RCCL) is used only for the barrier and the max-over-ranks time.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def usable_cpus():
    """CPUs this process can actually run on: the affinity mask capped by the cgroup CPU quota (the GPU
    boxes expose every host thread but grant a fixed quota, and threads beyond it only add contention)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return n


def cpu_baseline(comp, off, ln, n_units, unit_bytes, budget_s=12.0):
    """Reference CPU path (oracle/_ref = the real libmspack lzxd, built from /root/reference in the
    dev container) over the same units on the host cores; falls back to our CPU restatement."""
    import ctypes as C
    cores = usable_cpus()
    try:
        import helpers
        if helpers.have_ref():
            R = helpers.ref()
            off64 = np.ascontiguousarray(off, dtype=np.uint64)
            ilen = np.ascontiguousarray(ln + 4, dtype=np.uint32)
            olen = np.full(n_units, unit_bytes, dtype=np.uint32)
            def run(threads, reps):
                b = C.c_ulonglong(0); e = C.c_int(0)
                t = R.refh_bench(0, comp.ctypes.data, off64.ctypes.data, ilen.ctypes.data, olen.ctypes.data,
                                 n_units, 21, unit_bytes // 32768, threads, reps, C.byref(b), C.byref(e))
                if e.value:
                    raise RuntimeError("reference failed on %d units" % e.value)
                return b.value / t / 1e6, t
            one, _t = run(1, 1)                                   # one core, one pass: MB/s per core
            # one thread per usable CPU; enough passes for ~15 s of wall time
            est_pass_s = n_units * unit_bytes / (one * 1e6) / cores
            reps = max(2, min(256, int(15.0 / max(est_pass_s, 1e-3))))
            allv, tall = run(cores, reps)
            return {"value": round(allv, 1), "unit": "MB/s", "cores": cores, "kind": "reference",
                    "one_core_MBps": round(one, 1), "host_threads": os.cpu_count(),
                    "sample": "%d passes over the same %d-unit batch (%.0f MiB decoded per pass) on %d threads (= usable CPUs: affinity and cgroup quota) in "
                              "%.2f s, threads released together; libmspack lzxd_decompress memory-to-memory, one "
                              "decompressor per thread" % (reps, n_units, n_units * unit_bytes / 2**20, cores, tall)}
    except Exception as ex:          # pragma: no cover
        sys.stderr.write("cpu_baseline: reference unavailable (%s); using the port\n" % ex)
    import helpers
    sample = min(n_units, 256)
    t0 = time.perf_counter()
    for i in range(sample):
        e, o, _ = helpers.oracle_lzx(comp[int(off[i]):int(off[i]) + int(ln[i]) + 4].tobytes(), unit_bytes, 21,
                                     unit_bytes // 32768)
        assert e == 0
    dt = time.perf_counter() - t0
    return {"value": round(sample * unit_bytes / dt / 1e6, 1), "unit": "MB/s", "cores": 1, "kind": "port",
            "sample": "first %d units, single thread, oracle/liboracle.so" % sample}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--units", type=int, default=4096, help="reset intervals per GPU")
    ap.add_argument("--unit-kib", type=int, default=64)
    ap.add_argument("--text", type=int, default=0, help="plaintext family (0 = mix)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--exp", action="store_true", help="kernel experiments: skip the parity gate and the CPU leg (the line is then NOT a valid result)")
    args = ap.parse_args()

    import torch
    import libmspack_amd as M

    from libmspack_amd import dist as D
    rank, world, local = D.env_rank_world()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists)")
    torch.cuda.set_device(local)
    dist = D.init("nccl", torch.device("cuda", local))

    n, ub = args.units, args.unit_kib * 1024
    # ---- synthetic corpus: every rank its own seeds (weak scaling: fixed work per GPU) ----
    t0 = time.perf_counter()
    plain, comp, off, ln = M.corpus_lzx_units(D.unit_seed_base(0xBA5E11, rank), args.text, n, ub, 21,
                                              n_threads=max(1, usable_cpus() // max(world, 1)))
    gen_s = time.perf_counter() - t0
    units, out_bytes = M.make_units(M.KIND_LZX, off, ln + 4, np.full(n, ub), window_bits=21,
                                    reset_frames=ub // 32768)
    order = np.argsort(-(ln.astype(np.int64)), kind="stable").astype(np.uint32)   # longest first
    n_frames = int(M.frames_of(units).sum())

    dev = torch.device("cuda", local)
    d_in = torch.zeros(comp.size + 64, dtype=torch.uint8, device=dev)
    d_in[:comp.size] = torch.from_numpy(comp).to(dev)
    d_units = torch.from_numpy(units.view(np.uint8)).to(dev)
    d_order = torch.from_numpy(order.view(np.uint8)).to(dev)
    d_out = torch.zeros(out_bytes + 64, dtype=torch.uint8, device=dev)
    d_res = torch.zeros(n * M.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    d_fm = torch.zeros(M.lib().mspack_hip_frame_scratch_bytes(n_frames), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    L = M.lib()

    def step():
        rc = L.mspack_hip_decode_batch_device(d_units.data_ptr(), d_order.data_ptr(), n, d_in.data_ptr(), comp.size,
                                              d_out.data_ptr(), out_bytes, d_res.data_ptr(), d_fm.data_ptr(),
                                              n_frames, 1 << M.KIND_LZX, stream)
        if rc:
            raise RuntimeError(L.mspack_hip_last_error().decode())

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed, total_out = D.reduce_scalars(dist, dev, elapsed, float(n * ub))

    # ---- kernel-only duration with HIP events on the launch stream (roofline numerator) ----
    ms_kernel = L.mspack_hip_time_batch_device(d_units.data_ptr(), d_order.data_ptr(), n, d_in.data_ptr(), comp.size,
                                               d_out.data_ptr(), out_bytes, d_res.data_ptr(), d_fm.data_ptr(),
                                               n_frames, 1 << M.KIND_LZX, stream, max(3, min(args.steps, 10)))
    torch.cuda.synchronize()

    # ---- parity: every unit, every byte, outside the timed region ----
    res = d_res.cpu().numpy().view(M.RESULT_DTYPE)
    out = d_out[:n * ub].cpu().numpy()
    ok = bool((res["err"] == 0).all() and (res["out_len"] == ub).all() and np.array_equal(out, plain))
    if not ok and not args.exp:
        raise SystemExit("rank %d: GPU output is NOT bit-exact; refusing to report a number" % rank)

    comp_bytes = float(ln.sum())
    # HBM traffic per launch: PMC counters cannot be collected from inside this process; the latest
    # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same workload are kept in profiles/
    traffic = None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        w = tj["workload"]
        if (w["units_per_gpu"], w["unit_bytes"], w["text"]) == (n, ub, args.text):
            # FETCH_SIZE counts 64 B per 128-B request on gfx950 (the guide's x2; calibrated on this
            # kernel's stored-block copy, see profiles/traffic.json)
            traffic = int((tj["fetch_kib_per_launch"] * tj.get("fetch_correction", 1.0) + tj["write_kib_per_launch"]) * 1024)
    except Exception:
        traffic = None
    algo_bytes = comp_bytes + n * ub                      # in + out, per launch
    if rank == 0:
        line = {
            "metric": "decompressed MB/s (whole node), LZX 21-bit window, 4096-interval batch",
            "value": round(total_out * args.steps / elapsed / 1e6, 1),
            "unit": "MB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "CHM-style LZX, window_bits=21, reset interval %d frames (%d KiB), %d "
                                   "intervals per GPU, plaintext family %d, ratio %.3f" %
                                   (ub // 32768, args.unit_kib, n, args.text, comp_bytes / (n * ub)),
                       "units_per_gpu": n, "unit_bytes": ub, "bit_exact": ok,
                       "corpus_gen_s": round(gen_s, 2)},
            "roofline": {"bound": "hbm", "achieved": round(algo_bytes / (ms_kernel * 1e-3) / 1e9, 2),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(algo_bytes / (ms_kernel * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                         "traffic": traffic, "kernel": "mspack_decode_lzx", "kernel_ms": round(ms_kernel, 4),
                         "algorithmic_bytes_per_launch": int(algo_bytes)},
        }
        if world == 1 and not args.no_cpu and not args.exp:
            line["cpu_baseline"] = cpu_baseline(comp, off, ln, n, ub)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
