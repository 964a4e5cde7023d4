# scratch (GPU box): ONE big LZX-21 CAB folder (N frames = N CFDATA blocks) through mspack_create_cab_decompressor ->
# extract(): serial (MSPACK_HIP_NO_FRAME_PARSE=1) vs frame-parallel parse (the driver passes the block sizes as the
# folder's frame table).  Usage: python tools/exp_bigfolder.py [n_frames]
import hashlib, os, subprocess, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
if len(sys.argv) > 2 and sys.argv[2] == "child":
    import numpy as np
    import libmspack_amd as M
    from libmspack_amd import api
    nf = int(sys.argv[1])
    data = M.gen_plaintext(77, 0, nf * 32768 - 1234)
    lz, fo = M.lzx_encode(data, 21, 0)
    blocks = [lz[int(fo[i]):int(fo[i + 1])].tobytes() for i in range(len(fo) - 1)]
    us = [min(32768, data.size - i * 32768) for i in range(len(blocks))]
    cab = M.cab_write([(0x1503, blocks, us)], [(b"big.bin", data.size, 0, 0)])
    for rep in range(3):
        with api.Cab(cab, mem=True) as c:
            t0 = time.perf_counter()
            err, out = c.extract(0)
            dt = time.perf_counter() - t0
        ok = err == 0 and hashlib.md5(out).digest() == hashlib.md5(data.tobytes()).digest()
        print("  extract: %.1f ms  %.1f MB/s  %s" % (dt * 1e3, data.size / dt / 1e6, "bit-exact" if ok else "MISMATCH err=%d" % err), flush=True)
else:
    nf = sys.argv[1] if len(sys.argv) > 1 else "512"
    for label, env in (("serial (one wavefront)", {"MSPACK_HIP_NO_FRAME_PARSE": "1"}), ("frame-parallel parse", {})):
        print("%s frames, %s:" % (nf, label), flush=True)
        subprocess.run([sys.executable, __file__, nf, "child"], env=dict(os.environ, **env))
