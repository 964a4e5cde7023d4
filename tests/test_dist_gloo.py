"""CPU, world_size = 2 over gloo: the multi-GPU plumbing of bench.py (libmspack_amd/dist.py) --
disjoint per-rank corpora (weak scaling), static sharding, max-over-ranks time and summed bytes.
The decode path itself has no collective to test: units never exchange data."""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
sys.path.insert(0, %r)
import numpy as np, torch
import libmspack_amd as M
from libmspack_amd import dist as D
rank, world, local = D.env_rank_world()
dist = D.init("gloo")
assert dist is not None and world == 2
# weak scaling: each rank its own corpus; the two must differ
plain, comp, off, ln = M.corpus_lzx_units(D.unit_seed_base(0xBA5E11, rank), 0, 4, 65536, 21, n_threads=1)
digest = int(np.frombuffer(plain.tobytes()[:8], dtype=np.uint64)[0] %% (1 << 52))
t = torch.tensor([float(digest)], dtype=torch.float64)
lst = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
dist.all_gather(lst, t)
assert lst[0].item() != lst[1].item(), "ranks generated the same corpus"
# strong-scaling shard plan covers every unit exactly once
lo, hi = D.shard_range(1001, rank, world)
cover = torch.zeros(1001); cover[lo:hi] = 1
dist.all_reduce(cover)
assert bool((cover == 1).all())
# reductions used for the JSON line
el, total = D.reduce_scalars(dist, torch.device("cpu"), 1.0 + rank, 100.0 * (rank + 1))
assert el == 2.0 and total == 300.0
dist.barrier()
if rank == 0:
    print("GLOO_OK")
dist.destroy_process_group()
''' % ROOT


def test_two_rank_gloo(built, tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "GLOO_OK" in outs[0]
