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
assert D.gather_scalar(dist, torch.device("cpu"), 10.0 + rank) == [10.0, 11.0]
assert D.all_true(dist, torch.device("cpu"), True) is True
assert D.all_true(dist, torch.device("cpu"), rank == 0) is False
# strong-scaling corpus: the two shards of a 6-unit global list are the halves of the list one rank would make
lo, hi = D.shard_range(6, rank, world)
pl, _c, _o, _l = M.corpus_lzx_units(0xC0F165, 0, hi - lo, 65536, 21, n_threads=1, first_unit=lo)
full, _c, _o, _l = M.corpus_lzx_units(0xC0F165, 0, 6, 65536, 21, n_threads=1)
assert np.array_equal(pl, full[lo * 65536:hi * 65536])
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


def test_bench_refuses_wrong_world(built):
    """bench.py never prints a line whose n_gpus differs from the request: more ranks asked for than GPUs present
    (here: none) is an error before anything is spawned, and a launcher world size != --gpus is one too."""
    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
    import torch
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        assert p.returncode != 0 and b"refusing" in p.stderr and b'"metric"' not in p.stdout
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env2,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
    assert p.returncode != 0 and b"refusing" in p.stderr and b'"metric"' not in p.stdout
