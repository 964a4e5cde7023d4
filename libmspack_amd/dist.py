"""Multi-GPU plumbing for bench.py: one process per GPU (torch.distributed; backend "nccl" = RCCL on
ROCm, "gloo" in CPU tests).  The decode path itself has NO collective -- units are independent
(SURVEY.md sec. 8(e)); the process group is only used for the barrier around the timed region and
for the max-over-ranks / sum-over-ranks of the scalars that go into the JSON line."""
import os


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), \
        int(os.environ.get("LOCAL_RANK", "0"))


def init(backend, device=None):
    """returns the torch.distributed module (initialised) or None when WORLD_SIZE == 1"""
    rank, world, _local = env_rank_world()
    if world <= 1:
        return None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if device is not None:
        dist.init_process_group(backend, device_id=device)
    else:
        dist.init_process_group(backend)
    return dist


def unit_seed_base(base_seed, rank):
    """weak scaling: every rank generates its own, disjoint corpus"""
    return base_seed + (rank << 32)


def shard_range(n_total, rank, world):
    """static sharding of a global unit list (strong-scaling mode): contiguous, sizes differ by <= 1"""
    lo = n_total * rank // world
    hi = n_total * (rank + 1) // world
    return lo, hi


def reduce_scalars(dist, device, elapsed, bytes_out):
    """-> (max elapsed over ranks, sum of bytes over ranks)"""
    if dist is None:
        return elapsed, bytes_out
    import torch
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    b = torch.tensor([float(bytes_out)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(b, op=dist.ReduceOp.SUM)
    return float(t.item()), float(b.item())
