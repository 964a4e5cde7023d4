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


def gather_scalar(dist, device, value):
    """-> [value of rank 0, value of rank 1, ...] on every rank (per-rank step times: imbalance made visible)"""
    if dist is None:
        return [float(value)]
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(x.item()) for x in out]


def all_true(dist, device, flag):
    """logical AND over ranks (the bit-exactness gate: one bad rank voids the line)"""
    if dist is None:
        return bool(flag)
    import torch
    t = torch.tensor([1.0 if flag else 0.0], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item() > 0.5)
