"""Multi-GPU sharding: independent lanes / replicas split contiguously over ranks, one RCCL all-reduce of the
flat [loss-gradient || loss] buffer per optimiser iteration.  No per-step exchange exists:
lanes in the straight-road configurations are independent, so the tape and the state stay GPU-local.
"""
import os

import torch
import torch.distributed as dist


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend=None):
    """Initialise torch.distributed from the torchrun environment (backend 'nccl' = RCCL on ROCm)."""
    rank, world, local = env_rank_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            # DHTS_DIST_BACKEND=gloo lets two ranks share one GPU in a smoke test; the default on GPUs is RCCL ("nccl")
            backend = os.environ.get("DHTS_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(n_units, rank, world):
    """Contiguous [begin, end) of `n_units` independent lanes / replicas owned by `rank`."""
    base, rem = divmod(n_units, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def allreduce_sum_(flat):
    """In-place sum over ranks of a flat float32 buffer [d loss / d theta_shared || loss]; no-op at world size 1."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(seconds, device):
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
