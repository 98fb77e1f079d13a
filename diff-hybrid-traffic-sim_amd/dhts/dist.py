"""Multi-GPU sharding: independent lanes / replicas split contiguously over ranks, one RCCL all-reduce of the
flat [loss-gradient || loss] buffer per optimiser iteration.  No per-step exchange exists:
lanes in the straight-road configurations are independent, so the tape and the state stay GPU-local.
"""
import os

import torch
import torch.distributed as dist


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def _forced():
    """DHTS_DIST_FORCE=1: build the process group and run the collectives even with ONE rank -- so that a single-GPU box
    exercises the RCCL path (communicator set-up, device-buffer all-reduce) the multi-GPU bench will take."""
    return os.environ.get("DHTS_DIST_FORCE", "0") == "1"


def local_device(local=None):
    """The GPU of this rank: LOCAL_RANK on a full node; wraps only where several ranks share a device (single-GPU smoke tests).
    None without a GPU."""
    n_dev = torch.cuda.device_count()           # (counting devices does not initialise the GPU)
    if n_dev == 0:
        return None
    if local is None:
        local = env_rank_world()[2]
    return torch.device("cuda", local % n_dev)


def init(backend=None):
    """Initialise torch.distributed from the torchrun environment (backend 'nccl' = RCCL on ROCm).  The rank's device is made
    current BEFORE the process group exists and handed to it (`device_id`): RCCL then binds its communicator to that device at
    once instead of guessing at the first collective (with eight ranks on a node a lazily bound communicator lands on device 0
    for every rank that has not called set_device yet, and `barrier()` has to pick a device by itself)."""
    rank, world, local = env_rank_world()
    dev = local_device(local)
    if dev is not None:
        torch.cuda.set_device(dev)
    if (world > 1 or _forced()) and not dist.is_initialized():
        if backend is None:
            # the default on GPUs is RCCL ("nccl"); DHTS_DIST_BACKEND=gloo -- or more ranks than devices -- lets several ranks
            # share one GPU in a smoke test (RCCL refuses two ranks on one device)
            backend = os.environ.get("DHTS_DIST_BACKEND")
            if not backend:
                n_dev = torch.cuda.device_count()
                backend = "nccl" if n_dev >= world else "gloo"
                if n_dev and backend == "gloo" and rank == 0:
                    print("dhts.dist: %d ranks on %d GPU(s): ranks share devices, collectives over gloo" % (world, n_dev), flush=True,
                          file=__import__("sys").stderr)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {}
        if backend == "nccl" and dev is not None:
            kw["device_id"] = dev
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def shard_range(n_units, rank, world):
    """Contiguous [begin, end) of `n_units` independent lanes / replicas owned by `rank`."""
    base, rem = divmod(n_units, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def _active():
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _forced())


def _staged(t):
    """gloo moves host memory: a device tensor goes through a host copy there (RCCL takes it as it is)."""
    return t.is_cuda and dist.get_backend() == "gloo"


def allreduce_sum_(flat):
    """In-place sum over ranks of a flat float32 buffer [d loss / d theta_shared || loss]; no-op at world size 1."""
    if _active():
        if _staged(flat):
            host = flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            flat.copy_(host)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


def gather_to_rank0(flat):
    """[world][n] host copy of every rank's flat buffer on rank 0 (None elsewhere); diagnostics only, not on the timed path."""
    if not _active():
        return flat.detach().cpu()[None]
    world = dist.get_world_size()
    src = flat.detach().cpu() if _staged(flat) else flat.detach()
    parts = [torch.empty_like(src) for _ in range(world)]
    dist.all_gather(parts, src)
    return torch.stack([p.cpu() for p in parts]) if dist.get_rank() == 0 else None


def barrier():
    if _active():
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()


def max_over_ranks(seconds, device):
    if _active() and dist.get_backend() == "gloo":
        device = torch.device("cpu")
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    if _active():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def spawn_ranks(n, argv, module=None, script=None):
    """`--gpus n` without a launcher: start n children (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, one per GPU) of
    `python -m module argv...` or `python script argv...`.  The parent never touches a GPU and never execs; rank 0's stdout is the
    parent's, the other ranks' goes to stderr; a rank that dies ends the others (by their own PIDs) instead of leaving them in a
    collective.  Returns the worst exit code."""
    import socket
    import subprocess
    import sys
    import time
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable] + (["-m", module] if module else [os.path.abspath(script)]) + list(argv)
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen(cmd, env=env, stdout=None if r == 0 else sys.stderr))
    rcs = [None] * n
    while any(c is None for c in rcs):
        time.sleep(0.2)
        for r, p in enumerate(procs):
            if rcs[r] is None:
                rcs[r] = p.poll()
        if any(c not in (None, 0) for c in rcs):
            deadline = time.time() + 15.0
            while time.time() < deadline and any(p.poll() is None for p in procs):
                time.sleep(0.2)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            rcs = [p.wait() for p in procs]
    return max(abs(int(c)) for c in rcs)
