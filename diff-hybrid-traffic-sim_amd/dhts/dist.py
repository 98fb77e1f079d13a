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
            # the default on GPUs is RCCL ("nccl").  Several ranks on ONE device (a single-GPU smoke test) need gloo, because RCCL
            # refuses two ranks on one device -- but only when the caller asks for it by name: a node that shows fewer GPUs than
            # ranks must not turn an RCCL measurement into a gloo one behind the caller's back
            backend = os.environ.get("DHTS_DIST_BACKEND")
            if not backend:
                n_dev = torch.cuda.device_count()
                if 0 < n_dev < world:
                    raise RuntimeError("dhts.dist: %d ranks but only %d GPU(s) visible: refusing to fall back to gloo silently; "
                                       "set DHTS_DIST_BACKEND=gloo to let ranks share devices (smoke tests only)" % (world, n_dev))
                backend = "nccl" if n_dev else "gloo"        # (no GPU at all: host-only ranks, e.g. the launcher self-test)
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


def device_identity(device):
    """What tells this rank's GPU from its neighbours': marketing name, gfx arch, PCI bus id, UUID where the runtime gives one
    (diagnostics for the N > 1 bench line: the reader can see that N ranks sat on N distinct devices)."""
    ident = {"rank": dist.get_rank() if dist.is_initialized() else 0, "host": os.uname().nodename,
             "visible": os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")}
    if device is None or device.type != "cuda":
        ident["device"] = "cpu"
        return ident
    p = torch.cuda.get_device_properties(device)
    ident.update(device="cuda:%d" % device.index, name=p.name, arch=getattr(p, "gcnArchName", None),
                 uuid=str(getattr(p, "uuid", "")) or None)
    bus = None
    if all(hasattr(p, k) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id")):
        bus = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
    else:
        try:                                               # the HIP runtime torch already loaded
            import ctypes
            buf = ctypes.create_string_buffer(64)
            if ctypes.CDLL("libamdhip64.so").hipDeviceGetPCIBusId(buf, 64, int(device.index)) == 0:
                bus = buf.value.decode()
        except OSError:
            pass
    ident["pci_bus_id"] = bus
    return ident


def gather_objects(obj):
    """[world] list of every rank's (picklable) `obj` on every rank; [obj] without a process group.  Diagnostics only."""
    if not (dist.is_available() and dist.is_initialized()):
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


def collective_record(device):
    """{"backend", "world", "devices": [per-rank device_identity], "distinct_devices"} -- every rank must call it (a gather).
    Raises when the backend is RCCL and two ranks report the same device: such a line would not be an N-GPU number."""
    devs = gather_objects(device_identity(device))
    backend = dist.get_backend() if dist.is_initialized() else None
    keys = {(d.get("host"), d.get("pci_bus_id") or d.get("uuid") or d.get("device")) for d in devs}
    rec = {"backend": backend, "backend_is": "RCCL (torch.distributed 'nccl' on ROCm)" if backend == "nccl" else backend,
           "world": len(devs), "devices": devs, "distinct_devices": len(keys)}
    if backend == "nccl" and len(keys) != len(devs):
        raise RuntimeError("dhts.dist: %d ranks over RCCL but only %d distinct devices: %s" % (len(devs), len(keys), devs))
    return rec


def barrier():
    if _active():
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()


def shutdown():
    """Destroy the process group this module built (no-op without one)."""
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


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
