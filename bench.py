#!/usr/bin/env python3
"""bench.py -- differentiable cell-steps/s (forward + adjoint) of the time-fused HIP stepper.

    python bench.py --gpus N --steps K --warmup W [--workload macro|micro]

One "step" = one pass of the hot path over one batch of synthetic input: a full differentiable rollout
(forward sweep writing the Jacobian tape + reverse sweep reading it back) of BASELINE.json configs[1]
(macro: 1024 lanes x 512 ARZ cells x 1000 time steps) or, with --workload micro,
configs[2] (4096 lanes x 256 IDM vehicles x 1000 time steps; C3), inputs resident in HBM.
For N > 1 there is one process per GPU: either the caller launches them (torch.distributed.run sets RANK /
LOCAL_RANK / WORLD_SIZE) or, when `--gpus N` is given without that environment, this script starts N child
processes of itself before anything touches a GPU, relays rank 0's JSON line and exits non-zero if a rank fails.
Every rank owns its own shard of independent lanes / replicas (weak scaling: the per-GPU batch is the
configuration above), there is no data-path collective, and the flat [d loss / d theta_shared || loss] buffer is
all-reduced over RCCL once per pass.  Rank 0 prints ONE JSON line.

roofline.achieved / frac are priced in the bytes the dominant kernel MOVES (its compact tape, checked against the PMC
passes under profiles/); the reference's algorithmic tape bytes (48 B per cell-step, 32 B per vehicle-step) divided by
the same time are reported beside them as achieved_algorithmic / frac_algorithmic.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "diff-hybrid-traffic-sim_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured-achievable copy rate
MACRO_TAPE_B = 48           # float32 [3][2][2] per cell-step   (road/lane/dmacro_lane.py:56): the ALGORITHMIC bytes
MICRO_TAPE_B = 32           # float32 [2][2][2] per vehicle-step (road/lane/dmicro_lane.py:54)
# What the rollout kernels actually move is their compact tape (the interface tape, the second rows of dEgo /
# dLeading -- the same information, rebuilt into the blocks by the reverse sweep); its size comes from the library
# (dhts_macro_tape_bytes / dhts_micro_tape_bytes), not from a constant here.


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=["macro", "micro", "itscp_macro", "itscp_hybrid"], default="macro")
    ap.add_argument("--lanes", type=int, default=0, help="override lanes per GPU")
    ap.add_argument("--cells", type=int, default=0, help="override cells / vehicles per lane")
    ap.add_argument("--time-steps", type=int, default=0, help="override simulated time steps per rollout")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the short config 3 / config 4 sub-records of the default run")
    return ap.parse_args()


def spawn_ranks(n):
    """`--gpus n` without a launcher: start n children of this script (RANK / LOCAL_RANK / WORLD_SIZE set), one per GPU.
    The parent never touches the GPU and never execs; it relays rank 0's stdout and returns the worst exit code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    import tempfile
    procs = []
    with tempfile.TemporaryFile() as rank0_out:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port))
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=rank0_out if r == 0 else sys.stderr))
        # a rank that dies leaves the others waiting in a collective: end them (by their own PIDs) instead of hanging
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
        rank0_out.seek(0)
        sys.stdout.write(rank0_out.read().decode())
        sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(rcs) if c != 0]
    if bad:
        print("bench.py: rank(s) failed: %s" % bad, file=sys.stderr)
        return 1
    return 0


def make_workload(name, dev, rank, lanes=0, cells=0, time_steps=0):
    if name == "macro":
        return MacroWorkload(dev, rank, lanes or 1024, cells or 512, time_steps or 1000)
    if name == "micro":
        return MicroWorkload(dev, rank, lanes or 4096, cells or 256, time_steps or 1000)
    if name == "itscp_hybrid":
        return ItscpHybridWorkload(dev, rank, lanes or 256, 0, 0)
    return ItscpMacroWorkload(dev, rank, lanes or 256, 0, 0)


def pmc_traffic(w, kernel):
    """HBM bytes per launch of `kernel` from the PMC passes committed under profiles/ (rocprofv3 --pmc FETCH_SIZE and
    --pmc WRITE_SIZE, gfx950 FETCH correction applied).  Quoted only for the configuration the passes were taken on and only
    while the tape the library allocates still has the size the passes saw (a changed layout needs fresh passes)."""
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        hbm = pmc[w.name][kernel]["hbm_bytes"]
    except (OSError, ValueError, KeyError):
        return None
    moved = w.moved_bytes_per_launch()
    if abs(hbm - moved) > 0.02 * moved:
        print("bench.py: profiles/pmc_traffic.json (%d B) disagrees with the library's tape size (%d B) by more than 2 %%: "
              "traffic not quoted; re-take the PMC passes" % (hbm, moved), file=sys.stderr)
        return None
    return hbm


def kernel_records(w):
    """Per-kernel records from the HIP events one_pass(record=True) left on the launch stream (the torch current stream
    is the stream the C ABI is handed, ops._stream)."""
    fwd_ms = [e[0].elapsed_time(e[1]) for e in w.ev]
    bwd_ms = [e[2].elapsed_time(e[3]) for e in w.ev]
    fwd_avg, bwd_avg = sum(fwd_ms) / len(fwd_ms), sum(bwd_ms) / len(bwd_ms)
    moved = w.moved_bytes_per_launch()
    algo = w.units * w.unit_bytes
    rec = {}
    for k, ms in (("rollout_fwd", fwd_avg), ("rollout_bwd", bwd_avg)):
        rec[k] = {"ms": ms, "moved_GB": moved / 1e9, "GBps": moved / ms / 1e6, "frac_of_peak": moved / ms / 1e6 / HBM_PEAK_GBS,
                  "algorithmic_GB": algo / 1e9, "algorithmic_GBps": algo / ms / 1e6}
    return rec, ("rollout_fwd" if fwd_avg >= bwd_avg else "rollout_bwd")


def also_record(name, dev, passes=5):
    """A short run of another BASELINE configuration in the same process (not the headline; no collective)."""
    w = make_workload(name, dev, 0)
    for _ in range(2):
        w.one_pass()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(passes):
        w.one_pass(record=True)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    fault = w.err.tolist()
    assert fault[0] in (0, 2), "simulation fault during the bench: %s" % (fault,)
    kernels, dom = kernel_records(w)
    out = {"workload": w.name, "value": w.units * passes / el, "unit": w.unit_name, "passes": passes,
           "ms_per_pass": el / passes * 1e3, "dominant_kernel": dom, "limiter": w.limiter.get(dom, "hbm"), "kernels": kernels}
    if getattr(w, "counts", None) is not None:
        out["vehicles_spawned_replica0"] = int(w.counts[0, 0])
    return out


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))        # before any GPU call: the children own the devices
    from dhts import dist as D
    rank, world, local = D.init()
    if world != args.gpus and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    local = local % torch.cuda.device_count()      # one GPU per rank on a full node; wraps only in single-GPU smoke tests
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    w = make_workload(args.workload, dev, rank, args.lanes, args.cells, args.time_steps)
    L, N, T = w.L, w.N, w.T

    # the per-pass RCCL all-reduce: [loss] for the straight-lane workloads (every lane owns its unknowns); for the network
    # workloads the gradient summed over the rank's replicas as if the signal schedule were shared (BASELINE config 5) + loss
    shared_grad = args.workload.startswith("itscp")
    flat = torch.zeros((w.action.shape[1] if shared_grad else 0) + 1, dtype=torch.float32, device=dev)

    def reduce_pass(loss, g_a):
        if shared_grad:
            flat[:-1] = g_a.sum(dim=0)
        flat[-1] = loss
        local_part = flat.clone() if world > 1 else None
        D.allreduce_sum_(flat)
        return local_part
    for _ in range(args.warmup):
        loss, g_a, _ = w.one_pass()
        reduce_pass(loss, g_a)
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, g_a, g_b = w.one_pass(record=True)
        local_part = reduce_pass(loss, g_a)
    D.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    elapsed = D.max_over_ranks(elapsed, dev)

    fault = w.err.tolist()
    assert fault[0] in (0, 2), "simulation fault during the bench: %s" % (fault,)
    assert torch.isfinite(g_a).all() and torch.isfinite(g_b).all() and bool(torch.isfinite(flat).all())
    # every rank's own [gradient || loss] of the last pass, gathered so that rank 0 can show the all-reduce summed them
    parts = D.gather_to_rank0(local_part) if world > 1 else None

    if rank == 0:
        kernels, dom = kernel_records(w)
        k = kernels[dom]
        value = w.units * args.steps * world / elapsed
        out = {
            "metric": "differentiable cell-steps/s (fwd+bwd)",
            "value": value,
            "unit": "cell-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "dtype_detail": "float32 state widened to double for the step, Jacobian tape and adjoint in float32 (the reference's ladder)",
            "data": "synthetic",
            "config": {"workload": w.name, "lanes_per_gpu": L, "units_per_lane": N, "time_steps": T,
                       "parallelism": "%s sharded over %d GPU(s), no data-path collective; one all-reduce of [%s] per pass"
                                      % ("replicas" if shared_grad else "lanes", world,
                                         "d loss / d action || loss" if shared_grad else "loss")},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": k["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": k["frac_of_peak"], "traffic": pmc_traffic(w, dom), "traffic_unit": "bytes per launch (PMC)",
                         "moved_bytes_per_launch": w.moved_bytes_per_launch(),
                         "algorithmic_bytes_per_launch": w.units * w.unit_bytes,
                         "achieved_algorithmic": k["algorithmic_GBps"], "frac_algorithmic": k["algorithmic_GBps"] / HBM_PEAK_GBS,
                         "limiter": w.limiter.get(dom, "hbm"),
                         "note": "achieved = bytes the kernel moves (its compact tape: the same information as the reference's dqs "
                                 "blocks, which the reverse sweep rebuilds) / HIP-event time of the launch; *_algorithmic = the "
                                 "reference's tape bytes (48 B per cell-step, 32 B per vehicle-step) / the same time"},
            "whole_path": {"moved_GBps": 2 * w.moved_bytes_per_launch() * args.steps / elapsed / 1e9,
                           "frac_of_peak": 2 * w.moved_bytes_per_launch() * args.steps / elapsed / 1e9 / HBM_PEAK_GBS,
                           "algorithmic_GBps": w.units * 2 * w.unit_bytes * args.steps / elapsed / 1e9},
            "kernels": kernels,
            "loss_last_pass": flat.tolist()[-1],          # summed over ranks by the all-reduce
        }
        if parts is not None:
            out["allreduce_check"] = {"reduced": flat.tolist()[-1], "sum_of_rank_parts": float(parts[:, -1].double().sum()),
                                      "rank_parts": parts[:, -1].tolist(),
                                      "grad_max_abs_diff": float((parts[:, :-1].double().sum(dim=0) - flat[:-1].double().cpu()).abs().max())
                                      if shared_grad else 0.0}
        if world == 1:
            if args.workload == "macro" and not args.lanes and not args.cells and not args.time_steps and not args.no_also:
                del w.tape            # 17 GB back to the allocator before the other workloads take theirs
                torch.cuda.empty_cache()
                out["also"] = [also_record("micro", dev), also_record("itscp_hybrid", dev)]
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = w.cpu_baseline()
        print(json.dumps(out))
    D.barrier()


if __name__ == "__main__":
    main()
