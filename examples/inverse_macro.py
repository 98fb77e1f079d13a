#!/usr/bin/env python3
"""Inverse problem on macroscopic lanes, gradient-descent arm: the harness counterpart of the reference's
example/inverse/macro.py + _inverse.solve_gd (macro.py:34-241, _inverse.py:68-99,185-242), on the fused HIP path.

Same flags as the reference script (--n_trial --n_cell --n_timestep --cell_length --speed_limit --delta_time
--n_episode), plus --n_lane: every trial solves n_lane independent single-lane problems at once (the reference's
problem is n_lane = 1).  Truth state r ~ U[0,1], u ~ U[0,u_max], ghosts likewise; estimate = truth + N(0, 1e-2)
clamped; loss = sum (r - r*)^2 + sum (u - u*)^2 at t = T; Adam lr 1e-3; clamp to the bounds after every step; one log
line "{beg_error} {end_error}" per episode in result/inverse/<run>/gd/trial_<k>.txt (_inverse.py:504-514).
PyTorch does the optimiser step; every simulated step (forward and adjoint) runs in the HIP kernels.
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diff-hybrid-traffic-sim_amd"))

import torch as th  # noqa: E402

import dhts  # noqa: E402
from dhts import dist as D  # noqa: E402


def main():
    ap = argparse.ArgumentParser("Inverse problem in macroscopic traffic simulation (gradient descent, MI355X)")
    ap.add_argument("--n_trial", type=int, default=1)
    ap.add_argument("--n_cell", type=int, default=10)
    ap.add_argument("--n_timestep", type=int, default=500)
    ap.add_argument("--cell_length", type=float, default=5.0)
    ap.add_argument("--speed_limit", type=float, default=30.0)
    ap.add_argument("--delta_time", type=float, default=0.01)
    ap.add_argument("--n_episode", type=int, default=100)
    ap.add_argument("--n_lane", type=int, default=1)
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--run_name", default=None)
    ap.add_argument("--graph", action="store_true",
                    help="capture one whole episode (rollout, loss, adjoint, Adam step, clamp) in a HIP graph and replay it: "
                         "for small problems the episode is launch-bound (single GPU only)")
    args = ap.parse_args()

    rank, world, local = D.init()
    th.cuda.set_device(local)
    dev = th.device("cuda", local)
    if args.seed is not None:
        th.manual_seed(args.seed + rank)
    b, e = D.shard_range(args.n_lane, rank, world)          # independent problems shard over GPUs
    L, N, T, um = e - b, args.n_cell, args.n_timestep, args.speed_limit
    run = args.run_name or "macro_{}".format(time.strftime("%Y%m%d_%H%M%S"))
    log_dir = os.path.join("result", "inverse", run, "gd")
    if rank == 0:
        os.makedirs(log_dir, exist_ok=True)

    for trial in range(args.n_trial):
        ghost_r = th.rand(L, 2, device=dev)
        ghost_u = th.rand(L, 2, device=dev) * um
        r_true = th.rand(L, N, device=dev)
        u_true = th.rand(L, N, device=dev) * um
        with th.no_grad():
            r_tgt, _, u_tgt, _ = dhts.macro_rollout(r_true, u_true, ghost_r, ghost_u, T, args.delta_time, args.cell_length, um)
        r_est = (r_true + th.randn(L, N, device=dev) * 1e-2).clamp(0.0, 1.0).requires_grad_(True)
        u_est = (u_true + th.randn(L, N, device=dev) * 1e-2).clamp(0.0, um).requires_grad_(True)
        opt = th.optim.Adam([r_est, u_est], lr=1e-3, capturable=args.graph)
        lines = []

        def episode(check_faults):
            rT, _, uT, _ = dhts.macro_rollout(r_est, u_est, ghost_r, ghost_u, T, args.delta_time, args.cell_length, um,
                                              check_faults=check_faults)
            beg = ((r_est - r_true) ** 2).sum() + ((u_est - u_true) ** 2).sum()
            end = ((rT - r_tgt) ** 2).sum() + ((uT - u_tgt) ** 2).sum()
            opt.zero_grad(set_to_none=False)
            end.backward()
            opt.step()
            with th.no_grad():
                r_est.clamp_(0.0, 1.0)
                u_est.clamp_(0.0, um)
            return th.stack([beg.detach(), end.detach()]).float()

        if args.graph:
            assert world == 1, "--graph is a single-GPU mode"
            log = th.zeros(args.n_episode + 4, 2, device=dev)
            slot = th.zeros(1, dtype=th.long, device=dev)
            for p_ in (r_est, u_est):
                p_.grad = th.zeros_like(p_)
            keep = (r_est.detach().clone(), u_est.detach().clone())
            side = th.cuda.Stream()
            side.wait_stream(th.cuda.current_stream())
            with th.cuda.stream(side):                       # warm-up outside the capture (allocator, Adam state)
                for _ in range(3):
                    episode(False)
            th.cuda.current_stream().wait_stream(side)
            with th.no_grad():                               # the warm-up episodes do not count
                r_est.copy_(keep[0]); u_est.copy_(keep[1])
            opt = th.optim.Adam([r_est, u_est], lr=1e-3, capturable=True)
            side.wait_stream(th.cuda.current_stream())
            with th.cuda.stream(side):
                episode(False)                               # creates the new optimiser's state tensors
            th.cuda.current_stream().wait_stream(side)
            with th.no_grad():
                r_est.copy_(keep[0]); u_est.copy_(keep[1])
                for st in opt.state.values():
                    for v_ in st.values():
                        if isinstance(v_, th.Tensor):
                            v_.zero_()
            graph = th.cuda.CUDAGraph()
            with th.cuda.graph(graph):
                flat = episode(False)
                log.index_copy_(0, slot, flat.reshape(1, 2))
                slot.add_(1)
            th.cuda.synchronize()
            t0 = time.time()
            for ep in range(args.n_episode):
                graph.replay()
            th.cuda.synchronize()
            lines = ["{} {}\n".format(a, b) for a, b in log[:args.n_episode].tolist()]
        else:
            t0 = time.time()
            for ep in range(args.n_episode):
                flat = episode(True)
                D.allreduce_sum_(flat)                        # every lane owns its unknowns: only the errors are reduced
                lines.append("{} {}\n".format(flat[0].item(), flat[1].item()))
        th.cuda.synchronize()
        dt_wall = time.time() - t0
        if rank == 0:
            with open(os.path.join(log_dir, "trial_{}.txt".format(trial)), "w") as f:
                f.writelines(lines)
            first, last = lines[0].split(), lines[-1].split()
            print("Trial # {}: end error {:.6f} -> {:.6f} in {} episodes, {:.2f} s ({:.3e} differentiable cell-steps/s)".format(
                trial, float(first[1]), float(last[1]), args.n_episode, dt_wall,
                args.n_lane * N * T * args.n_episode / dt_wall))


if __name__ == "__main__":
    main()
