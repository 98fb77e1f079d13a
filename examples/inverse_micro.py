#!/usr/bin/env python3
"""Inverse problem on microscopic lanes, gradient-descent arm: the harness counterpart of the reference's
example/inverse/micro.py + _inverse.solve_gd (micro.py:36-236, _inverse.py:185-242), on the fused HIP path.

Same flags as the reference script (--n_trial --n_vehicle --n_timestep --vehicle_length --speed_limit --delta_time
--n_episode) plus --n_lane.  Vehicles at 4 len spacing + U[0, 2 len) jitter, v ~ lerp(0.3, 0.7) u_max,
default_micro_vehicle parameters, lane length 1e10 (nobody leaves), head gap = lane defaults (1000, 0);
loss = sum (p - p*)^2 + sum (v - v*)^2 at t = T; Adam lr 1e-2 (micro.py:264); positions clamped to their
[4 i len, 4 i len + 2 len] boxes, speeds to [0, u_max]; log lines "{beg_error} {end_error}".
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
from road.vehicle.micro_vehicle import MicroVehicle  # noqa: E402


def main():
    ap = argparse.ArgumentParser("Inverse problem in microscopic traffic simulation (gradient descent, MI355X)")
    ap.add_argument("--n_trial", type=int, default=1)
    ap.add_argument("--n_vehicle", type=int, default=10)
    ap.add_argument("--n_timestep", type=int, default=500)
    ap.add_argument("--vehicle_length", type=float, default=5.0)
    ap.add_argument("--speed_limit", type=float, default=30.0)
    ap.add_argument("--delta_time", type=float, default=0.01)
    ap.add_argument("--n_episode", type=int, default=100)
    ap.add_argument("--n_lane", type=int, default=1)
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--run_name", default=None)
    args = ap.parse_args()

    rank, world, local = D.init()
    th.cuda.set_device(local)
    dev = th.device("cuda", local)
    if args.seed is not None:
        th.manual_seed(args.seed + rank)
    b, e = D.shard_range(args.n_lane, rank, world)
    L, V, T, um, ln = e - b, args.n_vehicle, args.n_timestep, args.speed_limit, args.vehicle_length
    run = args.run_name or "micro_{}".format(time.strftime("%Y%m%d_%H%M%S"))
    log_dir = os.path.join("result", "inverse", run, "gd")
    if rank == 0:
        os.makedirs(log_dir, exist_ok=True)

    par = th.tensor(MicroVehicle.default_micro_vehicle(um).params(), dtype=th.float64, device=dev)
    params = par[:, None, None].expand(6, L, V).contiguous()
    head = th.tensor([[1000.0, 0.0]], dtype=th.float64, device=dev).expand(L, 2).contiguous()
    p_lb = (th.arange(V, device=dev) * 4.0 * ln)[None, :].expand(L, V)
    p_ub = p_lb + 2.0 * ln

    for trial in range(args.n_trial):
        p_true = p_lb + th.rand(L, V, device=dev) * 2.0 * ln
        v_true = th.lerp(th.tensor(0.3 * um, device=dev), th.tensor(0.7 * um, device=dev), th.rand(L, V, device=dev))
        with th.no_grad():
            p_tgt, v_tgt = dhts.micro_rollout(p_true, v_true, params, head, T, args.delta_time)
        p_est = th.max(th.min(p_true + th.randn(L, V, device=dev) * 0.1 * ln, p_ub), p_lb).requires_grad_(True)
        v_est = (v_true + th.randn(L, V, device=dev) * 1e-2 * um).clamp(0.0, um).requires_grad_(True)
        opt = th.optim.Adam([p_est, v_est], lr=1e-2)
        lines = []
        t0 = time.time()
        for ep in range(args.n_episode):
            pT, vT = dhts.micro_rollout(p_est, v_est, params, head, T, args.delta_time)
            beg = ((p_est - p_true) ** 2).sum() + ((v_est - v_true) ** 2).sum()
            end = ((pT - p_tgt) ** 2).sum() + ((vT - v_tgt) ** 2).sum()
            opt.zero_grad()
            end.backward()
            opt.step()
            with th.no_grad():
                p_est.copy_(th.max(th.min(p_est, p_ub), p_lb))
                v_est.clamp_(0.0, um)
            flat = th.stack([beg.detach(), end.detach()]).float()
            D.allreduce_sum_(flat)
            lines.append("{} {}\n".format(flat[0].item(), flat[1].item()))
        th.cuda.synchronize()
        dt_wall = time.time() - t0
        if rank == 0:
            with open(os.path.join(log_dir, "trial_{}.txt".format(trial)), "w") as f:
                f.writelines(lines)
            first, last = lines[0].split(), lines[-1].split()
            print("Trial # {}: end error {:.6f} -> {:.6f} in {} episodes, {:.2f} s ({:.3e} differentiable vehicle-steps/s)".format(
                trial, float(first[1]), float(last[1]), args.n_episode, dt_wall,
                args.n_lane * V * T * args.n_episode / dt_wall))


if __name__ == "__main__":
    main()
