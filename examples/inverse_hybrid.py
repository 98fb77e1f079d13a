#!/usr/bin/env python3
"""Inverse problem on the hybrid three-lane network, gradient-descent arm: the harness counterpart of the reference's
example/inverse/hybrid.py + _inverse.solve_gd (hybrid.py:14-254, _inverse.py:68-99,185-242).

Network: macro lane 0 -> micro lane 1 -> macro lane 2 (hybrid.py:37-82).  The unknown is lane 0's initial (density, speed);
the target is lane 0's state after n_timestep steps of RoadNetwork.forward, during which lane 0's outflow fills the flux
capacitor, vehicles are spawned on lane 1, driven by the IDM operator and handed to lane 2 (road/network/conversion.py).
Same flags as the reference script (--n_trial --n_cell --n_timestep --cell_length --speed_limit --delta_time --n_episode);
Adam lr 1e-3, clamp to the bounds after every step, one log line "{beg_error} {end_error}" per episode in
result/inverse/<run>/gd/trial_<k>.txt (_inverse.py:504-514).

An episode = ONE launch each way of the fused hybrid network kernels (dhts.ops.net_hybrid_state_rollout: the network starts
from the estimate, the loss taps the final state; round 4).  `--lane_by_lane` runs the same episodes through the drop-in
classes instead (road.network.road_network.RoadNetwork over dMacroLane / dMicroLane: one operator call per lane and step +
the conversions on the host, 3 T launches per episode) -- same numbers, the launch-bound way of using the kernels.
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diff-hybrid-traffic-sim_amd"))

import numpy as np  # noqa: E402
import torch as th  # noqa: E402

from dhts import device  # noqa: E402
from road.lane.dmacro_lane import dMacroLane  # noqa: E402
from road.lane.dmicro_lane import dMicroLane  # noqa: E402
from road.network.road_network import RoadNetwork  # noqa: E402


def build_network(n_cell, cell_length, speed_limit, bd_r, bd_u):
    length = n_cell * cell_length
    net = RoadNetwork(speed_limit)
    first = dMacroLane(0, length, speed_limit, cell_length)
    first.set_leftmost_cell(bd_r[0], bd_u[0])
    first.set_rightmost_cell(bd_r[1], bd_u[1])
    net.add_lane(first)
    net.add_lane(dMicroLane(1, length, speed_limit))
    last = dMacroLane(2, length, speed_limit, cell_length)
    last.set_leftmost_cell(bd_r[2], bd_u[2])
    last.set_rightmost_cell(bd_r[3], bd_u[3])
    net.add_lane(last)
    net.connect_lane(0, 1)
    net.connect_lane(1, 2)
    net.macro_route = net.create_random_macro_route()
    return net


def clear_network(net):
    """_inverse.py:312-326: lanes back to empty, vehicles and routes forgotten."""
    for lane in net.lane.values():
        lane.clear()
    net.vehicle.clear()
    net.micro_route.clear()
    net.num_vehicle = 0


def rollout(net, state, n_timestep, dt, differentiable):
    clear_network(net)
    net.lane[0].set_state_vector_u(state[0], state[1])
    for _ in range(n_timestep):
        net.forward(dt, differentiable)
    r, _, u = net.lane[0].get_state_vector()
    return r, u


class FusedNetwork:
    """The same network as tables for the fused kernels: lane 0 starts from the state handed to rollout(), lanes 1 and 2
    empty, the four stored ghosts as build_network sets them."""

    def __init__(self, n_cell, cell_length, speed_limit, bd_r, bd_u, n_timestep, dev):
        from dhts import ops
        from dhts.network import HybridNetworkTables
        self.ops, self.N, self.um, self.dev = ops, n_cell, speed_limit, dev
        length = n_cell * cell_length
        tab = HybridNetworkTables.plain([1, 0, 1], [n_cell, 0, n_cell], [length] * 3, [(0, 1), (1, 2)], n_timestep, macro_route=[1, -1, -1])
        self.tab = ops.DeviceHybridTables(tab, np.array([[1, 2]], dtype=np.int32), dev)      # a spawned vehicle's route: lane 1, lane 2
        g = th.zeros(1, 3, 4, device=dev)
        g[0, 0] = th.stack([bd_r[0], bd_u[0], bd_r[1], bd_u[1]])
        g[0, 1] = th.tensor([0.0, speed_limit, 0.0, speed_limit], device=dev)
        g[0, 2] = th.stack([bd_r[2], bd_u[2], bd_r[3], bd_u[3]])
        self.ghost0 = g
        self.num_vehicle = 0

    def rollout(self, state, dt):
        N, um, dev = self.N, self.um, self.dev
        r_all = th.cat([state[0], th.zeros(N, device=dev)])[None]
        u_all = th.cat([state[1], th.full((N,), um, device=dev)])[None]
        rT, _, uT, _, _, counts = self.ops.net_hybrid_state_rollout(r_all, u_all, self.tab, dt, um, ghost0=self.ghost0, plain=True)
        self.num_vehicle = int(counts[0, 0])
        return rT[0, :N], uT[0, :N]


def sq_error(a, b):
    return ((a[0] - b[0]) ** 2.0).sum() + ((a[1] - b[1]) ** 2.0).sum()


def main():
    ap = argparse.ArgumentParser("Inverse problem in hybrid traffic simulation (gradient descent, MI355X)")
    ap.add_argument("--n_trial", type=int, default=1)
    ap.add_argument("--n_cell", type=int, default=10)
    ap.add_argument("--n_timestep", type=int, default=500)
    ap.add_argument("--cell_length", type=float, default=5.0)
    ap.add_argument("--speed_limit", type=float, default=30.0)
    ap.add_argument("--delta_time", type=float, default=0.01)
    ap.add_argument("--n_episode", type=int, default=100)
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--run_name", default=None)
    ap.add_argument("--lane_by_lane", action="store_true", help="step the drop-in classes (3 T launches per episode) instead of the fused kernels")
    args = ap.parse_args()
    dev = device.get()
    if args.seed is not None:
        th.manual_seed(args.seed)
        np.random.seed(args.seed)
    N, T, um = args.n_cell, args.n_timestep, args.speed_limit
    run = args.run_name or "hybrid_{}".format(time.strftime("%Y%m%d_%H%M%S"))
    log_dir = os.path.join("result", "inverse", run, "gd")
    os.makedirs(log_dir, exist_ok=True)

    for trial in range(args.n_trial):
        bd_r = th.rand(4, device=dev)
        bd_u = th.rand(4, device=dev) * um
        if args.lane_by_lane:
            net = build_network(N, args.cell_length, um, bd_r, bd_u)
            run_episode = lambda state, diff: rollout(net, state, T, args.delta_time, diff)      # noqa: E731
        else:
            net = FusedNetwork(N, args.cell_length, um, bd_r, bd_u, T, dev)
            run_episode = lambda state, diff: net.rollout(state, args.delta_time)                # noqa: E731
        truth = (th.rand(N, device=dev), th.rand(N, device=dev) * um)
        with th.no_grad():
            target = tuple(x.detach().clone() for x in run_episode(truth, False))
        spawned_truth = net.num_vehicle
        est = ((truth[0] + th.randn(N, device=dev) * 1e-2).clamp(0.0, 1.0).requires_grad_(True),
               (truth[1] + th.randn(N, device=dev) * 1e-2).clamp(0.0, um).requires_grad_(True))
        opt = th.optim.Adam(est, lr=1e-3)
        lines = []
        t0 = time.time()
        for ep in range(args.n_episode):
            end_state = run_episode(est, True)
            beg = sq_error(truth, est)
            end = sq_error(target, end_state)
            lines.append("{} {}\n".format(beg.item(), end.item()))
            if not end.requires_grad:
                raise SystemExit("no gradient reaches the estimate (hybrid.py / _inverse.py:225-231)")
            opt.zero_grad()
            end.backward()
            opt.step()
            with th.no_grad():
                est[0].clamp_(0.0, 1.0)
                est[1].clamp_(0.0, um)
        dt_wall = time.time() - t0
        with open(os.path.join(log_dir, "trial_{}.txt".format(trial)), "w") as f:
            f.writelines(lines)
        first, last = lines[0].split(), lines[-1].split()
        print("Trial # {}: end error {:.6f} -> {:.6f} in {} episodes ({} vehicles spawned in the truth run), {:.2f} s = {:.2f} ms per "
              "episode ({:.0f} lane-steps/s, {})".format(
                  trial, float(first[1]), float(last[1]), args.n_episode, spawned_truth, dt_wall, 1e3 * dt_wall / max(args.n_episode, 1),
                  3 * T * args.n_episode / dt_wall,
                  "lane by lane: 3 T operator launches per episode" if args.lane_by_lane else "fused: one launch each way per episode"))


if __name__ == "__main__":
    main()
