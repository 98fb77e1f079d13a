#!/usr/bin/env python3
"""Differentiable traffic-signal control on the itscp environment, gradient arm: the harness counterpart of the reference's
example/control/itscp/run.py (flags of run_itscp_macro.sh / run_itscp_hybrid.sh) with the signal schedule itself as the
optimisation variable.

Every iteration calls the reference's entry point ItscpEnv.step(action, True); on this build the whole differentiable episode
(signals -> boundaries -> lane steps -> hand-offs -> queue loss, and its reverse sweep) runs in two fused kernel launches
(dhts_net_macro_rollout_* / dhts_net_hybrid_rollout_*).  PyTorch keeps the optimiser (Adam on the actions, clamped to [0, 1]).
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diff-hybrid-traffic-sim_amd"))

import numpy as np  # noqa: E402
import torch as th  # noqa: E402

from example.control.itscp._env import ItscpEnv  # noqa: E402
from example.control.itscp import problem as P  # noqa: E402


def main():
    ap = argparse.ArgumentParser("itscp: gradient-based signal control (MI355X)")
    ap.add_argument("--mode", choices=["macro", "hybrid"], default="hybrid")
    ap.add_argument("--problem", type=int, default=1)
    ap.add_argument("--n_intersection", type=int, default=3)
    ap.add_argument("--n_lane", type=int, default=1)
    ap.add_argument("--lane_length", type=float, default=5.0)
    ap.add_argument("--speed_limit", type=float, default=60.0)
    ap.add_argument("--simulation_length", type=int, default=20)
    ap.add_argument("--signal_length", type=int, default=4)
    ap.add_argument("--n_episode", type=int, default=100)
    ap.add_argument("--lr", type=float, default=1e-2)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--n_replica", type=int, default=1,
                    help="> 1: optimise that many random initial schedules of the same problem at once (one workgroup each in the "
                         "fused kernels) and report the best")
    ap.add_argument("--graph", action="store_true",
                    help="with --n_replica > 1: capture one whole iteration (both fused launches, Adam step, clamp, log) in a HIP "
                         "graph and replay it -- nothing is read back inside an iteration (check_faults=False)")
    args = ap.parse_args()
    assert th.cuda.is_available(), "needs a GPU (no CPU fallback)"
    dev = th.device("cuda")
    env = ItscpEnv()
    env.schedule_callback = getattr(P, "problem_%d" % args.problem)
    for k, v in dict(num_intersection=args.n_intersection, lane_length=args.lane_length, num_lane=args.n_lane,
                     policy_length=args.simulation_length, signal_length=args.signal_length, mode=args.mode,
                     speed_limit=args.speed_limit, random_seed=args.seed).items():
        env.config[k] = v
    env.reset()
    if args.n_replica > 1:
        return batch(args, env, dev)
    cache = None
    rng = np.random.default_rng(args.seed)
    action = th.tensor(rng.uniform(0.3, 0.7, env.action_size()).astype(np.float32), device=dev, requires_grad=True)
    opt = th.optim.Adam([action], lr=args.lr)
    t0 = time.time()
    for ep in range(args.n_episode):
        # a fresh episode of the same problem: keep the drawn schedules / routes (and the uploaded tables), rewind the env
        env.steps = 0
        env.time = 0
        env._fused_done = False
        env._fused_cache = cache if cache is not None else env._fused_cache
        opt.zero_grad()
        _, reward, _, _ = env.step(action, True)
        cache = env._fused_cache
        (-reward).backward()
        opt.step()
        with th.no_grad():
            action.clamp_(0.0, 1.0)
        if ep % 10 == 0 or ep == args.n_episode - 1:
            print("episode %4d  reward %.6f  (%.1f ms / episode)" % (ep, float(reward.detach()), 1e3 * (time.time() - t0) / (ep + 1)))


def batch(args, env, dev):
    """Many restarts in one launch: the network tables are shared, every replica owns an action vector."""
    from dhts import ops
    from dhts.network import HybridNetworkTables
    tab = HybridNetworkTables.from_env(env)
    routes = []
    for l in range(tab.n_lanes):
        if tab.lane_macro[l] == 0 and any(tab.lane_macro[a] for a in tab.prev_lanes[l]):
            for _ in range(8):
                r = list(env.simulator.create_random_route(l).route)[:32]
                routes.append(r + [-1] * (32 - len(r)))
    dev_tab = ops.DeviceHybridTables(tab, np.asarray(routes if routes else [[-1, -1]], dtype=np.int32), dev)
    sim_args = (env.num_intersection ** 2, env.config["signal_length"] * env.config["simulation_frequency"],
                1.0 / env.config["simulation_frequency"], args.speed_limit)
    gen = th.Generator(device="cpu").manual_seed(args.seed)
    action = (0.1 + 0.8 * th.rand(args.n_replica, env.action_size(), generator=gen)).to(dev).requires_grad_(True)
    opt = th.optim.Adam([action], lr=args.lr, capturable=args.graph)
    action.grad = th.zeros_like(action)
    # Sticky fault records, allocated outside any graph capture and read every so many episodes (a read synchronises).  The
    # forward's faults (CFL violation: the reference asserts, _macro_lane.py:141-146; an exhausted kernel capacity) end the run;
    # the reverse sweep's NaN fault is the one thing a batch tolerates (that replica sits the episode out), so it gets its own
    # record -- the first fault wins a record, and a tolerated NaN must not hide a later CFL violation.
    err_fwd, err_bwd = ops.new_error_record(dev), ops.new_error_record(dev)
    n_bad_sweeps = [0]

    def check_faults():
        ops.raise_on_fault(err_fwd)
        if err_bwd.tolist()[0] != 0:
            n_bad_sweeps[0] += 1
            err_bwd.zero_()

    def iteration():
        reward, _, _, _ = ops.net_hybrid_rollout(action, dev_tab, *sim_args, check_faults=False, err=err_fwd, err_bwd=err_bwd)
        opt.zero_grad(set_to_none=False)
        (-reward.sum()).backward()
        action.grad.nan_to_num_(0.0, 0.0, 0.0)       # a replica whose reverse sweep hit 0 * inf sits this episode out
        opt.step()
        with th.no_grad():
            action.clamp_(0.0, 1.0)
        r = reward.detach()
        return th.stack([r.max(), r.mean(), r.min()])

    def report(ep, stats, t0):
        print("episode %4d  reward best %.6f  mean %.6f  worst %.6f  (%.2f ms / episode of %d replicas)" % (
            ep, stats[0], stats[1], stats[2], 1e3 * (time.time() - t0) / (ep + 1), args.n_replica))

    if args.graph:
        log = th.zeros(args.n_episode + 4, 3, device=dev)
        slot = th.zeros(1, dtype=th.long, device=dev)
        keep = action.detach().clone()
        side = th.cuda.Stream()
        side.wait_stream(th.cuda.current_stream())
        with th.cuda.stream(side):                       # warm-up outside the capture (allocator, Adam state)
            for _ in range(3):
                iteration()
        th.cuda.current_stream().wait_stream(side)
        with th.no_grad():                               # the warm-up iterations do not count
            action.copy_(keep)
            for st in opt.state.values():
                for v_ in st.values():
                    if isinstance(v_, th.Tensor):
                        v_.zero_()
        graph = th.cuda.CUDAGraph()
        with th.cuda.graph(graph):
            stats = iteration()
            log.index_copy_(0, slot, stats.reshape(1, 3))
            slot.add_(1)
        th.cuda.synchronize()
        t0 = time.time()
        for ep in range(args.n_episode):
            graph.replay()
            if ep % 25 == 24:
                check_faults()
        th.cuda.synchronize()
        check_faults()
        rows = log[:args.n_episode].tolist()
        for ep in sorted(set(list(range(0, args.n_episode, 10)) + [args.n_episode - 1])):
            print("episode %4d  reward best %.6f  mean %.6f  worst %.6f" % (ep, rows[ep][0], rows[ep][1], rows[ep][2]))
        print("%.2f ms / episode of %d replicas (HIP graph replay)" % (1e3 * (time.time() - t0) / args.n_episode, args.n_replica))
        return
    t0 = time.time()
    for ep in range(args.n_episode):
        stats = iteration()
        if ep % 10 == 0 or ep == args.n_episode - 1:
            check_faults()
            report(ep, stats.tolist(), t0)
    if n_bad_sweeps[0]:
        print("%d fault check(s) found a non-finite reverse sweep (those replicas' gradients were dropped)" % n_bad_sweeps[0])


if __name__ == "__main__":
    main()
