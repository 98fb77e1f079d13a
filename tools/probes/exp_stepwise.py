"""Timing of the stepwise device path (dhts/stepwise.py) on networks beyond one workgroup, beside the batched-lane path (macro).
Run on the GPU box: python tools/probes/exp_stepwise.py [reps]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-hybrid-traffic-sim_amd"), os.path.join(ROOT, "tests")]
from test_oracle_golden import itscp_hybrid_tables, itscp_micro_tables, itscp_tables, meta_of      # noqa: E402
from dhts.stepwise import StepwiseNetwork, default_lane_capacity      # noqa: E402
from dhts.batched import BatchedMacroNetwork      # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
cuda = torch.device("cuda:0")
G = os.path.join(ROOT, "tests", "golden")


def args_of(m):
    return (m["num_intersection"] ** 2, m["simulation_frequency"] * m["signal_length"], 1.0 / m["simulation_frequency"], m["speed_limit"],
            m["static_speed"], m["vehicle_length"])


def timed(fn):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


for name in ("hybrid_n2l30", "hybrid_5x5", "micro_2x2", "macro_3x3x3", "hybrid_half"):
    g = np.load(os.path.join(G, "itscp_%s.npz" % name))
    if name.startswith("micro"):
        t, m, routes = itscp_micro_tables(g)
    else:
        t, m = itscp_hybrid_tables(g)
        routes = g["spawn_routes"] if g["spawn_routes"].shape[0] else -np.ones((1, 2), np.int32)
    a = torch.tensor(g["action"], device=cuda, requires_grad=True)
    for persistent in (False, True):
        net = StepwiseNetwork(t, routes, cuda, lane_capacity=default_lane_capacity(t, m["vehicle_length"]), persistent=persistent)

        def episode():
            a.grad = None
            cut, _, _, _ = net.rollout(a, *args_of(m), check_faults=False)
            cut.backward()

        def evaluation():
            with torch.no_grad():
                net.rollout(a.detach(), *args_of(m), differentiable=False, check_faults=False)
        print("%-14s %4d lanes %5d cells %4d IDM lanes %4d steps, %-10s form: differentiable episode %6.2f ms, evaluation episode %6.2f ms" % (
            name, t.n_lanes, t.n_cells, net.n_micro, t.T, "persistent" if persistent else "stepwise", 1e3 * timed(episode), 1e3 * timed(evaluation)), flush=True)
    R = 64
    netr = StepwiseNetwork([t] * R, routes, cuda, lane_capacity=default_lane_capacity(t, m["vehicle_length"]), persistent=True)
    ar = torch.tensor(np.tile(g["action"][None], (R, 1)), device=cuda, requires_grad=True)

    def batch():
        ar.grad = None
        cut, _, _, _ = netr.rollout(ar, *args_of(m), check_faults=False)
        cut.sum().backward()
    print("%-14s persistent form, %d replicas in one launch pair: %.2f ms per differentiable batch episode" % (name, R, 1e3 * timed(batch)), flush=True)
    if name.startswith("macro"):
        tab, _ = itscp_tables(g)
        bn = BatchedMacroNetwork(tab, cuda)

        def ep_b():
            a.grad = None
            r, _ = bn.graphed_rollout(a, *args_of(m))
            r.backward()
        print("%-14s batched-lane path, replayed HIP graph: %.2f ms" % (name, 1e3 * timed(ep_b)), flush=True)
