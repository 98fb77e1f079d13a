"""Threads per workgroup of the persistent kernels (DHTS_OPT_NETSTEP_BLOCK): differentiable / evaluation episode times at 256, 512 and
1 024 threads over the goldens above the fused limits and two inside them.  Run on the GPU box: python tools/probes/exp_persist_block.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-hybrid-traffic-sim_amd"), os.path.join(ROOT, "tests")]
from test_oracle_golden import itscp_hybrid_tables, itscp_micro_tables      # noqa: E402
from dhts import _lib      # noqa: E402
from dhts.stepwise import StepwiseNetwork, default_lane_capacity      # noqa: E402

cuda = torch.device("cuda:0")
G = os.path.join(ROOT, "tests", "golden")


def args_of(m):
    return (m["num_intersection"] ** 2, m["simulation_frequency"] * m["signal_length"], 1.0 / m["simulation_frequency"], m["speed_limit"],
            m["static_speed"], m["vehicle_length"])


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


for name in ("hybrid_n2l30", "hybrid_5x5", "micro_2x2", "macro_3x3x3", "hybrid_half", "hybrid_p2", "macro_small", "micro_small", "hybrid_l30"):
    g = np.load(os.path.join(G, "itscp_%s.npz" % name))
    if name.startswith("micro"):
        t, m, routes = itscp_micro_tables(g)
    else:
        t, m = itscp_hybrid_tables(g)
        routes = g["spawn_routes"] if g["spawn_routes"].shape[0] else -np.ones((1, 2), np.int32)
    a = torch.tensor(g["action"], device=cuda, requires_grad=True)
    net = StepwiseNetwork(t, routes, cuda, lane_capacity=default_lane_capacity(t, m["vehicle_length"]), persistent=True)

    def episode():
        a.grad = None
        cut, _, _, _ = net.rollout(a, *args_of(m), check_faults=False)
        cut.backward()
    out = []
    for block in (256, 512, 1024):
        _lib.lib().dhts_set_option(_lib.OPT_NETSTEP_BLOCK, block)
        out.append("%d: %6.2f ms" % (block, 1e3 * timed(episode)))
    _lib.lib().dhts_set_option(_lib.OPT_NETSTEP_BLOCK, 0)
    print("%-14s %4d lanes %5d cells %4d IDM lanes %4d steps | %s" % (name, t.n_lanes, t.n_cells, net.n_micro, t.T, " | ".join(out)), flush=True)
