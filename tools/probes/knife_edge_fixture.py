#!/usr/bin/env python3
"""itscp_hybrid.npz (the first 600-step reference run of BASELINE config 4's episode): d reward / d action of the fused kernels
against the reference's, full horizon and restricted to the first t0 steps -- and against the oracle's."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "diff-hybrid-traffic-sim_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from oracle import oracle as O  # noqa: E402
from test_oracle_golden import itscp_hybrid_tables  # noqa: E402
from dhts import ops  # noqa: E402
from dhts.network import group_routes  # noqa: E402

dev = torch.device("cuda", 0)
g = np.load(os.path.join(ROOT, "tests", "golden", "itscp_hybrid.npz"))
t, m = itscp_hybrid_tables(g)
routes, ptr = group_routes(g["spawn_routes"], t.n_lanes)
args = (m["num_intersection"] ** 2, m["simulation_frequency"] * m["signal_length"], 1.0 / m["simulation_frequency"], m["speed_limit"],
        m["static_speed"], m["vehicle_length"])
scale = np.abs(g["g_action"]).max()
tab = ops.DeviceHybridTables(t, g["spawn_routes"], dev)


def kern(loss_steps=0):
    a = torch.tensor(g["action"][None, :], device=dev, requires_grad=True)
    cut, reward, queue, counts = ops.net_hybrid_rollout(a, tab, *args, loss_steps)
    cut.sum().backward()
    return a.grad[0].cpu().numpy(), float(reward[0])


gk, rk = kern()
o = O.net_hybrid(t, routes, ptr, g["action"], *args)
print("full horizon: kernels vs reference %.3e   oracle vs reference %.3e   kernels vs oracle %.3e   reward rel %.2e" % (
    np.abs(gk - g["g_action"]).max() / scale, np.abs(o["g_action"] - g["g_action"]).max() / scale, np.abs(gk - o["g_action"]).max() / scale,
    abs(rk - float(g["reward"])) / abs(float(g["reward"]))))
for t0, ref in zip(g["g_action_cut_steps"], g["g_action_cut"]):
    gc, _ = kern(int(t0))
    oc = O.net_hybrid(t, routes, ptr, g["action"], *args, t_cut=int(t0))
    print("t0 = %3d: kernels vs reference %.3e   oracle vs reference %.3e   kernels vs oracle %.3e" % (
        t0, np.abs(gc - ref).max() / scale, np.abs(oc["g_action"] - ref).max() / scale, np.abs(gc - oc["g_action"]).max() / scale))
