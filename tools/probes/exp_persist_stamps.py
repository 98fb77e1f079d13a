"""Phase stamps of the persistent stepwise kernels (a -DDHTS_NS_STAMPS build of netstep_hybrid.hip: s_memtime at the phase boundaries of
thread 0 of replica 0, printed by the kernels).  SRC=netstep_hybrid tools/build_variants.sh stamps:"-DDHTS_NS_STAMPS", then on the GPU box
DHTS_LIB=.../variants/libdhts_stamps.so python tools/probes/exp_persist_stamps.py [golden name]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-hybrid-traffic-sim_amd"), os.path.join(ROOT, "tests")]
from test_oracle_golden import itscp_hybrid_tables      # noqa: E402
from dhts.stepwise import StepwiseNetwork, default_lane_capacity      # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "hybrid_n2l30"
g = np.load(os.path.join(ROOT, "tests", "golden", "itscp_%s.npz" % name))
t, m = itscp_hybrid_tables(g)
cuda = torch.device("cuda:0")
routes = g["spawn_routes"] if g["spawn_routes"].shape[0] else -np.ones((1, 2), np.int32)
net = StepwiseNetwork(t, routes, cuda, lane_capacity=default_lane_capacity(t, m["vehicle_length"]), persistent=True)
args = (m["num_intersection"] ** 2, m["simulation_frequency"] * m["signal_length"], 1.0 / m["simulation_frequency"], m["speed_limit"], m["static_speed"],
        m["vehicle_length"])
print(name, t.n_lanes, "lanes", t.n_cells, "cells", net.n_micro, "IDM lanes", t.T, "steps")
for _ in range(2):
    a = torch.tensor(g["action"], device=cuda, requires_grad=True)
    cut, _, _, _ = net.rollout(a, *args)
    cut.backward()
    torch.cuda.synchronize()
