import os, sys
import numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-hybrid-traffic-sim_amd"), os.path.join(ROOT, "tests")]
from test_oracle_golden import itscp_hybrid_tables
from dhts.stepwise import StepwiseNetwork, default_lane_capacity
from dhts import _lib
g = np.load(os.path.join(ROOT, "tests", "golden", "itscp_hybrid_n2l30.npz"))
t, m = itscp_hybrid_tables(g)
cuda = torch.device("cuda:0")
net = StepwiseNetwork(t, g["spawn_routes"], cuda, lane_capacity=default_lane_capacity(t, m["vehicle_length"]), persistent=True)
args = (m["num_intersection"] ** 2, m["simulation_frequency"] * m["signal_length"], 1.0 / m["simulation_frequency"], m["speed_limit"], m["static_speed"], m["vehicle_length"])
for kb in (0, 120, 96, 72, 56, 40, 24, 8):
    _lib.lib().dhts_set_option(_lib.OPT_NETSTEP_LDS_KB, kb)
    print("budget", kb, flush=True)
    a = torch.tensor(g["action"], device=cuda, requires_grad=True)
    cut, _, _, _ = net.rollout(a, *args)
    cut.backward()
    torch.cuda.synchronize()
