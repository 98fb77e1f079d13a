"""Fuzz: persistent form against stepwise form on random signal schedules over every itscp fixture (queues and counts bit for bit, gradient
to 1e-6), at the geometric lane capacity and at 32.  Run on the GPU box: python tools/probes/fuzz_forms.py [actions per fixture]"""
import glob
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-hybrid-traffic-sim_amd"), os.path.join(ROOT, "tests")]
from test_oracle_golden import itscp_hybrid_tables, itscp_micro_tables, itscp_vehicle_params      # noqa: E402
from dhts import ops      # noqa: E402
from dhts.stepwise import StepwiseNetwork, default_lane_capacity      # noqa: E402

n_act = int(sys.argv[1]) if len(sys.argv) > 1 else 6
cuda = torch.device("cuda:0")
rng = np.random.default_rng(2026)
bad = 0
names = sorted(os.path.basename(f)[6:-4] for f in glob.glob(os.path.join(ROOT, "tests", "golden", "itscp_*.npz")))
for name in names:
    g = np.load(os.path.join(ROOT, "tests", "golden", "itscp_%s.npz" % name))
    vp = itscp_vehicle_params(g)           # (round 6: fixtures of seeded random_micro_vehicle runs carry the vehicles' attributes)
    if "micro" in name:
        t, m, routes = itscp_micro_tables(g)
        t.set_micro_sources(np.concatenate([g["rand_draws"], rng.random(8 * len(g["rand_draws"]) + 64)]))
        routes = np.concatenate([routes] * 3)
        vp = None if vp is None else np.concatenate([vp] * 3)
    else:
        t, m = itscp_hybrid_tables(g)
        routes = np.concatenate([g["spawn_routes"]] * 4) if g["spawn_routes"].shape[0] else -np.ones((1, 2), np.int32)
        vp = None if vp is None else np.concatenate([vp] * 4)
    args = (m["num_intersection"] ** 2, m["simulation_frequency"] * m["signal_length"], 1.0 / m["simulation_frequency"], m["speed_limit"],
            m["static_speed"], m["vehicle_length"])
    hard = name.startswith("eval")
    worst = 0.0
    skipped = 0
    for cap in (default_lane_capacity(t, m["vehicle_length"]), 32):
        nets = [StepwiseNetwork(t, routes, cuda, lane_capacity=cap, persistent=p, vehicle_params=vp) for p in (True, False)]
        for k in range(n_act):
            act = rng.uniform(0.05, 0.95, len(g["action"])).astype(np.float32)
            outs = []
            try:
                for net in nets:
                    a = torch.tensor(act, device=cuda, requires_grad=not hard)
                    cut, reward, queue, counts = net.rollout(a, *args, differentiable=not hard)
                    grad = None
                    if not hard:
                        cut.backward()
                        grad = a.grad.cpu().numpy()
                    outs.append((queue.cpu().numpy(), counts.cpu().numpy(), grad))
            except (ops.CapacityError, AssertionError):     # (a schedule the reference would refuse too: capacity, CFL)
                skipped += 1
                continue
            ok = np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
            if not hard:
                scale = max(np.abs(outs[1][2]).max(), 1e-30)
                worst = max(worst, np.abs(outs[0][2] - outs[1][2]).max() / scale)
                ok = ok and worst <= 1e-6
            if not ok:
                bad += 1
                print("MISMATCH", name, "capacity", cap, "action", k)
    print("%-22s %4d lanes %5d cells: forms agree on %d schedules x 2 capacities (%d skipped: capacity / CFL), gradient %.1e" % (
        name, t.n_lanes, t.n_cells, n_act, skipped, worst), flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
