"""Fuzz: the persistent kernels against the CPU oracle on random signal schedules over every itscp fixture (training and evaluation
episodes): vehicle counts and hand-off events equal, queues and reward within 1e-5; the gradient's worst entry is printed (a schedule
can sit on a knife edge of the float32 arithmetic: tests/test_stepwise_gpu.py masks those).  GPU box: python tools/probes/fuzz_oracle.py [n]"""
import glob
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-hybrid-traffic-sim_amd"), os.path.join(ROOT, "tests")]
from test_oracle_golden import itscp_hybrid_tables, itscp_micro_tables, itscp_vehicle_params      # noqa: E402
from dhts import ops      # noqa: E402
from dhts.network import group_routes      # noqa: E402
from dhts.stepwise import StepwiseNetwork, default_lane_capacity      # noqa: E402
from oracle import oracle as O      # noqa: E402

O.build()
n_act = int(sys.argv[1]) if len(sys.argv) > 1 else 3
only = set(sys.argv[2].split(",")) if len(sys.argv) > 2 else None      # fixtures to run (the others still draw their random numbers)
from dhts import _lib      # noqa: E402
# the oracle forms the reward as ItscpEnv._reward does (one float32 chain over lanes and steps): so do the kernels here
assert _lib.lib().dhts_set_option(_lib.OPT_REWARD_CHAIN, 1) == 0
cuda = torch.device("cuda:0")
rng = np.random.default_rng(777)
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
    if vp is None:
        gr, ptr = group_routes(routes, t.n_lanes)
        gvp = None
    else:
        gr, ptr, gvp = group_routes(routes, t.n_lanes, vp)
    args = (m["num_intersection"] ** 2, m["simulation_frequency"] * m["signal_length"], 1.0 / m["simulation_frequency"], m["speed_limit"],
            m["static_speed"], m["vehicle_length"])
    hard = name.startswith("eval")
    if only is not None and name not in only:
        for k in range(n_act):
            rng.uniform(0.05, 0.95, len(g["action"]))
        continue
    net = StepwiseNetwork(t, routes, cuda, lane_capacity=max(16, default_lane_capacity(t, m["vehicle_length"])), persistent=True, vehicle_params=vp)
    wq = wg = 0.0
    done = skipped = 0
    why = {}
    for k in range(n_act):
        act = rng.uniform(0.05, 0.95, len(g["action"])).astype(np.float32)
        ref = O.net_hybrid(t, gr, ptr, act, *args, hard=hard, want_grad=not hard, vehicle_params=gvp)
        if ref["rc"] != 0:
            skipped += 1
            why["oracle rc %d" % ref["rc"]] = why.get("oracle rc %d" % ref["rc"], 0) + 1
            continue
        a = torch.tensor(act, device=cuda, requires_grad=not hard)
        try:
            cut, reward, queue, counts = net.rollout(a, *args, differentiable=not hard)
            if not hard:
                cut.backward()
        except (ops.CapacityError, AssertionError) as e:
            skipped += 1
            why[type(e).__name__] = why.get(type(e).__name__, 0) + 1
            continue
        q = queue.cpu().numpy()
        c = counts.cpu().numpy()
        eq = np.abs(q - ref["queue"]).max() / max(np.abs(ref["queue"]).max(), 1e-30)
        er = abs(float(reward) - ref["reward"]) / max(abs(ref["reward"]), 1e-30)
        ok = (int(c[0]), int(c[1])) == (ref["n_spawned"], ref["n_deposits"]) and eq <= 1e-5 and er <= 1e-5
        wq = max(wq, eq)
        if not hard:
            wg = max(wg, np.abs(a.grad.cpu().numpy() - ref["g_action"]).max() / max(np.abs(ref["g_action"]).max(), 1e-30))
        done += 1
        if not ok:
            bad += 1
            d = np.abs(q - ref["queue"])
            tt, ll = np.unravel_index(int(np.argmax(d)), d.shape)
            first = np.argwhere(d > 1e-6 * np.abs(ref["queue"]).max())
            print("MISMATCH", name, "schedule", k, "counts", c[:2], (ref["n_spawned"], ref["n_deposits"]), "queues %.1e reward %.1e" % (eq, er),
                  "| largest at step %d lane %d (kernel %.9g, oracle %.9g); first entry above 1e-6: step %s" % (
                      tt, ll, q[tt, ll], ref["queue"][tt, ll], None if not len(first) else tuple(first[0])))
    print("%-22s %4d lanes %5d cells: %d schedules vs the oracle (%d refused%s): queues %.1e, gradient %.1e" % (
        name, t.n_lanes, t.n_cells, done, skipped, ": %s" % why if why else "", wq, wg), flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
