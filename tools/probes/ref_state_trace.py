"""Where the oracle and the REFERENCE first part ways inside an itscp hybrid episode (this container only: imports /root/reference like
tools/gen_goldens.py).  Runs the reference's differentiable episode of a golden's configuration, records every macro cell's (r, y, u)
and the loss constants' inputs after every step, and compares with the oracle's state history step by step.
    python tools/probes/ref_state_trace.py hybrid_short [--source-ghost-f32] [--numpy-mean] [--torch-sqrt]
(--source-ghost-f32: the oracle rounds a source lane's upstream ghost to float32 as it did until the end of round 5; the default feeds it
to the solve in double, as the reference's Python floats do; --numpy-mean / --torch-sqrt: the oracle's running means as numpy's
float32 summation computes them / its float32 glue square root through this torch build's kernel -- docs/history/round_5_design_notebook.md section 8, "What is left")"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
name = sys.argv[1] if len(sys.argv) > 1 else "hybrid_short"
g = np.load(os.path.join(ROOT, "tests", "golden", "itscp_%s.npz" % name))
m = json.loads(str(g["meta"]))

# ---- the reference's run ----
sys.path[:0] = ["/root/reference", os.path.join(ROOT, "tools", "ref_stubs")]
import torch as th      # noqa: E402
from example.control.itscp._env import ItscpEnv      # noqa: E402
from example.control.itscp import problem as problems      # noqa: E402
env = ItscpEnv()
env.schedule_callback = getattr(problems, "problem_%d" % m.get("problem", 1))
env.render_eval = False
for k, v in dict(num_intersection=m["num_intersection"], lane_length=m["lane_length"], num_lane=m["num_lane"], render=False,
                 policy_length=m["policy_length"], signal_length=m["signal_length"], mode=m["mode"], speed_limit=60.0,
                 random_seed=m["seed"]).items():
    env.config[k] = v
env.reset()
keys = list(env.lane.keys())
action = th.tensor(g["action"], requires_grad=True)
states = []
orig_step = env._simulate_step


def step(a, differentiable):
    r = orig_step(a, differentiable)
    row = []
    for k in keys:
        sl = env.lane[k].sim_lane
        if sl.is_macro():
            for c in sl.curr_cell:
                row.append((float(c.state.q.r), float(c.state.q.y), float(c.state.u)))
    states.append(row)
    return r


env._simulate_step = step
env.queue_length.clear()
env._simulate(action, True)
ref = np.array(states, dtype=np.float64)          # [T][cells in lane order][3]
queue_ref = np.array([[float(x) for x in env.queue_length[k]] for k in keys])
print("reference run: %d steps, %d cells; queues equal the fixture's: %s" % (ref.shape[0], ref.shape[1], np.array_equal(queue_ref, g["queue"])))
for p in list(sys.path):
    if p.startswith("/root/reference"):
        sys.path.remove(p)
for mod in [k for k in sys.modules if k.split(".")[0] in ("example", "road", "model", "dmath")]:
    del sys.modules[mod]

# ---- the oracle's run ----
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-hybrid-traffic-sim_amd"), os.path.join(ROOT, "tests")]
from test_oracle_golden import itscp_hybrid_tables      # noqa: E402
from dhts.network import group_routes      # noqa: E402
from oracle import oracle as O      # noqa: E402
O.build()
O.set_source_ghost_f64("--source-ghost-f32" not in sys.argv)
if "--numpy-mean" in sys.argv:              # the running means as numpy sums them (O(window) per sample)
    O.set_numpy_mean(1)
if "--torch-sqrt" in sys.argv:              # the glue's float32 square root as this torch build evaluates it
    import torch as _th
    _buf = _th.zeros((), dtype=_th.float32)

    def _torch_sqrt(x):
        _buf.fill_(x)
        return _th.sqrt(_buf).item()
    O.set_sqrtf_hook(_torch_sqrt)
t, mm = itscp_hybrid_tables(g)
routes, route_ptr = group_routes(g["spawn_routes"], t.n_lanes)
o = O.net_hybrid(t, routes, route_ptr, g["action"], mm["num_intersection"] ** 2, mm["simulation_frequency"] * mm["signal_length"],
                 1.0 / mm["simulation_frequency"], mm["speed_limit"], mm["static_speed"], mm["vehicle_length"], want_hist=True)
hist = o["hist"]                                   # [T + 1][4][C], cells in the tables' order
# cells in lane order = the tables' order (lane_off ascending with the lane id)
order = []
for l in range(t.n_lanes):
    if t.lane_macro[l]:
        order += list(range(t.lane_off[l], t.lane_off[l] + t.lane_ncell[l]))
order = np.array(order)
T = ref.shape[0]
first = None
for s in range(T):
    orc = np.stack([hist[s + 1, 0, order], hist[s + 1, 1, order], hist[s + 1, 2, order]], axis=1).astype(np.float64)
    d = np.abs(orc - ref[s])
    if d.max() > 0 and first is None:
        first = s
        c, q = np.unravel_index(np.argmax(d), d.shape)
        cell = order[c]
        lane = int(np.searchsorted(np.asarray(t.lane_off) + np.asarray(t.lane_ncell) * np.asarray(t.lane_macro), cell, side="right"))
        print("first difference after step %d: cell %d (component %s): oracle %.9g reference %.9g (%.1e relative); %d of %d values differ"
              % (s, cell, "ryu"[q], orc[c, q], ref[s, c, q], d[c, q] / max(abs(ref[s, c, q]), 1e-30), int((d > 0).sum()), d.size))
    if s in (T // 4, T // 2, T - 1):
        print("after step %3d: max |d| %.2e, values differing %d of %d" % (s, d.max(), int((d > 0).sum()), d.size))
if first is None:
    print("the macro cells' state is bit-identical over the whole episode")
print("queues: max |d| / max |ref| = %.2e" % (np.abs(o["queue"].T - g["queue"]).max() / np.abs(g["queue"]).max()))
# detail of the first step
s = first if first is not None else 0
orc = np.stack([hist[s + 1, 0, order], hist[s + 1, 1, order], hist[s + 1, 2, order]], axis=1).astype(np.float64)
d = np.abs(orc - ref[s])
for c in np.nonzero((d > 0).any(axis=1))[0][:12]:
    cell = order[c]
    lane = [l for l in range(t.n_lanes) if t.lane_macro[l] and t.lane_off[l] <= cell < t.lane_off[l] + t.lane_ncell[l]][0]
    print("  cell %d = lane %d cell %d of %d: oracle (r, y, u) = %.9g %.9g %.9g | reference = %.9g %.9g %.9g"
          % (cell, lane, cell - t.lane_off[lane], t.lane_ncell[lane], *orc[c], *ref[s, c]))
