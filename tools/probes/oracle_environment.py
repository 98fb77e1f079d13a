"""The oracle against the reference fixtures at its defaults and with the two LIBRARY behaviours under the reference switched in:
numpy's float32 summation tree for the running means (oracle.set_numpy_mean) and this torch build's float32 square root for the glue
(oracle.set_sqrtf_hook(torch.sqrt ...)) -- docs/history/round_5_design_notebook.md section 8, "What is left between the oracle and the reference".  CPU only.
    python tools/probes/oracle_environment.py [fixture names ...]          (default: every itscp fixture, training and evaluation episodes)
Prints, per fixture, the queue terms that differ from the reference's / all and the largest difference relative to the largest term."""
import glob
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-hybrid-traffic-sim_amd"), os.path.join(ROOT, "tests")]
from test_oracle_golden import itscp_hybrid_tables, itscp_micro_tables, itscp_vehicle_params      # noqa: E402
from dhts.network import group_routes      # noqa: E402
from oracle import oracle as O      # noqa: E402
from util import rel_max      # noqa: E402

O.build()
_buf = torch.zeros((), dtype=torch.float32)


def torch_sqrt(x):
    _buf.fill_(x)
    return torch.sqrt(_buf).item()


probe = torch.full((), 0.16979104280471802, dtype=torch.float32)
print("torch %s: sqrt(0.16979104f) = %.9g (IEEE: 0.412057102)" % (torch.__version__, float(torch.sqrt(probe))))
names = sys.argv[1:] or sorted(os.path.basename(p)[6:-4] for p in glob.glob(os.path.join(ROOT, "tests", "golden", "itscp_*.npz")))
for name in names:
    g = np.load(os.path.join(ROOT, "tests", "golden", "itscp_%s.npz" % name))
    hard = name.startswith("eval_")              # an evaluation episode (Trainer.evaluate): hard thresholds, no running mean, no gradient
    if "micro" in name:
        t, m, rows = itscp_micro_tables(g)
    else:
        t, m = itscp_hybrid_tables(g)
        rows = g["spawn_routes"] if g["spawn_routes"].shape[0] else -np.ones((1, 2), np.int32)
    vp = itscp_vehicle_params(g)             # (round 6: runs whose vehicles carry seeded random_micro_vehicle attributes)
    if vp is None:
        routes, route_ptr = group_routes(rows, t.n_lanes)
        gvp = None
    else:
        routes, route_ptr, gvp = group_routes(rows, t.n_lanes, vp)
    args = (t, routes, route_ptr, g["action"], m["num_intersection"] ** 2, m["simulation_frequency"] * m["signal_length"],
            1.0 / m["simulation_frequency"], m["speed_limit"], m["static_speed"], m["vehicle_length"])
    out = []
    try:
        for env in (0, 1):
            O.set_sqrtf_hook(torch_sqrt if env else None)
            O.set_numpy_mean(env)
            t0 = time.time()
            o = O.net_hybrid(*args, hard=hard, vehicle_params=gvp)
            q, gq = o["queue"].T.astype(np.float32), g["queue"].astype(np.float32)
            grad = "gradient %.1e, " % (np.abs(o["g_action"] - g["g_action"]).max() / np.abs(g["g_action"]).max()) if not hard else ""
            out.append("%6d of %6d terms differ, %.1e, %sreward %s (%3.0f s)"
                       % (int((q != gq).sum()), q.size, rel_max(q, gq), grad,
                          "equal" if np.float32(o["reward"]) == np.float32(float(g["reward"])) else "differs", time.time() - t0))
    finally:
        O.set_sqrtf_hook(None)
        O.set_numpy_mean(0)
    print("%-14s defaults: %s | numpy mean + torch sqrt: %s" % (name, out[0], out[1]), flush=True)
