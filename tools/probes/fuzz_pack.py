#!/usr/bin/env python3
"""Fuzz: the fused hybrid kernels one against two replicas per compute unit (dhts_hybrid_tables::two_per_cu = -1 / 1) on random signal
schedules over every fused-size itscp fixture, with the fixture's vehicle attributes: counts, queues, reward and gradient bit for bit,
and the same faults.  GPU box: python tools/probes/fuzz_pack.py [schedules per fixture]"""
import glob
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-hybrid-traffic-sim_amd"), os.path.join(ROOT, "tests")]
from test_oracle_golden import itscp_hybrid_tables, itscp_micro_tables, itscp_vehicle_params      # noqa: E402
from dhts import ops      # noqa: E402

n_act = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cuda = torch.device("cuda:0")
rng = np.random.default_rng(31)
bad = tested = 0
for f in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "itscp_*.npz"))):
    name = os.path.basename(f)[6:-4]
    if name.startswith("eval") or name.startswith("macro"):
        continue
    g = np.load(f)
    vp = itscp_vehicle_params(g)
    if "micro" in name:
        t, m, rows = itscp_micro_tables(g)
        t.set_micro_sources(np.concatenate([g["rand_draws"], rng.random(8 * len(g["rand_draws"]) + 64)]))
        rows = np.concatenate([rows] * 3)
        vp = None if vp is None else np.concatenate([vp] * 3)
    else:
        t, m = itscp_hybrid_tables(g)
        rows = np.concatenate([g["spawn_routes"]] * 4)
        vp = None if vp is None else np.concatenate([vp] * 4)
    try:
        t.check_kernel_limits()
    except ValueError:
        continue
    args = (m["num_intersection"] ** 2, m["simulation_frequency"] * m["signal_length"], 1.0 / m["simulation_frequency"], m["speed_limit"],
            m["static_speed"], m["vehicle_length"])
    tabs = {}
    for mode in (-1, 1):
        tabs[mode] = ops.DeviceHybridTables(t, rows, cuda, vehicle_params=vp)
        tabs[mode].two_per_cu = mode
    if not ops.net_hybrid_plan(3, len(g["action"]), tabs[1], args[0])["packed"]:
        print("%-16s the packed plan does not fit: skipped" % name)
        continue
    acts = rng.uniform(0.1, 0.9, (n_act, 3, len(g["action"]))).astype(np.float32)
    n_bad = 0
    for k in range(n_act):
        res = {}
        for mode in (-1, 1):
            a = torch.tensor(acts[k], device=cuda, requires_grad=True)
            err, err_b = ops.new_error_record(cuda), ops.new_error_record(cuda)
            cut, reward, queue, counts = ops.net_hybrid_rollout(a, tabs[mode], *args, check_faults=False, err=err, err_bwd=err_b)
            cut.sum().backward()
            res[mode] = (reward, queue, counts, a.grad, err.tolist()[0], err_b.tolist()[0])
        x, y = res[-1], res[1]
        same = all(torch.equal(p, q) or (torch.isnan(p) == torch.isnan(q)).all() and torch.equal(torch.nan_to_num(p), torch.nan_to_num(q))
                   for p, q in zip(x[:4], y[:4])) and x[4:] == y[4:]
        n_bad += not same
    tested += 1
    bad += n_bad
    print("%-16s %3d lanes %4d cells: %d schedules x 3 replicas, one vs two replicas per unit: %s" % (name, t.n_lanes, t.n_cells, n_act, "equal" if not n_bad else "%d DIFFER" % n_bad), flush=True)
print("fixtures: %d, mismatches: %d" % (tested, bad))
sys.exit(1 if bad else 0)
