"""The network kernels against a directory of REFERENCE itscp episodes at random shapes (the fixtures tools/probes/ref_sweep.py left under
/tmp/dhts_ref_sweep/*/, copied in the build container to the untracked gpurun_in/itscp/):   python tools/probes/itscp_cases.py <dir>
Every episode through the persistent stepwise form (any size, any mode) and, where the network fits one workgroup, through the fused
kernels as well: vehicle counts, queue terms <= 1e-5, reward <= 1e-5, d reward / d action <= 1e-4 -- the product against the reference
itself, not through the oracle.  A line outside those is listed with the oracle-side finding of profiles/r06z_reference_sweep.log in mind
(two episodes there are the reference's own library environment)."""
import glob
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-hybrid-traffic-sim_amd"), os.path.join(ROOT, "tests")]
from dhts import _lib, ops      # noqa: E402
from dhts.stepwise import StepwiseNetwork, default_lane_capacity      # noqa: E402
from test_oracle_golden import itscp_hybrid_tables, itscp_micro_tables      # noqa: E402
from util import meta_of      # noqa: E402

assert _lib.lib().dhts_set_option(_lib.OPT_REWARD_CHAIN, 1) == 0        # the reward as ItscpEnv._reward forms it
cuda = torch.device("cuda:0")
bad = n = 0
worst = [0.0, 0.0, 0.0]
for f in sorted(glob.glob(os.path.join(sys.argv[1], "itscp_*.npz"))):
    g = np.load(f)
    m = meta_of(g)
    mode, hard = m["mode"], not m.get("differentiable", True)
    if mode == "micro":
        t, m, rows = itscp_micro_tables(g)
    else:
        t, m = itscp_hybrid_tables(g)
        rows = g["spawn_routes"] if g["spawn_routes"].shape[0] else -np.ones((1, 2), np.int32)
    args = (m["num_intersection"] ** 2, m["simulation_frequency"] * m["signal_length"], 1.0 / m["simulation_frequency"], m["speed_limit"],
            m["static_speed"], m["vehicle_length"])
    tag = "%-10s %-6s %dx%d x%d %3.0f m %2.0f m/s %2d s / %d s %s" % (os.path.basename(f)[6:-4], mode, m["num_intersection"], m["num_intersection"], m["num_lane"],
                                                                   m["lane_length"], m["speed_limit"], m["policy_length"], m["signal_length"], "eval " if hard else "train")
    gq = g["queue"].astype(np.float32)
    res = []
    forms = [("persistent", None)]
    try:
        t.check_kernel_limits()
        forms.append(("fused", ops.DeviceHybridTables(t, rows, cuda, lane_capacity=0)))
    except (ValueError, _lib.DhtsError):
        pass
    for form, dtab in forms:
        a = torch.tensor(g["action"], device=cuda, requires_grad=not hard)
        try:
            if form == "persistent":
                net = StepwiseNetwork(t, rows, cuda, lane_capacity=max(16, default_lane_capacity(t, m["vehicle_length"])), persistent=True)
                cut, reward, queue, counts = net.rollout(a, *args, differentiable=not hard)
                q, c0 = queue.cpu().numpy().T, int(counts[0])
            elif hard:
                reward, queue, counts = ops.net_hybrid_eval(a[None], dtab, *args)
                reward, q, c0 = reward[0], queue[0].cpu().numpy().T, int(counts[0, 0])
            else:
                cut, reward, queue, counts = ops.net_hybrid_rollout(a[None], dtab, *args)
                cut, reward, q, c0 = cut[0], reward[0], queue[0].detach().cpu().numpy().T, int(counts[0, 0])
            if not hard:
                cut.backward()
        except (ops.CapacityError, _lib.DhtsError, AssertionError) as e:
            res.append("%s: %s" % (form, type(e).__name__))
            continue
        eq = np.abs(q - gq).max() / max(np.abs(gq).max(), 1e-30)
        er = abs(float(reward) - float(g["reward"])) / max(abs(float(g["reward"])), 1e-30)
        eg = 0.0 if hard else np.abs(a.grad.cpu().numpy() - g["g_action"]).max() / max(np.abs(g["g_action"]).max(), 1e-30)
        ok = eq <= 1e-5 and er <= 1e-5 and eg <= 1e-4 and (mode == "macro" or c0 == m["n_vehicle_spawned"])
        worst = [max(worst[0], eq), max(worst[1], er), max(worst[2], eg)] if ok else worst
        bad += not ok
        res.append("%s: queues %.1e reward %.1e gradient %.1e%s" % (form, eq, er, eg, "" if ok else "  <-- LOOK"))
    n += 1
    print("%s | %3d lanes %4d cells | %s" % (tag, t.n_lanes, t.n_cells, " | ".join(res)), flush=True)
print("episodes: %d, outside the tolerances: %d; worst inside: queues %.1e reward %.1e gradient %.1e" % (n, bad, *worst))
