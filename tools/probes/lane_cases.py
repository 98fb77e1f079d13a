"""The straight-lane rollouts (dhts.macro_rollout / dhts.micro_rollout: dhts_macro_rollout_fwd / _bwd, dhts_micro_rollout_fwd / _bwd) against
a directory of REFERENCE runs at random shapes (tools/gen_goldens.py --only G4x,G6x: 48 ARZ lanes of 3 .. 96 cells x 10 .. 400 steps, cell
lengths 2.5 .. 100 m, three kinds of initial state, loss on the final state or on every step; 48 IDM lanes of 1 .. 48 vehicles x 10 .. 600
steps, default and random vehicle attributes, spacings down to 1.3 vehicle lengths -- written in the build container to the untracked
gpurun_in/):      python tools/probes/lane_cases.py <dir>
What tests/test_gpu_parity.py asserts on the committed G4 / G6 fixtures: states <= 1e-5 (micro: 1e-6), recorded steps, loss, gradients <= 1e-4
(micro: 1e-5)."""
import glob
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-hybrid-traffic-sim_amd"), os.path.join(ROOT, "tests")]
import dhts      # noqa: E402
from test_gpu_parity import T_, micro_inputs      # noqa: E402
from util import meta_of, rel_max      # noqa: E402

cuda = torch.device("cuda:0")
bad = n = 0
for f in sorted(glob.glob(os.path.join(sys.argv[1], "macro_rollout_*.npz"))):
    g = np.load(f)
    m = meta_of(g)
    r0, u0 = T_(g["r0"][None], cuda, grad=True), T_(g["u0"][None], cuda, grad=True)
    gr, gu = T_(g["ghost_r"][None], cuda, grad=True), T_(g["ghost_u"][None], cuda, grad=True)
    rT, yT, uT, _, hist = dhts.macro_rollout(r0, u0, gr, gu, m["T"], m["dt"], m["dx"], m["u_max"], want_hist=True)
    loss = hist.sum() if m["tap"] == "every_sum" else (rT ** 2).sum() + (uT ** 2).sum()
    loss.backward()
    es = max(rel_max(x.detach().cpu().numpy()[0], g[k]) for x, k in ((rT, "rT"), (yT, "yT"), (uT, "uT")))
    h = hist.detach().cpu().numpy()
    for t in range(len(g["steps_r"])):
        es = max(es, rel_max(h[t, 0, 0], g["steps_r"][t]), rel_max(h[t, 0, 1], g["steps_y"][t]), rel_max(h[t, 0, 2], g["steps_u"][t]))
    el = abs(float(loss) - float(g["loss"])) / max(abs(float(g["loss"])), 1e-30)
    eg = max(rel_max(r0.grad.cpu().numpy()[0], g["g_r0"]), rel_max(u0.grad.cpu().numpy()[0], g["g_u0"]),
             rel_max(gr.grad.cpu().numpy()[0], g["g_ghost_r"]), rel_max(gu.grad.cpu().numpy()[0], g["g_ghost_u"]))
    ok = es <= 1e-5 and el <= 1e-5 and eg <= 1e-4
    bad += not ok
    n += 1
    print("macro %-8s N=%2d T=%3d dx=%5.1f dt=%.4f u_max=%2.0f %-8s %-9s: states %.1e loss %.1e gradient %.1e%s" % (
        os.path.basename(f)[14:-4], len(g["r0"]), m["T"], m["dx"], m["dt"], m["u_max"], m.get("init", ""), m["tap"], es, el, eg, "" if ok else "  <-- MISMATCH"), flush=True)
for f in sorted(glob.glob(os.path.join(sys.argv[1], "micro_rollout_*.npz"))):
    g = np.load(f)
    m = meta_of(g)
    p0, v0 = T_(g["p0"][None], cuda, grad=True), T_(g["v0"][None], cuda, grad=True)
    head = T_(np.array([m["head"]], dtype=np.float64), cuda)
    pT, vT, hist = dhts.micro_rollout(p0, v0, micro_inputs(g, cuda), head, m["T"], m["dt"], want_hist=True)
    loss = hist.sum() if m["tap"] == "every_sum" else 1e-4 * (pT ** 2).sum() + (vT ** 2).sum()
    loss.backward()
    es = max(rel_max(pT.detach().cpu().numpy()[0], g["pT"]), rel_max(vT.detach().cpu().numpy()[0], g["vT"]))
    h = hist.detach().cpu().numpy()
    for t in range(len(g["steps_p"])):
        es = max(es, rel_max(h[t, 0, 0], g["steps_p"][t]), rel_max(h[t, 0, 1], g["steps_v"][t]))
    el = abs(float(loss) - float(g["loss"])) / max(abs(float(g["loss"])), 1e-30)
    eg = max(rel_max(p0.grad.cpu().numpy()[0], g["g_p0"]), rel_max(v0.grad.cpu().numpy()[0], g["g_v0"]))
    ok = es <= 1e-6 and el <= 1e-5 and eg <= 1e-5
    bad += not ok
    n += 1
    print("micro %-8s V=%2d T=%3d dt=%.4f limit=%4.1f %-7s %-9s: states %.1e loss %.1e gradient %.1e%s" % (
        os.path.basename(f)[14:-4], len(g["p0"]), m["T"], m["dt"], m["speed_limit"], m["params"], m["tap"], es, el, eg, "" if ok else "  <-- MISMATCH"), flush=True)
print("cases: %d, mismatches: %d" % (n, bad))
sys.exit(1 if bad else 0)
