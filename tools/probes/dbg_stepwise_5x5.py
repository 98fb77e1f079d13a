import os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/diff-hybrid-traffic-sim_amd"); sys.path.insert(0, "/root/repo/tests")
from test_oracle_golden import itscp_hybrid_tables
from dhts.network import group_routes
from dhts.stepwise import StepwiseNetwork
from oracle import oracle as O
g = np.load("/root/repo/tests/golden/itscp_hybrid_5x5.npz")
t, m = itscp_hybrid_tables(g)
args = (m["num_intersection"] ** 2, m["simulation_frequency"] * m["signal_length"], 1.0 / m["simulation_frequency"], m["speed_limit"], m["static_speed"], m["vehicle_length"])
routes = np.concatenate([g["spawn_routes"]] * 4)
gr, ptr = group_routes(routes, t.n_lanes)
cuda = torch.device("cuda:0")
net = StepwiseNetwork(t, routes, cuda)
rng = np.random.default_rng(5)
for k in range(4):
    act = rng.uniform(0.1, 0.9, len(g["action"])).astype(np.float32)
    def run(ls):
        a = torch.tensor(act, device=cuda, requires_grad=True)
        cut, reward, queue, counts = net.rollout(a, *args, loss_steps=ls)
        cut.backward()
        return a.grad.cpu().numpy(), counts.cpu().numpy()
    gd, cn = run(0)
    ref = O.net_hybrid(t, gr, ptr, act, *args)
    e = np.abs(gd - ref["g_action"]).max() / np.abs(ref["g_action"]).max()
    print("k", k, "err", e, "counts", cn, ref["n_spawned"], ref["n_deposits"])
    if e > 2e-5:
        lo, hi = 1, t.T
        for ls in (30, 60, 90, 120, 150, 180, 210, 240):
            gd, _ = run(ls)
            r2 = O.net_hybrid(t, gr, ptr, act, *args, t_cut=ls)
            print("  loss_steps", ls, "err", np.abs(gd - r2["g_action"]).max() / max(np.abs(r2["g_action"]).max(), 1e-30), "idx", int(np.argmax(np.abs(gd - r2["g_action"]))))
        # events
        ws = net._ws

# ---- state comparison for the failing action (k = 2) ----
rng = np.random.default_rng(5)
for k in range(3):
    act = rng.uniform(0.1, 0.9, len(g["action"])).astype(np.float32)
a = torch.tensor(act, device=cuda, requires_grad=True)
cut, reward, queue, counts = net.rollout(a, *args)
H = net.hist_in_network_order(net.last_hist).cpu().numpy()
ref = O.net_hybrid(t, gr, ptr, act, *args, want_hist=True)
R = ref["hist"]
for tt in (32, 97, 141, 211):
    for l in (376, 62, 331, 8):
        n, off = t.lane_ncell[l], t.lane_off[l]
        ls = t.left_src[tt, l]
        last = t.lane_off[ls] + t.lane_ncell[ls] - 1 if ls >= 0 else -1
        print("step", tt, "lane", l, "u dev", H[tt, 2, off:off + n], "u ref", R[tt, 2, off:off + n], "up", (H[tt, 2, last], R[tt, 2, last]) if last >= 0 else None)
d = (H != R)
print("cells differing bitwise per plane:", d.reshape(d.shape[0], 4, -1).sum(axis=(0, 2)), "first step with a difference:", int(np.argmax(d.any(axis=(1, 2)))))
