"""Macro networks of any size through the batched-lane path (dhts/batched.py: all lanes of the network as the batch of the
straight-lane step operator, one call per step): the reference's own itscp runs (G8), and a network that does not fit the fused
kernels' one workgroup (3 x 3 intersections with 3 lanes per approach: 360 lanes) against the CPU oracle."""
import os
import time

import numpy as np
import pytest

from test_oracle_golden import itscp_tables
from util import TOL_GRAD, TOL_STATE, grad_report, rel_max, state_report

pytestmark = pytest.mark.gpu


def _args(m):
    return (m["num_intersection"] ** 2, m["simulation_frequency"] * m["signal_length"], 1.0 / m["simulation_frequency"],
            m["speed_limit"], m["static_speed"], m["vehicle_length"])


@pytest.mark.parametrize("name", ["macro_small", "macro", "macro_2x2", "macro_half", "macro_long", "macro_3x3x3"])
def test_batched_network_vs_reference(cuda, golden_dir, name):
    import torch
    from dhts import ops
    from dhts.batched import BatchedMacroNetwork
    g = np.load(os.path.join(golden_dir, "itscp_%s.npz" % name))
    tab, m = itscp_tables(g)
    net = BatchedMacroNetwork(tab, cuda)
    action = torch.tensor(g["action"], device=cuda, requires_grad=True)
    reward, queue = net.rollout(action, *_args(m))
    reward.backward()
    assert state_report("batched %s: queues vs reference" % name, queue.detach().cpu().numpy().T, g["queue"]) <= TOL_STATE
    assert abs(float(reward) - float(g["reward"])) <= 1e-5 * abs(float(g["reward"]))
    assert grad_report("batched %s d reward / d action" % name, action.grad.cpu().numpy(), g["g_action"]) <= TOL_GRAD
    # the same episode with the ghost exchange as torch gathers and blends (the reference's glue op by op)
    a3 = torch.tensor(g["action"], device=cuda, requires_grad=True)
    r3, q3 = net.rollout_torch(a3, *_args(m))
    r3.backward()
    assert rel_max(queue.detach().cpu().numpy(), q3.detach().cpu().numpy()) <= 1e-6
    assert rel_max(action.grad.cpu().numpy(), a3.grad.cpu().numpy()) <= 1e-5
    if tab.n_cells + tab.n_lanes > 1024:          # (macro_3x3x3: 360 lanes + 2 124 cells, beyond one workgroup -- the reference's own run
        return                                    # above is this network's pin; round 5)
    # and the fused one-workgroup kernels give the same episode
    a2 = torch.tensor(g["action"][None], device=cuda, requires_grad=True)
    r2, q2 = ops.net_macro_rollout(a2, ops.DeviceNetTables(tab, cuda), *_args(m))
    assert rel_max(queue.detach().cpu().numpy(), q2[0].detach().cpu().numpy()) <= TOL_STATE


@pytest.mark.parametrize("name", ["eval_macro", "eval_macro_2x2"])
def test_batched_network_evaluation_episode_vs_reference(cuda, golden_dir, name):
    import torch
    from dhts.batched import BatchedMacroNetwork
    g = np.load(os.path.join(golden_dir, "itscp_%s.npz" % name))
    tab, m = itscp_tables(g)
    with torch.no_grad():
        reward, queue = BatchedMacroNetwork(tab, cuda).rollout(torch.tensor(g["action"], device=cuda), *_args(m), differentiable=False)
    assert state_report("batched %s: queues vs reference" % name, queue.cpu().numpy().T, g["queue"]) <= TOL_STATE
    assert abs(float(reward) - float(g["reward"])) <= 1e-5 * abs(float(g["reward"]))


def test_env_step_runs_a_network_beyond_one_workgroup_batched(cuda, oracle):
    """run_itscp_macro.sh with --n_intersection=3 --n_lane=3 (reference _env.py:221-439): 360 lanes, > 1 024 cells + lanes.
    ItscpEnv.step takes the batched path (no lane-by-lane stepping) and its reward, queue terms and d reward / d action are the
    oracle's for the same tables."""
    import torch
    from dhts.network import MacroNetworkTables
    from example.control.itscp._env import ItscpEnv
    env = ItscpEnv()
    for k, v in dict(num_intersection=3, num_lane=3, mode="macro", random_seed=5, macro_path="batched").items():      # (round 5: the
        env.config[k] = v                                       # default for such networks is the stepwise path, dhts/stepwise.py)
    env.reset()
    tab = MacroNetworkTables.from_env(env)
    assert tab.n_cells + tab.n_lanes > 1024 and tab.n_lanes == 360
    rng = np.random.default_rng(2)
    act = rng.uniform(0.2, 0.8, env.action_size()).astype(np.float32)
    action = torch.tensor(act, device=cuda, requires_grad=True)
    t0 = time.time()
    obs, reward, done, info = env.step(action, True)
    reward.backward()
    torch.cuda.synchronize()
    t_diff = time.time() - t0
    assert env._fused_cache[0] == "batched" and env._fused_done
    sq, F = env.num_intersection ** 2, env.config["signal_length"] * env.config["simulation_frequency"]
    o = oracle.net_macro(tab, act, sq, F, 1.0 / env.config["simulation_frequency"], env.simulator.speed_limit,
                         env.config["static_speed"], env.simulator.vehicle_length)
    assert o["rc"] == 0
    keys = list(env.lane.keys())
    queue = np.array([env.queue_length[k] for k in keys])            # [L][T]
    assert state_report("360-lane network, batched vs oracle: queues", queue.T, o["queue"]) <= TOL_STATE
    assert abs(float(reward.detach()) - o["reward"]) <= 1e-5 * abs(o["reward"])
    assert grad_report("360-lane network d reward / d action", action.grad.cpu().numpy(), o["g_action"]) <= TOL_GRAD
    env.reset()
    with torch.no_grad():
        t0 = time.time()
        _, reward_e, _, _ = env.step(torch.tensor(act, device=cuda), False)
        torch.cuda.synchronize()
        t_eval = time.time() - t0
    oe = oracle.net_macro(tab, act, sq, F, 1.0 / env.config["simulation_frequency"], env.simulator.speed_limit,
                          env.config["static_speed"], env.simulator.vehicle_length, hard=True)
    assert abs(float(reward_e) - oe["reward"]) <= 1e-5 * abs(oe["reward"])
    print("360 lanes, %d cells, %d steps: differentiable episode (forward + backward) %.2f s, evaluation episode %.2f s on the batched path (first calls: HIP-graph capture included)"
          % (tab.n_cells, tab.T, t_diff, t_eval))
    # a second pair of episodes after reset(): new schedules and routes go into the device tables in place, the captured graphs
    # are replayed; the numbers are again the oracle's for the new tables
    net = env._batched_net
    env.config["random_seed"] = 6              # (another inflow schedule and other per-step routes)
    env.reset()
    assert env._batched_net is net
    tab2 = MacroNetworkTables.from_env(env)
    action2 = torch.tensor(act, device=cuda, requires_grad=True)
    torch.cuda.synchronize()
    t0 = time.time()
    _, reward2, _, _ = env.step(action2, True)
    reward2.backward()
    torch.cuda.synchronize()
    t_replay = time.time() - t0
    assert env._fused_cache[1] is net and not getattr(env, "_batched_graph_failed", False) and len(net._graphs) == 2
    o2 = oracle.net_macro(tab2, act, sq, F, 1.0 / env.config["simulation_frequency"], env.simulator.speed_limit,
                          env.config["static_speed"], env.simulator.vehicle_length)
    assert abs(float(reward2.detach()) - o2["reward"]) <= 1e-5 * abs(o2["reward"])
    assert grad_report("360-lane network, replayed graph: d reward / d action", action2.grad.cpu().numpy(), o2["g_action"]) <= TOL_GRAD
    assert abs(o2["reward"] - o["reward"]) > 1e-6 * abs(o["reward"])          # (the new episode is a different one)
    print("  replayed differentiable episode with new tables: %.3f s (table upload + replay + oracle-checked)" % t_replay)


def test_graphed_rollout_equals_eager(cuda, golden_dir):
    import torch
    from dhts.batched import BatchedMacroNetwork
    g = np.load(os.path.join(golden_dir, "itscp_macro_2x2.npz"))
    tab, m = itscp_tables(g)
    net = BatchedMacroNetwork(tab, cuda)
    a1 = torch.tensor(g["action"], device=cuda, requires_grad=True)
    r1, q1 = net.rollout(a1, *_args(m))
    r1.backward()
    for _ in range(2):          # capture, then replay
        a2 = torch.tensor(g["action"], device=cuda, requires_grad=True)
        r2, q2 = net.graphed_rollout(a2, *_args(m))
        (3.0 * r2).backward()
        assert torch.equal(r1.detach(), r2.detach()) and torch.equal(q1.detach(), q2)
        assert torch.equal(3.0 * a1.grad, a2.grad)
    with torch.no_grad():
        re, qe = net.rollout(torch.tensor(g["action"], device=cuda), *_args(m), differentiable=False)
        rg, qg = net.graphed_rollout(torch.tensor(g["action"], device=cuda), *_args(m), differentiable=False)
    assert torch.equal(re, rg) and torch.equal(qe, qg)


def _random_network(rng, n_inter, T):
    """A random signalled macro network: every intersection has approaching lanes (signalled, west-east or north-south), mid
    lanes behind them (several per approaching lane: the per-step route picks one) and leaving lanes that feed approaching
    lanes of other intersections or end; cells per lane 1 .. 6, two cell lengths."""
    from dhts.network import SIG_ALWAYS, SIG_NS, SIG_WE, MacroNetworkTables
    ncell, length, kinds, inter, edges = [], [], [], [], []

    def lane(n, dx, kind, it):
        ncell.append(n); length.append(n * dx); kinds.append(kind); inter.append(it)
        return len(ncell) - 1
    leaving = []
    approaching = []
    for it in range(n_inter):
        for _ in range(int(rng.integers(2, 4))):
            a = lane(int(rng.integers(1, 7)), float(rng.choice([5.0, 4.0])), int(rng.choice([SIG_WE, SIG_NS])), it)
            approaching.append(a)
            outs = []
            for _ in range(int(rng.integers(1, 3))):
                m = lane(int(rng.integers(1, 4)), float(rng.choice([5.0, 4.0])), SIG_ALWAYS, it)
                o = lane(int(rng.integers(1, 7)), 5.0, SIG_ALWAYS, it)
                edges.append((a, m)); edges.append((m, o))
                outs.append(o)
            leaving += outs
    rng.shuffle(leaving)
    free = [a for a in approaching]
    rng.shuffle(free)
    for o in leaving[:len(leaving) // 2]:
        if not free:
            break
        a = free.pop()
        if inter[a] != inter[o]:
            edges.append((o, a))
    L = len(ncell)
    nxt = [[] for _ in range(L)]
    for a, b in edges:
        nxt[a].append(b)
    route = -np.ones((T, L), dtype=np.int32)
    for t in range(T):
        for l in range(L):
            if nxt[l]:
                route[t, l] = nxt[l][int(rng.integers(0, len(nxt[l])))]
    sched = rng.uniform(0.05, 0.6, (L, T))
    return MacroNetworkTables(ncell, length, edges, kinds, inter, route, sched)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_batched_random_network_vs_oracle(cuda, oracle, seed):
    """Random topologies (merging and splitting lanes with per-step routes, one-cell lanes, two cell lengths, sink lanes with their
    stored ghosts): queues, reward and d reward / d action of the batched path against the CPU oracle; where the network fits one
    workgroup, against the fused kernels too."""
    import torch
    from dhts import ops
    from dhts.batched import BatchedMacroNetwork
    rng = np.random.default_rng(seed)
    n_inter, T, F = 4, 90, 30
    tab = _random_network(rng, n_inter, T)
    act = rng.uniform(0.2, 0.8, 3 * n_inter).astype(np.float32)
    args = (n_inter, F, 0.1, 20.0, 0.2, 5.0)
    net = BatchedMacroNetwork(tab, cuda)
    a = torch.tensor(act, device=cuda, requires_grad=True)
    reward, queue = net.rollout(a, *args)
    reward.backward()
    o = oracle.net_macro(tab, act, *args)
    assert o["rc"] == 0
    assert state_report("random network %d (%d lanes, %d cells): queues vs oracle" % (seed, tab.n_lanes, tab.n_cells),
                        queue.detach().cpu().numpy(), o["queue"]) <= TOL_STATE
    assert abs(float(reward.detach()) - o["reward"]) <= 1e-5 * abs(o["reward"])
    assert grad_report("random network %d d reward / d action" % seed, a.grad.cpu().numpy(), o["g_action"]) <= TOL_GRAD
    if tab.n_cells + tab.n_lanes <= 1024:
        a2 = torch.tensor(act[None], device=cuda, requires_grad=True)
        r2, q2 = ops.net_macro_rollout(a2, ops.DeviceNetTables(tab, cuda), *args)
        r2.sum().backward()
        assert rel_max(queue.detach().cpu().numpy(), q2[0].detach().cpu().numpy()) <= TOL_STATE
        assert rel_max(a.grad.cpu().numpy(), a2.grad[0].cpu().numpy()) <= TOL_GRAD
    with torch.no_grad():
        re, qe = net.rollout(torch.tensor(act, device=cuda), *args, differentiable=False)
    oe = oracle.net_macro(tab, act, *args, hard=True)
    assert abs(float(re) - oe["reward"]) <= 1e-5 * max(abs(oe["reward"]), 1e-6)
