"""The fused macro-network rollout kernel (signals, ghost exchange, queue loss, action gradient; one workgroup per
replica) against the reference's itscp runs (G8) and against the CPU oracle on randomised actions / replica batches."""
import os

import numpy as np
import pytest

from test_oracle_golden import itscp_tables
from util import TOL_GRAD, TOL_STATE, grad_report, rel_elem, rel_max

pytestmark = pytest.mark.gpu


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


@pytest.mark.parametrize("name", ["macro_small", "macro", "macro_2x2", "macro_half", "macro_long"])
def test_network_rollout_vs_reference(cuda, golden_dir, name):
    import torch
    from dhts import ops
    g = load(golden_dir, "itscp_%s.npz" % name)
    tab, m = itscp_tables(g)
    dt = 1.0 / m["simulation_frequency"]
    dtab = ops.DeviceNetTables(tab, cuda)
    action = torch.tensor(g["action"][None], device=cuda, requires_grad=True)
    reward, queue = ops.net_macro_rollout(action, dtab, m["num_intersection"] ** 2, m["simulation_frequency"] * m["signal_length"],
                                          dt, m["speed_limit"], m["static_speed"], m["vehicle_length"])
    reward.sum().backward()
    assert rel_max(queue[0].cpu().numpy().T, g["queue"]) <= TOL_STATE
    assert rel_elem(queue[0].cpu().numpy().T, g["queue"]) <= 10 * TOL_STATE      # (sum of sigmoids)^2 dt: twice the state's relative error
    assert abs(float(reward[0]) - float(g["reward"])) <= 1e-5 * abs(float(g["reward"]))
    assert grad_report("G8 %s (kernels) d reward / d action" % name, action.grad[0].cpu().numpy(), g["g_action"]) <= TOL_GRAD


def test_network_replica_batch_vs_oracle(cuda, oracle, golden_dir):
    """64 replicas with their own actions (shared tables) and 3 replicas with their own schedules: every replica equals
    the oracle's single-network run; results are bitwise repeatable."""
    import torch
    from dhts import ops
    from dhts.network import MacroNetworkTables
    g = load(golden_dir, "itscp_macro_small.npz")
    tab, m = itscp_tables(g)
    dt, um = 1.0 / m["simulation_frequency"], m["speed_limit"]
    sq, F = m["num_intersection"] ** 2, m["simulation_frequency"] * m["signal_length"]
    rng = np.random.default_rng(4)
    R, A = 64, len(g["action"])
    acts = rng.uniform(0.1, 0.9, (R, A)).astype(np.float32)
    dtab = ops.DeviceNetTables(tab, cuda)
    a = torch.tensor(acts, device=cuda, requires_grad=True)
    w = torch.tensor(rng.uniform(0.5, 2.0, R).astype(np.float32), device=cuda)
    reward, queue = ops.net_macro_rollout(a, dtab, sq, F, dt, um)
    (reward * w).sum().backward()
    for r in (0, 17, 63):
        o = oracle.net_macro(tab, acts[r], sq, F, dt, um)
        assert rel_max(queue[r].cpu().numpy(), o["queue"]) <= TOL_STATE
        assert abs(float(reward[r]) - o["reward"]) <= 1e-5 * abs(o["reward"])
        assert rel_max(a.grad[r].cpu().numpy() / float(w[r]), o["g_action"]) <= TOL_GRAD
    a2 = torch.tensor(acts, device=cuda, requires_grad=True)
    reward2, queue2 = ops.net_macro_rollout(a2, dtab, sq, F, dt, um)
    (reward2 * w).sum().backward()
    assert torch.equal(reward, reward2) and torch.equal(queue, queue2) and torch.equal(a.grad, a2.grad)
    # per-replica schedules
    tabs = []
    for r in range(3):
        sched = tab.schedule.T * (0.5 + 0.25 * r)
        t = MacroNetworkTables.__new__(MacroNetworkTables)
        t.__dict__.update(tab.__dict__)
        t.schedule = np.ascontiguousarray(sched.T)
        tabs.append(t)
    dt3 = ops.DeviceNetTables(tabs, cuda)
    a3 = torch.tensor(acts[:3], device=cuda, requires_grad=True)
    reward3, _ = ops.net_macro_rollout(a3, dt3, sq, F, dt, um)
    reward3.sum().backward()
    for r in range(3):
        o = oracle.net_macro(tabs[r], acts[r], sq, F, dt, um)
        assert abs(float(reward3[r]) - o["reward"]) <= 1e-5 * max(abs(o["reward"]), 1e-6)
        assert rel_max(a3.grad[r].cpu().numpy(), o["g_action"]) <= TOL_GRAD


def test_multi_intersection_network_vs_oracle(cuda, oracle):
    """2 x 2 intersections (four signal entries per phase; lanes gated by another intersection's signal), tables straight
    from the environment classes: the kernels' per-intersection signal table and action reduction against the oracle."""
    import torch
    from dhts import ops
    from dhts.network import MacroNetworkTables
    from example.control.itscp._env import ItscpEnv
    from example.control.itscp.problem import problem_3
    np.random.seed(77)
    env = ItscpEnv()
    env.schedule_callback = problem_3
    for k, v in dict(num_intersection=2, lane_length=10.0, num_lane=2, policy_length=4, signal_length=1, mode="macro",
                     speed_limit=60.0).items():
        env.config[k] = v
    env.reset()
    tab = MacroNetworkTables.from_env(env)
    sq, F, dt, um = 4, env.config["signal_length"] * env.config["simulation_frequency"], 1.0 / env.config["simulation_frequency"], 60.0
    assert env.action_size() == 4 * sq and tab.n_cells * env.num_timestep <= 100000
    rng = np.random.default_rng(9)
    acts = rng.uniform(0.1, 0.9, (5, env.action_size())).astype(np.float32)
    a = torch.tensor(acts, device=cuda, requires_grad=True)
    reward, queue = ops.net_macro_rollout(a, ops.DeviceNetTables(tab, cuda), sq, F, dt, um)
    reward.sum().backward()
    for k in range(len(acts)):
        o = oracle.net_macro(tab, acts[k], sq, F, dt, um)
        assert rel_max(queue[k].cpu().numpy(), o["queue"]) <= TOL_STATE, k
        assert abs(float(reward[k]) - o["reward"]) <= 1e-5 * abs(o["reward"]), k
        assert rel_max(a.grad[k].cpu().numpy(), o["g_action"]) <= TOL_GRAD, k
        assert np.count_nonzero(o["g_action"]) >= 8, k            # several intersections and phases carry gradient


@pytest.mark.parametrize("name", ["eval_macro", "eval_macro_2x2"])
def test_network_evaluation_episode_vs_reference(cuda, oracle, golden_dir, name):
    """dhts_net_macro_rollout_eval = ItscpEnv.step(action, False) of the reference (Trainer.evaluate, trainer.py:94-142): hard
    signals, hard ghost switch, hard is_static.  Replica 0 runs the reference's own action (fixture: queues <= 1e-5, reward
    <= 1e-5), the others random actions against the oracle's evaluation mode."""
    import torch
    from dhts import ops
    g = load(golden_dir, "itscp_%s.npz" % name)
    tab, m = itscp_tables(g)
    args = (m["num_intersection"] ** 2, m["simulation_frequency"] * m["signal_length"], 1.0 / m["simulation_frequency"],
            m["speed_limit"], m["static_speed"], m["vehicle_length"])
    rng = np.random.default_rng(31)
    acts = np.concatenate([g["action"][None], rng.uniform(0.05, 0.95, (6, len(g["action"]))).astype(np.float32)])
    acts[1] = 0.5                                           # progress == action exactly at mid-phase: neither light is on
    a = torch.tensor(acts, device=cuda)
    reward, queue = ops.net_macro_eval(a, ops.DeviceNetTables(tab, cuda), *args)
    q = queue.cpu().numpy()
    assert rel_max(q[0].T, g["queue"]) <= TOL_STATE and rel_elem(q[0].T, g["queue"]) <= 10 * TOL_STATE
    assert abs(float(reward[0]) - float(g["reward"])) <= 1e-5 * abs(float(g["reward"]))
    for k in range(1, len(acts)):
        o = oracle.net_macro(tab, acts[k], *args, hard=True)
        assert rel_max(q[k], o["queue"]) <= TOL_STATE, k
        assert abs(float(reward[k]) - o["reward"]) <= 1e-5 * abs(o["reward"]), k
    # bitwise repeatable, and not the differentiable episode
    reward2, queue2 = ops.net_macro_eval(a, ops.DeviceNetTables(tab, cuda), *args)
    assert torch.equal(reward, reward2) and torch.equal(queue, queue2)
    soft, _ = ops.net_macro_rollout(a, ops.DeviceNetTables(tab, cuda), *args)
    assert abs(float(soft[0]) - float(reward[0])) > 1e-3 * abs(float(reward[0]))
