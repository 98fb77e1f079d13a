"""DHTS_OPT_REWARD_CHAIN: the reward as ItscpEnv._reward forms it (reference example/control/itscp/_env.py:770-797) -- ONE running
sum over lanes (outermost) and steps, float32 from the first tensor term on -- behind every network rollout: fused macro, fused
hybrid, stepwise in both forms, differentiable and evaluation episodes.  Checked three ways: against the same chain walked in
numpy over the queue terms the kernel itself returned (equality: this IS the reference's order), against the reference's own
reward of the fixture, and -- by default, option off -- that nothing changes."""
import os

import numpy as np
import pytest

from test_oracle_golden import itscp_hybrid_tables, itscp_tables
from util import meta_of

pytestmark = pytest.mark.gpu


def chain(queue_tl, lane_macro=None, hard=False, dt=None):
    """reward = 0; for lane: for x in queue_length[lane]: reward = reward + (-1.0) * x   (_env.py:770-797).  queue_tl [T][L] float32.
    Differentiable episode: every x a float32 tensor.  Evaluation episode: an IDM lane's x is the Python float (n ** 2.0) * dt
    (_env.py:709-738), a cell lane's a float32 tensor: a double sum until the first tensor joins it, float32 from then on."""
    T, L = queue_tl.shape
    q = np.asarray(queue_tl, dtype=np.float32)
    rf, rd, tensor = np.float32(0.0), 0.0, not hard
    for l in range(L):
        macro = lane_macro is None or bool(lane_macro[l])
        if macro and not tensor:
            rf, tensor = np.float32(rd), True
        col = q[:, l]
        if tensor:
            for x in col:
                rf = np.float32(rf + np.float32(-1.0) * x)
        else:
            for x in col:
                n = np.rint(np.sqrt(float(x) / dt))
                rd = rd + -1.0 * ((n * n) * dt)
    return np.float32(rf) if tensor else np.float32(rd)


@pytest.fixture()
def chain_on():
    from dhts import _lib
    assert _lib.lib().dhts_set_option(_lib.OPT_REWARD_CHAIN, 1) == 0
    yield
    _lib.lib().dhts_set_option(_lib.OPT_REWARD_CHAIN, 0)


def _args(m):
    return (m["num_intersection"] ** 2, m["simulation_frequency"] * m["signal_length"], 1.0 / m["simulation_frequency"], m["speed_limit"],
            m["static_speed"], m["vehicle_length"])


@pytest.mark.parametrize("name", ["macro", "macro_2x2", "macro_long"])
def test_macro_network_reward_in_reference_order(cuda, golden_dir, chain_on, name):
    import torch
    from dhts import ops
    g = np.load(os.path.join(golden_dir, "itscp_%s.npz" % name))
    tab, m = itscp_tables(g)
    a = torch.tensor(np.tile(g["action"][None], (3, 1)), device=cuda)
    reward, queue = ops.net_macro_rollout(a, ops.DeviceNetTables(tab, cuda), *_args(m))
    for r in range(3):
        assert np.float32(reward[r].item()) == chain(queue[r].cpu().numpy())
    assert abs(float(reward[0]) - float(g["reward"])) <= 1e-6 * abs(float(g["reward"]))
    ev, q_ev = ops.net_macro_eval(a[:1], ops.DeviceNetTables(tab, cuda), *_args(m))
    assert np.float32(ev[0].item()) == chain(q_ev[0].cpu().numpy(), None, True, 1.0 / m["simulation_frequency"])


@pytest.mark.parametrize("name", ["hybrid_p2", "hybrid", "hybrid_s3", "hybrid_n2"])
def test_hybrid_network_reward_in_reference_order(cuda, golden_dir, chain_on, name):
    import torch
    from dhts import _lib, ops
    g = np.load(os.path.join(golden_dir, "itscp_%s.npz" % name))
    t, m = itscp_hybrid_tables(g)
    tab = ops.DeviceHybridTables(t, g["spawn_routes"], cuda)
    a = torch.tensor(np.tile(g["action"][None], (2, 1)), device=cuda, requires_grad=True)
    cut, reward, queue, counts = ops.net_hybrid_rollout(a, tab, *_args(m))
    for r in range(2):
        assert np.float32(reward[r].item()) == chain(queue[r].cpu().numpy(), t.lane_macro)
        assert np.float32(cut[r].item()) == np.float32(reward[r].item())
    got = float(reward[0])
    assert abs(got - float(g["reward"])) <= 2e-6 * abs(float(g["reward"]))          # (the queue terms themselves differ by <= 2.3e-6)
    # the order is the only thing the option changes: same queue terms, same gradient
    cut.sum().backward()
    _lib.lib().dhts_set_option(_lib.OPT_REWARD_CHAIN, 0)
    a2 = torch.tensor(np.tile(g["action"][None], (2, 1)), device=cuda, requires_grad=True)
    cut2, reward2, queue2, _ = ops.net_hybrid_rollout(a2, tab, *_args(m))
    cut2.sum().backward()
    assert torch.equal(queue, queue2) and torch.equal(a.grad, a2.grad)
    assert abs(float(reward2[0]) - got) <= 2e-5 * abs(got)


@pytest.mark.parametrize("name", ["eval_hybrid", "eval_hybrid_4x4", "eval_macro"])
def test_evaluation_episode_reward_in_reference_order(cuda, golden_dir, chain_on, name):
    import torch
    from dhts import ops
    g = np.load(os.path.join(golden_dir, "itscp_%s.npz" % name))
    t, m = itscp_hybrid_tables(g)
    routes = g["spawn_routes"] if g["spawn_routes"].shape[0] else -np.ones((1, 2), np.int32)
    a = torch.tensor(g["action"][None], device=cuda)
    reward, queue, counts = ops.net_hybrid_eval(a, ops.DeviceHybridTables(t, routes, cuda), *_args(m))
    assert np.float32(reward[0].item()) == chain(queue[0].cpu().numpy(), t.lane_macro, True, 1.0 / m["simulation_frequency"])
    assert abs(float(reward[0]) - float(g["reward"])) <= 1e-6 * abs(float(g["reward"]))


@pytest.mark.parametrize("persistent", [False, True])
@pytest.mark.parametrize("name", ["hybrid_n2l30", "micro_2x2", "eval_micro_2x2", "eval_hybrid_5x5"])
def test_stepwise_reward_in_reference_order(cuda, golden_dir, chain_on, name, persistent):
    import torch
    from dhts.stepwise import StepwiseNetwork
    from test_oracle_golden import itscp_micro_tables
    g = np.load(os.path.join(golden_dir, "itscp_%s.npz" % name))
    hard = name.startswith("eval")
    if "micro" in name:
        t, m, routes = itscp_micro_tables(g)
    else:
        t, m = itscp_hybrid_tables(g)
        routes = g["spawn_routes"]
    net = StepwiseNetwork(t, routes, cuda, lane_capacity=32, persistent=persistent)
    a = torch.tensor(g["action"], device=cuda, requires_grad=not hard)
    cut, reward, queue, counts = net.rollout(a, *_args(m), differentiable=not hard, loss_steps=0 if hard else 100)
    q = queue.cpu().numpy()
    dt = 1.0 / m["simulation_frequency"]
    assert np.float32(float(reward.detach())) == chain(q, t.lane_macro, hard, dt)
    if not hard:
        assert np.float32(float(cut.detach())) == chain(q[:100], t.lane_macro)
    else:
        assert float(cut) == float(reward)                    # (an evaluation episode has no restricted reward)
    assert abs(float(reward.detach()) - float(g["reward"])) <= 2e-6 * abs(float(g["reward"]))
