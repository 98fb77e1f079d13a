"""INTEGRATION.md's marked code blocks are executed as written, so the document cannot drift from the code (VERDICT r1:
its macro-network snippet had the wrong signature).  A block is marked by `<!-- snippet: NAME -->` in front of its fence."""
import os
import re

import numpy as np
import pytest

from conftest import PKG, ROOT


def snippets():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    out = {}
    for m in re.finditer(r"<!-- snippet: (\w+) -->\s*```python\n(.*?)```", text, re.S):
        out[m.group(1)] = m.group(2)
    return out


def test_every_marked_snippet_is_known():
    assert set(snippets()) == {"binding", "rollouts", "net_macro", "net_batched", "net_hybrid", "net_eval", "net_micro", "net_state", "net_stepwise",
                               "trainer_replicas", "net_vehicle_params"}


def test_binding_snippet_loads_the_library():
    """The ctypes stub of section 2 against the built library (no GPU call: signatures and symbols only)."""
    code = snippets()["binding"].replace("/path/to/libdhts.so", os.path.join(PKG, "csrc", "libdhts.so"))
    ns = {}
    exec(compile(code, "INTEGRATION.md:binding", "exec"), ns)
    d = ns["MacroDesc"](1, 100, 0.01, 5.0, 30.0)
    import ctypes as C
    assert ns["_lib"].dhts_macro_step_tape_bytes(C.byref(d)) == 3 * 128 * 16


def _itscp_env(mode, **cfg):
    from example.control.itscp._env import ItscpEnv
    from example.control.itscp.problem import problem_1
    np.random.seed(3)
    env = ItscpEnv()
    env.schedule_callback = problem_1
    for k, v in dict(mode=mode, speed_limit=60.0, **cfg).items():
        env.config[k] = v
    env.reset()
    return env


@pytest.mark.gpu
def test_rollout_snippet(cuda):
    import torch
    L, N, V, T = 3, 40, 12, 15
    g = torch.Generator().manual_seed(1)
    ns = dict(r0=(0.1 + 0.8 * torch.rand(L, N, generator=g)).to(cuda).requires_grad_(True),
              u0=(30 * torch.rand(L, N, generator=g)).to(cuda).requires_grad_(True),
              ghost_r=torch.rand(L, 2, generator=g).to(cuda).requires_grad_(True),
              ghost_u=(30 * torch.rand(L, 2, generator=g)).to(cuda).requires_grad_(True),
              r_target=torch.zeros(L, N, device=cuda), u_target=torch.zeros(L, N, device=cuda),
              T=T, dt=0.01, dx=5.0, u_max=30.0,
              p0=(torch.arange(V)[None, :] * 20.0 + 5 * torch.rand(L, V, generator=g)).to(cuda).requires_grad_(True),
              v0=(10 + 5 * torch.rand(L, V, generator=g)).to(cuda).requires_grad_(True),
              params=torch.tensor([30.0, 24.0, 27.0, 0.5, 0.1, 5.0], dtype=torch.float64, device=cuda)[:, None, None].expand(6, L, V).contiguous(),
              head=torch.tensor([[1000.0, 0.0]] * L, dtype=torch.float64, device=cuda))
    code = snippets()["rollouts"].replace('sys.path.insert(0, "diff-hybrid-traffic-sim_amd")', "pass")
    exec(compile(code, "INTEGRATION.md:rollouts", "exec"), ns)
    assert ns["r0"].grad is not None and torch.isfinite(ns["r0"].grad).all() and ns["ghost_u"].grad is not None
    assert ns["pT"].shape == (L, V) and torch.isfinite(ns["vT"]).all()


@pytest.mark.gpu
def test_net_macro_snippet(cuda):
    import torch
    env = _itscp_env("macro", num_intersection=1, lane_length=10.0, num_lane=1, policy_length=2, signal_length=1)
    action = (0.1 + 0.8 * torch.rand(4, env.action_size())).to(cuda).requires_grad_(True)
    ns = dict(env=env, action=action)
    exec(compile(snippets()["net_macro"], "INTEGRATION.md:net_macro", "exec"), ns)
    assert ns["reward"].shape == (4,) and action.grad is not None and torch.isfinite(action.grad).all()
    assert ns["queue"].shape == (4, env.num_timestep, ns["tab"].n_lanes)
    # the batched-lane path on the same tables: replica 0's episode
    fused_reward, fused_queue = ns["reward"].detach(), ns["queue"].detach()
    exec(compile(snippets()["net_batched"], "INTEGRATION.md:net_batched", "exec"), ns)
    assert ns["queue"].shape == (env.num_timestep, ns["tab"].n_lanes)
    assert abs(float(ns["reward"]) - float(fused_reward[0])) <= 1e-5 * abs(float(fused_reward[0]))
    assert float((ns["queue"] - fused_queue[0]).abs().max()) <= 1e-5 * float(fused_queue[0].abs().max())


@pytest.mark.gpu
def test_net_hybrid_snippet(cuda):
    import torch
    env = _itscp_env("hybrid", num_intersection=3, lane_length=5.0, num_lane=1, policy_length=4, signal_length=2)
    action = (0.1 + 0.8 * torch.rand(2, env.action_size())).to(cuda).requires_grad_(True)
    ns = dict(env=env, action=action)
    exec(compile(snippets()["net_hybrid"], "INTEGRATION.md:net_hybrid", "exec"), ns)
    assert ns["reward"].shape == (2,) and action.grad is not None and torch.isfinite(action.grad).all()
    assert ns["counts"].shape == (2, 4)


@pytest.mark.gpu
def test_net_vehicle_params_snippet(cuda):
    """Vehicles with their own IDM attributes beside the routes, on the tables the hybrid snippet builds."""
    import torch
    env = _itscp_env("hybrid", num_intersection=3, lane_length=5.0, num_lane=1, policy_length=8, signal_length=2)
    action = (0.1 + 0.8 * torch.rand(2, env.action_size())).to(cuda).requires_grad_(True)
    ns = dict(env=env, action=action)
    exec(compile(snippets()["net_hybrid"], "INTEGRATION.md:net_hybrid", "exec"), ns)
    np.random.seed(11)
    exec(compile(snippets()["net_vehicle_params"], "INTEGRATION.md:net_vehicle_params", "exec"), ns)
    assert ns["reward_rv"].shape == (2,) and torch.isfinite(ns["reward_rv"]).all()
    assert int(ns["counts_rv"][0, 0]) >= 1 and not torch.equal(ns["reward_rv"].detach(), ns["full_reward"])      # other vehicles, another episode


@pytest.mark.gpu
def test_net_eval_snippet(cuda):
    """Evaluation episodes of both network kinds, on the tables the two snippets above build."""
    import torch
    ns_m = dict(env=_itscp_env("macro", num_intersection=1, lane_length=10.0, num_lane=1, policy_length=2, signal_length=1))
    ns_m["action"] = (0.1 + 0.8 * torch.rand(4, ns_m["env"].action_size())).to(cuda).requires_grad_(True)
    exec(compile(snippets()["net_macro"], "INTEGRATION.md:net_macro", "exec"), ns_m)
    ns_h = dict(env=_itscp_env("hybrid", num_intersection=3, lane_length=5.0, num_lane=1, policy_length=4, signal_length=2))
    ns_h["action"] = (0.1 + 0.8 * torch.rand(2, ns_h["env"].action_size())).to(cuda).requires_grad_(True)
    exec(compile(snippets()["net_hybrid"], "INTEGRATION.md:net_hybrid", "exec"), ns_h)
    ns = dict(action=ns_m["action"].detach(), dev_tab_macro=ns_m["dev_tab"], n_inter_sq=ns_m["n_inter_sq"],
              frames_per_phase=ns_m["frames_per_phase"], dt=ns_m["dt"], u_max=ns_m["u_max"], action_hyb=ns_h["action"].detach(),
              dev_tab_hybrid=ns_h["dev_tab"], n_inter_sq_hyb=ns_h["n_inter_sq"], frames_per_phase_hyb=ns_h["frames_per_phase"])
    exec(compile(snippets()["net_eval"], "INTEGRATION.md:net_eval", "exec"), ns)
    assert ns["reward_macro"].shape == (4,) and ns["queue_macro"].shape == ns_m["queue"].shape
    assert ns["reward_hyb"].shape == (2,) and ns["counts"].shape == (2, 4) and torch.isfinite(ns["reward_hyb"]).all()


@pytest.mark.gpu
def test_net_micro_snippet(cuda):
    import torch
    env = _itscp_env("micro", num_intersection=1, lane_length=30.0, num_lane=1, policy_length=4, signal_length=2)
    action = (0.1 + 0.8 * torch.rand(3, env.action_size())).to(cuda).requires_grad_(True)
    ns = dict(env=env, action=action, np=np)
    exec(compile(snippets()["net_micro"], "INTEGRATION.md:net_micro", "exec"), ns)
    assert ns["reward"].shape == (3,) and action.grad is not None and torch.isfinite(action.grad).all()
    assert ns["tab"].n_cells == 0 and int(ns["counts"][:, 0].min()) > 0          # every replica admitted vehicles


@pytest.mark.gpu
def test_net_state_snippet(cuda):
    """The plain three-lane network from a given state: the snippet runs, vehicles are spawned and the gradient reaches (r0, u0)."""
    import torch
    n_cell, T = 10, 300
    g = torch.Generator().manual_seed(4)
    ns = dict(n_cell=n_cell, T=T, dx=5.0, dt=0.01, u_max=30.0,
              r0=(0.4 + 0.5 * torch.rand(n_cell, generator=g)).to(cuda).requires_grad_(True),
              u0=(10.0 + 15.0 * torch.rand(n_cell, generator=g)).to(cuda).requires_grad_(True),
              r_target=torch.zeros(n_cell, device=cuda), u_target=torch.zeros(n_cell, device=cuda),
              ghost0=torch.tensor([[[0.5, 12.0, 0.3, 20.0], [0.0, 30.0, 0.0, 30.0], [0.1, 25.0, 0.2, 22.0]]], device=cuda))
    exec(compile(snippets()["net_state"], "INTEGRATION.md:net_state", "exec"), ns)
    assert ns["n_spawned"] >= 1 and ns["n_events"] >= ns["n_spawned"]
    assert bool(torch.isfinite(ns["r0"].grad).all()) and float(ns["r0"].grad.abs().max()) > 0 and float(ns["u0"].grad.abs().max()) > 0


@pytest.mark.gpu
def test_net_stepwise_snippet(cuda):
    """The stepwise path on a hybrid network beyond the fused limits (two lanes per approach, 30 m lanes), single and as three replicas."""
    import torch
    env = _itscp_env("hybrid", num_intersection=3, lane_length=30.0, num_lane=2, policy_length=2, signal_length=1)
    action = (0.1 + 0.8 * torch.rand(env.action_size())).to(cuda).requires_grad_(True)
    ns = dict(env=env, action=action)
    exec(compile(snippets()["net_stepwise"], "INTEGRATION.md:net_stepwise", "exec"), ns)
    assert ns["tab"].n_cells + ns["tab"].n_lanes > 960
    assert action.grad is not None and bool(torch.isfinite(action.grad).all()) and ns["queue"].shape == (env.num_timestep, ns["tab"].n_lanes)
    assert ns["rewards"].shape == (3,) and ns["queues"].shape == (3, env.num_timestep, ns["tab"].n_lanes)
    assert float(ns["rewards"][0]) == float(ns["rewards"][2]) == float(ns["full_reward"])          # same tables, same action: same episode


@pytest.mark.gpu
def test_trainer_replicas_snippet(cuda):
    import torch
    env = _itscp_env("hybrid", num_intersection=3, lane_length=5.0, num_lane=1, policy_length=2, signal_length=1, random_seed=5)
    ns = dict(env=env)
    exec(compile(snippets()["trainer_replicas"], "INTEGRATION.md:trainer_replicas", "exec"), ns)
    assert ns["trainer"].batch.path == "fused x8" and bool(torch.isfinite(ns["flat"]).all()) and ns["flat"].numel() > 64 * 64
