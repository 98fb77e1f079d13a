"""Replica-batched, data-parallel controller training (example/control/trainer.py: Trainer(env, n_replica=R), this build's consumer of
BASELINE config 5's pattern): the batched step's controller gradient equals the mean of the single-environment trainers'
(reference-style, trainer.py:144-205), on one rank and over two ranks with one all-reduce of the flat gradient.
CPU: two gloo ranks, the replicas' rollouts from the CPU oracle (test infrastructure).  GPU: the fused kernels."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import PKG, ROOT

COMMON = r'''
import json, os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(pkg)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np, torch


def make_env(seed, mode, n_int, sim_len):
    from example.control.itscp._env import ItscpEnv
    from example.control.itscp import problem as problems
    env = ItscpEnv()
    env.schedule_callback = problems.problem_2
    for k, v in dict(num_intersection=n_int, lane_length=5.0 if mode == "hybrid" else 10.0, num_lane=1, policy_length=sim_len, signal_length=1,
                     mode=mode, speed_limit=60.0, random_seed=seed).items():
        env.config[k] = v
    env.reset()
    return env


def flat_params(trainer):
    return torch.cat([p.detach().reshape(-1).cpu() for p in trainer.controller.parameters()])
'''

ORACLE_BATCH = r'''
class StubEnv:
    # what Trainer reads of an environment (observation / action boxes, observe()); no lanes: nothing here touches a GPU
    def __init__(self, n_obs, n_action, seed):
        from example.control.itscp._env import Box
        self.observation_space = Box(0, 1, shape=(n_obs,))
        self.action_space = Box(0.1, 0.9, shape=(n_action,))
        self._obs = np.random.default_rng(seed).uniform(0, 1, n_obs).astype(np.float32)

    def observe(self):
        return self._obs


class OracleBatch:
    # The ReplicaBatch interface with the rollouts from the CPU oracle on the reference's macro_small network, one inflow scaling per
    # replica: what a rank's fused launch computes.

    def __init__(self, seeds):
        from test_oracle_golden import itscp_tables
        g = np.load(os.path.join(%(root)r, "tests", "golden", "itscp_macro_small.npz"))
        self.tabs, self.envs = [], []
        for sd in seeds:
            t, m = itscp_tables(g)
            t.schedule = np.ascontiguousarray(t.schedule * np.random.default_rng(sd).uniform(0.5, 1.0))
            self.tabs.append(t)
            self.envs.append(StubEnv(12, len(g["action"]), sd))
        self.args = (m["num_intersection"] ** 2, m["simulation_frequency"] * m["signal_length"], 1.0 / m["simulation_frequency"], m["speed_limit"],
                     m["static_speed"], m["vehicle_length"])
        self.path = "oracle x%%d" %% len(seeds)

    def observe(self):
        return np.stack([e.observe() for e in self.envs])

    def rollout(self, actions, differentiable=True):
        from oracle import oracle as O
        tabs, args = self.tabs, self.args

        class F(torch.autograd.Function):
            @staticmethod
            def forward(ctx, a):
                outs = [O.net_macro(t, a[r].detach().numpy(), *args) for r, t in enumerate(tabs)]
                ctx.g = torch.tensor(np.stack([o["g_action"] for o in outs]))
                return torch.tensor([o["reward"] for o in outs], dtype=torch.float32)

            @staticmethod
            def backward(ctx, g):
                return g[:, None] * ctx.g
        return F.apply(actions)
'''

CPU_WORKER = COMMON + ORACLE_BATCH + r'''
from dhts import dist as D
from example.control.trainer import Trainer
rank, world, local = D.init(backend="gloo")
R = 2
torch.manual_seed(100 + rank)                      # different initial weights per rank: the trainer must broadcast rank 0's
batch = OracleBatch([11 + rank * R + r for r in range(R)])
tr = Trainer(batch.envs[0], network_size=(16,), lr=1e-2, device="cpu", n_replica=R)
tr.batch = batch
tr._ensure_batch()
w0 = flat_params(tr)
flat = tr.flat_gradient(tr.batch_loss(1))
tr.optimizer.step()
if rank == 0:
    print("RESULT " + json.dumps({"flat": flat.tolist(), "w0": w0.tolist(), "w1": flat_params(tr).tolist()}))
else:
    print("RESULT1 " + json.dumps({"w0": w0.tolist(), "w1": flat_params(tr).tolist()}))
'''


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def run_ranks(script, n, extra_env=None, timeout=600):
    port = free_port()
    procs = []
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(n), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="2", **(extra_env or {}))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=timeout) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    return [so for so, _ in outs]


def result(out, tag="RESULT "):
    line = [l for l in out.splitlines() if l.startswith(tag)][0]
    return json.loads(line[len(tag):])


def test_two_gloo_ranks_train_one_controller_on_oracle_replicas(oracle, tmp_path):
    """2 ranks x 2 replicas on CPU: rank 0's weights reach rank 1, the all-reduced flat gradient is the gradient of the mean reward
    over all four environments (computed here in one process), and both ranks take the same Adam step."""
    script = tmp_path / "worker.py"
    script.write_text(CPU_WORKER % {"root": ROOT, "pkg": PKG})
    outs = run_ranks(script, 2)
    r0, r1 = result(outs[0]), result(outs[1], "RESULT1 ")
    assert r0["w0"] == r1["w0"] and r0["w1"] == r1["w1"] and r0["w0"] != r0["w1"]
    # the same in one process: four environments, one controller (rank 0's initial weights)
    ns = {}
    exec(compile((COMMON + ORACLE_BATCH) % {"root": ROOT, "pkg": PKG}, "common", "exec"), ns)
    import torch
    from example.control.trainer import Trainer
    batch = ns["OracleBatch"]([11 + k for k in range(4)])
    torch.manual_seed(100)
    tr = Trainer(batch.envs[0], network_size=(16,), lr=1e-2, device="cpu", n_replica=4)
    tr.batch = batch
    assert np.array_equal(np.asarray(r0["w0"], np.float32), ns["flat_params"](tr).numpy())
    flat = tr.flat_gradient(tr.batch_loss(1)).numpy()
    ref = np.asarray(r0["flat"], np.float32)
    assert np.abs(flat - ref).max() <= 1e-5 * np.abs(flat).max()
    assert np.abs(flat[:-1]).max() > 0


GPU_WORKER = COMMON + r'''
from dhts import dist as D
from example.control.trainer import Trainer
rank, world, local = D.init()
R = 2
torch.manual_seed(7)
env = make_env(21, "hybrid", 3, 2)
tr = Trainer(env, network_size=(32,), lr=1e-2, n_replica=R)
flat = tr.flat_gradient(tr.batch_loss(1))
if rank == 0:
    print("RESULT " + json.dumps({"flat": flat.cpu().tolist(), "path": tr.batch.path}))
'''


@pytest.mark.gpu
def test_replica_batch_gradient_equals_single_environment_trainers(cuda, tmp_path):
    """R = 4 hybrid environments (own inflow schedules and per-step routes each) in ONE fused launch pair: the controller gradient of
    the batched step = the mean of four reference-style single-environment trainer gradients (<= 1e-5 of its largest entry); the same
    four environments as 2 ranks x 2 replicas on this one GPU over gloo give the same flat gradient after the all-reduce; and the
    episode rate of the batch against one environment per episode is printed."""
    import time
    import torch
    from example.control.trainer import Trainer
    ns = {}
    exec(compile(COMMON % {"root": ROOT, "pkg": PKG}, "common", "exec"), ns)
    torch.manual_seed(7)
    env = ns["make_env"](21, "hybrid", 3, 2)
    tr = Trainer(env, network_size=(32,), lr=1e-2, n_replica=4)
    flat = tr.flat_gradient(tr.batch_loss(1)).cpu().numpy()
    assert tr.batch.path == "fused x4"
    # four single-environment trainers with the same controller
    acc = None
    for k in range(4):
        e = tr.batch.envs[k]
        e.rewind() if getattr(e, "_fused_done", False) else None
        single = Trainer(e, network_size=(32,), lr=1e-2)
        single.controller.load_state_dict(tr.controller.state_dict())
        reward, _, _ = single.run_episode(True)
        single.optimizer.zero_grad()
        (-reward).backward()
        g = torch.cat([p.grad.reshape(-1) for p in single.controller.parameters()] + [(-reward).detach().reshape(1)]).cpu().numpy()
        acc = g if acc is None else acc + g
    acc /= 4
    err = np.abs(flat - acc).max() / np.abs(acc).max()
    print("batched controller gradient vs mean of 4 single-environment trainers: %.2e of max|g|" % err)
    assert err <= 1e-5
    # two ranks x two replicas on this GPU (gloo: RCCL refuses two ranks on one device)
    script = tmp_path / "worker.py"
    script.write_text(GPU_WORKER % {"root": ROOT, "pkg": PKG})
    outs = run_ranks(script, 2, {"DHTS_DIST_BACKEND": "gloo"})
    r0 = result(outs[0])
    ref = np.asarray(r0["flat"], np.float32)
    assert r0["path"] == "fused x2"
    assert np.abs(ref - flat).max() <= 1e-5 * np.abs(flat).max()
    # episode rates: 256 environments per launch pair against one
    for R in (1, 256):
        torch.manual_seed(7)
        t2 = Trainer(ns["make_env"](21, "hybrid", 3, 2), network_size=(256, 256), lr=1e-4, n_replica=R)
        step = (lambda: t2.train_epoch_batched(1)) if R > 1 else (lambda: t2.train_epoch(1))
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        dt_ = (time.perf_counter() - t0) / n
        print("Trainer, %3d environment(s) per optimiser step: %.1f ms per step, %.0f environment-episodes/s" % (R, 1e3 * dt_, R / dt_))


@pytest.mark.gpu
def test_replica_batch_of_a_network_beyond_the_fused_limits(cuda):
    """Trainer(env, n_replica = 3) on a hybrid network the fused kernels cannot hold (two lanes per approach, 30 m lanes: 252 lanes +
    1 152 cells): the replicas run as workgroups of the stepwise path's persistent kernels, and the batched controller gradient equals
    the mean of the single-environment trainers' (each on the same path through ItscpEnv.step)."""
    import torch
    from example.control.itscp._env import ItscpEnv
    from example.control.itscp import problem as problems
    from example.control.trainer import Trainer

    def make(seed):
        env = ItscpEnv()
        env.schedule_callback = problems.problem_1
        for k, v in dict(num_intersection=3, lane_length=30.0, num_lane=2, policy_length=2, signal_length=1, mode="hybrid", speed_limit=60.0,
                         random_seed=seed).items():
            env.config[k] = v
        env.reset()
        return env
    torch.manual_seed(3)
    tr = Trainer(make(31), network_size=(32,), lr=1e-2, n_replica=3)
    flat = tr.flat_gradient(tr.batch_loss(1)).cpu().numpy()
    assert tr.batch.path == "stepwise x3"
    acc = None
    for k in range(3):
        e = tr.batch.envs[k]
        single = Trainer(e, network_size=(32,), lr=1e-2)
        single.controller.load_state_dict(tr.controller.state_dict())
        reward, _, _ = single.run_episode(True)
        assert e.last_path == "stepwise"
        single.optimizer.zero_grad()
        (-reward).backward()
        g = torch.cat([p.grad.reshape(-1) for p in single.controller.parameters()] + [(-reward).detach().reshape(1)]).cpu().numpy()
        acc = g if acc is None else acc + g
    acc /= 3
    assert np.abs(flat - acc).max() <= 1e-5 * np.abs(acc).max()


@pytest.mark.gpu
def test_trainer_epoch_of_two_episodes_on_a_stepwise_network(cuda):
    """Trainer.train_epoch(num_episode = 2) sums two episodes and calls backward() once -- on a network that takes the
    stepwise path (one workspace per differentiable rollout now), single environment and replica batch."""
    import torch
    from example.control.itscp._env import ItscpEnv
    from example.control.itscp import problem as problems
    from example.control.trainer import Trainer

    def make(seed):
        env = ItscpEnv()
        env.schedule_callback = problems.problem_1
        for k, v in dict(num_intersection=3, lane_length=30.0, num_lane=2, policy_length=2, signal_length=1, mode="hybrid", speed_limit=60.0,
                         random_seed=seed).items():
            env.config[k] = v
        env.reset()
        return env
    torch.manual_seed(5)
    tr = Trainer(make(31), network_size=(32,), lr=1e-2)
    before = flat = torch.cat([p.detach().reshape(-1).clone() for p in tr.controller.parameters()])
    reward, _, _ = tr.run_episode(True)
    tr.optimizer.zero_grad()
    (-reward).backward()
    g1 = torch.cat([p.grad.reshape(-1).clone() for p in tr.controller.parameters()])
    assert tr.env.last_path == "stepwise"
    # two identical episodes (rewind keeps schedules and routes), one backward: the mean loss' gradient is one episode's
    total = 0
    for _ in range(2):
        r, _, _ = tr.run_episode(True)
        total = total + r
    tr.optimizer.zero_grad()
    ((-total) / 2).backward()
    g2 = torch.cat([p.grad.reshape(-1) for p in tr.controller.parameters()])
    assert torch.allclose(g2, g1, rtol=1e-6, atol=1e-7 * float(g1.abs().max()))
    loss = tr.train_epoch(2)                                            # the trainer's own loop, optimiser step included
    assert torch.isfinite(loss) and not torch.equal(before, torch.cat([p.detach().reshape(-1) for p in tr.controller.parameters()]))
    # replica batch on the same path
    tb = Trainer(make(41), network_size=(32,), lr=1e-2, n_replica=2)
    f1 = tb.flat_gradient(tb.batch_loss(1)).clone()
    f2 = tb.flat_gradient(tb.batch_loss(2))
    assert tb.batch.path == "stepwise x2"
    assert torch.allclose(f2, f1, rtol=1e-6, atol=1e-7 * float(f1.abs().max()))
