"""the controller-training driver around the hot path -- CLI flags of the reference's
example/control/itscp/run.py:12-24, Trainer surface, eval.txt / checkpoint formats (trainer.py:131-132, 213-216)."""
import os
import re

import numpy as np
import pytest


REFERENCE_FLAGS = {          # run.py:13-23: flag -> (type, default)
    "mode": (str, "macro"), "problem": (int, 1), "n_trial": (int, 5), "n_intersection": (int, 1), "n_lane": (int, 3),
    "lane_length": (float, 20.0), "speed_limit": (float, 60.0), "simulation_length": (int, 10), "signal_length": (int, 2),
    "n_episode": (int, 200), "lr": (float, 1e-3),
}


def test_run_flags_match_reference():
    from example.control.itscp.run import build_parser
    ns = build_parser().parse_args([])
    for flag, (ty, default) in REFERENCE_FLAGS.items():
        assert isinstance(getattr(ns, flag), ty) and getattr(ns, flag) == default, flag
    # the run_itscp_hybrid.sh line parses
    ns = build_parser().parse_args("--mode=hybrid --problem=1 --n_trial=1 --n_intersection=3 --n_lane=1 --lane_length=5 "
                                   "--speed_limit=60 --simulation_length=20 --signal_length=4 --n_episode=100 --lr=1e-4".split())
    assert (ns.mode, ns.n_intersection, ns.lane_length, ns.lr) == ("hybrid", 3, 5.0, 1e-4)
    with pytest.raises(SystemExit):
        build_parser().parse_args(["--mode=lwr"])
    with pytest.raises(SystemExit):
        build_parser().parse_args(["--problem=4"])


class _StubEnv:
    """A one-step environment with the attributes the trainer touches; reward = -|action - target|^2."""

    def __init__(self):
        from example.control.itscp._env import Box
        self.observation_space = Box(0, 1, shape=(6,))
        self.action_space = Box(0.1, 0.9, shape=(4,))
        self.target = np.array([0.2, 0.8, 0.5, 0.3], dtype=np.float32)
        self.calls = []

    def __deepcopy__(self, memo):          # episodes of this stub leave no state behind
        return self

    def observe(self):
        return np.linspace(0, 1, 6).astype(np.float32)

    def step(self, action, differentiable):
        import torch as th
        self.calls.append(bool(differentiable))
        reward = -((action - th.as_tensor(self.target, device=action.device)) ** 2).sum()
        return self.observe(), reward, True, {"img": []}


def test_trainer_surface_and_file_formats(tmp_path):
    import torch as th
    from example.control.trainer import Trainer
    th.manual_seed(0)
    env = _StubEnv()
    tr = Trainer(env, network_size=[16, 16], lr=1e-2, device="cpu")
    assert [n for n, _ in tr.controller.named_parameters()][:2] == ["network.0.weight", "network.0.bias"]
    a = tr.policy_action(env.observe())
    assert a.shape == (4,) and float(a.min()) >= 0.1 and float(a.max()) <= 0.9        # squashed into the action box
    log = str(tmp_path / "trial_0")
    tr.train(2, 41, 10, 1, log, progress=False)
    lines = open(os.path.join(log, "eval.txt")).read().split("\n")
    assert lines[-1] == "" and len(lines) == 6                                          # epochs 0, 10, 20, 30, 40
    assert all(re.fullmatch(r"-?\d+\.\d{6}", x) for x in lines[:-1])                    # "{:08f}" of -average reward
    vals = [float(x) for x in lines[:-1]]
    assert vals[-1] < 0.5 * vals[0]                                                     # the loss goes down
    for path in ("model.zip", "best/model.zip"):
        ck = th.load(os.path.join(log, path))
        assert sorted(ck) == ["controller_state_dict", "optimizer_state_dict"]
        assert sorted(ck["controller_state_dict"]) == sorted(tr.controller.state_dict())
    assert env.calls.count(False) == 5 and env.calls.count(True) == 82
    other = Trainer(env, network_size=[16, 16], lr=1e-2, device="cpu")
    other.load(os.path.join(log, "model.zip"))
    assert th.equal(other.policy_action(env.observe()), tr.policy_action(env.observe()))
    tags = {ln.split("\t")[0] for ln in open(os.path.join(log, "scalars.tsv"))} if os.path.exists(os.path.join(log, "scalars.tsv")) else {"loss/train", "loss/eval"}
    assert tags == {"loss/train", "loss/eval"}


@pytest.mark.gpu
def test_run_driver_end_to_end(cuda, tmp_path):
    """run.py on a small macro problem: result layout, formats, and the training episodes go through the fused kernels."""
    import torch as th
    from example.control.itscp import run
    root = str(tmp_path / "result")
    name = run.main(["--mode=macro", "--problem=1", "--n_trial=2", "--n_intersection=1", "--n_lane=3", "--lane_length=30",
                     "--simulation_length=2", "--signal_length=1", "--n_episode=4", "--lr=1e-3", "--seed=5", "--result_root", root])
    assert re.fullmatch(re.escape(root) + r"/macro_\d+", name)
    for trial in range(2):
        log = os.path.join(name, "trial_%d" % trial)
        lines = open(os.path.join(log, "eval.txt")).read().split()
        assert len(lines) == 5 and all(np.isfinite(float(x)) and float(x) >= 0 for x in lines)     # -reward = queue loss >= 0
        ck = th.load(os.path.join(log, "best", "model.zip"))
        assert sorted(ck) == ["controller_state_dict", "optimizer_state_dict"]


@pytest.mark.gpu
def test_trainer_hybrid_episode_gradient(cuda):
    """One training epoch on the run_itscp_hybrid.sh network: the episode is one fused rollout, its gradient reaches every
    controller parameter, rewinding reproduces the episode bit for bit."""
    import torch as th
    from example.control.itscp import run
    from example.control.trainer import Trainer
    args = run.build_parser().parse_args("--mode=hybrid --problem=1 --n_intersection=3 --n_lane=1 --lane_length=5 --speed_limit=60 "
                                         "--simulation_length=20 --signal_length=4 --seed=7".split())
    env = run.make_env(args)
    th.manual_seed(1)
    tr = Trainer(env, lr=1e-4)
    r1, a1, _ = tr.run_episode(True)
    assert env._fused_done and a1.shape == (45,) and a1.is_cuda
    r2, a2, _ = tr.run_episode(True)
    assert th.equal(r1.detach(), r2.detach()) and th.equal(a1.detach(), a2.detach())
    before = [p.detach().clone() for p in tr.controller.parameters()]
    loss = tr.train_epoch(1)
    assert abs(float(loss) + float(r1.detach())) <= 1e-6 * abs(float(r1.detach()))
    grads = [p.grad for p in tr.controller.parameters()]
    assert all(g is not None and th.isfinite(g).all() and float(g.abs().max()) > 0 for g in grads)
    assert any(not th.equal(b, p.detach()) for b, p in zip(before, tr.controller.parameters()))


@pytest.mark.gpu
def test_itscp_control_graph_replay_matches_eager(cuda):
    """examples/itscp_control.py --graph: a whole optimiser iteration of a replica batch (both fused launches, Adam, clamp)
    captured in a HIP graph -- possible because check_faults=False reads nothing back -- gives the rewards of the eager loop."""
    import subprocess
    import sys
    from conftest import ROOT
    base = [sys.executable, os.path.join(ROOT, "examples", "itscp_control.py"), "--mode", "hybrid", "--n_intersection", "3",
            "--lane_length", "5", "--simulation_length", "6", "--signal_length", "2", "--n_episode", "11", "--n_replica", "6",
            "--seed", "4"]
    outs = []
    for extra in ([], ["--graph"]):
        p = subprocess.run(base + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, cwd=ROOT)
        assert p.returncode == 0, p.stderr[-2000:]
        rows = [[float(x) for x in re.findall(r"(?:best|mean|worst) (-?\d+\.\d+)", l)] for l in p.stdout.splitlines() if l.startswith("episode")]
        outs.append(np.array([r for r in rows if len(r) == 3]))
    assert outs[0].shape == outs[1].shape and outs[0].shape[0] >= 2
    assert np.allclose(outs[0], outs[1], rtol=1e-5, atol=1e-6)
    assert outs[0][-1, 1] > outs[0][0, 1]          # the mean reward improved
