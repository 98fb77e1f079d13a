"""Gradient-based training of a signal controller (reference example/control/trainer.py:14-226).

Same surface and on-disk formats as the reference's Trainer:
  train(num_episode_per_epoch, num_epoch, num_eval_epoch, num_eval_episode, log_path)
  evaluate / train_epoch / run_episode(differentiable) -> (episode_reward, action, info) / save / load
  <log_path>/eval.txt          one line "{:08f}" of -average reward per evaluation           (trainer.py:131-132)
  <log_path>/model.zip         th.save({"controller_state_dict", "optimizer_state_dict"})      (trainer.py:213-216)
  <log_path>/best/model.zip    the same, at the best evaluation so far                        (trainer.py:134-142)
Scalars "loss/train" and "loss/eval" go to TensorBoard when it is installed, else to <log_path>/scalars.tsv.
<log_path>/paths.txt (this build only) says, per epoch, which path the episodes took: "fused" (one workgroup per network, two
launches per episode), "batched" / "stepwise" (step by step on the device: networks beyond one workgroup) or "lane-by-lane" (the
operator per lane and step: minutes per episode) -- so that a flag that moves a run off the fast path is visible in its logs.

What differs, on purpose: the controller and the action live on the GPU and env.step(action, True) runs the fused network
kernels, so a training episode is two kernel launches; instead of deep-copying the environment for every episode
(trainer.py:172) the episode state is rewound (ItscpEnv.rewind), which keeps the uploaded tables."""
import os
from copy import deepcopy

import torch as th

from example.control.controller import Controller


class _TsvWriter:
    def __init__(self, path):
        self.path = os.path.join(path, "scalars.tsv")

    def add_scalar(self, tag, value, step):
        with open(self.path, "a") as f:
            f.write("{}\t{}\t{}\n".format(tag, int(step), float(value)))


def _make_writer(path):
    try:
        from torch.utils.tensorboard import SummaryWriter
        return SummaryWriter(path)
    except Exception:                                   # tensorboard is not part of this image
        return _TsvWriter(path)


class Trainer:

    def __init__(self, env, network_size=(256, 256), lr=1e-3, device=None):
        self.env = env
        in_shape, out_shape = env.observation_space.shape, env.action_space.shape
        assert len(in_shape) == 1 and len(out_shape) == 1, "flat observation and action vectors"
        self.device = th.device(device) if device is not None else th.device("cuda" if th.cuda.is_available() else "cpu")
        self.controller = Controller(in_shape[0], out_shape[0], list(network_size)).to(self.device)
        self.optimizer = th.optim.Adam(self.controller.parameters(), lr)
        self.best_eval_result = -float("inf")
        self.writer = None
        self.paths = {}                                   # (kind, path) -> episodes since the last log line

    def _log_paths(self, epoch, log_path):
        if self.paths:
            with open(log_path + "/paths.txt", "a") as f:
                f.write("epoch {} {}\n".format(epoch, " ".join("{}:{}={}".format(k[0], k[1], n) for k, n in sorted(self.paths.items()))))
            self.paths = {}

    def train(self, num_episode_per_epoch, num_epoch, num_eval_epoch, num_eval_episode, log_path, progress=True):
        os.makedirs(log_path, exist_ok=True)
        self.writer = _make_writer(log_path)
        self.best_eval_result = -float("inf")
        epochs = range(num_epoch)
        bar = None
        if progress:
            try:
                from tqdm import tqdm
                bar = epochs = tqdm(epochs)
            except Exception:
                bar = None
        for epoch in epochs:
            if epoch % max(int(num_eval_epoch), 1) == 0:
                self.evaluate(epoch, num_eval_episode, log_path)
            self.controller.train(True)
            loss = self.train_epoch(num_episode_per_epoch)
            self.writer.add_scalar("loss/train", float(loss), epoch)
            if bar is not None:
                bar.set_description("Loss: {:.6f}".format(float(loss)))
            self.save(log_path + "/model.zip")
            self._log_paths(epoch, log_path)

    def evaluate(self, epoch, num_episode, log_path):
        self.controller.train(False)
        total = 0.0
        with th.no_grad():
            for _ in range(num_episode):
                reward, _, _ = self.run_episode(False)
                total += float(reward)
        avg_reward = total / num_episode
        if self.writer is not None:
            self.writer.add_scalar("loss/eval", -avg_reward, epoch)
        with open(log_path + "/eval.txt", "a") as f:
            f.write("{:08f}\n".format(-avg_reward))
        if avg_reward > self.best_eval_result:
            self.best_eval_result = avg_reward
            os.makedirs(log_path + "/best", exist_ok=True)
            self.save(log_path + "/best/model.zip")
        return avg_reward

    def train_epoch(self, num_episode):
        total = 0
        for _ in range(num_episode):
            reward, _, _ = self.run_episode(True)
            total = total + reward
        loss = (-total) / num_episode
        self.optimizer.zero_grad()
        loss.backward()
        self.optimizer.step()
        return loss.detach()

    def policy_action(self, obs):
        """Controller logits squashed into the action box: low + (high - low) * sigmoid (trainer.py:182-187)."""
        logits = self.controller(th.as_tensor(obs, device=self.device))
        space = self.env.action_space
        low = th.as_tensor(space.low, device=self.device)
        high = th.as_tensor(space.high, device=self.device)
        return low + (high - low) * th.sigmoid(logits)

    def run_episode(self, differentiable):
        env = self.env
        if hasattr(env, "rewind"):
            env.rewind()                                # same schedules / routes, episode state back to reset()
            if not differentiable:
                # an evaluation episode must leave the training environment alone: a twin that shares everything a fused
                # episode only reads (and copies the lanes if it has to step lane by lane after all)
                env = env.episode_copy() if hasattr(env, "episode_copy") else deepcopy(env)
        else:
            env = deepcopy(env)
        obs = env.observe()
        episode_reward = 0
        while True:
            action = self.policy_action(obs)
            obs, reward, terminal, info = env.step(action, differentiable)
            episode_reward = episode_reward + reward
            if terminal:
                break
        key = ("train" if differentiable else "eval", getattr(env, "last_path", "lane-by-lane"))
        self.paths[key] = self.paths.get(key, 0) + 1
        return episode_reward, action, info

    def save(self, path):
        th.save({"controller_state_dict": self.controller.state_dict(),
                 "optimizer_state_dict": self.optimizer.state_dict()}, path)

    def load(self, path):
        checkpoint = th.load(path, map_location=self.device)
        self.controller.load_state_dict(checkpoint["controller_state_dict"])
        if "optimizer_state_dict" in checkpoint:
            self.optimizer.load_state_dict(checkpoint["optimizer_state_dict"])
