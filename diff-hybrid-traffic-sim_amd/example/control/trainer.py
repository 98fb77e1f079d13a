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
(trainer.py:172) the episode state is rewound (ItscpEnv.rewind), which keeps the uploaded tables.

Replica-batched, data-parallel training (this build only; BASELINE config 5 -- "optimisation-batch replicas shard across the GPUs
... all-reduce of the loss gradient"): Trainer(env, n_replica=R) holds R environments of the same topology with their own drawn
inflow schedules and routes (example/control/replicas.py).  An "episode" is then ONE controller forward on [R][obs], ONE fused
launch pair over the R replicas (a launch sized for 256 replicas takes as long as one for a single replica) and the mean reward;
with torch.distributed initialised (dhts.dist.init; run.py --gpus N starts the ranks) every rank holds R replicas of its own, the
flat controller gradient [d loss / d theta || loss] is summed over the ranks by one all-reduce (RCCL; <= 1.8 MB, latency-bound)
and every rank takes the same Adam step."""
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

    def __init__(self, env, network_size=(256, 256), lr=1e-3, device=None, n_replica=1):
        self.env = env
        self.n_replica = int(n_replica)
        self.batch = None                                 # ReplicaBatch, built at the first batched episode
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
            if self._batched():
                from dhts import dist as D
                acc = th.zeros(2, dtype=th.float32, device=self.device)
                for _ in range(num_episode):
                    rewards, _ = self.run_batch(False)
                    acc[0] += rewards.sum()
                    acc[1] += rewards.numel()
                D.allreduce_sum_(acc)
                total, num_episode = float(acc[0]), float(acc[1])
            else:
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
        if self._batched():
            return self.train_epoch_batched(num_episode)
        total = 0
        for _ in range(num_episode):
            reward, _, _ = self.run_episode(True)
            total = total + reward
        loss = (-total) / num_episode
        self.optimizer.zero_grad()
        loss.backward()
        self.optimizer.step()
        return loss.detach()

    # ---- replica-batched, data-parallel training ----------------------------------------------------------------------
    def _batched(self):
        from dhts import dist as D
        return self.n_replica > 1 or D._active()

    def _ensure_batch(self):
        from dhts import dist as D
        rank, world, _ = D.env_rank_world() if D._active() else (0, 1, 0)
        if self.batch is None:
            from example.control.replicas import ReplicaBatch
            # every rank draws its own replicas: seeds offset by rank * R (an unseeded environment draws from np.random as it comes)
            self.batch = ReplicaBatch(self.env, self.n_replica, self.device, seed_offset=rank * self.n_replica)
        if world > 1 and not getattr(self, "_controller_synced", False):
            # ... and all ranks start from the SAME controller: rank 0's weights
            import torch.distributed as dist
            for p_ in self.controller.parameters():
                buf = p_.data.cpu() if (p_.is_cuda and dist.get_backend() == "gloo") else p_.data
                dist.broadcast(buf, src=0)
                if buf is not p_.data:
                    p_.data.copy_(buf)
            self._controller_synced = True
        return self.batch

    def run_batch(self, differentiable):
        """One episode of every replica: rewards [R] (differentiable w.r.t. the controller's weights) and the actions [R][A]."""
        b = self._ensure_batch()
        obs = th.as_tensor(b.observe(), device=self.device)
        actions = self.policy_action(obs)
        rewards = b.rollout(actions, differentiable)
        key = ("train" if differentiable else "eval", b.path)
        self.paths[key] = self.paths.get(key, 0) + 1
        return rewards, actions

    def batch_loss(self, num_episode=1):
        """- mean reward over the episodes, the replicas and the ranks' share: the quantity whose gradient the ranks sum."""
        from dhts import dist as D
        world = D.env_rank_world()[1] if D._active() else 1
        total = 0
        for _ in range(num_episode):
            rewards, _ = self.run_batch(True)
            total = total + rewards.sum()
        return (-total) / (num_episode * self.n_replica * world)

    def flat_gradient(self, loss):
        """[d loss / d theta || loss] as one flat float32 buffer, summed over the ranks (one all-reduce); the parameters' .grad are set
        from it."""
        from dhts import dist as D
        params = [p_ for p_ in self.controller.parameters() if p_.requires_grad]
        self.optimizer.zero_grad()
        loss.backward()
        flat = th.cat([(p_.grad if p_.grad is not None else th.zeros_like(p_)).reshape(-1) for p_ in params] + [loss.detach().reshape(1)])
        D.allreduce_sum_(flat)
        o = 0
        for p_ in params:
            n = p_.numel()
            p_.grad = flat[o:o + n].reshape(p_.shape).clone()
            o += n
        return flat

    def train_epoch_batched(self, num_episode):
        flat = self.flat_gradient(self.batch_loss(num_episode))
        self.optimizer.step()
        return flat[-1].detach()

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
