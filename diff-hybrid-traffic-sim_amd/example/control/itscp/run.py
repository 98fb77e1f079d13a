"""Intersection signal control, controller-training driver: the flags, defaults and result layout of the reference's
example/control/itscp/run.py:12-70 (./result/control/itscp/<mode>_<unix time>/trial_<k>/{eval.txt, model.zip, best/}).

    python -m example.control.itscp.run --mode=hybrid --problem=1 --n_trial=1 --n_intersection=3 --n_lane=1 \
        --lane_length=5 --speed_limit=60 --simulation_length=20 --signal_length=4 --n_episode=100 --lr=1e-4

(the run_itscp_hybrid.sh line).

Build extras: --n_replica R trains on R environments of the same topology at once (own inflow schedules each, one fused launch
pair per episode of the batch: example/control/replicas.py), --gpus N starts N ranks (one per GPU, R replicas each) whose
controller gradients are summed by one RCCL all-reduce per optimiser step (dhts.dist) -- BASELINE config 5's pattern with a real
consumer:  python -m example.control.itscp.run --mode=hybrid --n_intersection=3 --n_lane=1 --lane_length=5 --simulation_length=20
--signal_length=4 --n_trial=1 --n_episode=20 --n_replica 256 --gpus 8"""
import argparse
import os
from time import time


def build_parser():
    parser = argparse.ArgumentParser("Script to solve intersection signal control problem")
    parser.add_argument("--mode", type=str, choices=["macro", "micro", "hybrid"], default="macro")
    parser.add_argument("--problem", type=int, choices=[1, 2, 3], default=1)
    parser.add_argument("--n_trial", type=int, default=5)
    parser.add_argument("--n_intersection", type=int, default=1)
    parser.add_argument("--n_lane", type=int, default=3)
    parser.add_argument("--lane_length", type=float, default=20.)
    parser.add_argument("--speed_limit", type=float, default=60.)
    parser.add_argument("--simulation_length", type=int, default=10)
    parser.add_argument("--signal_length", type=int, default=2)
    parser.add_argument("--n_episode", type=int, default=200)
    parser.add_argument("--lr", type=float, default=1e-3)
    # build extras (not in the reference)
    parser.add_argument("--seed", type=int, default=0, help="> 0: np.random / torch seed for the drawn schedules, routes and weights")
    parser.add_argument("--result_root", type=str, default="./result/control/itscp")
    parser.add_argument("--n_replica", type=int, default=1, help="environments per rank trained on at once (one fused launch pair per batch)")
    parser.add_argument("--gpus", type=int, default=1, help="ranks (one per GPU); without a launcher the script starts them itself")
    return parser


def make_env(args):
    from example.control.itscp._env import ItscpEnv
    from example.control.itscp import problem as problems
    env = ItscpEnv()
    env.schedule_callback = getattr(problems, "problem_{}".format(args.problem))
    env.config["num_intersection"] = args.n_intersection
    env.config["lane_length"] = args.lane_length
    env.config["num_lane"] = args.n_lane
    env.config["render"] = False
    env.config["policy_length"] = args.simulation_length
    env.config["signal_length"] = args.signal_length
    env.config["mode"] = args.mode
    env.config["speed_limit"] = args.speed_limit
    if args.seed > 0:
        env.config["random_seed"] = args.seed
    env.reset()
    return env


def main(argv=None):
    import sys
    args = build_parser().parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        from dhts import dist as D                 # (importing it does not touch a GPU; the children own the devices)
        rc = D.spawn_ranks(args.gpus, sys.argv[1:] if argv is None else list(argv), module="example.control.itscp.run")
        if rc:
            raise SystemExit(rc)
        return None
    import torch as th
    from dhts import dist as D
    from example.control.trainer import Trainer
    rank, world, _ = D.init() if (args.gpus > 1 or "WORLD_SIZE" in os.environ) else (0, 1, 0)
    if args.seed > 0:
        th.manual_seed(args.seed)
    stamp = int(time())
    if world > 1:                                   # one run directory for all ranks: rank 0's clock
        t_ = th.tensor([stamp], dtype=th.int64)
        import torch.distributed as dist
        if dist.get_backend() == "nccl":
            t_ = t_.cuda()
        dist.broadcast(t_, src=0)
        stamp = int(t_.item())
    run_name = "{}/{}_{}".format(args.result_root, args.mode, stamp)
    if world > 1 and rank > 0:
        run_name += "/rank_{}".format(rank)        # (every rank keeps its own logs; the weights are the same on all of them)
    os.makedirs(run_name, exist_ok=True)
    env = make_env(args)
    for trial in range(args.n_trial):
        log_path = run_name + "/trial_{}".format(trial)
        env.render_eval = True
        trainer = Trainer(env, lr=args.lr, n_replica=args.n_replica)
        t0 = time()
        trainer.train(1, args.n_episode + 1, max(args.n_episode // 10, 1), 1, log_path, progress=rank == 0)
        if rank == 0 and (args.n_replica > 1 or world > 1):
            n_env = (args.n_episode + 1) * args.n_replica * world
            print("trial {}: {} training episodes of {} environment(s) on {} rank(s) in {:.2f} s: {:.1f} environment-episodes/s".format(
                trial, args.n_episode + 1, args.n_replica * world, world, time() - t0, n_env / (time() - t0)), flush=True)
    return run_name


if __name__ == "__main__":
    main()
