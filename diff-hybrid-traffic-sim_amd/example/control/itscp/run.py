"""Intersection signal control, controller-training driver: the flags, defaults and result layout of the reference's
example/control/itscp/run.py:12-70 (./result/control/itscp/<mode>_<unix time>/trial_<k>/{eval.txt, model.zip, best/}).

    python -m example.control.itscp.run --mode=hybrid --problem=1 --n_trial=1 --n_intersection=3 --n_lane=1 \
        --lane_length=5 --speed_limit=60 --simulation_length=20 --signal_length=4 --n_episode=100 --lr=1e-4

(the run_itscp_hybrid.sh line).  `micro` mode is the reference's autodiff MicroLane path and is not part of this build."""
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
    args = build_parser().parse_args(argv)
    if args.mode == "micro":
        raise SystemExit("mode 'micro' (autodiff MicroLane) is outside this build; use macro or hybrid")
    import torch as th
    from example.control.trainer import Trainer
    if args.seed > 0:
        th.manual_seed(args.seed)
    run_name = "{}/{}_{}".format(args.result_root, args.mode, int(time()))
    os.makedirs(run_name, exist_ok=True)
    env = make_env(args)
    for trial in range(args.n_trial):
        log_path = run_name + "/trial_{}".format(trial)
        env.render_eval = True
        trainer = Trainer(env, lr=args.lr)
        trainer.train(1, args.n_episode + 1, max(args.n_episode // 10, 1), 1, log_path)
    return run_name


if __name__ == "__main__":
    main()
