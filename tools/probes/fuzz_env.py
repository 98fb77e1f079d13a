"""Fuzz: ItscpEnv.step -- the reference's entry point -- on RANDOM environments (grid size, lanes per road, lane length, speed limit,
episode and signal length, demand problem, mode macro / hybrid / micro, training and evaluation episodes) against the CPU oracle run on the
tables the environment itself builds.  Whatever path the environment picks (fused kernels, persistent or stepwise form) is the one
checked: queue terms and reward within 1e-5, vehicle counts equal; the gradient's worst entry is printed (an episode can sit on a
knife edge of the float32 arithmetic; above 1e-4 it is listed, not counted).  The goldens pin 46 reference episodes; this covers the
shapes between them.  GPU box:  python tools/probes/fuzz_env.py [n_envs] [seed]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-hybrid-traffic-sim_amd"), os.path.join(ROOT, "tests")]
from dhts import _lib      # noqa: E402
from dhts.network import MacroNetworkTables, group_routes      # noqa: E402
from example.control.itscp import problem as problems      # noqa: E402
from example.control.itscp._env import ItscpEnv      # noqa: E402
from oracle import oracle as O      # noqa: E402

O.build()
n_env = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2026
assert _lib.lib().dhts_set_option(_lib.OPT_REWARD_CHAIN, 1) == 0        # the reward as ItscpEnv._reward forms it (one float32 chain)
cuda = torch.device("cuda:0")
rng = np.random.default_rng(seed)
bad, listed, t_start = 0, 0, time.time()
modes = tuple(os.environ.get("FUZZ_ENV_MODES", "macro,hybrid,micro").split(","))
only = set(int(x) for x in os.environ["FUZZ_ENV_ONLY"].split(",")) if os.environ.get("FUZZ_ENV_ONLY") else None      # trial numbers to run
paths = {}
for trial in range(n_env):
    mode = modes[int(rng.integers(len(modes)))]
    n_int = int(rng.integers(1, 5))
    n_lane = int(rng.integers(1, 3 if n_int >= 3 else 4))
    cfg = dict(mode=mode, num_intersection=n_int, num_lane=n_lane, lane_length=float(rng.choice([5, 10, 15, 20, 30, 40, 60])),
               speed_limit=float(rng.choice([30, 45, 60])), policy_length=int(rng.choice([2, 3, 4, 6, 8, 12, 16, 20])),
               signal_length=int(rng.choice([1, 2, 4])), random_seed=int(rng.integers(1 << 20)))
    if mode == "micro":
        cfg["policy_length"] = min(cfg["policy_length"], 8)
    cfg["signal_length"] = min(cfg["signal_length"], cfg["policy_length"])      # (at least one signal phase: an episode without one has no action)
    hard = bool(rng.integers(4) == 0)
    if os.environ.get("FUZZ_ENV_TRACE"):
        print("trial %d: %s hard=%s" % (trial, cfg, hard), flush=True)
    env = ItscpEnv()
    env.schedule_callback = getattr(problems, "problem_%d" % int(rng.integers(1, 4)))
    env.config.update(cfg)
    env.reset()
    act = rng.uniform(0.1, 0.9, env.action_size()).astype(np.float32)
    args = (env.num_intersection ** 2, env.config["signal_length"] * env.config["simulation_frequency"],
            1.0 / env.config["simulation_frequency"], env.simulator.speed_limit, env.config["static_speed"], env.simulator.vehicle_length)
    # the checker's run, on the environment's own host tables
    # a quarter of the hybrid / micro environments: every vehicle with attributes of its own (micro_vehicle.py:75-121's ranges; hybrid: drawn
    # for 0.7 x the speed limit, so that a deposited vehicle stays inside the ARZ cells' CFL bound) -- dhts_hybrid_tables::veh_params
    own = mode != "macro" and bool(rng.integers(4) == 0)

    def attributes(limit):
        u = rng.random(5)
        return [limit * (1.5 + 0.5 * u[0]), limit * (1.0 + 0.5 * u[1]), limit * (0.8 + 0.4 * u[2]), 5.0 * (0.2 + 0.2 * u[3]), 0.2 + 0.4 * u[4], 5.0]
    if own and mode == "micro":
        for waiting in env.simulator.lane_waiting_micro_vehicle.values():
            for v in waiting:
                v.accel_max, v.accel_pref, v.target_speed, v.min_space, v.time_pref, v.length = attributes(env.simulator.speed_limit)
    if mode == "macro":
        tab = MacroNetworkTables.from_env(env)
    else:
        tab, routes, vp = env._fused_episode_inputs()
        if own and mode == "hybrid":
            vp = env.fused_vehicle_params = np.array([attributes(0.7 * env.simulator.speed_limit) for _ in range(len(routes))])
        if mode == "micro":
            draws = rng.random(env._fused_n_draws)
            env.fused_draws = draws
            tab.set_micro_sources(draws)
        else:
            env.fused_routes = routes
    if only is not None and trial not in only:
        continue            # (its random numbers are drawn: the selected trials see the streams of the full run)
    if mode == "macro":
        ref = O.net_macro(tab, act, *args, hard=hard)
        ref.setdefault("n_spawned", 0)
    else:
        if vp is None:
            gr, ptr = group_routes(routes, tab.n_lanes)
            gvp = None
        else:
            gr, ptr, gvp = group_routes(routes, tab.n_lanes, vp)
        ref = O.net_hybrid(tab, gr, ptr, act, *args, hard=hard, want_grad=not hard, vehicle_params=gvp)
    tag = "%-6s %dx%d x%d lanes of %4.0f m, %2.0f m/s, %d s / %d s, %s%s" % (
        mode, n_int, n_int, n_lane, cfg["lane_length"], cfg["speed_limit"], cfg["policy_length"], cfg["signal_length"], "eval " if hard else "train",
        ", own attributes" if own else "")
    if ref["rc"] != 0:
        # a CFL violation (rc 1) is the reference's assert: the product must fault as well, not return numbers
        verdict = ""
        if ref["rc"] == 1:
            try:
                env.step(torch.tensor(act, device=cuda, requires_grad=not hard), not hard)
                verdict = "; the product returned numbers  <-- MISMATCH"
                bad += 1
            except Exception as e:          # noqa: BLE001 (whatever the fault is raised as, it is printed)
                verdict = "; the product: %s: %s" % (type(e).__name__, str(e)[:90])
        print("%s: the checker refuses the episode (rc %d)%s" % (tag, ref["rc"], verdict), flush=True)
        continue
    if os.environ.get("FUZZ_ENV_TRACE"):
        print("  checker done", flush=True)
    # the product's run
    a = torch.tensor(act, device=cuda, requires_grad=not hard)
    obs, reward, done, info = env.step(a, not hard)
    path = getattr(env, "last_path", "lane by lane") if getattr(env, "_fused_done", False) else "lane by lane"
    paths[path] = paths.get(path, 0) + 1
    if not hard:
        reward.backward()
    q = np.array([env.queue_length[k] for k in env.lane.keys()], dtype=np.float32).T          # [T][L]
    # env.step's reward carries -reward_queue_c; the checker's is the plain chain
    r = float(reward.detach()) / (-env.reward_queue_c)
    eq = np.abs(q - ref["queue"]).max() / max(np.abs(ref["queue"]).max(), 1e-30)
    er = abs(r - ref["reward"]) / max(abs(ref["reward"]), 1e-30)
    ok = eq <= 1e-5 and er <= 1e-5
    if mode != "macro":
        ok = ok and int(env.fused_counts[0]) == ref["n_spawned"]
    eg = 0.0
    note = ""
    if not hard:
        g = a.grad.cpu().numpy() / (-env.reward_queue_c)
        fin = (bool(np.isfinite(g).all()), bool(np.isfinite(ref["g_action"]).all()))
        if fin == (True, True):
            eg = np.abs(g - ref["g_action"]).max() / max(np.abs(ref["g_action"]).max(), 1e-30)
        else:
            # a non-finite gradient: both sides or neither (where they are non-finite is compared too)
            eg = float("nan")
            same = np.array_equal(np.isfinite(g), np.isfinite(ref["g_action"]))
            note = "  <-- non-finite gradient: product %s, checker %s%s" % (
                "finite" if fin[0] else "%d of %d entries" % ((~np.isfinite(g)).sum(), g.size),
                "finite" if fin[1] else "%d of %d entries" % ((~np.isfinite(ref["g_action"])).sum(), g.size), "" if same else " -- DIFFERENT")
            if os.environ.get("FUZZ_ENV_TRACE"):
                print("      product %s\n      checker %s" % (g, ref["g_action"]))
            if not same:
                ok = False
    if not ok:
        bad += 1
        note += "  <-- MISMATCH (counts %s / %s)" % (getattr(env, "fused_counts", None), ref.get("n_spawned"))
    elif eg > 1e-4:
        # conditioning: how far does the CHECKER's own gradient move when one action changes by one float32 ulp?  (an episode on a knife edge
        # of the float32 arithmetic amplifies rounding differences between any two implementations; only a distance beyond that is a finding)
        cond = 0.0
        if mode != "macro":
            for k in np.argsort(-np.abs(g - ref["g_action"])):          # (the entries that differ most first; stops once the noise covers the distance)
                ak = act.copy()
                ak[k] = np.nextafter(ak[k], np.float32(2.0))
                rk = O.net_hybrid(tab, gr, ptr, ak, *args, hard=False, want_grad=True, vehicle_params=gvp)
                if rk["rc"] == 0:
                    cond = max(cond, np.abs(rk["g_action"] - ref["g_action"]).max() / max(np.abs(ref["g_action"]).max(), 1e-30))
                if cond > eg:
                    break
        if cond > eg:
            listed += 1
            note = "  <-- gradient above 1e-4; one ulp of an action moves the checker's own gradient by %.1e: ill-conditioned (listed)" % cond
        else:
            bad += 1
            note = "  <-- gradient above 1e-4 (one ulp of an action moves the checker's gradient by %.1e only)  MISMATCH" % cond
    if note and mode != "macro" and os.environ.get("FUZZ_ENV_DUMP"):          # the case for a closer look: host tables, routes, attributes, action
        import pickle
        os.makedirs(os.environ["FUZZ_ENV_DUMP"], exist_ok=True)
        with open(os.path.join(os.environ["FUZZ_ENV_DUMP"], "case_%d_%d.pkl" % (seed, trial)), "wb") as f:
            pickle.dump(dict(tab=tab, routes=routes, vp=vp, act=act, args=args, hard=hard, cfg=cfg), f)
    print("%3d %s | %4d lanes %5d cells, %-10s: queues %.1e reward %.1e gradient %.1e%s" % (
        trial, tag, tab.n_lanes, tab.n_cells, path, eq, er, eg, note), flush=True)
print("environments: %d, paths %s, mismatches: %d, gradients above 1e-4: %d (%.0f s)" % (n_env, paths, bad, listed, time.time() - t_start))
sys.exit(1 if bad else 0)
