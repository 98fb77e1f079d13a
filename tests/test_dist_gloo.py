"""The N > 1 path on CPU: world_size-2 gloo processes shard independent lanes with dhts.dist.shard_range and
all-reduce the flat [d loss / d theta_shared || loss] buffer.  The per-shard numbers come from the
CPU oracle here (test infrastructure); on the GPU box the same host logic drives the HIP kernels (bench.py)."""
import os
import socket
import subprocess
import sys

import numpy as np

from conftest import PKG, ROOT

WORKER = r'''
import json, os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(pkg)r)
import numpy as np, torch
from dhts import dist as D
from oracle import oracle as O

rank, world, local = D.init(backend="gloo")
assert world == 2
L, N, T, dt, dx, um = 7, 40, 30, 0.01, 5.0, 30.0          # 7 lanes over 2 ranks: ragged shards 4 + 3
rng = np.random.default_rng(123)
r0 = rng.uniform(0.05, 0.95, (L, N)).astype(np.float32)
u0 = rng.uniform(0.0, um, (L, N)).astype(np.float32)
theta = np.array([0.4, 12.0, 0.6, 7.0], np.float32)       # shared parameters: the ghost (r, u) of every lane
b, e = D.shard_range(L, rank, world)
gr = np.tile(theta[[0, 2]], (e - b, 1)); gu = np.tile(theta[[1, 3]], (e - b, 1))
f = O.macro_rollout_fwd(r0[b:e], u0[b:e], gr, gu, T, dt, dx, um)
g = O.macro_rollout_bwd(f, g_rT=2 * f["rT"], g_uT=2 * f["uT"])
loss = float(np.sum(f["rT"].astype(np.float64) ** 2) + np.sum(f["uT"].astype(np.float64) ** 2))
flat = torch.tensor([g["g_ghost_r"][:, 0].sum(), g["g_ghost_u"][:, 0].sum(), g["g_ghost_r"][:, 1].sum(),
                     g["g_ghost_u"][:, 1].sum(), loss], dtype=torch.float32)
D.allreduce_sum_(flat)
D.barrier()
tmax = D.max_over_ranks(1.0 + rank, torch.device("cpu"))
if rank == 0:
    print("RESULT " + json.dumps({"flat": flat.tolist(), "tmax": tmax, "shard": [b, e]}))
'''


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_shard_range_covers_all_units():
    from dhts import dist as D
    for n in (0, 1, 7, 8, 1024, 2048):
        for world in (1, 2, 3, 8):
            spans = [D.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gloo_allreduce_matches_single_process(oracle, tmp_path):
    import json
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT, "pkg": PKG})
    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=240) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
    line = [l for l in outs[0][0].splitlines() if l.startswith("RESULT ")][0]
    res = json.loads(line[len("RESULT "):])
    assert res["tmax"] == 2.0 and res["shard"] == [0, 4]
    # single-process reference over all 7 lanes
    L, N, T, dt, dx, um = 7, 40, 30, 0.01, 5.0, 30.0
    rng = np.random.default_rng(123)
    r0 = rng.uniform(0.05, 0.95, (L, N)).astype(np.float32)
    u0 = rng.uniform(0.0, um, (L, N)).astype(np.float32)
    theta = np.array([0.4, 12.0, 0.6, 7.0], np.float32)
    gr = np.tile(theta[[0, 2]], (L, 1))
    gu = np.tile(theta[[1, 3]], (L, 1))
    f = oracle.macro_rollout_fwd(r0, u0, gr, gu, T, dt, dx, um)
    g = oracle.macro_rollout_bwd(f, g_rT=2 * f["rT"], g_uT=2 * f["uT"])
    loss = float(np.sum(f["rT"].astype(np.float64) ** 2) + np.sum(f["uT"].astype(np.float64) ** 2))
    ref = [g["g_ghost_r"][:, 0].sum(), g["g_ghost_u"][:, 0].sum(), g["g_ghost_r"][:, 1].sum(), g["g_ghost_u"][:, 1].sum(), loss]
    assert np.allclose(res["flat"], ref, rtol=2e-6)


WORKER_NET = r'''
import json, os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(pkg)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np, torch
from dhts import dist as D
from dhts.network import group_routes
from oracle import oracle as O
from test_oracle_golden import itscp_hybrid_tables
import copy

rank, world, local = D.init(backend="gloo")
g = np.load(os.path.join(%(root)r, "tests", "golden", "itscp_hybrid_short.npz"))
t, m = itscp_hybrid_tables(g)
routes, ptr = group_routes(g["spawn_routes"], t.n_lanes)
args = (m["num_intersection"] ** 2, m["simulation_frequency"] * m["signal_length"], 1.0 / m["simulation_frequency"],
        m["speed_limit"], m["static_speed"], m["vehicle_length"])
R = 3                                                        # replicas: own inflow schedules, ONE shared signal schedule
scale = [1.0, 0.8, 0.6]
b, e = D.shard_range(R, rank, world)
flat = torch.zeros(len(g["action"]) + 1, dtype=torch.float32)
for r in range(b, e):
    x = copy.copy(t); x.schedule = np.ascontiguousarray(t.schedule * scale[r])
    o = O.net_hybrid(x, routes, ptr, g["action"], *args)
    assert o["rc"] == 0
    flat[:-1] += torch.tensor(o["g_action"]); flat[-1] += o["reward"]
D.allreduce_sum_(flat)
D.barrier()
if rank == 0:
    print("RESULT " + json.dumps({"flat": flat.tolist(), "shard": [b, e]}))
'''


def test_two_rank_gloo_replica_sharding(oracle, tmp_path):
    """BASELINE config 5's pattern on CPU: network replicas sharded over two ranks, the gradient w.r.t. the shared signal
    schedule and the reward all-reduced in one call."""
    import copy
    import json
    from dhts.network import group_routes
    from test_oracle_golden import itscp_hybrid_tables
    script = tmp_path / "worker_net.py"
    script.write_text(WORKER_NET % {"root": ROOT, "pkg": PKG})
    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=240) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
    res = json.loads([l for l in outs[0][0].splitlines() if l.startswith("RESULT ")][0][len("RESULT "):])
    assert res["shard"] == [0, 2]
    g = np.load(os.path.join(ROOT, "tests", "golden", "itscp_hybrid_short.npz"))
    t, m = itscp_hybrid_tables(g)
    routes, ptr = group_routes(g["spawn_routes"], t.n_lanes)
    args = (m["num_intersection"] ** 2, m["simulation_frequency"] * m["signal_length"], 1.0 / m["simulation_frequency"],
            m["speed_limit"], m["static_speed"], m["vehicle_length"])
    ref = np.zeros(len(g["action"]) + 1)
    for sc in (1.0, 0.8, 0.6):
        x = copy.copy(t)
        x.schedule = np.ascontiguousarray(t.schedule * sc)
        o = oracle.net_hybrid(x, routes, ptr, g["action"], *args)
        ref[:-1] += o["g_action"]
        ref[-1] += o["reward"]
    assert np.allclose(res["flat"], ref, rtol=1e-5, atol=1e-6 * np.abs(ref).max())


def _run_bench(args, extra_env=None, timeout=240):
    import time
    env = dict(os.environ, OMP_NUM_THREADS="1", DHTS_DIST_BACKEND="gloo")       # (by name: eight ranks never fit a test box's GPUs)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)                       # no launcher environment: bench.py starts its own ranks
    env.update(extra_env or {})
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    return p, time.time() - t0


def test_bench_launcher_eight_ranks_over_gloo():
    """bench.py --gpus 8 without a launcher: eight child ranks (own port, RANK / LOCAL_RANK / WORLD_SIZE set), the per-pass
    all-reduce of [gradient || loss], max-over-ranks timing, the gather of every rank's part and rank 0's ONE JSON line relayed
    by the parent -- the stub workload on CPU, collectives over gloo (the launcher and collective code of the 8-GPU run)."""
    import json
    p, _ = _run_bench(["--workload", "stub", "--gpus", "8", "--steps", "3", "--warmup", "1"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout                        # one line, from rank 0
    o = json.loads(lines[0])
    assert o["n_gpus"] == 8 and o["steps"] == 3 and o["warmup"] == 1
    chk = o["allreduce_check"]
    assert len(chk["rank_parts"]) == 8 and len(set(chk["rank_parts"])) == 8          # every rank contributed its own part
    assert chk["reduced"] == chk["sum_of_rank_parts"] == o["loss_last_pass"] and chk["grad_max_abs_diff"] == 0.0
    # rank r's loss: sum over (2 x 3) entries of (k + 100 r)^2
    want = [float(sum((k + 100.0 * r) ** 2 for k in range(6))) for r in range(8)]
    assert chk["rank_parts"] == want
    # who took part: the backend by name and one identity per rank (on a GPU node: name, PCI bus id, uuid of eight distinct devices)
    col = o["collective"]
    assert col["backend"] == "gloo" and col["world"] == 8 and [d["rank"] for d in col["devices"]] == list(range(8))
    # N > 1 appends the second workload (on GPUs: BASELINE config 5, 256 hybrid replicas per rank) with its own per-pass all-reduce
    # of [gradient || loss]: here the stub at another shape (3 x 3 per rank), timed and checked the same way
    assert len(o["also"]) == 1
    sec = o["also"][0]
    assert sec["n_gpus"] == 8 and sec["passes"] == 3 and sec["value"] > 0 and "x 8 ranks" in sec["config"]
    chk2 = sec["allreduce_check"]
    want2 = [float(sum((k + 100.0 * r) ** 2 for k in range(9))) for r in range(8)]
    assert chk2["rank_parts"] == want2 and chk2["buffer_floats"] == 4
    assert chk2["reduced"] == chk2["sum_of_rank_parts"] == sec["loss_last_pass"] and chk2["grad_max_abs_diff"] == 0.0


def test_more_ranks_than_gpus_is_refused_unless_gloo_is_named(monkeypatch):
    """dhts.dist.init never picks gloo behind the caller's back on a node that shows some, but fewer, GPUs than ranks."""
    import pytest
    import torch
    from dhts import dist as D
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: None)
    monkeypatch.setenv("WORLD_SIZE", "8")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("LOCAL_RANK", "0")
    monkeypatch.delenv("DHTS_DIST_BACKEND", raising=False)
    with pytest.raises(RuntimeError, match="refusing to fall back to gloo"):
        D.init()


def test_bench_launcher_ends_the_ranks_when_one_dies():
    """A rank that dies behind the warm-up leaves the others inside a collective: the launcher ends them (its 15 s window) and
    returns non-zero instead of hanging."""
    p, took = _run_bench(["--workload", "stub", "--gpus", "8", "--steps", "3", "--warmup", "1"], {"DHTS_STUB_FAIL_RANK": "5"}, timeout=120)
    assert p.returncode != 0
    assert "rank(s) failed" in p.stderr and "(5, 7)" in p.stderr, p.stderr[-1500:]
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]               # no result line from a broken run
    assert took < 90.0
