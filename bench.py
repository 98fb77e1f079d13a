#!/usr/bin/env python3
"""bench.py -- differentiable cell-steps/s (forward + adjoint) of the time-fused HIP stepper.

    python bench.py --gpus N --steps K --warmup W [--workload macro|micro]

One "step" = one pass of the hot path over one batch of synthetic input: a full differentiable rollout
(forward sweep writing the Jacobian tape + reverse sweep reading it back) of BASELINE.json configs[1]
(macro: 1024 lanes x 512 ARZ cells x 1000 time steps) or, with --workload micro,
configs[2] (4096 lanes x 256 IDM vehicles x 1000 time steps; C3), inputs resident in HBM.
For N > 1 there is one process per GPU: either the caller launches them (torch.distributed.run sets RANK /
LOCAL_RANK / WORLD_SIZE) or, when `--gpus N` is given without that environment, this script starts N child
processes of itself before anything touches a GPU, relays rank 0's JSON line and exits non-zero if a rank fails.
Every rank owns its own shard of independent lanes / replicas (weak scaling: the per-GPU batch is the
configuration above), there is no data-path collective, and the flat [d loss / d theta_shared || loss] buffer is
all-reduced over RCCL once per pass.  Rank 0 prints ONE JSON line.

roofline.achieved / frac are priced in the bytes the dominant kernel MOVES (its compact tape, checked against PMC passes:
at N = 1 the run takes them itself before it touches the GPU -- live_counters -- and falls back to the ones under profiles/); the reference's algorithmic tape bytes (48 B per cell-step, 32 B per vehicle-step) divided by
the same time are reported beside them as achieved_algorithmic / frac_algorithmic.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "diff-hybrid-traffic-sim_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured-achievable copy rate
MACRO_TAPE_B = 48           # float32 [3][2][2] per cell-step   (road/lane/dmacro_lane.py:56): the ALGORITHMIC bytes
MICRO_TAPE_B = 32           # float32 [2][2][2] per vehicle-step (road/lane/dmicro_lane.py:54)
# What the rollout kernels actually move is their compact tape (the interface tape, the second rows of dEgo /
# dLeading -- the same information, rebuilt into the blocks by the reverse sweep); its size comes from the library
# (dhts_macro_tape_bytes / dhts_micro_tape_bytes), not from a constant here.


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)      # the first passes of a process run ~8 % slower (clock ramp)
    ap.add_argument("--workload", choices=["macro", "micro", "itscp_macro", "itscp_hybrid", "itscp_stepwise", "stub"], default="macro",
                    help="stub: launcher / collective self-test on any device (tests/test_dist_gloo.py), not a measurement")
    ap.add_argument("--lanes", type=int, default=0, help="override lanes per GPU")
    ap.add_argument("--cells", type=int, default=0, help="override cells / vehicles per lane")
    ap.add_argument("--time-steps", type=int, default=0, help="override simulated time steps per rollout")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the sub-records of the default run (N = 1: configs 3 and 4, the stepwise "
                    "network, the replica sweep, config 2 x 8 lanes; N > 1: config 5)")
    ap.add_argument("--no-live-counters", action="store_true", help="N = 1: skip the rocprofv3 --pmc child passes taken before the timed region "
                    "(roofline.traffic / issue_side then quote the committed passes under profiles/, fingerprint permitting)")
    ap.add_argument("--also-replicas", type=int, default=0, help="N > 1: replicas per rank of the config 5 sub-record (default 256); "
                    "giving it forces the sub-record even when the headline's shape is overridden (tests)")
    return ap.parse_args()


def spawn_ranks(n):
    """`--gpus n` without a launcher: start n children of this script (RANK / LOCAL_RANK / WORLD_SIZE set), one per GPU.
    The parent never touches the GPU and never execs; it relays rank 0's stdout and returns the worst exit code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    import tempfile
    procs = []
    with tempfile.TemporaryFile() as rank0_out:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port))
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=rank0_out if r == 0 else sys.stderr))
        # a rank that dies leaves the others waiting in a collective: end them (by their own PIDs) instead of hanging
        rcs = [None] * n
        while any(c is None for c in rcs):
            time.sleep(0.2)
            for r, p in enumerate(procs):
                if rcs[r] is None:
                    rcs[r] = p.poll()
            if any(c not in (None, 0) for c in rcs):
                deadline = time.time() + 15.0
                while time.time() < deadline and any(p.poll() is None for p in procs):
                    time.sleep(0.2)
                for p in procs:
                    if p.poll() is None:
                        p.kill()
                rcs = [p.wait() for p in procs]
        rank0_out.seek(0)
        # rank 0's JSON line goes to stdout; anything a backend printed beside it (gloo's connection banner) to stderr
        for line in rank0_out.read().decode().splitlines(True):
            (sys.stdout if line.lstrip().startswith("{") else sys.stderr).write(line)
        sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(rcs) if c != 0]
    if bad:
        print("bench.py: rank(s) failed: %s" % bad, file=sys.stderr)
        return 1
    return 0


class MacroWorkload:
    """r0 ~ U[0.05, 0.95], u0 ~ U[0, u_max], fixed random ghosts, dx = 5, dt = 0.01, u_max = 30,
    loss = sum r_T^2 + sum u_T^2."""
    name = "macro_straight_1024x512x1000"
    unit_bytes = MACRO_TAPE_B
    unit_name = "cell-steps/s"
    # what each kernel runs into, qualitatively; every NUMBER quoted beside it comes from the counter passes
    # committed as profiles/issue_counters.json (issue_side below) or from this run's own events
    limiter = {"rollout_fwd": "the dependent instruction stream of the wavefronts that solve the queued interfaces (phase 2, 40 % of a step) and "
                              "instruction issue in phase 1 at 4 wavefronts per SIMD, not HBM",
               "rollout_bwd": "instruction issue + LDS / barrier latency at 4 wavefronts per SIMD (three steps of tape in flight)"}

    @staticmethod
    def inputs(rank, L, N, um=30.0):
        """The seeded synthetic inputs of a rank (CPU tensors): r0, u0 [L][N], ghost r, ghost u [L][2].  tests/test_gpu_parity.py
        takes lanes of exactly these tensors to pin the benchmarked kernel instantiations against the oracle."""
        gen = torch.Generator(device="cpu").manual_seed(2026 + rank)
        r0 = 0.05 + 0.9 * torch.rand(L, N, generator=gen)
        u0 = um * torch.rand(L, N, generator=gen)
        gr = 0.05 + 0.9 * torch.rand(L, 2, generator=gen)
        gu = um * torch.rand(L, 2, generator=gen)
        return r0, u0, gr, gu

    def __init__(self, dev, rank, L, N, T):
        from dhts import ops
        self.ops, self.L, self.N, self.T = ops, L, N, T
        self.dt, self.dx, self.um = 0.01, 5.0, 30.0
        r0, u0, gr, gu = self.inputs(rank, L, N, self.um)
        self.r0, self.u0, gr, gu = r0.to(dev), u0.to(dev), gr.to(dev), gu.to(dev)
        gy, gq = ops.macro_state_from_ru(gr, gu, self.um)
        self.ghost = torch.stack([gr, gy, gu, gq], dim=-1).contiguous()
        self.desc = ops.macro_desc(L, N, self.dt, self.dx, self.um)
        # zeros, not empty: the first touch of 17 GB would otherwise be billed to the first warm-up launch (and to the average
        # of a rocprofv3 --stats run of this command)
        self.tape = torch.zeros(ops.macro_tape_numel(self.desc, T), dtype=torch.float32, device=dev)
        self.tape_bytes = self.tape.numel() * 4
        self.err = ops.new_error_record(dev)
        self.out = tuple(torch.empty(L, N, device=dev) for _ in range(4))
        self.gout = (torch.empty(L, N, device=dev), torch.empty(L, N, device=dev))
        self.g_ghost = torch.zeros(L, 2, 2, dtype=torch.float64, device=dev)
        self.units = L * N * T                       # cell-steps per pass
        self.name = "macro_straight_%dx%dx%d" % (L, N, T)
        self.ev = []

    def tape_census(self):
        """What one launch moves of the tape, read off the tape's own row headers after a forward pass (include/dhts.h): the
        trivial flux Jacobians S (12 B per cell), the used part of the header (count, cnt indices) and the cnt exception
        entries (32 B each), each block rounded up to whole 128-byte lines -- HBM moves lines, not bytes."""
        if getattr(self, "_census", None) is None:
            N, rows = self.N, self.T * self.L
            s_f4 = ((3 * N + 3) // 4 + 7) // 8 * 8
            h_f4 = (8 + 2 * (N + 1) + 127) // 128 * 8
            e_f4 = (2 * (N + 1) + 7) // 8 * 8
            row = self.tape.view(rows, (s_f4 + h_f4 + e_f4) * 4)
            cnt = row[:, s_f4 * 4].contiguous().view(torch.int32).to(torch.int64)
            lines = lambda nbytes: (nbytes + 127) // 128 * 128
            s_bytes = rows * lines(12 * N)
            h_bytes = int(lines(8 + 2 * cnt).sum().item())
            e_bytes = int(lines(32 * cnt).sum().item())
            self._census = {"s_bytes": s_bytes, "header_bytes": h_bytes, "exception_bytes": e_bytes,
                            "exceptions": int(cnt.sum().item()), "interfaces": rows * (N + 1)}
        return self._census

    def moved_bytes_per_launch(self):
        """the tape bytes one launch writes (forward) / reads (reverse); state, ghosts and cotangents are O(cells), not
        O(cells x T)"""
        c = self.tape_census()
        return c["s_bytes"] + c["header_bytes"] + c["exception_bytes"]

    def one_pass(self, record=False):
        ops = self.ops
        y0, q0 = ops.macro_state_from_ru(self.r0, self.u0, self.um)
        if record:
            e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            e[0].record()
        rT, yT, uT, _ = ops.macro_rollout_fwd(self.desc, self.T, self.r0, y0, self.u0, q0, self.ghost,
                                              tape=self.tape, err=self.err, out=self.out)
        if record:
            e[1].record()
        loss = (rT * rT).sum() + (uT * uT).sum()
        g_r, g_y = 2.0 * rT, torch.zeros_like(rT)
        ops.macro_u_tap_bwd(rT, yT, 2.0 * uT, g_r, g_y, self.um)
        if record:
            e[2].record()
        g_r0, g_y0, _ = ops.macro_rollout_bwd(self.desc, self.T, self.tape, g_r, g_y, err=self.err, out=self.gout,
                                              g_ghost=self.g_ghost)
        if record:
            e[3].record()
            self.ev.append(e)
        g_u0 = ops.macro_state_from_ru_bwd(self.r0, self.u0, g_y0, g_r0, self.um)
        return loss, g_r0, g_u0

    def parity_check(self, g_a, g_b):
        """The last timed pass of THIS run against the oracle's run of the same lanes (cpu_baseline keeps its first pass): final
        (r, u) and d loss / d (r0, u0) of lanes 0 .. cores - 1, norm-relative per tensor."""
        o = getattr(self, "oracle_sample", None)
        if o is None:
            return None
        n = o["lanes"]
        st = max(_rel(self.out[0][:n].cpu().numpy(), o["state"][0]), _rel(self.out[2][:n].cpu().numpy(), o["state"][1]))
        gr = max(_rel(g_a[:n].cpu().numpy(), o["grad"][0]), _rel(g_b[:n].cpu().numpy(), o["grad"][1]))
        return {"state_rel": st, "grad_rel": gr, "lanes": n, "against": "oracle (C port, pinned by tests/golden) on the same inputs, all %d steps" % self.T,
                "what": "r_T, u_T | d loss / d r0, d loss / d u0 of the last timed pass", "tol_state": TOL_STATE, "tol_grad": TOL_GRAD}

    def cpu_baseline(self, seconds=10.0):
        """The C oracle (a port of the reference's algorithm) on the host cores: bounded sample of the same workload -- one
        lane per core x 512 cells x 1000 steps (a lane's tape stays with the thread that wrote it; the arrays of the first
        pass are reused by the later ones, so the passes do not fault in fresh pages), repeated for >= 10 s -- and the same
        code on one core; parallel_efficiency = all-cores rate / (cores x one-core rate)."""
        import numpy as np
        from oracle import oracle as O
        cores = host_cores()
        os.environ.setdefault("OMP_NUM_THREADS", str(cores))
        Lc, N, T = cores, self.N, self.T
        if Lc * N * T * 48 > 16e9:                   # bound the sample's tape to 16 GB of host memory
            T = max(50, int(16e9 // (Lc * N * 48)))
        # the first Lc lanes of the very tensors the GPU passes ran on (rank 0's batch)
        r0, u0, gr, gu = (np.ascontiguousarray(t.numpy()[:Lc]) for t in self.inputs(0, self.L, N, self.um))
        O.macro_rollout_fwd(r0[:2], u0[:2], gr[:2], gu[:2], 2, self.dt, self.dx, self.um)   # load + warm
        f = O.macro_rollout_fwd(r0, u0, gr, gu, T, self.dt, self.dx, self.um)               # first touch of the tape: not timed
        b = O.macro_rollout_bwd(f, g_rT=2 * f["rT"], g_uT=2 * f["uT"])
        if T == self.T and f["rc"] == 0:
            self.oracle_sample = {"lanes": Lc, "state": (f["rT"].copy(), f["uT"].copy()), "grad": (b["g_r0"].copy(), b["g_u0"].copy())}
        done, t0 = 0, time.perf_counter()
        while True:
            f = O.macro_rollout_fwd(r0, u0, gr, gu, T, self.dt, self.dx, self.um, out=f)
            O.macro_rollout_bwd(f, g_rT=2 * f["rT"], g_uT=2 * f["uT"])
            done += Lc * N * T
            el = time.perf_counter() - t0
            if el >= seconds:
                break
        del f
        # the same code on one core: a single lane leaves the OpenMP loop over lanes with one iteration
        f1 = O.macro_rollout_fwd(r0[:1], u0[:1], gr[:1], gu[:1], T, self.dt, self.dx, self.um)
        one, t1 = 0, time.perf_counter()
        while time.perf_counter() - t1 < 0.2 * seconds:
            f1 = O.macro_rollout_fwd(r0[:1], u0[:1], gr[:1], gu[:1], T, self.dt, self.dx, self.um, out=f1)
            O.macro_rollout_bwd(f1, g_rT=2 * f1["rT"], g_uT=2 * f1["uT"])
            one += N * T
        one_rate = one / (time.perf_counter() - t1)
        return {"value": done / el, "unit": "cell-steps/s", "cores": cores, "kind": "port",
                "sample": "%d lanes (one per core) x %d cells x %d steps fwd+bwd, tape reused, repeated %.1f s (OpenMP over lanes)" % (Lc, N, T, el),
                "one_core_value": one_rate, "parallel_efficiency": done / el / (cores * one_rate)}


class MicroWorkload:
    """256 default_micro_vehicle(30) per lane, p_i = 20 i + U[0, 10), v ~ U[9, 21], head gap 1000 / 0,
    dt = 0.01, loss = sum 1e-4 p_T^2 + sum v_T^2."""
    name = "micro_idm_4096x256x1000"
    unit_bytes = MICRO_TAPE_B
    unit_name = "vehicle-steps/s"
    limiter = {"rollout_fwd": "instruction issue (vector) beside the HBM write stream", "rollout_bwd": "hbm (the tape read stream)"}

    PARAMS = (30.0 * 1.0, 30.0 * 0.8, 30.0 * 0.9, 5.0 * 0.1, 0.1, 5.0)     # default_micro_vehicle(30), micro_vehicle.py:31-72

    @staticmethod
    def inputs(rank, L, V):
        """The seeded synthetic inputs of a rank (CPU tensors): p0, v0 [L][V] (tests pin the benchmarked instantiations on them)."""
        gen = torch.Generator(device="cpu").manual_seed(3026 + rank)
        p0 = torch.arange(V)[None, :] * 20.0 + 10.0 * torch.rand(L, V, generator=gen)
        v0 = 9.0 + 12.0 * torch.rand(L, V, generator=gen)
        return p0, v0

    def __init__(self, dev, rank, L, V, T):
        from dhts import ops
        self.ops, self.L, self.V, self.T = ops, L, V, T
        self.dt = 0.01
        p0, v0 = self.inputs(rank, L, V)
        self.p0, self.v0 = p0.to(dev), v0.to(dev)
        par = torch.tensor(self.PARAMS, dtype=torch.float64, device=dev)
        self.params = par[:, None, None].expand(6, L, V).contiguous()
        self.head = torch.tensor([[1000.0, 0.0]], dtype=torch.float64, device=dev).expand(L, 2).contiguous()
        self.desc = ops.micro_desc(L, V, self.dt)
        self.tape = torch.zeros(ops.micro_tape_numel(self.desc, T), dtype=torch.float32, device=dev)
        self.tape_bytes = self.tape.numel() * 4
        self.N = V
        self.err = ops.new_error_record(dev)
        self.out = (torch.empty(L, V, device=dev), torch.empty(L, V, device=dev))
        self.gout = (torch.empty(L, V, device=dev), torch.empty(L, V, device=dev))
        self.g_head = torch.zeros(L, 2, dtype=torch.float64, device=dev)
        self.units = L * V * T
        self.name = "micro_idm_%dx%dx%d" % (L, V, T)
        self.ev = []

    def moved_bytes_per_launch(self):
        return self.tape_bytes

    def one_pass(self, record=False):
        ops = self.ops
        if record:
            e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            e[0].record()
        pT, vT = ops.micro_rollout_fwd(self.desc, self.T, self.p0, self.v0, self.params, self.head, tape=self.tape,
                                       err=self.err, out=self.out)
        if record:
            e[1].record()
        loss = 1e-4 * (pT * pT).sum() + (vT * vT).sum()
        g_p, g_v = 2e-4 * pT, 2.0 * vT
        if record:
            e[2].record()
        g_p0, g_v0, _ = ops.micro_rollout_bwd(self.desc, self.T, self.tape, g_p, g_v, err=self.err, out=self.gout,
                                              g_head=self.g_head)
        if record:
            e[3].record()
            self.ev.append(e)
        return loss, g_p0, g_v0

    def parity_check(self, g_a, g_b):
        """As MacroWorkload.parity_check: final (p, v) and d loss / d (p0, v0) of lanes 0 .. cores - 1."""
        o = getattr(self, "oracle_sample", None)
        if o is None:
            return None
        n = o["lanes"]
        st = max(_rel(self.out[0][:n].cpu().numpy(), o["state"][0]), _rel(self.out[1][:n].cpu().numpy(), o["state"][1]))
        gr = max(_rel(g_a[:n].cpu().numpy(), o["grad"][0]), _rel(g_b[:n].cpu().numpy(), o["grad"][1]))
        return {"state_rel": st, "grad_rel": gr, "lanes": n, "against": "oracle (C port, pinned by tests/golden) on the same inputs, all %d steps" % self.T,
                "what": "p_T, v_T | d loss / d p0, d loss / d v0 of the last timed pass", "tol_state": TOL_STATE, "tol_grad": TOL_GRAD}

    def cpu_baseline(self, seconds=10.0):
        """As MacroWorkload.cpu_baseline: one lane per core x 256 vehicles x 1000 steps, tape reused."""
        import numpy as np
        from oracle import oracle as O
        cores = host_cores()
        os.environ.setdefault("OMP_NUM_THREADS", str(cores))
        Lc, V, T = cores, self.V, self.T
        if Lc * V * T * 32 > 16e9:
            T = max(50, int(16e9 // (Lc * V * 32)))
        p0, v0 = (np.ascontiguousarray(t.numpy()[:Lc]) for t in self.inputs(0, self.L, V))     # lanes of rank 0's own batch
        par = np.tile(np.array(self.PARAMS), (Lc, V, 1))
        O.micro_rollout_fwd(p0[:2], v0[:2], par[:2], 2, self.dt)
        f = O.micro_rollout_fwd(p0, v0, par, T, self.dt)
        b = O.micro_rollout_bwd(f, g_pT=2e-4 * f["pT"], g_vT=2 * f["vT"])
        if T == self.T and f["rc"] == 0:
            self.oracle_sample = {"lanes": Lc, "state": (f["pT"].copy(), f["vT"].copy()), "grad": (b["g_p0"].copy(), b["g_v0"].copy())}
        done, t0 = 0, time.perf_counter()
        while True:
            f = O.micro_rollout_fwd(p0, v0, par, T, self.dt, out=f)
            O.micro_rollout_bwd(f, g_pT=2e-4 * f["pT"], g_vT=2 * f["vT"])
            done += Lc * V * T
            el = time.perf_counter() - t0
            if el >= seconds:
                break
        del f
        f1 = O.micro_rollout_fwd(p0[:1], v0[:1], par[:1], T, self.dt)
        one, t1 = 0, time.perf_counter()
        while time.perf_counter() - t1 < 0.2 * seconds:
            f1 = O.micro_rollout_fwd(p0[:1], v0[:1], par[:1], T, self.dt, out=f1)
            O.micro_rollout_bwd(f1, g_pT=2e-4 * f1["pT"], g_vT=2 * f1["vT"])
            one += V * T
        one_rate = one / (time.perf_counter() - t1)
        return {"value": done / el, "unit": "vehicle-steps/s", "cores": cores, "kind": "port",
                "sample": "%d lanes (one per core) x %d vehicles x %d steps fwd+bwd, tape reused, repeated %.1f s (OpenMP over lanes)" % (Lc, V, T, el),
                "one_core_value": one_rate, "parallel_efficiency": done / el / (cores * one_rate)}


def problem_1_array(keys, T):
    """problem_1's inflow schedule (example/control/itscp/problem.py:5-70 with one session: one direction draws 0.9 + 0.1 U per
    lane, the other 0.01 U, constant over the episode) as a [T][lanes] array in one numpy call per replica -- the same
    distribution, without 86 400 list appends (replicas beyond the 256th of a batch: the 2 048-replica sweep)."""
    import numpy as np
    ns = np.random.random() > 0.5
    hot = np.array([(k.loc in ("north", "south")) if ns else (k.loc in ("west", "east")) for k in keys])
    r = np.random.random(len(keys))
    row = np.where(hot, 0.9 + 0.1 * r, 0.01 * r)
    return np.ascontiguousarray(np.broadcast_to(row[None, :], (T, len(keys))), dtype=np.float64)


class ItscpMacroWorkload:
    """run_itscp_macro.sh's network (1 intersection, 3 lanes, 30 m, 10 s, signal 2 s: 40 lanes, 236 cells, 300 steps, 5
    actions) x 256 replicas with per-replica problem_1 schedules and actions U[0.1, 0.9]: reward and d reward / d action
    of every replica in one fused launch each way (a stepping stone to BASELINE config 4, which adds micro lanes)."""
    name = "itscp_macro_256x(40 lanes, 236 cells)x300"
    limiter = {"rollout_fwd": "latency (300 dependent steps, one workgroup per replica)", "rollout_bwd": "latency (300 dependent steps, one workgroup per replica)"}
    unit_bytes = MACRO_TAPE_B
    unit_name = "cell-steps/s"

    def moved_bytes_per_launch(self):
        """per-cell-step blocks (48 B) + state history (16 B) + loss constant (4 B), per-lane-step queue terms"""
        return self.R * self.T * (self.N * (48 + 16 + 4) + self.n_lanes * 4)

    def __init__(self, dev, rank, R, _n, _t):
        import numpy as np
        from dhts import ops
        from dhts.network import MacroNetworkTables
        from example.control.itscp._env import ItscpEnv
        from example.control.itscp.problem import problem_1
        self.ops, self.R = ops, R
        np.random.seed(1000 * rank + 1)
        env = ItscpEnv()
        env.schedule_callback = problem_1
        for k, v in dict(num_intersection=1, lane_length=30.0, num_lane=3, policy_length=10, signal_length=2, mode="macro",
                         speed_limit=60.0).items():
            env.config[k] = v
        env.reset()
        base = MacroNetworkTables.from_env(env)
        tabs = [base]
        keys = list(env.lane.keys())
        for r in range(1, R):       # same topology and per-step routes, a fresh problem_1 inflow schedule per replica
            sched = env.schedule_callback(keys, env.num_timestep)
            t = MacroNetworkTables.__new__(MacroNetworkTables)
            t.__dict__.update(base.__dict__)
            t.schedule = np.ascontiguousarray(np.array([sched[k] for k in keys], dtype=np.float64).T)
            tabs.append(t)
        self.tab = ops.DeviceNetTables(tabs, dev)
        self.host_tab = base
        self.sq, self.F, self.dt, self.um = 1, 60, 1.0 / 30.0, 60.0
        gen = torch.Generator(device="cpu").manual_seed(77 + rank)
        self.action = (0.1 + 0.8 * torch.rand(R, env.action_size(), generator=gen)).to(dev).requires_grad_(True)
        self.units = R * tabs[0].n_cells * tabs[0].T
        self.L, self.N, self.T, self.n_lanes = R, tabs[0].n_cells, tabs[0].T, tabs[0].n_lanes
        self.name = "itscp_macro_%dx(%d lanes, %d cells)x%d" % (R, self.n_lanes, self.N, self.T)
        self.err = ops.new_error_record(dev)
        self.ev = []

    def one_pass(self, record=False):
        self.action.grad = None
        if record:
            e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            e[0].record()
        reward, _ = self.ops.net_macro_rollout(self.action, self.tab, self.sq, self.F, self.dt, self.um, err=self.err)
        self.reward = reward.detach()
        if record:
            e[1].record()
        loss = -reward.sum()
        if record:
            e[2].record()
        loss.backward()
        if record:
            e[3].record()
            self.ev.append(e)
        return loss.detach(), self.action.grad, self.action.grad

    def parity_check(self, g_a, g_b):
        """Replica 0 of the last timed pass against the oracle's episode: reward and d reward / d action."""
        o = getattr(self, "oracle_sample", None)
        if o is None:
            return None
        rew, rew_chain = float(self.reward[0]), reward_in_reference_order(self)
        return {"state_rel": abs(rew_chain - o["reward"]) / max(abs(o["reward"]), 1e-30), "grad_rel": _rel(-g_a[0].cpu().numpy(), o["g_action"]),
                "state_rel_kernel_order": abs(rew - o["reward"]) / max(abs(o["reward"]), 1e-30),
                "lanes": 1, "against": "oracle (C port, pinned by tests/golden) on replica 0's schedule and action, all %d steps" % self.T,
                "what": "reward (summed in the reference's order, DHTS_OPT_REWARD_CHAIN, on one more untimed pass; state_rel_kernel_order: as the "
                        "timed passes sum it) | d reward / d action of replica 0 in the last timed pass", "tol_state": TOL_STATE, "tol_grad": TOL_GRAD}

    def cpu_baseline(self, seconds=10.0):
        """The C oracle of the macro network (scalar, one core): whole episodes of replica 0, repeated for >= 10 s."""
        from oracle import oracle as O
        a = self.action[0].detach().cpu().numpy()
        done, t0 = 0, time.perf_counter()
        while True:
            o = O.net_macro(self.host_tab, a, self.sq, self.F, self.dt, self.um)
            if done == 0:
                self.oracle_sample = {"reward": float(o["reward"]), "g_action": o["g_action"].copy()}
            done += self.N * self.T
            el = time.perf_counter() - t0
            if el >= seconds:
                break
        return {"value": done / el, "unit": "cell-steps/s", "cores": 1, "kind": "port",
                "sample": "replica 0's episode (%d cells x %d steps) fwd+bwd, repeated %.1f s on one core" % (self.N, self.T, el)}


class ItscpHybridWorkload:
    """run_itscp_hybrid.sh's network (3 x 3 intersections, 1 lane, 20 s, signal 4 s: 144 lanes of which the 16 of the
    centre intersection are micro, 256 cells, 600 steps, 45 actions) x 256 replicas with per-replica problem_1 schedules and actions U[0.1, 0.9]: reward and
    d reward / d action of every replica in one fused launch each way (BASELINE config 4)."""
    name = "itscp_hybrid_256x(144 lanes, 256 cells, 16 micro lanes)x600"
    limiter = {"rollout_fwd": "latency (600 dependent steps, one workgroup per replica)", "rollout_bwd": "latency (600 dependent steps, one workgroup per replica)"}
    unit_bytes = MACRO_TAPE_B
    unit_name = "cell-steps/s"

    def moved_bytes_per_launch(self):
        """the interface tape (two 2x2 products = 32 B per interface slot and step), state history (16 B) + loss constant (4 B)
        per cell-step, per-lane-step queue terms, and the 36-byte records of the micro side (counts[:, 2] = records per replica)"""
        recs = int(self.counts[:, 2].sum()) if self.counts is not None else 0
        nip = (self.N + self.n_lanes + 63) // 64 * 64
        return self.R * self.T * (nip * 32 + self.N * (16 + 4) + self.n_lanes * 4) + recs * 36

    def __init__(self, dev, rank, R, _n, _t):
        import numpy as np
        from dhts import ops
        from dhts.network import HybridNetworkTables
        from example.control.itscp._env import ItscpEnv
        from example.control.itscp.problem import problem_1
        self.ops, self.R = ops, R
        np.random.seed(1000 * rank + 9)
        env = ItscpEnv()
        env.schedule_callback = problem_1
        for k, v in dict(num_intersection=3, lane_length=5.0, num_lane=1, policy_length=20, signal_length=4, mode="hybrid",
                         speed_limit=60.0).items():
            env.config[k] = v
        env.reset()
        tab = HybridNetworkTables.from_env(env)
        # pre-drawn routes: 8 per micro lane that a macro lane feeds (RoadNetwork.create_random_route)
        routes = []
        for l in range(tab.n_lanes):
            if tab.lane_macro[l] == 0 and any(tab.lane_macro[a] for a in tab.prev_lanes[l]):
                for _ in range(8):
                    r = env.simulator.create_random_route(l).route
                    routes.append(list(r) + [-1] * (32 - len(r)))
        tabs = [tab]
        keys = list(env.lane.keys())
        for r in range(1, R):       # same topology and per-step routes, a fresh problem_1 inflow schedule per replica
            t = HybridNetworkTables.__new__(HybridNetworkTables)
            t.__dict__.update(tab.__dict__)
            if r < 256:
                sched = env.schedule_callback(keys, env.num_timestep)
                t.schedule = np.ascontiguousarray(np.array([sched[k] for k in keys], dtype=np.float64).T)
            else:
                t.schedule = problem_1_array(keys, env.num_timestep)
            tabs.append(t)
        self.tab = ops.DeviceHybridTables(tabs, np.array(routes, dtype=np.int32), dev)
        self.host_tab, self.host_routes = tab, np.array(routes, dtype=np.int32)
        self.sq, self.F, self.dt, self.um = 9, 120, 1.0 / 30.0, 60.0
        gen = torch.Generator(device="cpu").manual_seed(177 + rank)
        self.action = (0.1 + 0.8 * torch.rand(R, env.action_size(), generator=gen)).to(dev).requires_grad_(True)
        self.units = R * tab.n_cells * tab.T
        self.L, self.N, self.T, self.n_lanes = R, tab.n_cells, tab.T, tab.n_lanes
        self.name = "itscp_hybrid_%dx(%d lanes, %d cells, %d micro lanes)x%d" % (R, self.n_lanes, self.N,
                                                                                 int((np.asarray(tab.lane_macro) == 0).sum()), self.T)
        self.err = ops.new_error_record(dev)
        self.ev = []
        self.counts = None

    def restrict(self, R):
        """The first R replicas of this batch as the batch (same uploaded tables, a prefix of the actions): the replica sweep
        times several batch sizes of ONE instance."""
        if getattr(self, "_full", None) is None:
            self._full = (self.tab, self.action.detach(), self.n_lanes)
        tab, action, _ = self._full
        assert R <= action.shape[0]
        self.tab, self.action = tab.first(R), action[:R].clone().requires_grad_(True)
        self.R = self.L = R
        self.units = R * self.N * self.T
        self.name = self.name.replace(self.name.split("x(")[0], "itscp_hybrid_%d" % R, 1)
        self.ev, self.counts = [], None
        self.err.zero_()
        return self

    def one_pass(self, record=False):
        self.action.grad = None
        if record:
            e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            e[0].record()
        # faults go to the workload's own sticky record, read once behind the timed region (as for the straight lanes)
        reward, _, _, self.counts = self.ops.net_hybrid_rollout(self.action, self.tab, self.sq, self.F, self.dt, self.um, err=self.err)
        self.reward = reward.detach()
        if record:
            e[1].record()
        loss = -reward.sum()
        if record:
            e[2].record()
        loss.backward()
        if record:
            e[3].record()
            self.ev.append(e)
        # A replica whose reverse sweep met 0 x inf (a head gap clamped to exactly 0, didm.py:60-70: the reference ASSERTS on such an
        # action, dmacro_lane.py:308) leaves NaN in its own row and a DHTS_FAULT_NAN record: a batch drops that member, as the
        # replica-batched trainer does, and the line says how many there were (no host sync here)
        g = self.drop_nonfinite(self.action.grad)
        return loss.detach(), g, g

    def drop_nonfinite(self, g):
        bad = ~torch.isfinite(g).all(dim=1)
        self.dropped = bad.sum()
        return torch.where(bad[:, None], torch.zeros_like(g), g)

    def dropped_replicas(self):
        d = getattr(self, "dropped", None)
        return 0 if d is None else int(d)

    def parity_check(self, g_a, g_b):
        """Replica 0 of the last timed pass against the oracle's episode of the same schedule and action: reward, number of
        vehicles spawned, d reward / d action (the pass differentiates -sum reward)."""
        o = getattr(self, "oracle_sample", None)
        if o is None:
            return None
        rew, rew_chain = float(self.reward[0]), reward_in_reference_order(self)
        return {"state_rel": abs(rew_chain - o["reward"]) / max(abs(o["reward"]), 1e-30), "grad_rel": _rel(-g_a[0].cpu().numpy(), o["g_action"]),
                "state_rel_kernel_order": abs(rew - o["reward"]) / max(abs(o["reward"]), 1e-30),
                "lanes": 1, "vehicles_spawned": [int(self.counts[0, 0]), o["n_spawned"]],
                "against": "oracle (C port, pinned by tests/golden) on replica 0's schedule and action, all %d steps" % self.T,
                "what": "reward (summed in the reference's order, DHTS_OPT_REWARD_CHAIN, on one more untimed pass; state_rel_kernel_order: as the "
                        "timed passes sum it) | d reward / d action [%d] of replica 0 in the last timed pass; vehicles spawned (kernel, oracle)" % g_a.shape[1],
                "tol_state": TOL_STATE, "tol_grad": TOL_GRAD}

    def cpu_baseline(self, seconds=10.0):
        """The C oracle of the hybrid network (scalar, one core): whole episodes of replica 0, repeated for >= 10 s."""
        from dhts.network import group_routes
        from oracle import oracle as O
        a = self.action[0].detach().cpu().numpy()
        routes, ptr = group_routes(self.host_routes, self.host_tab.n_lanes)
        done, t0 = 0, time.perf_counter()
        while True:
            o = O.net_hybrid(self.host_tab, routes, ptr, a, self.sq, self.F, self.dt, self.um)
            assert o["rc"] == 0
            if done == 0:
                self.oracle_sample = {"reward": float(o["reward"]), "n_spawned": int(o["n_spawned"]), "g_action": o["g_action"].copy()}
            done += self.N * self.T
            el = time.perf_counter() - t0
            if el >= seconds:
                break
        return {"value": done / el, "unit": "cell-steps/s", "cores": 1, "kind": "port",
                "sample": "replica 0's episode (%d cells x %d steps, %d vehicles) fwd+bwd, repeated %.1f s on one core"
                          % (self.N, self.T, o["n_spawned"], el)}


class ItscpStepwiseWorkload(ItscpHybridWorkload):
    """A hybrid network BEYOND the fused kernels' one-workgroup limits -- run_itscp_hybrid.sh with --n_lane=2 --lane_length=30 over 8 s:
    252 lanes (28 IDM lanes), 1 152 cells, 240 steps, 36 actions -- x 256 replicas like config 4 (own problem_1 inflow schedules and actions) on the
    stepwise path's persistent kernels (dhts_netstep_rollout_fwd / _bwd, one workgroup per replica; dhts/stepwise.py).  Not a BASELINE
    configuration: the reference's CLI reaches it with two flags, and until round 5 it ran lane by lane (minutes per episode)."""
    limiter = {"rollout_fwd": "instruction issue + barriers of ONE compute unit per replica (16 wavefronts, ~17 phases per step, 240 steps; docs/history/round_5_design_notebook.md section 9)",
               "rollout_bwd": "instruction issue + barriers of ONE compute unit per replica (16 wavefronts, 240 steps; docs/history/round_5_design_notebook.md section 9)"}

    def moved_bytes_per_launch(self):
        """blocks dqs[c][3][2][2] (48 B) + state history (16 B) + loss constant (4 B) per cell-step, queue terms per lane-step"""
        return self.R * self.T * (self.N * (48 + 16 + 4) + 4 * self.n_lanes)

    def __init__(self, dev, rank, R, _n, _t):
        import numpy as np
        from dhts import ops
        from dhts.network import HybridNetworkTables
        from dhts.stepwise import StepwiseNetwork, default_lane_capacity
        from example.control.itscp._env import ItscpEnv
        from example.control.itscp.problem import problem_1
        self.ops, self.R = ops, R
        np.random.seed(1000 * rank + 11)
        env = ItscpEnv()
        env.schedule_callback = problem_1
        for k, v in dict(num_intersection=3, lane_length=30.0, num_lane=2, policy_length=8, signal_length=2, mode="hybrid", speed_limit=60.0).items():
            env.config[k] = v
        env.reset()
        tab = HybridNetworkTables.from_env(env)
        routes = []
        for l in range(tab.n_lanes):
            if tab.lane_macro[l] == 0 and any(tab.lane_macro[a] for a in tab.prev_lanes[l]):
                for _ in range(8):
                    r = env.simulator.create_random_route(l).route
                    routes.append(list(r) + [-1] * (32 - len(r)))
        tabs = [tab]
        keys = list(env.lane.keys())
        for r in range(1, R):
            sched = env.schedule_callback(keys, env.num_timestep)
            t = HybridNetworkTables.__new__(HybridNetworkTables)
            t.__dict__.update(tab.__dict__)
            t.schedule = np.ascontiguousarray(np.array([sched[k] for k in keys], dtype=np.float64).T)
            tabs.append(t)
        self.host_tab, self.host_routes = tab, np.array(routes, dtype=np.int32)
        self.net = StepwiseNetwork(tabs, self.host_routes, dev, lane_capacity=default_lane_capacity(tab, env.simulator.vehicle_length), persistent=True)
        self.sq, self.F, self.dt, self.um = 9, 60, 1.0 / 30.0, 60.0
        gen = torch.Generator(device="cpu").manual_seed(277 + rank)
        self.action = (0.1 + 0.8 * torch.rand(R, env.action_size(), generator=gen)).to(dev).requires_grad_(True)
        self.units = R * tab.n_cells * tab.T
        self.L, self.N, self.T, self.n_lanes = R, tab.n_cells, tab.T, tab.n_lanes
        self.name = "itscp_stepwise_%dx(%d lanes, %d cells, %d micro lanes)x%d" % (R, self.n_lanes, self.N, int((np.asarray(tab.lane_macro) == 0).sum()), self.T)
        self.err = self.net.err
        self.ev = []
        self.counts = None

    def one_pass(self, record=False):
        self.action.grad = None
        if record:
            e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            e[0].record()
        reward, _, _, self.counts = self.net.rollout(self.action, self.sq, self.F, self.dt, self.um, check_faults=False)
        self.reward = reward.detach()
        if record:
            e[1].record()
        loss = -reward.sum()
        if record:
            e[2].record()
        loss.backward()
        if record:
            e[3].record()
            self.ev.append(e)
        g = self.drop_nonfinite(self.action.grad)
        return loss.detach(), g, g


def reward_in_reference_order(w):
    """Replica 0's reward of one more (untimed) pass with DHTS_OPT_REWARD_CHAIN on: ItscpEnv._reward's one float32 chain over lanes
    (outermost) and steps (example/control/itscp/_env.py:770-797) instead of the rollout kernels' own order."""
    from dhts import _lib as _L
    assert _L.lib().dhts_set_option(_L.OPT_REWARD_CHAIN, 1) == 0
    try:
        w.one_pass()
        return float(w.reward[0])
    finally:
        _L.lib().dhts_set_option(_L.OPT_REWARD_CHAIN, 0)


TOL_STATE, TOL_GRAD = 1e-5, 1e-4     # BASELINE.json north_star: state <= 1e-5 relative, gradients <= 1e-4 (norm-relative)


def _rel(a, ref):
    """max |a - ref| / max |ref| (the criterion of tests/util.py:rel_max)"""
    import numpy as np
    a, ref = np.asarray(a, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    return float(np.max(np.abs(a - ref)) / max(float(np.max(np.abs(ref))), 1e-30))


class StubWorkload:
    """Not a benchmark: a few float operations on whatever device there is, with the interface of the network workloads (a
    shared "schedule" gradient per rank).  `--workload stub` lets tests/test_dist_gloo.py drive the launcher, the rank
    environment, the per-pass all-reduce, the gather and rank 0's relay with eight CPU ranks over gloo;
    DHTS_STUB_FAIL_RANK=k makes rank k die behind the warm-up (the launcher must end the others and return non-zero)."""
    name = "stub"
    unit_bytes = 1
    unit_name = "units/s"
    limiter = {}

    def __init__(self, dev, rank, R, _n, _t):
        self.L, self.N, self.T, self.rank = R, 4, 1, rank
        self.units = R * 4
        self.action = (torch.arange(R * 3, dtype=torch.float32, device=dev).reshape(R, 3) + 100.0 * rank).requires_grad_(True)
        self.err = torch.zeros(4, dtype=torch.int32, device=dev)
        self.ev = []
        self.passes = 0

    def moved_bytes_per_launch(self):
        return self.units

    def one_pass(self, record=False):
        self.passes += 1
        if os.environ.get("DHTS_STUB_FAIL_RANK") == str(self.rank) and self.passes == 2:
            os._exit(7)
        self.action.grad = None
        loss = (self.action * self.action).sum()
        loss.backward()
        return loss.detach(), self.action.grad, self.action.grad

    def cpu_baseline(self, seconds=0.0):
        return None

    def parity_check(self, g_a, g_b):
        return None


def make_workload(name, dev, rank, lanes=0, cells=0, time_steps=0):
    if name == "stub":
        return StubWorkload(dev, rank, lanes or 2, 0, 0)
    if name == "macro":
        return MacroWorkload(dev, rank, lanes or 1024, cells or 512, time_steps or 1000)
    if name == "micro":
        return MicroWorkload(dev, rank, lanes or 4096, cells or 256, time_steps or 1000)
    if name == "itscp_hybrid":
        return ItscpHybridWorkload(dev, rank, lanes or 256, 0, 0)
    if name == "itscp_stepwise":
        return ItscpStepwiseWorkload(dev, rank, lanes or 256, 0, 0)
    return ItscpMacroWorkload(dev, rank, lanes or 256, 0, 0)


# Counter passes of THIS run (live_counters), keyed like profiles/issue_counters.json / pmc_traffic.json; None = not taken.
_LIVE = {"issue": None, "traffic": None, "seconds": None}
# FETCH_SIZE and WRITE_SIZE each alone (MI355X_MICROARCH.md, HBM section); the other two passes are the instruction-issue side
LIVE_GROUPS = (("FETCH_SIZE",), ("WRITE_SIZE",),
               ("GRBM_GUI_ACTIVE", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY",
                "SQ_WAIT_INST_ANY"),
               ("SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_VMEM_WR",
                "SQ_INSTS_VMEM_RD", "SQ_INSTS_VALU_CVT"))


def under_profiler():
    env = os.environ
    return any(k.startswith(("ROCPROF", "ROCP_")) for k in env) or "rocprof" in env.get("LD_PRELOAD", "")


def live_counters(key, deadline):
    """HBM bytes and the instruction-issue side of the headline workload's rollout kernels, MEASURED BY THIS RUN: four child
    processes `rocprofv3 --kernel-trace --pmc <group> -- python3 tools/run_workload.py <key> 3` (FETCH_SIZE alone, WRITE_SIZE alone,
    two SQ groups; never a trace domain beside --pmc), started before this process touches the GPU and summarised by
    tools/pmc_workloads_summary.py exactly like the committed passes of tools/pmc_workloads.sh.  Any failure (no rocprofv3, a
    time-out, a counter the box refuses) leaves _LIVE empty and the record falls back to the committed passes, labelled so."""
    import shutil
    import tempfile
    exe = os.environ.get("DHTS_ROCPROFV3") or shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        print("bench.py: no rocprofv3: counters not taken by this run", file=sys.stderr)
        return False
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import pmc_workloads_summary
        out = tempfile.mkdtemp(prefix="dhts_pmc_", dir="/tmp" if os.access("/tmp", os.W_OK) else None)
    except (OSError, ImportError) as e:
        print("bench.py: counter passes of this run not possible (%s): falling back to the committed ones" % e, file=sys.stderr)
        return False
    t0 = time.time()
    try:
        for i, group in enumerate(LIVE_GROUPS, 1):
            left = deadline - time.time()          # (one budget for all workloads of the run: a box that is slow at this is not waited for)
            if left < 15.0:
                raise subprocess.TimeoutExpired(exe, left)
            cmd = [exe, "--kernel-trace", "--pmc", *group, "--output-format", "csv", "-d", os.path.join(out, key, "g%d" % i), "--",
                   sys.executable, os.path.join(ROOT, "tools", "run_workload.py"), key, "3"]
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), timeout=left, stdout=subprocess.PIPE,
                               stderr=subprocess.STDOUT)
            with open(os.path.join(out, "%s.g%d.log" % (key, i)), "wb") as f:
                f.write(r.stdout)
            if r.returncode != 0:
                print("bench.py: counter pass %d (%s) ended with code %d: %s" % (i, " ".join(group), r.returncode,
                      r.stdout.decode(errors="replace")[-400:]), file=sys.stderr)
                return False
        issue, traffic = pmc_workloads_summary.summarise(out, quiet=True)
        if issue.get("library_code_sha16") != library_code_sha16():
            return False
        for slot, got in (("issue", issue), ("traffic", traffic)):
            _LIVE[slot] = dict(_LIVE[slot] or {}, **{k: v for k, v in got.items() if isinstance(v, dict)})
        _LIVE["seconds"] = round((_LIVE["seconds"] or 0.0) + time.time() - t0, 1)
        return True
    except Exception as e:          # noqa: BLE001 -- whatever goes wrong in a counter pass or its summary, the timed run must still happen
        print("bench.py: counter passes of this run failed (%s: %s): falling back to the committed ones" % (type(e).__name__, e), file=sys.stderr)
        return False
    finally:
        shutil.rmtree(out, ignore_errors=True)


def pmc_source(live):
    return ("measured by this run: rocprofv3 --pmc child passes (FETCH_SIZE and WRITE_SIZE each alone, two SQ groups) over 3 passes of this "
            "workload before the timed region; %s s for all workloads of this run" % _LIVE["seconds"]) if live else "committed passes under profiles/ (tools/pmc_workloads.sh)"


def pmc_traffic(w, kernel, moved=None):
    """HBM bytes per launch of `kernel` from the PMC passes committed under profiles/ (rocprofv3 --pmc FETCH_SIZE and
    --pmc WRITE_SIZE, gfx950 FETCH correction applied).  Quoted only for the configuration the passes were taken on and only
    while the tape the library allocates still has the size the passes saw (a changed layout needs fresh passes)."""
    try:
        pmc = _LIVE["traffic"] if _LIVE["traffic"] is not None and w.name in _LIVE["traffic"] else \
            json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        hbm = pmc[w.name][kernel]["hbm_bytes"]
    except (OSError, ValueError, KeyError):
        return None
    if moved is None:
        moved = w.moved_bytes_per_launch()
    if abs(hbm - moved) > 0.02 * moved:
        print("bench.py: profiles/pmc_traffic.json (%d B) disagrees with the library's tape size (%d B) by more than 2 %%: "
              "traffic not quoted; re-take the PMC passes" % (hbm, moved), file=sys.stderr)
        return None
    return hbm


def host_cores():
    """CPUs this process may actually use: the scheduler affinity capped by the container's CPU quota (cgroup cpu.max /
    cfs_quota), so that the CPU baseline starts as many threads as can run and divides by that number."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(p)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / p
        except (OSError, ValueError):
            pass
    if quota is not None and quota >= 1:
        n = min(n, int(quota))
    return max(1, n)


def library_code_sha16():
    """Fingerprint of the device code libdhts.so holds: sha256 over its .hip_fatbin section (the gfx950 code objects) -- a kernel
    whose body changed under the same name changes it.  Counter summaries under profiles/ carry the fingerprint of the library
    they were taken on and are not quoted for another."""
    import hashlib
    import struct
    from dhts import _lib
    try:
        with open(_lib.SO_PATH, "rb") as f:
            b = f.read()
        if b[:4] != b"\x7fELF" or b[4] != 2:
            return None
        shoff, = struct.unpack_from("<Q", b, 0x28)
        shentsize, shnum, shstrndx = struct.unpack_from("<HHH", b, 0x3A)
        sec = [struct.unpack_from("<IIQQQQIIQQ", b, shoff + i * shentsize) for i in range(shnum)]
        stroff = sec[shstrndx][4]
        for name, _t, _f, _a, off, size, *_ in sec:
            end = b.index(b"\0", stroff + name)
            if b[stroff + name:end] == b".hip_fatbin":
                return hashlib.sha256(b[off:off + size]).hexdigest()[:16]
    except (OSError, ValueError, struct.error, IndexError):
        pass
    return None


def issue_counters(w, kernel):
    """Instruction-issue side of `kernel` from the counter passes committed as profiles/issue_counters.json (tools/pmc_workloads.sh:
    rocprofv3 --pmc SQ_* passes over each workload at its BASELINE shape).  Not measured by this run; refused when the library's
    kernels are not the ones the passes saw (fingerprint of the device code, not of the kernel names)."""
    if _LIVE["issue"] is not None and w.name in _LIVE["issue"] and kernel in _LIVE["issue"][w.name]:
        return dict(_LIVE["issue"][w.name][kernel], source=pmc_source(True))
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "issue_counters.json")))
        side = rec[w.name][kernel]
    except (OSError, ValueError, KeyError):
        return None
    have = library_code_sha16()
    if have is None or rec.get("library_code_sha16") != have:
        print("bench.py: profiles/issue_counters.json was taken on another build of libdhts.so (device code %s, now %s): "
              "issue_side not quoted; re-run tools/pmc_workloads.sh" % (rec.get("library_code_sha16"), have), file=sys.stderr)
        return None
    return dict(side, source="%s (rocprofv3 --pmc passes of tools/pmc_workloads.sh on this configuration), not measured by this run"
                % rec.get("source", "profiles/issue_counters.json"))


def kernel_records(w):
    """Per-kernel records from the HIP events one_pass(record=True) left on the launch stream (the torch current stream
    is the stream the C ABI is handed, ops._stream)."""
    fwd_ms = [e[0].elapsed_time(e[1]) for e in w.ev]
    bwd_ms = [e[2].elapsed_time(e[3]) for e in w.ev]
    fwd_avg, bwd_avg = sum(fwd_ms) / len(fwd_ms), sum(bwd_ms) / len(bwd_ms)
    moved = w.moved_bytes_per_launch()
    algo = w.units * w.unit_bytes
    rec = {}
    for k, ms in (("rollout_fwd", fwd_avg), ("rollout_bwd", bwd_avg)):
        rec[k] = {"ms": ms, "moved_GB": moved / 1e9, "GBps": moved / ms / 1e6, "frac_of_peak": moved / ms / 1e6 / HBM_PEAK_GBS,
                  "algorithmic_GB": algo / 1e9, "algorithmic_GBps": algo / ms / 1e6}
    return rec, ("rollout_fwd" if fwd_avg >= bwd_avg else "rollout_bwd")


def check_parity(rec, name):
    """True (and a line on stderr) when a parity_check record is outside the contract's tolerances."""
    if rec is None:
        return False
    bad = not (rec["state_rel"] <= rec["tol_state"] and rec["grad_rel"] <= rec["tol_grad"])
    if "vehicles_spawned" in rec:
        bad |= rec["vehicles_spawned"][0] != rec["vehicles_spawned"][1]
    rec["ok"] = not bad
    if bad:
        print("bench.py: PARITY FAILURE on %s: %s" % (name, json.dumps(rec)), file=sys.stderr)
    return bad


def also_record(name, dev, passes=5, lanes=0):
    """A short run of another BASELINE configuration in the same process (not the headline; no collective)."""
    w = make_workload(name, dev, 0, lanes=lanes)
    for _ in range(2):
        w.one_pass()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(passes):
        _, g_a, g_b = w.one_pass(record=True)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    check_faults(w, "the bench")
    kernels, dom = kernel_records(w)
    for k in kernels:          # the counters behind each kernel's `limiter` (this run's passes, else the committed ones, fingerprinted)
        side = issue_counters(w, k)
        if side is not None:
            kernels[k]["issue_side"] = side
            if side.get("hbm_bytes_per_launch"):        # HBM bytes of the launch (FETCH_SIZE x 2 + WRITE_SIZE) over THIS run's event time
                kernels[k]["traffic_pmc_bytes"] = side["hbm_bytes_per_launch"]
                kernels[k]["traffic_GBps"] = side["hbm_bytes_per_launch"] / kernels[k]["ms"] / 1e6
                kernels[k]["traffic_frac_of_peak"] = kernels[k]["traffic_GBps"] / HBM_PEAK_GBS
    out = {"workload": w.name, "value": w.units * passes / el, "unit": w.unit_name, "passes": passes,
           "ms_per_pass": el / passes * 1e3, "dominant_kernel": dom, "limiter": w.limiter.get(dom, "hbm"), "kernels": kernels}
    if getattr(w, "counts", None) is not None:
        out["vehicles_spawned_replica0"] = int(w.counts[0, 0])
        out["replicas_dropped_nonfinite_gradient"] = w.dropped_replicas()
    return out, w, (g_a, g_b)


def check_faults(w, where):
    """The workload's sticky fault record behind a timed region: collisions are tolerated like the reference (it prints and
    carries on); a batch of network replicas also tolerates a member whose REVERSE sweep went non-finite (it is dropped from
    the shared gradient and counted) -- never a forward fault (CFL, capacity)."""
    fault = w.err.tolist()
    ok = (0, 2, 3) if hasattr(w, "dropped_replicas") else (0, 2)
    assert fault[0] in ok, "simulation fault during %s: %s" % (where, fault)
    return fault


def allreduce_check(flat, parts, shared_grad):
    """rank 0's proof that the per-pass collective summed every rank's own [gradient || loss] of the last pass"""
    if parts is None:
        return None
    return {"reduced": flat.tolist()[-1], "sum_of_rank_parts": float(parts[:, -1].double().sum()), "rank_parts": parts[:, -1].tolist(),
            "buffer_floats": int(flat.numel()),
            "grad_max_abs_diff": float((parts[:, :-1].double().sum(dim=0) - flat[:-1].double().cpu()).abs().max()) if shared_grad else 0.0}


def collective_run(w, steps, warmup, shared_grad, dev, record):
    """W untimed passes, then exactly `steps` passes bracketed by barrier + synchronize on both sides, each followed by the ONE
    collective of the path: the all-reduce of [d loss / d theta_shared || loss] (straight lanes own their unknowns: [loss] alone).
    Returns (max-over-ranks seconds, the reduced buffer, every rank's own part on rank 0, the last pass's gradients)."""
    from dhts import dist as D
    sync = torch.cuda.synchronize if dev.type == "cuda" else (lambda: None)
    world = D.env_rank_world()[1]
    flat = torch.zeros((w.action.shape[1] if shared_grad else 0) + 1, dtype=torch.float32, device=dev)

    def reduce_pass(loss, g_a):
        if shared_grad:
            flat[:-1] = g_a.sum(dim=0)
        flat[-1] = loss
        local_part = flat.clone() if world > 1 else None
        D.allreduce_sum_(flat)
        return local_part
    for _ in range(warmup):
        loss, g_a, _ = w.one_pass()
        reduce_pass(loss, g_a)
    D.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss, g_a, g_b = w.one_pass(record=record)
        local_part = reduce_pass(loss, g_a)
    D.barrier()
    sync()
    elapsed = D.max_over_ranks(time.perf_counter() - t0, dev)
    check_faults(w, "the bench")
    assert torch.isfinite(g_a).all() and torch.isfinite(g_b).all() and bool(torch.isfinite(flat).all())
    # every rank's own [gradient || loss] of the last pass, gathered so that rank 0 can show the all-reduce summed them
    parts = D.gather_to_rank0(local_part) if world > 1 else None
    return elapsed, flat, parts, (g_a, g_b)


def replica_sweep(dev, sizes=(256, 512, 1024, 2048), passes=3):
    """Config 4's network at several replicas per GPU, ONE instance (2 048 replicas built once, prefixes timed): the saturated
    throughput of the hybrid path on one GPU and the one-GPU time of config 5's whole 2 048-replica problem (the strong-scaling
    denominator).  256 replicas = one workgroup per compute unit."""
    w = make_workload("itscp_hybrid", dev, 0, lanes=max(sizes))
    recs = []
    for R in sizes:
        w.restrict(R)
        for _ in range(2):
            w.one_pass()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(passes):
            w.one_pass(record=True)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        check_faults(w, "the replica sweep")
        kernels, _ = kernel_records(w)
        plan = w.ops.net_hybrid_plan(R, w.action.shape[1], w.tab, w.sq)
        recs.append({"replicas": R, "replicas_per_cu": R / float(plan["cus"]), "two_workgroups_per_cu": plan["packed"],
                     "lds_bytes_fwd_bwd": [plan["lds_fwd"], plan["lds_bwd"]], "ms_per_pass": el / passes * 1e3, "value": w.units * passes / el,
                     "unit": w.unit_name, "fwd_ms": kernels["rollout_fwd"]["ms"], "bwd_ms": kernels["rollout_bwd"]["ms"],
                     "fwd_GBps": kernels["rollout_fwd"]["GBps"], "bwd_GBps": kernels["rollout_bwd"]["GBps"],
                     "replicas_dropped_nonfinite_gradient": w.dropped_replicas()})
    base = recs[0]["value"]
    for r in recs:
        r["vs_256_replicas"] = r["value"] / base
    return {"workload": "itscp_hybrid replica sweep (%d lanes, %d cells x %d steps per replica)" % (w.n_lanes, w.N, w.T),
            "passes": passes, "points": recs,
            "note": "one GPU; the last point is BASELINE config 5's whole problem (2 048 replicas) on ONE device.  Beyond one replica "
                    "per compute unit the kernels are launched two to a unit (DHTS_OPT_HYB_PACK, include/dhts.h: half the LDS and 128 "
                    "registers per workgroup, same results bit for bit)"}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))        # before any GPU call: the children own the devices
    if (args.gpus == 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1 and args.workload != "stub" and not args.no_live_counters
            and not (args.lanes or args.cells or args.time_steps) and not under_profiler() and torch.cuda.device_count() > 0):
        # child processes; this one has not touched the GPU yet.  The default run takes them for its sub-records' workloads too
        deadline = time.time() + 180.0
        for key in ((args.workload,) if args.no_also or args.workload != "macro" else ("macro", "micro", "itscp_hybrid", "itscp_stepwise")):
            if not live_counters(key, deadline):
                break
    from dhts import dist as D
    rank, world, local = D.init()                  # (makes this rank's GPU current, then builds the process group on it)
    if world != args.gpus and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
    stub = args.workload == "stub"
    assert stub or torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    dev = D.local_device(local)                    # one GPU per rank on a full node; wraps only in single-GPU smoke tests
    if dev is None:
        dev = torch.device("cpu")                  # (the stub only)
    else:
        assert torch.cuda.current_device() == dev.index
    sync = torch.cuda.synchronize if dev.type == "cuda" else (lambda: None)
    # who took part: backend + every rank's device (gathered; raises when RCCL ranks share a device)
    collective = D.collective_record(dev) if (world > 1 or D._forced()) else None      # (DHTS_DIST_FORCE=1: the one-rank RCCL self-test)

    w = make_workload(args.workload, dev, rank, args.lanes, args.cells, args.time_steps)
    L, N, T = w.L, w.N, w.T
    default_shape = not args.lanes and not args.cells and not args.time_steps

    # the per-pass RCCL all-reduce: [loss] for the straight-lane workloads (every lane owns its unknowns); for the network
    # workloads the gradient summed over the rank's replicas as if the signal schedule were shared (BASELINE config 5) + loss
    shared_grad = args.workload.startswith("itscp") or stub
    # The pair kernel's priority rotation (DHTS_OPT_MACRO_FWD_ROTATE, include/dhts.h) encodes an observation about this pool's
    # dispatcher; correctness does not depend on it, speed may: untimed passes with and without it, interleaved, before the warm-up,
    # and the run takes what this box prefers (reported as roofline.fwd_rotate: best pass time of either setting)
    rotate_rec = None
    if args.workload == "macro" and dev.type == "cuda":
        from dhts import _lib as _L
        ms = {1: [], 0: []}
        for _ in range(3):                       # (the first passes of a process run slower -- clock ramp: warm up first, then interleave)
            w.one_pass()
        for _ in range(2):                       # (few passes: they are launches of the same kernel and enter a profiler's average)
            for setting in (1, 0):
                _L.lib().dhts_set_option(_L.OPT_MACRO_FWD_ROTATE, setting)
                w.one_pass()
                sync()
                tq = time.perf_counter()
                w.one_pass()
                sync()
                ms[setting].append((time.perf_counter() - tq) * 1e3)
        ms = {k: min(v) for k, v in ms.items()}
        chosen = 1 if ms[1] <= ms[0] else 0
        _L.lib().dhts_set_option(_L.OPT_MACRO_FWD_ROTATE, chosen)
        rotate_rec = {"ms_per_pass_with": ms[1], "ms_per_pass_without": ms[0], "chosen": chosen}
    elapsed, flat, parts, (g_a, g_b) = collective_run(w, args.steps, args.warmup, shared_grad, dev, record=not stub)

    # N > 1, the driver's default command: BASELINE config 5 behind the headline -- the itscp hybrid network, 256 replicas per rank
    # (2 048 over 8 GPUs), d reward / d (shared signal schedule) [45] || reward all-reduced once per pass.  Every rank runs it.
    second = None
    if world > 1 and not args.no_also and ((args.workload == "macro" and (default_shape or args.also_replicas)) or stub):
        if not stub:
            kernels_head = kernel_records(w)           # (the events and the census need the tape: read them before it goes)
            census_head = w.tape_census() if rank == 0 else None
            moved_head = w.moved_bytes_per_launch() if rank == 0 else None
            del w.tape
            torch.cuda.empty_cache()
        w5 = make_workload("stub" if stub else "itscp_hybrid", dev, rank, 3 if stub else (args.also_replicas or 256))
        passes5 = min(args.steps, 5)
        el5, flat5, parts5, _g5 = collective_run(w5, passes5, 2, True, dev, record=not stub)
        if rank == 0:
            second = {"workload": w5.name, "config": "BASELINE config 5: %d replicas per rank x %d ranks = %d replicas" % (w5.L, world, w5.L * world),
                      "value": w5.units * passes5 * world / el5, "unit": w5.unit_name, "n_gpus": world, "passes": passes5,
                      "ms_per_pass": el5 / passes5 * 1e3, "scaling": "weak",
                      "allreduce": "[d reward / d action (%d) || reward] summed over ranks once per pass" % (flat5.numel() - 1),
                      "allreduce_check": allreduce_check(flat5, parts5, True), "loss_last_pass": flat5.tolist()[-1]}
            if not stub:
                k5, dom5 = kernel_records(w5)
                second.update(dominant_kernel=dom5, kernels=k5, vehicles_spawned_replica0=int(w5.counts[0, 0]),
                              replicas_dropped_nonfinite_gradient_rank0=w5.dropped_replicas())
        del w5
    else:
        kernels_head = census_head = moved_head = None

    parity_failed = False
    if rank == 0 and stub:
        out = {"metric": "stub", "value": w.units * args.steps * world / elapsed, "unit": w.unit_name, "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
               "config": {"workload": w.name, "device": dev.type}, "loss_last_pass": flat.tolist()[-1],
               "allreduce_check": allreduce_check(flat, parts, True)}
        if collective is not None:
            out["collective"] = collective
        if second is not None:
            out["also"] = [second]
        print(json.dumps(out))
    elif rank == 0:
        kernels, dom = kernels_head if kernels_head is not None else kernel_records(w)
        k = kernels[dom]
        moved = moved_head if moved_head is not None else w.moved_bytes_per_launch()
        value = w.units * args.steps * world / elapsed
        out = {
            "metric": "differentiable cell-steps/s (fwd+bwd)",
            "value": value,
            "unit": "cell-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "dtype_detail": "float32 state widened to double for the step, Jacobian tape and adjoint in float32 (the reference's ladder)",
            "data": "synthetic",
            "config": {"workload": w.name, "lanes_per_gpu": L, "units_per_lane": N, "time_steps": T,
                       "parallelism": "%s sharded over %d GPU(s), no data-path collective; one all-reduce of [%s] per pass"
                                      % ("replicas" if shared_grad else "lanes", world,
                                         "d loss / d action || loss" if shared_grad else "loss")},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": k["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": k["frac_of_peak"], "traffic": pmc_traffic(w, dom, moved), "traffic_unit": "bytes per launch (PMC)",
                         "traffic_source": pmc_source(_LIVE["traffic"] is not None and w.name in _LIVE["traffic"]),
                         "moved_bytes_per_launch": moved,
                         "algorithmic_bytes_per_launch": w.units * w.unit_bytes,
                         "achieved_algorithmic": k["algorithmic_GBps"], "frac_algorithmic": k["algorithmic_GBps"] / HBM_PEAK_GBS,
                         "limiter": w.limiter.get(dom, "hbm"),
                         "note": "achieved = bytes the kernel moves (its compact tape: the same information as the reference's dqs "
                                 "blocks, which the reverse sweep rebuilds; macro: counted from the tape's own row headers after "
                                 "the run, in whole 128-byte lines) / HIP-event time of the launch; *_algorithmic = the "
                                 "reference's tape bytes (48 B per cell-step, 32 B per vehicle-step) / the same time"},
            "whole_path": {"moved_GBps": 2 * moved * args.steps / elapsed / 1e9,
                           "frac_of_peak": 2 * moved * args.steps / elapsed / 1e9 / HBM_PEAK_GBS,
                           "algorithmic_GBps": w.units * 2 * w.unit_bytes * args.steps / elapsed / 1e9},
            "kernels": kernels,
            "loss_last_pass": flat.tolist()[-1],          # summed over ranks by the all-reduce
        }
        if hasattr(w, "tape_census"):
            out["roofline"]["tape_census"] = census_head if census_head is not None else w.tape_census()
        if rotate_rec is not None:
            out["roofline"]["fwd_rotate"] = rotate_rec
        side = issue_counters(w, dom)
        if side is not None:
            out["roofline"]["issue_side"] = side
        if parts is not None:
            out["allreduce_check"] = allreduce_check(flat, parts, shared_grad)
        if collective is not None:
            out["collective"] = collective
        if second is not None:
            out["also"] = [second]
        if world == 1:
            if args.workload == "macro" and default_shape and not args.no_also:
                del w.tape            # 24 GB back to the allocator before the other workloads take theirs
                torch.cuda.empty_cache()
                out["also"] = []
                for name in ("micro", "itscp_hybrid", "itscp_stepwise"):
                    rec, w2, g2 = also_record(name, dev)
                    if not args.no_cpu_baseline:
                        # the oracle on a sample of this sub-record's own inputs (its rate is reported, the headline's is cpu_baseline)
                        rec["cpu_baseline"] = w2.cpu_baseline(seconds=3.0)
                        rec["parity_check"] = w2.parity_check(*g2)
                        parity_failed |= check_parity(rec["parity_check"], rec["workload"])
                    del w2, g2
                    torch.cuda.empty_cache()
                    out["also"].append(rec)
                # the one-GPU side of the scaling question: config 4's network at 256 ... 2 048 replicas on this device,
                # and config 2 at 8 x its lanes (what 8 GPUs run together), each as ms per pass and cell-steps/s
                # (two big extras -- 25 GB and 195 GB of tape: a device short of memory loses these records, never the line)
                try:
                    out["also"].append(replica_sweep(dev))
                except torch.OutOfMemoryError as e:
                    out["also"].append({"workload": "itscp_hybrid replica sweep", "skipped": "out of device memory: %s" % str(e)[:120]})
                torch.cuda.empty_cache()
                try:
                    rec, w2, _g = also_record("macro", dev, passes=3, lanes=8 * L)
                    rec["note"] = "config 2 x 8 lanes on ONE GPU (tape %.1f GB): the one-GPU time of the 8-GPU run's whole problem" % (w2.tape_bytes / 1e9)
                    del w2, _g
                except torch.OutOfMemoryError as e:
                    rec = {"workload": "macro_straight_%dx%dx%d" % (8 * L, N, T), "skipped": "out of device memory: %s" % str(e)[:120]}
                torch.cuda.empty_cache()
                out["also"].append(rec)
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = w.cpu_baseline()
                # what this run timed, checked on this box: the last timed pass against the oracle's run of the same inputs
                out["parity_check"] = w.parity_check(g_a, g_b)
                parity_failed |= check_parity(out["parity_check"], w.name)
        print(json.dumps(out))
    D.barrier()
    D.shutdown()                 # (the process group goes down in order on every rank: no teardown warnings, no straggler)
    if parity_failed:
        sys.exit(3)


if __name__ == "__main__":
    main()
