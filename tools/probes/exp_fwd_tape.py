#!/usr/bin/env python3
"""Forward rollout of BASELINE config 2 with and without the tape (what do the tape stores cost?), and the reverse sweep alone."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "diff-hybrid-traffic-sim_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device("cuda:0")
w = bench.MacroWorkload(dev, 0, 1024, 512, 1000)
ops = w.ops
y0, q0 = ops.macro_state_from_ru(w.r0, w.u0, w.um)


def timed(fn, n=8):
    for _ in range(2):
        fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))
    return ts[0], ts[len(ts) // 2]


print("fwd with tape    min / median ms: %.3f %.3f" % timed(lambda: ops.macro_rollout_fwd(w.desc, w.T, w.r0, y0, w.u0, q0, w.ghost, tape=w.tape, err=w.err, out=w.out)))
print("fwd without tape min / median ms: %.3f %.3f" % timed(lambda: ops.macro_rollout_fwd(w.desc, w.T, w.r0, y0, w.u0, q0, w.ghost, tape=None, err=w.err, out=w.out)))
g_r, g_y = 2.0 * w.out[0], torch.zeros_like(w.out[0])
print("bwd              min / median ms: %.3f %.3f" % timed(lambda: ops.macro_rollout_bwd(w.desc, w.T, w.tape, g_r, g_y, err=w.err, out=w.gout, g_ghost=w.g_ghost)))
print(w.tape_census())

# the two kernels alternating, as in a training loop (bench.py's one_pass without the glue kernels in between)
ev = []
for i in range(10):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    ops.macro_rollout_fwd(w.desc, w.T, w.r0, y0, w.u0, q0, w.ghost, tape=w.tape, err=w.err, out=w.out)
    e[1].record()
    ops.macro_rollout_bwd(w.desc, w.T, w.tape, g_r, g_y, err=w.err, out=w.gout, g_ghost=w.g_ghost)
    e[2].record()
    ev.append(e)
torch.cuda.synchronize()
print("alternating: fwd ms", ["%.3f" % e[0].elapsed_time(e[1]) for e in ev[2:]])
print("alternating: bwd ms", ["%.3f" % e[1].elapsed_time(e[2]) for e in ev[2:]])
w.ev = []
for _ in range(6):
    w.one_pass(record=True)
torch.cuda.synchronize()
print("one_pass: fwd ms", ["%.3f" % e[0].elapsed_time(e[1]) for e in w.ev[1:]])
print("one_pass: bwd ms", ["%.3f" % e[2].elapsed_time(e[3]) for e in w.ev[1:]])

# reverse sweep with per-step cotangents (a loss on the state history)
gh = torch.zeros(w.T, w.L, 2, w.N, device=dev)
print("bwd with g_hist  min / median ms: %.3f %.3f" % timed(lambda: ops.macro_rollout_bwd(w.desc, w.T, w.tape, g_r, g_y, g_hist=gh, err=w.err, out=w.gout, g_ghost=w.g_ghost)))
