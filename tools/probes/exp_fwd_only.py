#!/usr/bin/env python3
"""Forward rollout of BASELINE config 2 from an experiment build of the library: exp_fwd_only.py <path to .so>"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "diff-hybrid-traffic-sim_amd"))
sys.path.insert(0, ROOT)
from dhts import _lib  # noqa: E402
if len(sys.argv) > 1:
    _lib.SO_PATH = os.path.abspath(sys.argv[1])
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device("cuda:0")
w = bench.MacroWorkload(dev, 0, 1024, 512, 1000)
ops = w.ops
y0, q0 = ops.macro_state_from_ru(w.r0, w.u0, w.um)
n = 8
fn = lambda: ops.macro_rollout_fwd(w.desc, w.T, w.r0, y0, w.u0, q0, w.ghost, tape=w.tape, err=w.err, out=w.out)
for _ in range(2):
    fn()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
ev[0].record()
for i in range(n):
    fn()
    ev[i + 1].record()
torch.cuda.synchronize()
ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))
print("%s: fwd min / median ms: %.3f %.3f" % (os.path.basename(_lib.SO_PATH), ts[0], ts[n // 2]))
