#!/usr/bin/env python3
"""Time the macro forward kernel of BASELINE config 2 for each wavefronts-per-lane setting (GPU box)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diff-hybrid-traffic-sim_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from dhts import _lib  # noqa: E402

dev = torch.device("cuda:0")
L = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
w = bench.MacroWorkload(dev, 0, L, N, 1000)
sweep = [(1, 0)] + [(0, k) for k in (0, 1, 2, 4, 8, 16) if 64 * k <= N or k == 0]       # (kernel variant, wavefronts per lane)
for variant, waves in sweep:
    _lib.lib().dhts_set_option(_lib.OPT_MACRO_FWD_VARIANT, variant)
    _lib.lib().dhts_set_option(_lib.OPT_MACRO_FWD_WAVES, waves)
    w.ev = []
    for _ in range(2):
        w.one_pass()
    for _ in range(5):
        w.one_pass(record=True)
    torch.cuda.synchronize()
    fwd = sorted(e[0].elapsed_time(e[1]) for e in w.ev)
    bwd = sorted(e[2].elapsed_time(e[3]) for e in w.ev)
    print("variant %d waves/lane %d: fwd median %.3f ms (min %.3f)  bwd median %.3f ms" % (variant, waves, fwd[2], fwd[0], bwd[2]), flush=True)
