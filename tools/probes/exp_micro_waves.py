#!/usr/bin/env python3
"""Micro forward kernel of BASELINE config 3 for each wavefronts-per-lane setting."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "diff-hybrid-traffic-sim_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from dhts import _lib  # noqa: E402

dev = torch.device("cuda:0")
L = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
V = int(sys.argv[2]) if len(sys.argv) > 2 else 256
w = bench.make_workload("micro", dev, 0) if (L, V) == (4096, 256) else bench.MicroWorkload(dev, 0, L, V, 1000)
for waves in (0, 1, 2, 4):
    _lib.lib().dhts_set_option(_lib.OPT_MICRO_FWD_WAVES, waves)
    w.ev = []
    for _ in range(3):
        w.one_pass()
    for _ in range(7):
        w.one_pass(record=True)
    torch.cuda.synchronize()
    fwd = sorted(e[0].elapsed_time(e[1]) for e in w.ev)
    bwd = sorted(e[2].elapsed_time(e[3]) for e in w.ev)
    print("waves/lane %d: fwd median %.3f ms (min %.3f)  bwd median %.3f ms" % (waves, fwd[3], fwd[0], bwd[3]), flush=True)
_lib.lib().dhts_set_option(_lib.OPT_MICRO_FWD_WAVES, 0)
