#!/usr/bin/env python3
"""Phase table of the hybrid network kernels (config 4): cycles per step each role spends working / waiting in front of every
barrier.  Needs the instrumented build of the library:

    cd diff-hybrid-traffic-sim_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DDHTS_HYB_STAMPS \
        -c -o /tmp/hyb_st.o hybrid_kernels.hip && hipcc --offload-arch=gfx950 -shared -fPIC -o libdhts_stamps.so \
        dhts_common.o macro_kernels.o micro_kernels.o network_kernels.o /tmp/hyb_st.o
    python tools/probes/exp_hyb_stamps.py
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "diff-hybrid-traffic-sim_amd"))
sys.path.insert(0, ROOT)
from dhts import _lib  # noqa: E402
if not os.environ.get("DHTS_LIB"):
    _lib.SO_PATH = os.path.join(ROOT, "diff-hybrid-traffic-sim_amd", "csrc", "variants", "libdhts_hstamps.so")
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device("cuda:0")
w = bench.ItscpHybridWorkload(dev, 0, 256, 0, 0)
for _ in range(3):
    w.one_pass()
w.ev = []
for _ in range(5):
    w.one_pass(record=True)
torch.cuda.synchronize()
print("instrumented: fwd %.3f ms bwd %.3f ms" % (np.median([e[0].elapsed_time(e[1]) for e in w.ev]), np.median([e[2].elapsed_time(e[3]) for e in w.ev])))
buf = (C.c_longlong * (2 * 8 * 16 * 24))()
assert _lib.lib().dhts_debug_stamps(buf) == 0
a = np.array(buf[:], dtype=np.int64).reshape(2, 8, 16, 24)
T = 600
for kern, name, nb in ((0, "forward", 4), (1, "reverse", 5)):
    print(name, "(cycles per step, mean over replicas 0..7)")
    m = a[kern].mean(axis=0) / T
    for wave in range(8):
        nm = "wave %d%s" % (wave, " (flush)" if wave == 6 else (" (micro)" if wave == 7 else ""))
        print("  %-14s work %s   drain %s   barrier %s   total %d" % (nm, np.round(m[wave, :nb]).astype(int), np.round(m[wave, 8:8 + nb]).astype(int),
                                                                       np.round(m[wave, 16:16 + nb]).astype(int), int(m[wave].sum())))
        if kern == 0 and m[wave, 12:16].any():    # HYB_SUB2: cell waves, phase D = loss constants + history row | lane sums | (rest: flush)
            print("  %-14s  sub2 %s" % ("", np.round(m[wave, 12:16]).astype(int)))
        if kern == 0 and m[wave, 4:8].any():      # sub-phase stamps (HYB_SUB): cell waves = table fetch, ghosts, loss scan (phase A), interface
            print("  %-14s   sub %s" % ("", np.round(m[wave, 4:8]).astype(int)))     # solves (B); micro wave = capacitors, pre-screen, commits (D)
