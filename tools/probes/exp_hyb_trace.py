#!/usr/bin/env python3
"""Per-step barrier trace of the hybrid network kernels (config 4, replica 0): which wavefront arrives last at every barrier of
every step, how long each phase takes in steps with / without hand-off events.  Needs the instrumented build
(tools/build_variants.sh with SRC=hybrid_kernels hstamps:"-DDHTS_HYB_STAMPS").
    DHTS_LIB=diff-hybrid-traffic-sim_amd/csrc/variants/libdhts_hstamps.so python3 tools/probes/exp_hyb_trace.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "diff-hybrid-traffic-sim_amd"))
sys.path.insert(0, ROOT)
from dhts import _lib  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device("cuda:0")
w = bench.ItscpHybridWorkload(dev, 0, 256, 0, 0)
for _ in range(3):
    w.one_pass()
w.ev = []
for _ in range(3):
    w.one_pass(record=True)
torch.cuda.synchronize()
print("library:", _lib.SO_PATH)
print("instrumented: fwd %.3f ms bwd %.3f ms" % (np.median([e[0].elapsed_time(e[1]) for e in w.ev]), np.median([e[2].elapsed_time(e[3]) for e in w.ev])))
buf = (C.c_int * (2 * 640 * 16 * 10))()
assert _lib.lib().dhts_debug_trace(buf) == 0
a = np.array(buf[:], dtype=np.int64).reshape(2, 640, 16, 5, 2)
T, NW = 600, 8
names = ["w0", "w1", "w2", "w3", "w4", "w5", "flush", "micro"]
for kern, name, nb in ((0, "forward", 4), (1, "reverse", 5)):
    tr = a[kern, :T, :NW, :nb]                       # [T][wave][barrier][arrival / release]
    arr, rel = tr[..., 0], tr[..., 1]
    order = range(T) if kern == 0 else range(T - 1, -1, -1)
    steps = list(order)
    rel_all = rel.max(axis=1)                        # [T][nb] release (about the same on every wave)
    last_arr = arr.max(axis=1)                       # [T][nb]
    who = arr.argmax(axis=1)                         # [T][nb]
    # phase i of step t starts at the release of the barrier before
    start = np.zeros((T, nb))
    for j, t in enumerate(steps):
        for i in range(nb):
            if i > 0:
                start[t, i] = rel_all[t, i - 1]
            elif j > 0:
                start[t, i] = rel_all[steps[j - 1], nb - 1]
            else:
                start[t, i] = arr[t, :, 0].min()
    dur = rel_all - start                            # phase length by the barrier's release
    crit = last_arr - start                          # the last wave's work
    step_len = dur.sum(axis=1)
    print("%s: step %.0f cycles mean (median %.0f, p90 %.0f, max %.0f)" % (name, step_len[1:-1].mean(), np.median(step_len), np.percentile(step_len, 90), step_len.max()))
    ev = step_len > 1.5 * np.median(step_len)
    print("  steps longer than 1.5 medians: %d, they hold %.1f %% of the time" % (ev.sum(), 100. * step_len[ev].sum() / step_len.sum()))
    for sel, tag in ((~ev, "ordinary steps"), (ev, "long steps")):
        if sel.sum() == 0:
            continue
        print("  %s (%d): phase length %s   last arrival %s   release - last arrival %s" % (
            tag, sel.sum(), np.round(dur[sel].mean(axis=0)).astype(int), np.round(crit[sel].mean(axis=0)).astype(int),
            np.round((dur - crit)[sel].mean(axis=0)).astype(int)))
        for i in range(nb):
            h = np.bincount(who[sel, i], minlength=NW)
            print("    barrier %d last arriver: %s" % (i, "  ".join("%s %d" % (names[k], h[k]) for k in range(NW) if h[k])))
        # mean work per wave and phase (arrival - start)
        wk = arr - start[:, None, :]
        for k in range(NW):
            print("    %-6s work %s  (std %s)" % (names[k], np.round(wk[sel, k].mean(axis=0)).astype(int), np.round(wk[sel, k].std(axis=0)).astype(int)))
    top = np.argsort(-step_len)[:8]
    print("  longest steps:", [(int(t), int(step_len[t]), [int(x) for x in dur[t]]) for t in top])
