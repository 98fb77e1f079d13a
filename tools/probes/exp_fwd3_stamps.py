#!/usr/bin/env python3
"""Phase table of the macro pair forward kernel (config 2): cycles per step every wavefront of a workgroup spends in phase 1,
at barrier 1, in phase 2 and at barrier 2.  Needs the instrumented build:  tools/build_variants.sh stamps:"-DDHTS_FWD3_STAMPS"
    python3 tools/probes/exp_fwd3_stamps.py [variant] [group]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "diff-hybrid-traffic-sim_amd"))
sys.path.insert(0, ROOT)
from dhts import _lib  # noqa: E402
_lib.SO_PATH = os.path.join(ROOT, "diff-hybrid-traffic-sim_amd", "csrc", "variants", "libdhts_stamps.so")
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402

variant = int(sys.argv[1]) if len(sys.argv) > 1 else 0      # 0 = pair kernel, 2 = lane-group kernel (clock only)
group = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda:0")
lanes = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
w = bench.MacroWorkload(dev, 0, lanes, 512, 1000)
assert _lib.lib().dhts_set_option(_lib.OPT_MACRO_FWD_VARIANT, variant) == 0
assert _lib.lib().dhts_set_option(_lib.OPT_MACRO_FWD_GROUP, group) == 0
for _ in range(100):          # ~0.5 s of back-to-back launches before anything is read (clock settles)
    w.one_pass()
w.ev = []
for _ in range(5):
    w.one_pass(record=True)
torch.cuda.synchronize()
print("variant %d group %d lanes %d, instrumented: fwd %.3f ms" % (variant, group, lanes, np.median([e[0].elapsed_time(e[1]) for e in w.ev])))
buf = (C.c_longlong * (16 * 16 * 8))()
assert _lib.lib().dhts_debug_fwd3_stamps(buf) == 0
if variant != 0:
    buf = (C.c_longlong * (16 * 16 * 8))()
a = np.array(buf[:], dtype=np.int64).reshape(16, 16, 8).astype(np.float64)
nw = 16 if group == 4 else 8
T = w.T
for name, sl in (("workgroups 0..7", slice(0, 8)), ("workgroups G/2..G/2+7", slice(8, 16))):
    print("cycles per step (s_memtime), mean over %s, per wavefront" % name)
    print("           phase 1   barrier 1 | steps with queue entries: share, entries, phase 2, barrier 2 | other steps: phase 2, barrier 2 | per step total")
    m = a[sl].mean(axis=0)
    for wave in range(nw):
        r = m[wave]
        held = max(r[4], 1e-9)
        idle = max(T - r[4], 1e-9)
        print("  wave %2d: %7.0f %9.0f | %5.2f %6.1f %8.0f %8.0f | %8.0f %8.0f | %7.0f" % (
            wave, r[0] / T, r[1] / T, r[4] / T, r[5] / held, r[2] / held, r[3] / held, r[6] / idle, r[7] / idle, (r[0] + r[1] + r[2] + r[3] + r[6] + r[7]) / T))
cb = (C.c_longlong * (2 * 16 * 2))()
assert _lib.lib().dhts_debug_fwd_clock(cb) == 0
c = np.array(cb[:], dtype=np.float64).reshape(2, 16, 2)
k = 1 if variant == 0 else 0
print("in-kernel clock (s_memtime / s_memrealtime x 100 MHz): median %.0f MHz (min %.0f, max %.0f); step loop of workgroups 0..7: %.3f ms, of workgroups G/2..G/2+7: %.3f ms" % (
    np.median(c[k, :, 0] / c[k, :, 1] * 100), (c[k, :, 0] / c[k, :, 1] * 100).min(), (c[k, :, 0] / c[k, :, 1] * 100).max(), np.median(c[k, :8, 1]) / 1e5, np.median(c[k, 8:, 1]) / 1e5))
