#!/usr/bin/env python3
"""Times the macro forward kernel of BASELINE config 2 with and without DHTS_OPT_MACRO_FWD_GROUP (two traffic lanes per
workgroup, one phase-2 list) and compares final state, tape-driven gradient bit for bit.  GPU box: python3 tools/probes/exp_fwd_groups.py"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "diff-hybrid-traffic-sim_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
from dhts import _lib  # noqa: E402

dev = torch.device("cuda:0")
for pair in (1, 2, 4, 1, 2, 4):
    assert _lib.lib().dhts_set_option(_lib.OPT_MACRO_FWD_GROUP, pair) == 0
    w = bench.MacroWorkload(dev, 0, 1024, 512, 1000)
    for _ in range(4):
        w.one_pass()
    for _ in range(12):
        loss, g_r0, g_u0 = w.one_pass(record=True)
    torch.cuda.synchronize()
    fwd = sorted(e[0].elapsed_time(e[1]) for e in w.ev)
    bwd = sorted(e[2].elapsed_time(e[3]) for e in w.ev)
    h = hashlib.sha256()
    for t in (w.out[0], w.out[2], g_r0, g_u0):
        h.update(t.cpu().numpy().tobytes())
    print(json.dumps({"lanes_per_wg": pair, "fwd_min": fwd[0], "fwd_med": fwd[len(fwd) // 2], "bwd_med": bwd[len(bwd) // 2],
                      "sha": h.hexdigest()[:16], "fault": w.err.tolist()[0]}), flush=True)
