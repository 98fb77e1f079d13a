#!/usr/bin/env python3
"""Times the macro rollout forward kernel of BASELINE config 2 for the forward variants of ONE build of libdhts.so
(DHTS_OPT_MACRO_FWD_VARIANT: 0 = pair kernel, 2 = the lane kernel) and lanes per workgroup, and checks every
setting against the first one bit for bit (final state, and the gradient the reverse sweep makes of its tape).
GPU box:  [DHTS_LIB=<variant build>] [DHTS_EXP_LANES=<lanes, default 1024>] python3 tools/exp_fwd_pairs.py [variant:group ...]      (default: 2:0 0:0 0:4 0:1 2:0 0:0)"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "diff-hybrid-traffic-sim_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
from dhts import _lib  # noqa: E402


def main():
    specs = sys.argv[1:] or ["2:0", "0:0", "0:4", "0:1", "2:0", "0:0"]
    print("library:", _lib.SO_PATH, flush=True)
    dev = torch.device("cuda:0")
    w = bench.MacroWorkload(dev, 0, int(os.environ.get("DHTS_EXP_LANES", "1024")), 512, 1000)
    ref = None
    for spec in specs:
        v, g = (int(x) for x in spec.split(":"))
        assert _lib.lib().dhts_set_option(_lib.OPT_MACRO_FWD_VARIANT, v) == 0
        assert _lib.lib().dhts_set_option(_lib.OPT_MACRO_FWD_GROUP, g) == 0
        w.ev = []
        for _ in range(3):
            w.one_pass()
        for _ in range(10):
            loss, g_r0, g_u0 = w.one_pass(record=True)
        torch.cuda.synchronize()
        fwd = sorted(e[0].elapsed_time(e[1]) for e in w.ev)
        bwd = sorted(e[2].elapsed_time(e[3]) for e in w.ev)
        h = hashlib.sha256()
        for t in (w.out[0], w.out[2], g_r0, g_u0):
            h.update(t.cpu().numpy().tobytes())
        sha = h.hexdigest()[:16]
        ref = ref or sha
        census = w.tape_census() if hasattr(w, "tape_census") else None
        print(json.dumps({"variant": v, "group": g, "fwd_min": round(fwd[0], 4), "fwd_med": round(fwd[len(fwd) // 2], 4),
                          "bwd_med": round(bwd[len(bwd) // 2], 4), "sha": sha, "bitwise_equal_first": sha == ref,
                          "fault": w.err.tolist()[0], "loss": float(loss), "tape_bytes": census}), flush=True)


if __name__ == "__main__":
    main()
