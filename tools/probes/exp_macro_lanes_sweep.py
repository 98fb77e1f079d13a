#!/usr/bin/env python3
"""Config 2's kernels at 1 x ... 8 x its lanes on ONE GPU (the tape grows with the lanes: 24 GB of address space per 1024 lanes): forward
and reverse launch times per 1024 lanes, with and without the pair kernel's priority rotation."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402
from dhts import _lib  # noqa: E402

dev = torch.device("cuda", 0)
sizes = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "1024,2048,4096,8192".split(","))]
for L in sizes:
    w = bench.MacroWorkload(dev, 0, L, 512, 1000)
    for rot in (1, 0):
        _lib.lib().dhts_set_option(_lib.OPT_MACRO_FWD_ROTATE, rot)
        w.ev = []
        for _ in range(2):
            w.one_pass()
        torch.cuda.synchronize()
        for _ in range(3):
            w.one_pass(record=True)
        torch.cuda.synchronize()
        k, _ = bench.kernel_records(w)
        print(json.dumps({"lanes": L, "rotate": rot, "fwd_ms": k["rollout_fwd"]["ms"], "bwd_ms": k["rollout_bwd"]["ms"],
                          "fwd_ms_per_1024_lanes": k["rollout_fwd"]["ms"] * 1024 / L, "bwd_ms_per_1024_lanes": k["rollout_bwd"]["ms"] * 1024 / L,
                          "tape_GB": w.tape_bytes / 1e9}), flush=True)
    del w
    torch.cuda.empty_cache()
_lib.lib().dhts_set_option(_lib.OPT_MACRO_FWD_ROTATE, 1)
