#!/usr/bin/env python3
"""DHTS_OPT_HYB_PACK: config 4's network at 256 .. 2048 replicas with one and with two workgroups per compute unit -- forward and
reverse launch times (HIP events), and whether reward / gradient are the same bit for bit."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402
from dhts import _lib  # noqa: E402

dev = torch.device("cuda", 0)
sizes = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "256,512,1024".split(","))]
w = bench.make_workload("itscp_hybrid", dev, 0, max(sizes))
out = []
ref = {}
for pack in (0, 1, 0, 1):
    assert _lib.lib().dhts_set_option(_lib.OPT_HYB_PACK, pack) == 0
    for R in sizes:
        w.restrict(R)
        for _ in range(2):
            w.one_pass()
        torch.cuda.synchronize()
        for _ in range(4):
            _, g, _ = w.one_pass(record=True)
        torch.cuda.synchronize()
        k, _ = bench.kernel_records(w)
        rec = {"pack": pack, "replicas": R, "fwd_ms": k["rollout_fwd"]["ms"], "bwd_ms": k["rollout_bwd"]["ms"], "fault": w.err.tolist()}
        key = R
        if pack == 0 and key not in ref:
            ref[key] = (w.reward.clone(), g.clone())
        else:
            rec["reward_equal"] = bool(torch.equal(w.reward, ref[key][0]))
            rec["grad_equal"] = bool(torch.equal(g, ref[key][1]))
        out.append(rec)
        print(json.dumps(rec), flush=True)
