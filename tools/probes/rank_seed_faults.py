#!/usr/bin/env python3
"""Which of the eight ranks' config-5 batches (bench.py's itscp_hybrid workload, 256 replicas, rank seeds 0..7) hold a replica
whose reverse sweep goes non-finite (the reference asserts on such an action): run on ONE GPU before the 8-GPU bench."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402

dev = torch.device("cuda", 0)
out = []
for rank in range(8):
    w = bench.make_workload("itscp_hybrid", dev, rank, 256)
    loss, g, _ = w.one_pass()
    torch.cuda.synchronize()
    out.append({"rank": rank, "fault": w.err.tolist(), "dropped": w.dropped_replicas(), "loss": float(loss),
                "grad_finite": bool(torch.isfinite(g).all())})
    del w
    torch.cuda.empty_cache()
print(json.dumps(out))
