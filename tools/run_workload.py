#!/usr/bin/env python3
"""A few passes of one bench.py workload -- the program rocprofv3 is pointed at for counter passes (GPU box):

    rocprofv3 --kernel-trace --pmc SQ_WAVES ... -d <out> --output-format csv -- python3 tools/run_workload.py <workload> [passes]

workload: macro | micro | itscp_hybrid | itscp_stepwise | itscp_macro (bench.py's instances at their BASELINE shapes).  Prints
the workload's name, its units per pass and what one launch moves, for the summariser (tools/pmc_workloads_summary.py)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diff-hybrid-traffic-sim_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "macro"
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
w = bench.make_workload(name, dev, 0)
for _ in range(passes):
    w.one_pass(record=True)
torch.cuda.synchronize()
k, dom = bench.kernel_records(w)
print("WORKLOAD " + json.dumps({"key": name, "name": w.name, "units": w.units, "unit": w.unit_name, "moved_bytes": w.moved_bytes_per_launch(),
                                "fwd_ms": k["rollout_fwd"]["ms"], "bwd_ms": k["rollout_bwd"]["ms"], "library_code_sha16": bench.library_code_sha16()}))
