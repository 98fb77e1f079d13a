#!/usr/bin/env python3
"""Histogram of the record kinds one replica of config 4 writes (the stream the reverse sweep replays).  GPU box."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "diff-hybrid-traffic-sim_amd"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from dhts import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
w = bench.ItscpHybridWorkload(dev, 0, 1, 0, 0)
a = w.action.detach()
t = w.tab
R, A = a.shape
d = _lib.NetDesc(R, t.n_lanes, t.n_cells, t.T, w.sq, w.F, A, w.dt, w.um, 0.2, 5.0)
tc = t.c(0)
lib = _lib.lib()
ws_n = lib.dhts_net_hybrid_workspace_bytes(C.byref(d), C.byref(tc))
hist = torch.empty(R * (t.T + 1) * 4 * t.n_cells, dtype=torch.float32, device=dev)
tape = torch.empty(lib.dhts_net_hybrid_tape_bytes(C.byref(d)) // 4, dtype=torch.float32, device=dev)
kc = torch.empty(R * t.T * t.n_cells, dtype=torch.float32, device=dev)
queue = torch.empty(R, t.T, t.n_lanes, dtype=torch.float32, device=dev)
reward = torch.empty(R, dtype=torch.float32, device=dev)
counts = torch.zeros(R, 4, dtype=torch.int32, device=dev)
ws = torch.zeros(ws_n, dtype=torch.uint8, device=dev)
err = ops.new_error_record(dev)
P = lambda x: C.c_void_p(x.data_ptr())  # noqa: E731
rc = lib.dhts_net_hybrid_rollout_fwd(C.byref(d), C.byref(tc), P(a), P(hist), P(tape), P(kc), P(queue), P(reward), P(counts), P(ws), P(err), ops._stream())
torch.cuda.synchronize()
n = int(counts[0, 2])
T, L = t.T, t.n_lanes
off = ((4 * T * 2 * L + 15) // 16) * 16          # own_hist in front of rec_k (hybrid_kernels.hip: hyb_ws)
rk = ws[off:off + 4 * n].cpu().numpy().view(np.int32)
kinds = rk >> 24
names = {1: "NODE", 2: "COMMIT", 3: "DEPOSIT", 4: "CELLREAD", 5: "SIGNAL", 6: "SEED", 7: "IMPORT", 8: "IDM"}
print("records:", n, "=", n / T, "per step;", counts[0].tolist())
for k, c in sorted(zip(*np.unique(kinds, return_counts=True))):
    print("  %-8s %6d  (%.2f per step)" % (names.get(int(k), k), c, c / T))
