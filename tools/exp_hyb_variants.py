#!/usr/bin/env python3
"""Times the hybrid network kernels on BASELINE config 4 (256 replicas) for several builds of libdhts.so
(SRC=hybrid_kernels tools/build_variants.sh ...): forward, reverse and the evaluation kernel; prints a checksum of the gradient.
GPU box:  python3 tools/exp_hyb_variants.py [name ...]"""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VAR = os.path.join(ROOT, "diff-hybrid-traffic-sim_amd", "csrc", "variants")
CHILD = r"""
import hashlib, json, os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "diff-hybrid-traffic-sim_amd"))
import torch
import bench
dev = torch.device("cuda:0")
w = bench.ItscpHybridWorkload(dev, 0, 256, 0, 0)
for _ in range(3):
    w.one_pass()
for _ in range(10):
    loss, g, _ = w.one_pass(record=True)
torch.cuda.synchronize()
fwd = sorted(e[0].elapsed_time(e[1]) for e in w.ev)
bwd = sorted(e[2].elapsed_time(e[3]) for e in w.ev)
ev = []
for _ in range(8):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    w.ops.net_hybrid_eval(w.action.detach(), w.tab, w.sq, w.F, w.dt, w.um, err=w.err)
    b.record()
    torch.cuda.synchronize()
    ev.append(a.elapsed_time(b))
h = hashlib.sha256(g.cpu().numpy().tobytes()).hexdigest()[:16]
print(json.dumps({"fwd_med": fwd[len(fwd) // 2], "bwd_med": bwd[len(bwd) // 2], "eval_med": sorted(ev)[len(ev) // 2], "grad_sha": h,
                  "fault": w.err.tolist()[0]}))
"""


def main():
    names = sys.argv[1:] or sorted(os.path.basename(p)[len("libdhts_"):-3] for p in glob.glob(os.path.join(VAR, "libdhts_*.so")))
    for name in ["product"] + names:
        env = dict(os.environ)
        if name != "product":
            env["DHTS_LIB"] = os.path.join(VAR, "libdhts_%s.so" % name)
        p = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=900)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        print(name, json.loads(line[-1]) if line else {"error": p.stderr[-800:]}, flush=True)


if __name__ == "__main__":
    main()
