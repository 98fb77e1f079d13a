#!/usr/bin/env python3
"""Times the macro rollout kernels of BASELINE config 2 for several builds of libdhts.so (tools/build_variants.sh) and checks
every build against the product build bit for bit (final state, and the gradient the reverse sweep makes of its tape).
GPU box:  python3 tools/exp_fwd_variants.py [name ...]      (names of csrc/variants/libdhts_<name>.so; default: all)"""
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
print(json.dumps({"fwd_min": fwd[0], "fwd_med": fwd[len(fwd) // 2], "bwd_min": bwd[0], "bwd_med": bwd[len(bwd) // 2],
                  "sha": h.hexdigest()[:16], "fault": w.err.tolist()[0]}))
"""


def main():
    names = sys.argv[1:] or sorted(os.path.basename(p)[len("libdhts_"):-3] for p in glob.glob(os.path.join(VAR, "libdhts_*.so")))
    res = {}
    for name in ["product"] + names:
        env = dict(os.environ)
        if name != "product":
            env["DHTS_LIB"] = os.path.join(VAR, "libdhts_%s.so" % name)
        p = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=600)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        res[name] = json.loads(line[-1]) if line else {"error": p.stderr[-500:]}
        print(name, res[name], flush=True)
    ref = res["product"].get("sha")
    for name in names:
        print("%-12s fwd %.3f ms (min %.3f)  bwd %.3f  %s" % (name, res[name].get("fwd_med", -1), res[name].get("fwd_min", -1),
              res[name].get("bwd_med", -1), "bitwise = product" if res[name].get("sha") == ref else "DIFFERS from product"))


if __name__ == "__main__":
    main()
