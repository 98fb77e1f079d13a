import os, sys, ctypes as C
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "diff-hybrid-traffic-sim_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, bench
from dhts import _lib
dev = torch.device("cuda:0")
w = bench.ItscpHybridWorkload(dev, 0, 256, 0, 0)
for _ in range(3): w.one_pass()
w.ev = []
for _ in range(5): w.one_pass(record=True)
torch.cuda.synchronize()
print("fwd %.3f ms bwd %.3f ms" % (np.median([e[0].elapsed_time(e[1]) for e in w.ev]), np.median([e[2].elapsed_time(e[3]) for e in w.ev])))
lib = C.CDLL(os.path.join(ROOT, "diff-hybrid-traffic-sim_amd/csrc/libdhts.so"))
buf = (C.c_longlong * (8 * 3 * 8))()
assert lib.dhts_debug_read(buf) == 0
a = np.array(buf[:]).reshape(8, 3, 8)
for rep in (0, 3):
    for role, nm in enumerate(("cell wave 0", "flush wave", "micro wave")):
        print("replica %d %-12s work A/B/C/D %s   wait A/B/C/D %s   sum %d" % (rep, nm, a[rep, role, :4], a[rep, role, 4:], a[rep, role].sum()))

assert lib.dhts_debug_read2(buf) == 0
b = np.array(buf[:]).reshape(8, 3, 8)
for rep in (0,):
    for role, nm in enumerate(("cell wave 0", "flush wave", "micro wave")):
        print("replica %d %-12s A: ghosts %d scan %d lanes %d vsamples %d | D: caps %d prescreen %d events %d commits+publish %d" % ((rep, nm) + tuple(b[rep, role])))
