import os, sys
ROOT='/root/repo'
sys.path.insert(0, os.path.join(ROOT, "diff-hybrid-traffic-sim_amd")); sys.path.insert(0, ROOT)
import torch, bench
from dhts import _lib
dev = torch.device("cuda:0")
w = bench.MicroWorkload(dev, 0, 4096, 256, 1000)
for waves in (1, 2, 4, 0):
    _lib.lib().dhts_set_option(_lib.OPT_MICRO_FWD_WAVES, waves)
    w.ev = []
    for _ in range(2): w.one_pass()
    for _ in range(5): w.one_pass(record=True)
    torch.cuda.synchronize()
    fwd = sorted(e[0].elapsed_time(e[1]) for e in w.ev); bwd = sorted(e[2].elapsed_time(e[3]) for e in w.ev)
    print("micro waves/lane %d: fwd median %.3f ms (min %.3f)  bwd median %.3f ms" % (waves, fwd[2], fwd[0], bwd[2]), flush=True)
