import os, sys
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "diff-hybrid-traffic-sim_amd")); sys.path.insert(0, ROOT)
import torch, bench
from dhts import _lib
dev = torch.device("cuda:0")
w = bench.MacroWorkload(dev, 0, 1024, 512, 1000)
for dbg in (0, 4, 2, 6):
    for waves in (4, 8):
        _lib.lib().dhts_set_option(77, dbg)
        _lib.lib().dhts_set_option(_lib.OPT_MACRO_FWD_WAVES, waves)
        w.ev = []
        for _ in range(2): w.one_pass()
        for _ in range(5): w.one_pass(record=True)
        torch.cuda.synchronize()
        fwd = sorted(e[0].elapsed_time(e[1]) for e in w.ev)
        print("dbg %d waves %d: fwd median %.3f ms" % (dbg, waves, fwd[2]), flush=True)
