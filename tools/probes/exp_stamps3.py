import os, sys
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "diff-hybrid-traffic-sim_amd")); sys.path.insert(0, ROOT)
import torch, bench
from dhts import _lib
dev = torch.device("cuda:0")
w = bench.MacroWorkload(dev, 0, 1024, 512, 1000)
for waves in (4,):
    _lib.lib().dhts_set_option(_lib.OPT_MACRO_FWD_WAVES, waves)
    w.ev = []
    for _ in range(2): w.one_pass()
    for _ in range(3): w.one_pass(record=True)
    torch.cuda.synchronize()
    fwd = sorted(e[0].elapsed_time(e[1]) for e in w.ev)
    q = w.out[3][:16].cpu().numpy()
    print("waves %d: fwd median %.3f ms" % (waves, fwd[1]))
    for wg in (0, 3):
        for wv in range(8):
            v = q[2 * wg, wv * 8: wv * 8 + 4]
            print("  wg %d wave %d: work/interval %6.0f  wait %6.0f | when active %6.0f x %5.0f" % ((wg, wv) + tuple(v)))
