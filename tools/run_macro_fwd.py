#!/usr/bin/env python3
"""A few launches of the macro rollout kernels of BASELINE config 2 with a chosen forward variant / wave count -- the
program rocprofv3 is pointed at for counter passes (GPU box):

    rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES ... -d <out> --output-format csv -- python3 tools/run_macro_fwd.py 0 0
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diff-hybrid-traffic-sim_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from dhts import _lib  # noqa: E402

variant = int(sys.argv[1]) if len(sys.argv) > 1 else 0
waves = int(sys.argv[2]) if len(sys.argv) > 2 else 0
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda:0")
w = bench.MacroWorkload(dev, 0, 1024, 512, 1000)
_lib.lib().dhts_set_option(_lib.OPT_MACRO_FWD_VARIANT, variant)
_lib.lib().dhts_set_option(_lib.OPT_MACRO_FWD_WAVES, waves)
for _ in range(passes):
    w.one_pass(record=True)
torch.cuda.synchronize()
print("variant %d waves %d: fwd %s ms" % (variant, waves, ["%.3f" % e[0].elapsed_time(e[1]) for e in w.ev]))
