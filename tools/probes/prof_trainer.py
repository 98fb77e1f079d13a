#!/usr/bin/env python3
"""cProfile of the reference-style controller-training driver (python -m example.control.itscp.run with run_itscp_hybrid.sh's flags,
30 epochs): where the host time around the fused episodes goes.  GPU box:  python3 tools/probes/prof_trainer.py [all]"""
import cProfile
import io
import os
import pstats
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "diff-hybrid-traffic-sim_amd")
os.chdir(PKG)
sys.path.insert(0, PKG)
everything = len(sys.argv) > 1
sys.argv = ["run", "--mode=hybrid", "--n_trial=1", "--n_intersection=3", "--n_lane=1", "--lane_length=5", "--simulation_length=20",
            "--signal_length=4", "--n_episode=30", "--lr=1e-4"]
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_module("example.control.itscp.run", run_name="__main__")
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
st = pstats.Stats(pr, stream=s).sort_stats("cumulative")
if everything:
    st.print_stats(40)
else:
    st.print_stats("diff-hybrid-traffic-sim_amd|dhts", 30)
print(s.getvalue()[:10000])
