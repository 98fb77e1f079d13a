#!/usr/bin/env python3
"""The REFERENCE on a case a fuzz run found (build container only: imports /root/reference through tools/gen_goldens.py): the
micro_rv_2x2 network with the fuzz's action and admission-draw stream, written as a fixture under /tmp -- who is right about a queue
term, the oracle or the kernels?    python tools/probes/ref_replay_case.py <case.npz with action, draws> <out dir>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_goldens as G  # noqa: E402

case = np.load(sys.argv[1])
G.OUT = sys.argv[2]
os.makedirs(G.OUT, exist_ok=True)
os.environ["DHTS_FINE_CUTS"] = "60"
os.environ.pop("DHTS_LANE_LATE", None)
G.gen_itscp("case_micro_rv_2x2", "micro", 2, 2, 10.0, 4, 1, seed=183, action_kind="rand", problem=3, random_vehicles=1.0,
            action_override=case["action"], draws_override=case["draws"])
