#!/usr/bin/env python3
"""Macro forward rollout (with tape) for several shapes x kernel (DHTS_OPT_MACRO_FWD_VARIANT: 0 = pair kernel, 2 = lane / lane-group
kernels) x traffic lanes per workgroup (DHTS_OPT_MACRO_FWD_GROUP = 1, 2, 4).  GPU box: python3 tools/probes/exp_fwd_groups_shapes.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "diff-hybrid-traffic-sim_amd"))
import torch  # noqa: E402

from dhts import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
um, dt, dx, T = 30.0, 0.01, 5.0, 500
for L, N in ((8192, 128), (4096, 256), (2048, 256), (2048, 512), (1024, 512), (512, 512), (1024, 384), (512, 1024), (256, 1024), (300, 512)):
    gen = torch.Generator().manual_seed(5)
    r = (0.05 + 0.9 * torch.rand(L, N, generator=gen)).to(dev)
    u = (um * torch.rand(L, N, generator=gen)).to(dev)
    gr = (0.05 + 0.9 * torch.rand(L, 2, generator=gen)).to(dev)
    gu = (um * torch.rand(L, 2, generator=gen)).to(dev)
    y, q = ops.macro_state_from_ru(r, u, um)
    gy, gq = ops.macro_state_from_ru(gr, gu, um)
    ghost = torch.stack([gr, gy, gu, gq], dim=-1).contiguous()
    desc = ops.macro_desc(L, N, dt, dx, um)
    tape = torch.empty(ops.macro_tape_numel(desc, T), device=dev)
    line = "%5d lanes x %4d cells x %d steps:" % (L, N, T)
    ref = None
    for V, G in ((2, 0), (0, 1), (0, 2), (0, 4), (0, 0)):
        assert _lib.lib().dhts_set_option(_lib.OPT_MACRO_FWD_VARIANT, V) == 0
        assert _lib.lib().dhts_set_option(_lib.OPT_MACRO_FWD_GROUP, G) == 0
        plan = ops.macro_rollout_plan(desc, T)
        ts = []
        for rep in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            out = ops.macro_rollout_fwd(desc, T, r, y, u, q, ghost, tape=tape)
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        if ref is None:
            ref = [o.clone() for o in out]
        same = all(torch.equal(x, z) for x, z in zip(out, ref))
        line += "   V%d G=%d (kernel %d, %d lanes/wg): %.3f ms%s" % (V, G, plan["fwd_kernel"], plan["fwd_lanes_per_group"], sorted(ts)[2], "" if same else " DIFFERS")
    print(line, flush=True)
_lib.lib().dhts_set_option(_lib.OPT_MACRO_FWD_GROUP, 0)
_lib.lib().dhts_set_option(_lib.OPT_MACRO_FWD_VARIANT, 0)
