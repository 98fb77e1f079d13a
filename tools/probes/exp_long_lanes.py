#!/usr/bin/env python3
"""Forward + reverse rollout times for lanes longer than one workgroup's threads (256 lanes x 2048 cells x 1000 steps and
512 x 1536): the two-cells-per-thread reverse sweep against the general one (forced by asking for per-step cotangents that are
zero).  GPU box: python3 tools/probes/exp_long_lanes.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "diff-hybrid-traffic-sim_amd"))
import torch  # noqa: E402

from dhts import ops  # noqa: E402

dev = torch.device("cuda:0")
um, dt, dx, T = 30.0, 0.01, 5.0, 1000
for L, N in ((256, 2048), (512, 1536), (512, 1024)):
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
    zeros = torch.zeros(T, L, 2, N, device=dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    res = {}
    for rep in range(3):
        ev[0].record()
        out = ops.macro_rollout_fwd(desc, T, r, y, u, q, ghost, tape=tape)
        ev[1].record()
        g_r, g_y = 2 * out[0], torch.zeros_like(out[0])
        ev[2].record()
        a = ops.macro_rollout_bwd(desc, T, tape, g_r, g_y)
        ev[3].record()
        torch.cuda.synchronize()
        res["fwd"], res["bwd"] = ev[0].elapsed_time(ev[1]), ev[2].elapsed_time(ev[3])
        ev[2].record()
        b = ops.macro_rollout_bwd(desc, T, tape, g_r, g_y, g_hist=zeros)
        ev[3].record()
        torch.cuda.synchronize()
        res["bwd_general"] = ev[2].elapsed_time(ev[3])
    same = all(torch.equal(x, z) for x, z in zip(a, b))
    print("%d lanes x %d cells x %d steps: plan %s  fwd %.2f ms  reverse %.2f ms  (per-step-cotangent path %.2f ms)  same bits: %s" % (
        L, N, T, ops.macro_rollout_plan(desc, T)["bwd_pipelined"], res["fwd"], res["bwd"], res["bwd_general"], same), flush=True)
