#!/usr/bin/env python3
"""Every reference episode the fused hybrid kernels hold, launched with the packed plan forced (two_per_cu = 1): is the plan taken, how many
records per lane and step does its staging area hold (plan[2]), and does the episode fit (fault record).  GPU box."""
import glob, os, sys
ROOT='/root/repo'
sys.path[:0]=[ROOT, os.path.join(ROOT,'diff-hybrid-traffic-sim_amd'), os.path.join(ROOT,'tests')]
import numpy as np, torch
from test_oracle_golden import itscp_hybrid_tables, itscp_micro_tables
from dhts import ops
dev=torch.device('cuda',0)
for f in sorted(glob.glob(os.path.join(ROOT,'tests/golden/itscp_*.npz'))):
    name=os.path.basename(f)[6:-4]
    if name.startswith('eval') or name.startswith('macro'): continue
    g=np.load(f)
    if 'micro' in name:
        t,m,rows=itscp_micro_tables(g)
    else:
        t,m=itscp_hybrid_tables(g); rows=g['spawn_routes']
    try:
        t.check_kernel_limits()
    except ValueError:
        continue
    tab=ops.DeviceHybridTables(t,rows,dev)
    tab.two_per_cu=1
    args=(m['num_intersection']**2,m['simulation_frequency']*m['signal_length'],1.0/m['simulation_frequency'],m['speed_limit'],m['static_speed'],m['vehicle_length'])
    plan=ops.net_hybrid_plan(2,len(g['action']),tab,args[0])
    err=ops.new_error_record(dev)
    a=torch.tensor(g['action'][None],device=dev)
    ops.net_hybrid_rollout(a,tab,*args,check_faults=False,err=err)
    print(name, 'packed',plan['packed'],'stage_h',plan['stage_h'],'fault',err.tolist())
