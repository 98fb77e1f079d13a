#!/usr/bin/env python3
"""dhts_micro_step_fwd_tensor (itscp `micro` mode's float32 tensor ladder) against the oracle, bit for bit, on 81 920 vehicle-steps whose
vehicles carry random_micro_vehicle-style attributes (doubles with all their bits) and whose states include tiny gaps, collisions and
standing vehicles.  GPU box."""
import os, sys
ROOT='/root/repo'
sys.path[:0]=[ROOT, os.path.join(ROOT,'diff-hybrid-traffic-sim_amd'), os.path.join(ROOT,'tests')]
import numpy as np, torch
from dhts import ops
from oracle import oracle as O
dev=torch.device('cuda',0)
rng=np.random.default_rng(5)
L,V=64,64
dt=1.0/30.0
tot=bad_p=bad_v=0
for trial in range(20):
    sl=60.0
    prm=np.empty((L,V,6))
    u=rng.random((L,V,5))
    prm[...,0]=sl*1.5+u[...,0]*sl*0.5; prm[...,1]=sl*1.0+u[...,1]*sl*0.5; prm[...,2]=sl*0.8+u[...,2]*sl*0.4; prm[...,3]=1.0+u[...,3]*1.0; prm[...,4]=0.2+u[...,4]*0.4; prm[...,5]=5.0
    gaps=rng.choice([0.001,0.5,2.0,6.0,15.0,40.0],(L,V))*rng.uniform(0.5,1.5,(L,V))
    p=np.cumsum(gaps+5.0,axis=1).astype(np.float32)
    v=(rng.choice([0.0,0.1,0.2,1.0,10.0,40.0,70.0],(L,V))*rng.uniform(0.0,1.2,(L,V))).astype(np.float32)
    head=np.stack([rng.uniform(0.0,60.0,L),rng.uniform(-30,30,L)],1).astype(np.float32).astype(np.float64)
    desc=ops.micro_desc(L,V,dt)
    params_d=torch.tensor(np.ascontiguousarray(prm.transpose(2,0,1)),dtype=torch.float64,device=dev)
    np_t,nv_t=ops.micro_step_fwd(desc,torch.tensor(p,device=dev),torch.tensor(v,device=dev),params_d,torch.tensor(head,device=dev),tensor_ladder=True)
    np_t,nv_t=np_t.cpu().numpy(),nv_t.cpu().numpy()
    for l in range(L):
        o=O.micro_step_f32(p[l],v[l],prm[l],head[l,0],head[l,1],dt,want_tape=False)
        dp=(np_t[l]!=o['np']); dv=(nv_t[l]!=o['nv'])
        tot+=V; bad_p+=int(dp.sum()); bad_v+=int(dv.sum())
        if dv.any() and bad_v<=6:
            i=int(np.argmax(dv)); print('lane',l,'veh',i,'v',v[l,i],'gap',(p[l,i+1]-p[l,i]-5.0) if i<V-1 else head[l,0],'kernel nv',repr(nv_t[l,i]),'oracle nv',repr(o['nv'][i]),'prm',prm[l,i])
print('vehicle-steps',tot,'position mismatches',bad_p,'speed mismatches',bad_v)
