import os, sys, json, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/diff-hybrid-traffic-sim_amd'); sys.path.insert(0,'/root/repo/tests')
from test_itscp_gpu import build_env
g=np.load('/root/repo/tests/golden/itscp_hybrid.npz'); m=json.loads(str(g['meta']))
def run(delta, idx):
    env=build_env(g,m)
    a=g['action'].astype(np.float64).copy(); a[idx]+=delta
    action=torch.tensor(a.astype(np.float32),device='cuda')
    with torch.no_grad():
        env._simulate(action,True)
        keys=list(env.lane.keys())
        tot=0.0
        for k in keys:
            for x in env.queue_length[k]: tot+= -float(x)
    return tot
idx=2*9+4
h=4e-3
rp=run(+h,idx); rm=run(-h,idx)
print('FD slope idx',idx,(rp-rm)/(2*h),'ref',g['g_action'][idx],'mine earlier -18.5084')
