import sys, os, numpy as np, torch, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/diff-hybrid-traffic-sim_amd'); sys.path.insert(0,'/root/repo/tests')
from test_oracle_golden import itscp_hybrid_tables
from dhts import ops
dev=torch.device('cuda')
g=np.load('/root/repo/tests/golden/itscp_hybrid.npz')
t,m=itscp_hybrid_tables(g)
dt_=ops.DeviceHybridTables(t,g["spawn_routes"],dev)
R=256
a=torch.tensor(np.tile(g['action'][None,:],(R,1)),device=dev,requires_grad=True)
args=(9,120,1/30,60.0)
for it in range(2):
    a.grad=None
    cut,reward,queue,counts=ops.net_hybrid_rollout(a,dt_,*args)
    cut.sum().backward(); torch.cuda.synchronize()
k0=a.grad[1,:16].cpu().numpy(); k1=a.grad[2,:16].cpu().numpy()
print('thread0 phases R0..R4 us:',k0[:5].round(0),'sum',k0[:5].sum())
print('micro   phases R0..R4 us:',k1[:5].round(0),' replay R2:',k1[8].round(0),' replay R3+outbox:',k1[9].round(0))
