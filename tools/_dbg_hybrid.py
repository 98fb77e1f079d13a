import os, sys, json, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/diff-hybrid-traffic-sim_amd'); sys.path.insert(0,'/root/repo/tests')
from test_itscp_gpu import build_env
g=np.load('/root/repo/tests/golden/itscp_hybrid_short.npz'); m=json.loads(str(g['meta']))
env=build_env(g,m)
keys=list(env.lane.keys())
action=torch.tensor(g['action'],device='cuda',requires_grad=True)
env._simulate(action,True)
np.set_printoptions(precision=4,suppress=True,linewidth=220)
queue=np.array([[float(x) for x in env.queue_length[k]] for k in keys])
print('queue rel', np.max(np.abs(queue-g['queue']))/np.max(np.abs(g['queue'])), 'nveh', env.simulator.num_vehicle, m['n_vehicle_spawned'])
for tag,want in (('macro',True),('micro',False)):
    part=0
    for k in keys:
        if env.lane[k].sim_lane.is_macro()==want:
            for x in env.queue_length[k]: part=part+(-1.0)*x
    if isinstance(part,torch.Tensor) and part.requires_grad:
        ga=torch.autograd.grad(part,action,retain_graph=True,allow_unused=True)[0]
        ga=np.zeros(len(g['action'])) if ga is None else ga.cpu().numpy()
    else: ga=np.zeros(len(g['action']))
    ref=g['g_action_%s_lanes'%tag]
    print(tag,'max abs ref',np.abs(ref).max(),'max abs diff',np.abs(ga-ref).max())
    print(' mine',ga.reshape(-1,9)); print(' ref ',ref.reshape(-1,9))
