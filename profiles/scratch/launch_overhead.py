import sys; sys.path.insert(0,'.')
import torch, numpy as np
from bench import c2_config
from collectivecrossing_amd.batched import BatchedCollectiveCrossing
cfg=c2_config(); E,N=4096,8
env=BatchedCollectiveCrossing(cfg,E); env.make_reset_pool(0,4096); env.reset_from_pool()
acts=torch.randint(0,5,(64,E,N),dtype=torch.uint8,device=env.device)
for K in (1,2,4,8,16,32,64):
    traj=env.alloc_rollout(K)
    ts=[]
    for r in range(20):
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record(); env.rollout(acts[:K],auto_reset=True,out=traj); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1)*1e3)
    print(K, "median us %.1f  min %.1f"%(np.median(ts), min(ts)))
# no pool / no auto reset
ts=[]
traj=env.alloc_rollout(1)
for r in range(20):
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record(); env.rollout(acts[:1],auto_reset=False,out=traj); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1)*1e3)
print("K=1 no auto-reset: median us %.1f min %.1f"%(np.median(ts),min(ts)))
ts=[]
for r in range(20):
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record(); env.rollout(acts[:1],auto_reset=False,want_traj=False); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1)*1e3)
print("K=1 no outputs: median us %.1f min %.1f"%(np.median(ts),min(ts)))
