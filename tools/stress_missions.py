"""60 perturbed 64-agent circle-swap missions flown to completion through lsc_tick_device_fused: every agent arrives, no
failed plan, no pair closer than the collision model allows."""
import sys, numpy as np, torch, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import lsc_planner_amd as L
from lsc_planner_amd.planner import PlannerConfig
dev=torch.device("cuda",0)
rng=np.random.default_rng(1)
t0=time.time(); total=0; bad_missions=0
for rep in range(60):
    n=64
    ms=L.circle_swap(n, 8.0)
    ms.goal[:,:3]+=rng.uniform(0,0.05,(n,3)).astype(np.float32)        # goal noise like max_noise
    ms.start[:,:2]+=rng.uniform(-0.05,0.05,(n,2)).astype(np.float32)
    pl=L.SwarmPlanner(ms, PlannerConfig(goal_mode="prior_based", reset_threshold=0.15))
    f32=dict(dtype=torch.float32, device=dev)
    st=[torch.zeros((n,9),**f32), torch.zeros((n,9),**f32)]; st[0][:,:3]=torch.from_numpy(ms.start).to(dev)
    tj=[torch.zeros((n,90),**f32), torch.zeros((n,90),**f32)]
    goal=torch.from_numpy(ms.goal).to(dev)
    cost=torch.zeros(n,dtype=torch.float64,device=dev); status=torch.zeros(n,dtype=torch.int32,device=dev); iters=torch.zeros(n,dtype=torch.int32,device=dev)
    stream=torch.cuda.current_stream().cuda_stream
    nfail=0; mind=9
    for tick in range(1,301):
        pl.tick_device_fused(st[0], goal, tj[0], tj[1], st[1], cost, status, iters, tick, stream)
        st.reverse(); tj.reverse()
        if tick%10==0:
            s=status.cpu().numpy(); nfail+=int((s!=0).sum())
            p=st[0][:,:3].double().cpu().numpy().copy(); p[:,2]/=2
            D=np.linalg.norm(p[:,None]-p[None],axis=2)+np.eye(n)*9; mind=min(mind,D.min())
    total+=300*n
    fin=st[0].cpu().numpy()
    d=np.linalg.norm(fin[:,:3]-ms.goal,axis=1).max()
    okm = np.isfinite(fin).all() and d<0.11 and nfail==0 and mind>=0.3-1e-3
    if not okm: bad_missions+=1; print("rep",rep,"finite",np.isfinite(fin).all(),"max dist to goal %.3f"%d,"failed agent-ticks(sampled)",nfail,"min downwash-scaled distance %.3f"%mind, flush=True)
    pl.close()
print("stress: %d agent-ticks in %.1fs, missions with a problem: %d"%(total, time.time()-t0, bad_missions))
