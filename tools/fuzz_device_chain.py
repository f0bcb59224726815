"""Device-resident fused chain vs host-buffer chain, bit for bit, on random missions (large-count version of the test)."""
import sys, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import lsc_planner_amd as L
from lsc_planner_amd.planner import PlannerConfig, next_state_host
from lsc_planner_amd.mission import Mission
dev=torch.device("cuda",0)
bad=0; tot=0
for trial in range(120):
    rng=np.random.default_rng(90000+trial)
    n=int(rng.integers(1,40))
    side=float(rng.uniform(1.0,6.0)); zt=float(rng.uniform(0.8,3.0))
    wmin=np.array([-side,-side,0],np.float32); wmax=np.array([side,side,zt],np.float32)
    start=rng.uniform(wmin+0.05,wmax-0.05,(n,3)).astype(np.float32); goal=rng.uniform(wmin-0.2,wmax+0.2,(n,3)).astype(np.float32)
    radius=rng.uniform(0.05,0.3,n); dw=rng.uniform(1.0,3.0,n)
    vmax=np.repeat(rng.uniform(0.3,2.5,(n,1)),3,1); amax=np.repeat(rng.uniform(0.5,5.0,(n,1)),3,1); vnom=rng.uniform(0.3,2.0,n)
    ms=Mission(start,goal,wmin,wmax,radius,dw,vmax,amax,vnom,name="fuzz")
    mode="prior_based" if trial%2 else "static"
    thr=0.15 if trial%3==0 else 0.0
    cap=3 if trial%5==0 else 0
    cfg=dict(goal_mode=mode, reset_threshold=thr, max_rows_per_cp=cap)
    ph=L.SwarmPlanner(ms, PlannerConfig(**cfg)); pd=L.SwarmPlanner(ms, PlannerConfig(**cfg))
    state=np.zeros((n,9),np.float32); state[:,:3]=start
    traj=np.zeros((n,3,30),np.float32)
    f32=dict(dtype=torch.float32, device=dev)
    st_d=[torch.tensor(state,device=dev), torch.zeros((n,9),**f32)]
    tj_d=[torch.zeros((n,90),**f32), torch.zeros((n,90),**f32)]
    goal_d=torch.tensor(goal,device=dev)
    cost=torch.zeros(n,dtype=torch.float64,device=dev); status=torch.zeros(n,dtype=torch.int32,device=dev); iters=torch.zeros(n,dtype=torch.int32,device=dev)
    stream=torch.cuda.current_stream().cuda_stream
    for tick in range(1,11):
        g=ph.plan(state, goal, traj)
        pd.tick_device_fused(st_d[0], goal_d, tj_d[0], tj_d[1], st_d[1], cost, status, iters, tick, stream)
        torch.cuda.synchronize()
        tot+=n
        tn=tj_d[1].cpu().numpy().reshape(n,3,30)
        ns=next_state_host(g["traj"])
        if not (np.array_equal(tn,g["traj"]) and np.array_equal(status.cpu().numpy(),g["status"]) and np.array_equal(st_d[1].cpu().numpy(), ns)):
            bad+=1; print("MISMATCH trial",trial,"n",n,cfg,"tick",tick, np.abs(tn-g["traj"]).max(), np.abs(st_d[1].cpu().numpy()-ns).max(), flush=True); break
        okc = g["status"]==0
        if not np.array_equal(cost.cpu().numpy()[okc], g["cost"][okc]): bad+=1; print("COST MISMATCH", trial, tick); break
        traj=g["traj"]; state=ns
        st_d.reverse(); tj_d.reverse()
    ph.close(); pd.close()
print("device-vs-host fuzz: agent-ticks",tot,"mismatching trials",bad)
