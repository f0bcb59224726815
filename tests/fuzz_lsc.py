"""Seeded fuzzing of the LSC-mode tick against the oracle (the large-count version of tests/test_gpu_fuzz.py::test_fuzz_lsc_mode).
    python tests/fuzz_lsc.py SEED0 TRIALS [MAX_AGENTS] ['{"prune": 0}']
Needs a GPU and the built oracle (test infrastructure); prints one summary line."""
import os, sys, numpy as np, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import lsc_planner_amd as L
from lsc_planner_amd.planner import PlannerConfig, next_state_host
from lsc_planner_amd.mission import Mission
from oracle import oracle
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from tolerances import COST_ATOL, COST_RTOL, FUZZ_TRAJ_ATOL as TRAJ_ATOL
seed0=int(sys.argv[1]); ntr=int(sys.argv[2])
bad=0; tot=0; fails=0; maxd=0.0; maxc=0.0
for trial in range(ntr):
    rng=np.random.default_rng(seed0+trial)
    n=int(rng.integers(1,int(sys.argv[3]) if len(sys.argv)>3 else 14))
    side=float(rng.uniform(0.8,6.0)); zt=float(rng.uniform(0.6,3.0))
    wmin=np.array([-side,-side,0],np.float32); wmax=np.array([side,side,zt],np.float32)
    kind=rng.integers(0,4)
    start=rng.uniform(wmin+0.05,wmax-0.05,(n,3)).astype(np.float32)
    goal=rng.uniform(wmin-0.3,wmax+0.3,(n,3)).astype(np.float32)     # some goals outside the world
    if kind==1 and n>1: start[1]=start[0]+np.float32(1e-3)              # nearly coincident agents
    if kind==2: goal[:]=start                                          # already at goal
    radius=rng.uniform(0.05,0.4,n); dw=rng.uniform(1.0,3.0,n)
    vmax=np.repeat(rng.uniform(0.2,3.0,(n,1)),3,1); amax=np.repeat(rng.uniform(0.5,6.0,(n,1)),3,1)
    if kind==3: vmax[:,2]*=0.3; amax[:,2]*=0.5
    vnom=rng.uniform(0.3,2.0,n)
    ms=Mission(start,goal,wmin,wmax,radius,dw,vmax,amax,vnom,name="fuzz")
    mode="prior_based" if trial%2 else "static"
    try:
        import json as _j; pl=L.SwarmPlanner(ms, PlannerConfig(goal_mode=mode, **(_j.loads(sys.argv[4]) if len(sys.argv)>4 else {})))
    except Exception as e:
        print("trial",trial,"create failed",e); continue
    prm=oracle.make_params(world_min=wmin, world_max=wmax, obs_f32=True)
    sw=oracle.Swarm(prm, radius, dw, vmax, amax, vnom)
    state=np.zeros((n,9),np.float32); state[:,:3]=start
    if trial%3==0: state[:,3:6]=rng.uniform(-0.5,0.5,(n,3)).astype(np.float32)   # moving first tick
    traj=np.zeros((n,3,30),np.float32); stale=np.zeros_like(traj)
    for tick in range(1,9):
        g=pl.plan(state, goal, traj)
        goals=pl.last_goals() if mode=="prior_based" else goal
        sw.stale[:]=stale
        o=sw.tick(state, goals, traj, tick, want_lsc=False, nthreads=8)
        tot+=n
        ok=o["status"]==0; fails+=int((~ok).sum())
        msg=None
        if not np.array_equal(g["status"],o["status"]): msg="status %s vs %s"%(g["status"],o["status"])
        elif not np.isfinite(g["traj"]).all(): msg="non-finite traj"
        elif not (np.abs(g["cost"]-o["cost"])[ok] <= COST_RTOL*np.abs(o["cost"])[ok]+COST_ATOL).all(): msg="cost %s"%(np.abs(g["cost"]-o["cost"])[ok]/np.maximum(1e-30,np.abs(o["cost"])[ok])).max()
        elif np.abs(g["traj"]-o["traj"]).max()>TRAJ_ATOL: msg="traj %.2e"%np.abs(g["traj"]-o["traj"]).max()
        if msg is None:
            maxd=max(maxd, float(np.abs(g["traj"]-o["traj"]).max())); maxc=max(maxc, float((np.abs(g["cost"]-o["cost"])[ok]/np.maximum(1e-3,np.abs(o["cost"])[ok])).max(initial=0.0)))
        if msg:
            bad+=1; print("MISMATCH seed",seed0+trial,"n",n,"kind",kind,"mode",mode,"tick",tick,msg, flush=True); break
        stale=np.where(ok[:,None,None], g["traj"], stale).astype(np.float32); traj=g["traj"]; state=next_state_host(traj)
    pl.close()
print("fuzz done: trials",ntr,"agent-ticks",tot,"oracle failures",fails,"mismatching trials",bad,"| largest plan difference %.2e m, cost difference %.2e relative (|f| > 1e-3)"%(maxd,maxc))
