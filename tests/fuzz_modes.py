"""Seeded fuzzing of the alternate planner modes against the oracle (large-count version of test_fuzz_alternate_modes).
    python tests/fuzz_modes.py SEED0 TRIALS
A tick on which kernel and oracle differ in cost or plan (statuses equal) goes to a third solver when HiGHS is importable: the agents'
QPs are solved by HiGHS (tests/highs_qp.py), and the tick counts as ARBITRATED -- the oracle's miss, not a mismatch -- when the kernel
is inside the tolerance table against HiGHS.  (The oracle's normal equations lose the optimum of near-degenerate slack-mode QPs now and
then: 3.6e-6 relative in tests/golden/fuzz_found_6800157.npz, where HiGHS and the kernel agree to 2e-8.)  Every such tick, and every
mismatch, is dumped to gpurun_out/fuzz/."""
import os, sys, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import lsc_planner_amd as L
from lsc_planner_amd.planner import PlannerConfig, next_state_host
from lsc_planner_amd.mission import Mission
from oracle import oracle as O
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from tolerances import COST_ATOL, COST_RTOL, FUZZ_PLAN_COMPARED_BELOW_COST, FUZZ_TRAJ_ATOL as TRAJ_ATOL
seed0=int(sys.argv[1]); ntr=int(sys.argv[2])
bad=0; tot=0; fails=0; arbitrated=0; maxd=0.0; maxc=0.0
try:
    import highs_qp as H
    HAVE_H = H.available()
except Exception:
    HAVE_H = False


def highs_verdict(sw, prm, md_kw, state, goal, traj, tick, g, o2, vnom, vmax, amax):
    """True when, for every agent on which kernel and oracle differ, the kernel is inside the table against HiGHS."""
    n = len(state)
    bvc = md_kw.get("planner") == "bvc"
    okh = True
    for a in range(n):
        dc = abs(g["cost"][a] - o2["cost"][a]); dp = np.abs(g["traj"][a] - o2["traj"][a]).max()
        if dc <= COST_RTOL * abs(o2["cost"][a]) + COST_ATOL and dp <= TRAJ_ATOL:
            continue
        others = [j for j in range(n) if j != a]
        obs = []
        for j in others:
            pred = O.shift_traj(traj[j]) if tick >= 2 else O.const_vel_traj(state[j, :3], state[j, 3:6])
            moved = sw.slack_set[a, j] and np.linalg.norm(pred[:, 0] - state[j, :3]) > 0.15
            obs.append(np.repeat(state[j, :3, None], 30, axis=1) if (bvc or moved) else pred)
        qp = O.qp_assemble_ex(prm, O.make_modes(**md_kw), state[a], goal[a], float(vnom[a]), vmax[a], amax[a], np.array(obs, np.float32),
                              o2["normal"][a], o2["d"][a], slack_flags=sw.slack_set[a, others])
        verdict, xh, hc = H.solve_oracle_qp(qp)[:3]
        if verdict != "Optimal":
            return False
        xh = np.asarray(xh)[:90].reshape(3, 30)
        good = abs(g["cost"][a] - hc) <= COST_RTOL * abs(hc) + COST_ATOL and (abs(hc) >= FUZZ_PLAN_COMPARED_BELOW_COST or np.abs(xh - g["traj"][a]).max() <= TRAJ_ATOL)
        print("   HiGHS agent", a, "cost", hc, "kernel", g["cost"][a], "oracle", o2["cost"][a], "|plan - HiGHS| kernel %.2e oracle %.2e" % (np.abs(xh - g["traj"][a]).max(), np.abs(xh - o2["traj"][a]).max()), flush=True)
        okh = okh and good
    return okh
MODES=[(dict(planner_mode="bvc"), dict(planner="bvc")),
       (dict(planner_mode="bvc", slack_mode="collision_constraint"), dict(planner="bvc", slack="collision_constraint")),
       (dict(planner_mode="bvc", slack_mode="dynamical_limit"), dict(planner="bvc", slack="dynamical_limit")),
       (dict(planner_mode="bvc", n_constraint_segments=2), dict(planner="bvc", n_constraint_segments=2)),
       (dict(reset_threshold=0.15), dict(reset_threshold=0.15))]
for trial in range(ntr):
    rng=np.random.default_rng(seed0+trial)
    n=int(rng.integers(2,10))
    side=float(rng.uniform(1.0,4.0)); zt=float(rng.uniform(1.0,3.0))
    wmin=np.array([-side,-side,0],np.float32); wmax=np.array([side,side,zt],np.float32)
    # separated starts (BVC needs distinct positions)
    while True:
        start=rng.uniform(wmin+0.2,wmax-0.2,(n,3)).astype(np.float32)
        D=np.linalg.norm(start[:,None]-start[None],axis=2)+np.eye(n)*9
        if D.min()>0.45: break
    goal=rng.uniform(wmin+0.1,wmax-0.1,(n,3)).astype(np.float32)
    radius=rng.uniform(0.08,0.2,n); dw=rng.uniform(1.0,2.5,n)
    vmax=np.repeat(rng.uniform(0.4,2.0,(n,1)),3,1); amax=np.repeat(rng.uniform(1.0,4.0,(n,1)),3,1); vnom=rng.uniform(0.5,1.5,n)
    ms=Mission(start,goal,wmin,wmax,radius,dw,vmax,amax,vnom,name="fuzz")
    ck,mk=MODES[trial%len(MODES)]
    pl=L.SwarmPlanner(ms, PlannerConfig(goal_mode="static", **ck))
    prm=O.make_params(world_min=wmin, world_max=wmax, obs_f32=True)
    sw=O.SwarmEx(prm, O.make_modes(**mk), radius, dw, vmax, amax, vnom)
    state=np.zeros((n,9),np.float32); state[:,:3]=start
    traj=np.zeros((n,3,30),np.float32); stale=np.zeros_like(traj)
    gust_tick=int(rng.integers(3,7)) if "reset_threshold" in ck else -1
    for tick in range(1,11):
        if tick==gust_tick:
            q=int(rng.integers(0,n)); state[q,:3]+=rng.uniform(-0.4,0.4,3).astype(np.float32)
        own=sw.disturbance_update(state, traj, tick)
        g=pl.plan(state, goal, traj)
        sw.stale[:]=stale
        o=sw.tick(state, goal, traj, tick, want_lsc=False, nthreads=8)
        tot+=n; ok=o["status"]==0; fails+=int((~ok).sum())
        msg=None
        if not np.array_equal(g["status"],o["status"]): msg="status %s vs %s"%(g["status"],o["status"])
        elif not np.isfinite(g["traj"]).all(): msg="non-finite"
        elif not (np.abs(g["cost"]-o["cost"])[ok] <= COST_RTOL*np.abs(o["cost"])[ok]+COST_ATOL).all(): msg="cost rel %.2e"%(np.abs(g["cost"]-o["cost"])[ok]/np.maximum(1e-30,np.abs(o["cost"])[ok])).max()
        elif np.abs(g["traj"]-o["traj"])[(~ok) | (np.abs(o["cost"])<FUZZ_PLAN_COMPARED_BELOW_COST)].max(initial=0.0)>TRAJ_ATOL: msg="traj %.2e"%np.abs(g["traj"]-o["traj"]).max()   # (plans compared below |f| = 1e4 only, see tests/test_gpu_fuzz.py)
        elif (np.abs(o["cost"])>=FUZZ_PLAN_COMPARED_BELOW_COST).any(): big=locals().get("big",0)+1
        if msg is None:
            sel=(~ok) | (np.abs(o["cost"])<FUZZ_PLAN_COMPARED_BELOW_COST)
            maxd=max(maxd, float(np.abs(g["traj"]-o["traj"])[sel].max(initial=0.0))); maxc=max(maxc, float((np.abs(g["cost"]-o["cost"])[ok]/np.maximum(1e-3,np.abs(o["cost"])[ok])).max(initial=0.0)))
        if msg and HAVE_H and not msg.startswith("status") and msg != "non-finite":
            o2=sw.tick(state, goal, traj, tick, want_lsc=True, nthreads=8)
            if highs_verdict(sw, prm, mk, state, goal, traj, tick, g, o2, vnom, vmax, amax):
                arbitrated+=1; print("ARBITRATED seed",seed0+trial,"n",n,"mode",ck,"tick",tick,msg,"(the kernel is inside the table against HiGHS)",flush=True)
                import os; os.makedirs("gpurun_out/fuzz",exist_ok=True)
                np.savez("gpurun_out/fuzz/arbitrated_%d.npz"%(seed0+trial), state=state, traj=traj, stale=stale, goal=goal, tick=tick, gstatus=g["status"], ostatus=o["status"], gtraj=g["traj"], gcost=g["cost"], ocost=o["cost"], slack=sw.slack_set, which=trial%len(MODES), start=start, radius=radius, dw=dw, vmax=vmax, amax=amax, vnom=vnom, wmin=wmin, wmax=wmax, giters=g["iters"])
                msg=None
        if msg:
            bad+=1; print("MISMATCH seed",seed0+trial,"n",n,"mode",ck,"tick",tick,msg,flush=True)
            import os; os.makedirs("gpurun_out/fuzz",exist_ok=True)
            np.savez("gpurun_out/fuzz/mismatch_%d.npz"%(seed0+trial), state=state, traj=traj, stale=stale, goal=goal, tick=tick, gstatus=g["status"], ostatus=o["status"], gtraj=g["traj"], gcost=g["cost"], ocost=o["cost"], slack=sw.slack_set, which=trial%len(MODES), start=start, radius=radius, dw=dw, vmax=vmax, amax=amax, vnom=vnom, wmin=wmin, wmax=wmax, giters=g["iters"])
            break
        stale=np.where(ok[:,None,None], g["traj"], stale).astype(np.float32); traj=g["traj"]; state=next_state_host(traj)
    pl.close()
print("modes fuzz done: trials",ntr,"agent-ticks",tot,"oracle failures",fails,"mismatching trials",bad,"arbitrated by HiGHS (oracle off)",arbitrated,"| largest plan difference %.2e m, cost difference %.2e relative"%(maxd,maxc))
