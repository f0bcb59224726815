"""Seeded fuzzing of the round-4 builds against the oracle: planar worlds (the 60-variable QP), M = horizon / dt = 4
(liblsc_hip_m4.so against the oracle built for M = 4), and both at once -- tiny swarms with extreme parameters as in
tests/fuzz_lsc.py (goals outside the world, nearly coincident agents, agents at their goal, moving first ticks).
    python tests/fuzz_variants.py SEED0 TRIALS [planar|m4|planar_m4|all] [MAX_AGENTS]
Needs a GPU and the built oracle (test infrastructure); prints one summary line per variant.  tests/test_gpu_fuzz.py runs a
small count of the same generator in the GPU suite."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

from tolerances import COST_ATOL, COST_RTOL, FUZZ_PLAN_COMPARED_BELOW_COST, FUZZ_TRAJ_ATOL, FUZZ_TRAJ_ATOL_HALF_SECOND

Z2D = 0.9


def run_variant(L, O, seed0, trials, variant, max_agents=14, verbose=False):
    """-> (agent_ticks, oracle_failures, list of mismatch descriptions)."""
    from lsc_planner_amd.mission import Mission
    from lsc_planner_amd.planner import PlannerConfig, next_state_host
    planar, m4 = "planar" in variant, "m4" in variant
    M, dt = (4, 0.5) if m4 else (5, 0.2)
    TRAJ_ATOL = FUZZ_TRAJ_ATOL_HALF_SECOND if m4 else FUZZ_TRAJ_ATOL
    cfg = dict(dt=dt, horizon=M * dt)
    pkw = dict(dt=dt)
    if planar:
        cfg.update(world_dimension=2, world_z_2d=Z2D)
        pkw.update(world_dimension=2, world_z_2d=Z2D)
    agent_ticks = failures = 0
    bad = []
    run_variant.maxd = 0.0
    with O.segments(M):
        for trial in range(trials):
            rng = np.random.default_rng(seed0 + trial)
            n = int(rng.integers(1, max_agents))
            side, top = float(rng.uniform(0.8, 6.0)), float(rng.uniform(1.2, 3.0))
            wmin, wmax = np.array([-side, -side, 0], np.float32), np.array([side, side, top], np.float32)
            kind = int(rng.integers(0, 4))
            start = rng.uniform(wmin + 0.05, wmax - 0.05, (n, 3)).astype(np.float32)
            goal = rng.uniform(wmin - 0.3, wmax + 0.3, (n, 3)).astype(np.float32)       # some goals outside the world
            if planar:
                start[:, 2] = goal[:, 2] = np.float32(Z2D)
            if kind == 1 and n > 1:
                start[1] = start[0] + np.array([1e-3, 1e-3, 0 if planar else 1e-3], np.float32)   # nearly coincident agents
            if kind == 2:
                goal[:] = start                                                           # already there
            radius, dw = rng.uniform(0.05, 0.4, n), rng.uniform(1.0, 3.0, n)
            vmax, amax = np.repeat(rng.uniform(0.2, 3.0, (n, 1)), 3, 1), np.repeat(rng.uniform(0.5, 6.0, (n, 1)), 3, 1)
            if kind == 3:
                vmax[:, 2] *= 0.3
                amax[:, 2] *= 0.5
            vnom = rng.uniform(0.3, 2.0, n)
            ms = Mission(start, goal, wmin, wmax, radius, dw, vmax, amax, vnom, name="fuzz")
            mode = "prior_based" if trial % 2 else "static"
            pl = L.SwarmPlanner(ms, PlannerConfig(goal_mode=mode, **cfg))
            assert pl.M == M
            prm = O.make_params(world_min=wmin, world_max=wmax, obs_f32=True, **pkw)
            sw = O.SwarmEx(prm, O.make_modes(), radius, dw, vmax, amax, vnom)
            state = np.zeros((n, 9), np.float32)
            state[:, :3] = start
            if trial % 3 == 0:
                state[:, 3:6] = rng.uniform(-0.5, 0.5, (n, 3)).astype(np.float32)         # moving first tick
                if planar:
                    state[:, 5] = 0
            traj = np.zeros((n, 3, 6 * M), np.float32)
            stale = np.zeros_like(traj)
            if planar:
                # the optimiser's trajectory starts at zero (src/traj_optimizer.cpp:16-19); in a planar world the product starts its z block
                # at z_2d, so that an agent whose FIRST QP fails keeps a plan in the plane (round 5; lsc_set_agents) -- the oracle is told the same
                stale[:, 2, :] = np.float32(Z2D)
            for tick in range(1, 9):
                g = pl.plan(state, goal, traj)
                goals = goal
                msg = None
                if mode == "prior_based":
                    goals = sw.goal_prior_based(state, goal, traj, tick, dt=dt)
                    if not np.array_equal(pl.last_goals(), goals):
                        msg = "goals"
                sw.stale[:] = stale
                o = sw.tick(state, goals, traj, tick, want_lsc=False, nthreads=8)
                ok = o["status"] == 0
                agent_ticks += n
                failures += int((~ok).sum())
                tame = ~ok | (np.abs(o["cost"]) < FUZZ_PLAN_COMPARED_BELOW_COST)
                if msg:
                    pass
                elif not np.array_equal(g["status"], o["status"]):
                    msg = "status %s vs %s" % (g["status"], o["status"])
                elif not np.isfinite(g["traj"]).all():
                    msg = "non-finite plan"
                elif not (np.abs(g["cost"] - o["cost"])[ok] <= COST_RTOL * np.abs(o["cost"])[ok] + COST_ATOL).all():
                    msg = "cost %.3e" % (np.abs(g["cost"] - o["cost"])[ok] / np.maximum(1e-30, np.abs(o["cost"])[ok])).max()
                elif np.abs(g["traj"] - o["traj"])[tame].max(initial=0.0) > TRAJ_ATOL:
                    msg = "plan %.2e" % np.abs(g["traj"] - o["traj"])[tame].max()
                elif planar and not (g["traj"][ok][:, 2, :] == np.float32(Z2D)).all():
                    msg = "a planar plan left z_2d"
                if msg is None:
                    run_variant.maxd = max(run_variant.maxd, float(np.abs(g["traj"] - o["traj"])[tame].max(initial=0.0)))
                if msg:
                    os.makedirs("gpurun_out/fuzz", exist_ok=True)
                    np.savez("gpurun_out/fuzz/variant_%s_%d.npz" % (variant, seed0 + trial), state=state, goal=goals, traj=traj, stale=stale, tick=tick, dt=dt,
                             gtraj=g["traj"], gcost=g["cost"], gstatus=g["status"], otraj=o["traj"], ocost=o["cost"], ostatus=o["status"],
                             radius=radius, dw=dw, vmax=vmax, amax=amax, vnom=vnom, wmin=wmin, wmax=wmax)
                    bad.append("seed %d n %d kind %d mode %s tick %d: %s" % (seed0 + trial, n, kind, mode, tick, msg))
                    if verbose:
                        print("MISMATCH", variant, bad[-1], flush=True)
                    break
                stale = np.where(ok[:, None, None], g["traj"], stale).astype(np.float32)
                traj = g["traj"]
                state = next_state_host(traj, dt=dt)
                if planar and not (traj[:, 2, :] == np.float32(Z2D)).all():
                    bad.append("seed %d tick %d: a plan of a planar world left the plane (failed solves keep a stale plan whose z block is z_2d)" % (seed0 + trial, tick))
                    break
            pl.close()
    return agent_ticks, failures, bad


if __name__ == "__main__":
    import lsc_planner_amd as L
    from oracle import oracle as O
    seed0, trials = int(sys.argv[1]), int(sys.argv[2])
    which = sys.argv[3] if len(sys.argv) > 3 else "all"
    cap = int(sys.argv[4]) if len(sys.argv) > 4 else 14
    for v in (["planar", "m4", "planar_m4"] if which == "all" else [which]):
        at, fl, bad = run_variant(L, O, seed0, trials, v, cap, verbose=True)
        print("fuzz %s done: trials %d agent-ticks %d oracle failures %d mismatching trials %d | largest plan difference %.2e m" % (v, trials, at, fl, len(bad), run_variant.maxd), flush=True)
