"""-m gpu: many short random missions (dense and sparse, uniform and heterogeneous agents, both goal modes), every tick
of every agent against the oracle on the same inputs: equal statuses, cost and control points within the tolerances of
test_gpu_parity.py.  Catches what single hand-picked missions do not (rare solver failures, warm-start corner cases)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tolerances import COST_ATOL, COST_RTOL, TRAJ_ATOL


@pytest.mark.parametrize("dense", [False, True])
def test_random_missions_tick_by_tick(oracle, dense):
    """dense=True packs the agents so tightly that many QPs are infeasible: the failure verdicts (status 1, stale
    trajectory kept) must be the oracle's too."""
    import lsc_planner_amd as L
    from lsc_planner_amd.planner import PlannerConfig, next_state_host
    rng = np.random.default_rng(123 + int(dense))
    agent_ticks = failures = 0
    for trial in range(24):
        n = int(rng.integers(8, 28)) if dense else int(rng.integers(4, 40))
        side = float(rng.uniform(1.0, 1.8)) if dense else float(rng.uniform(2.5, 6.0))
        ms = L.random_swarm(n, world=(-side, -side, 0, side, side, 2.5), seed=int(rng.integers(1, 1 << 30)),
                            min_sep=0.33 if dense else 0.5, shrink=0.15 if dense else 0.4)
        if trial % 3 == 0:
            ms.radius[:] = rng.uniform(0.1, 0.25, n)
            ms.downwash[:] = rng.uniform(1.0, 2.5, n)
            ms.max_vel[:] = rng.uniform(0.6, 1.5, (n, 1))
            ms.max_acc[:] = rng.uniform(1.0, 3.0, (n, 1))
        mode = "prior_based" if trial % 2 else "static"
        pl = L.SwarmPlanner(ms, PlannerConfig(goal_mode=mode))
        prm = oracle.make_params(world_min=ms.world_min, world_max=ms.world_max, obs_f32=True)
        sw = oracle.Swarm(prm, ms.radius, ms.downwash, ms.max_vel, ms.max_acc, ms.nominal_velocity)
        state = np.zeros((n, 9), np.float32)
        state[:, :3] = ms.start
        traj = np.zeros((n, 3, 30), np.float32)
        stale = np.zeros_like(traj)               # TrajOptimizer::trajectory of every agent (kept on failure)
        for tick in range(1, 49):
            g = pl.plan(state, ms.goal, traj)
            goals = ms.goal
            if mode == "prior_based":                              # the oracle's own goals feed the oracle; the GPU's must equal them
                goals = oracle.goal_prior_based(state, ms.goal, traj, tick)
                assert np.array_equal(pl.last_goals(), goals), (trial, n, tick)
            sw.stale[:] = stale
            o = sw.tick(state, goals, traj, tick, want_lsc=False, nthreads=16)
            assert np.array_equal(g["status"], o["status"]), (trial, n, tick)
            ok = o["status"] == 0
            failures += int((~ok).sum())
            assert (np.abs(g["cost"] - o["cost"])[ok] <= COST_RTOL * np.abs(o["cost"])[ok] + COST_ATOL).all(), (trial, tick)
            assert np.abs(g["traj"] - o["traj"]).max() <= TRAJ_ATOL, (trial, tick)
            stale = np.where(ok[:, None, None], g["traj"], stale).astype(np.float32)
            traj = g["traj"]
            state = next_state_host(traj)
            agent_ticks += n
        pl.close()
    assert agent_ticks > 15000
    assert (failures > 100) == dense


def test_random_octomap_worlds_full_tick(oracle):
    """Octomap worlds in the reference's default goal mode, every stage of the tick chained over time on random maps:
    goals (grid A* + line of sight) and corridor boxes bit-exact, QP statuses equal and plans within tolerance.  The
    oracle tick is fed the goals it computed itself, so a goal mismatch cannot hide behind the later stages."""
    import lsc_planner_amd as L
    from lsc_planner_amd.planner import PlannerConfig, next_state_host
    rng = np.random.default_rng(77)
    wmin, wmax = (-4, -4, 0), (4, 4, 2.0)
    res = 0.1
    kmin = np.array([np.floor(wmin[k] / res) + 32768 for k in range(3)], np.int32)
    dims = [int(np.floor(wmax[k] / res) + 32768 - kmin[k] + 1) for k in range(3)]
    for trial in range(4):
        # blocky random obstacles, then a genuine distance field of them (so that corridors and rays make sense)
        coarse = rng.random((dims[0] // 5 + 1, dims[1] // 5 + 1, dims[2] // 5 + 1)) < 0.06
        occ = np.kron(coarse, np.ones((5, 5, 5), bool))[:dims[0], :dims[1], :dims[2]]
        occ[:, :, :2] = False
        idx = np.argwhere(occ)
        leaves = np.concatenate([idx + kmin, np.ones((len(idx), 1), int)], 1).astype(np.int32)
        dm = oracle.DistMap(leaves, res, wmin, wmax)
        n = 10
        ms = L.random_swarm(n, world=wmin + wmax, seed=300 + trial, edt=dm.dist, edt_key_min=dm.key_min, min_clearance=0.5)
        pl = L.SwarmPlanner(ms, PlannerConfig(use_octomap=True, goal_mode="prior_based"))
        pl.set_distmap(dm.dist, dm.key_min, res)
        prm = oracle.make_params(world_min=ms.world_min, world_max=ms.world_max, use_sfc=True, obs_f32=True)
        sw = oracle.Swarm(prm, ms.radius, ms.downwash, ms.max_vel, ms.max_acc, ms.nominal_velocity)
        sw.set_distmap(dm)
        state = np.zeros((n, 9), np.float32)
        state[:, :3] = ms.start
        traj = np.zeros((n, 3, 30), np.float32)
        stale = np.zeros_like(traj)
        for tick in range(1, 31):
            goals_ref = oracle.goal_prior_based_map(prm, dm, state, ms.goal, traj, tick, ms.radius, ms.downwash)
            g = pl.plan(state, ms.goal, traj, want_constraints=True)
            assert np.array_equal(pl.last_goals(), goals_ref), (trial, tick)
            sw.stale[:] = stale
            o = sw.tick(state, goals_ref, traj, tick, want_lsc=False, nthreads=8)
            assert np.array_equal(g["sfc"], o["sfc"]), (trial, tick)
            assert np.array_equal(g["status"], o["status"]), (trial, tick, g["status"], o["status"])
            ok = o["status"] == 0
            assert (np.abs(g["cost"] - o["cost"])[ok] <= COST_RTOL * np.abs(o["cost"])[ok] + COST_ATOL).all(), (trial, tick)
            assert np.abs(g["traj"] - o["traj"]).max() <= TRAJ_ATOL, (trial, tick)
            stale = np.where(ok[:, None, None], g["traj"], stale).astype(np.float32)
            traj = g["traj"]
            state = next_state_host(traj)
        pl.close()
