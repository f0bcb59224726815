"""-m gpu: seeded fuzzing of the tick through the C ABI against the oracle -- tiny swarms with extreme parameters (radii,
downwash, dynamic limits, goals outside the world, nearly coincident agents, agents already at their goal, moving first
ticks), in the default LSC mode and in the alternate modes (BVC, both slack modes, N_constraint_segments, disturbance
reset with a gust).  Equal statuses every tick (about a fifth of the LSC-mode QPs are infeasible), cost within 1e-6
relative, control points within 5e-5 m.

What this found when it was first run (round 2): two alternate-mode QPs in 7.9 k on which the ORACLE gave up -- its normal
equations lost definiteness close to a degenerate optimum and it reported "infeasible" -- while the kernel returned the
optimum HiGHS confirms (1.36186419 and 106.741176).  The oracle now shifts the diagonal and goes on
(oracle/lsc_oracle.c, chol_factor failure branch); the two instances are fixtures in tests/golden/fuzz_found_*.npz."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

from tolerances import COST_ATOL, COST_RTOL, FUZZ_PLAN_COMPARED_BELOW_COST, FUZZ_TRAJ_ATOL as TRAJ_ATOL
# with a slack penalty of 1e5 a grossly violated limit makes |f| ~ 1e7; both solvers stop on criteria relative to |f|, so
# beyond this cost only the cost is compared (it still pins the optimum: the QP is strictly convex)
TRAJ_COST_LIMIT = FUZZ_PLAN_COMPARED_BELOW_COST


def _check(g, o, where):
    assert np.array_equal(g["status"], o["status"]), (where, g["status"], o["status"])
    assert np.isfinite(g["traj"]).all(), where
    ok = o["status"] == 0
    assert (np.abs(g["cost"] - o["cost"])[ok] <= COST_RTOL * np.abs(o["cost"])[ok] + COST_ATOL).all(), (where, g["cost"], o["cost"])
    tame = ~ok | (np.abs(o["cost"]) < TRAJ_COST_LIMIT)
    assert np.abs(g["traj"] - o["traj"])[tame].max(initial=0.0) <= TRAJ_ATOL, where
    return ok


@pytest.mark.parametrize("seed0,trials,options", [(1000, 150, {}), (70000, 40, {"warm_start_mu": 0.0}), (71000, 40, {"prune": 0}),
                                                  (72000, 40, {"prune": 2}), (73000, 40, {"max_rows_per_cp": 3})])
def test_fuzz_lsc_mode(oracle, seed0, trials, options):
    """options: the solver's cold start only / no pruning / box-only pruning / an LDS row capacity of three rows per control
    point (most agents then take the second pass with their rows in HBM)."""
    import lsc_planner_amd as L
    from lsc_planner_amd.mission import Mission
    from lsc_planner_amd.planner import PlannerConfig, next_state_host
    agent_ticks = failures = 0
    for trial in range(trials):
        rng = np.random.default_rng(seed0 + trial)
        n = int(rng.integers(1, 14))
        side, top = float(rng.uniform(0.8, 6.0)), float(rng.uniform(0.6, 3.0))
        wmin, wmax = np.array([-side, -side, 0], np.float32), np.array([side, side, top], np.float32)
        kind = int(rng.integers(0, 4))
        start = rng.uniform(wmin + 0.05, wmax - 0.05, (n, 3)).astype(np.float32)
        goal = rng.uniform(wmin - 0.3, wmax + 0.3, (n, 3)).astype(np.float32)      # some goals outside the world
        if kind == 1 and n > 1:
            start[1] = start[0] + np.float32(1e-3)                                   # nearly coincident agents
        if kind == 2:
            goal[:] = start                                                          # already there
        radius, dw = rng.uniform(0.05, 0.4, n), rng.uniform(1.0, 3.0, n)
        vmax, amax = np.repeat(rng.uniform(0.2, 3.0, (n, 1)), 3, 1), np.repeat(rng.uniform(0.5, 6.0, (n, 1)), 3, 1)
        if kind == 3:
            vmax[:, 2] *= 0.3
            amax[:, 2] *= 0.5
        vnom = rng.uniform(0.3, 2.0, n)
        ms = Mission(start, goal, wmin, wmax, radius, dw, vmax, amax, vnom, name="fuzz")
        mode = "prior_based" if trial % 2 else "static"
        pl = L.SwarmPlanner(ms, PlannerConfig(goal_mode=mode, **options))
        sw = oracle.Swarm(oracle.make_params(world_min=wmin, world_max=wmax, obs_f32=True), radius, dw, vmax, amax, vnom)
        state = np.zeros((n, 9), np.float32)
        state[:, :3] = start
        if trial % 3 == 0:
            state[:, 3:6] = rng.uniform(-0.5, 0.5, (n, 3)).astype(np.float32)        # moving first tick
        traj = np.zeros((n, 3, 30), np.float32)
        stale = np.zeros_like(traj)
        for tick in range(1, 9):
            g = pl.plan(state, goal, traj)
            goals = goal
            if mode == "prior_based":                              # the oracle's own goals feed the oracle; the GPU's must equal them
                goals = oracle.goal_prior_based(state, goal, traj, tick)
                assert np.array_equal(pl.last_goals(), goals), (trial, n, kind, tick)
            sw.stale[:] = stale
            o = sw.tick(state, goals, traj, tick, want_lsc=False, nthreads=8)
            ok = _check(g, o, (trial, n, kind, mode, tick))
            failures += int((~ok).sum())
            agent_ticks += n
            stale = np.where(ok[:, None, None], g["traj"], stale).astype(np.float32)
            traj = g["traj"]
            state = next_state_host(traj)
        pl.close()
    assert agent_ticks > 45 * trials and failures > 6 * trials, (agent_ticks, failures)


MODES = [(dict(planner_mode="bvc"), dict(planner="bvc")),
         (dict(planner_mode="bvc", slack_mode="collision_constraint"), dict(planner="bvc", slack="collision_constraint")),   # (LSC fixes the slack
         (dict(planner_mode="bvc", slack_mode="dynamical_limit"), dict(planner="bvc", slack="dynamical_limit")),             #  mode to none)
         (dict(planner_mode="bvc", n_constraint_segments=2), dict(planner="bvc", n_constraint_segments=2)),
         (dict(reset_threshold=0.15), dict(reset_threshold=0.15))]


def test_fuzz_alternate_modes(oracle):
    import lsc_planner_amd as L
    from lsc_planner_amd.mission import Mission
    from lsc_planner_amd.planner import PlannerConfig, next_state_host
    agent_ticks = 0
    for trial in range(150):
        rng = np.random.default_rng(5000 + trial)
        n = int(rng.integers(2, 10))
        side, top = float(rng.uniform(1.0, 4.0)), float(rng.uniform(1.0, 3.0))
        wmin, wmax = np.array([-side, -side, 0], np.float32), np.array([side, side, top], np.float32)
        while True:                                                                  # BVC needs distinct positions
            start = rng.uniform(wmin + 0.2, wmax - 0.2, (n, 3)).astype(np.float32)
            if (np.linalg.norm(start[:, None] - start[None], axis=2) + np.eye(n) * 9).min() > 0.45:
                break
        goal = rng.uniform(wmin + 0.1, wmax - 0.1, (n, 3)).astype(np.float32)
        radius, dw = rng.uniform(0.08, 0.2, n), rng.uniform(1.0, 2.5, n)
        vmax, amax = np.repeat(rng.uniform(0.4, 2.0, (n, 1)), 3, 1), np.repeat(rng.uniform(1.0, 4.0, (n, 1)), 3, 1)
        vnom = rng.uniform(0.5, 1.5, n)
        ms = Mission(start, goal, wmin, wmax, radius, dw, vmax, amax, vnom, name="fuzz")
        ck, mk = MODES[trial % len(MODES)]
        pl = L.SwarmPlanner(ms, PlannerConfig(goal_mode="static", **ck))
        sw = oracle.SwarmEx(oracle.make_params(world_min=wmin, world_max=wmax, obs_f32=True), oracle.make_modes(**mk),
                            radius, dw, vmax, amax, vnom)
        state = np.zeros((n, 9), np.float32)
        state[:, :3] = start
        traj = np.zeros((n, 3, 30), np.float32)
        stale = np.zeros_like(traj)
        gust_tick = int(rng.integers(3, 7)) if "reset_threshold" in ck else -1
        for tick in range(1, 11):
            if tick == gust_tick:
                state[int(rng.integers(0, n)), :3] += rng.uniform(-0.4, 0.4, 3).astype(np.float32)
            sw.disturbance_update(state, traj, tick)
            g = pl.plan(state, goal, traj)
            sw.stale[:] = stale
            o = sw.tick(state, goal, traj, tick, want_lsc=False, nthreads=8)
            ok = _check(g, o, (trial, n, ck, tick))
            agent_ticks += n
            stale = np.where(ok[:, None, None], g["traj"], stale).astype(np.float32)
            traj = g["traj"]
            state = next_state_host(traj)
        pl.close()
    assert agent_ticks > 7000, agent_ticks


def test_fuzz_octomap_worlds(oracle):
    """Random block worlds (block size, density, world size), random grid resolution / margin of the goal planner, every
    fourth one planar (world/dimension = 2): goals and corridor boxes bit-exact, statuses equal, plans within tolerance,
    20 chained ticks each."""
    import lsc_planner_amd as L
    from lsc_planner_amd.planner import PlannerConfig, next_state_host
    agent_ticks = 0
    for trial in range(24):
        rng = np.random.default_rng(100 + trial)
        side, top = float(rng.choice([3.0, 4.0, 5.0])), float(rng.choice([1.5, 2.0, 2.5]))
        wmin, wmax = (-side, -side, 0), (side, side, top)
        res = 0.1
        kmin = np.array([np.floor(wmin[k] / res) + 32768 for k in range(3)], np.int32)
        dims = [int(np.floor(wmax[k] / res) + 32768 - kmin[k] + 1) for k in range(3)]
        bs, dens = int(rng.choice([3, 5, 7])), float(rng.uniform(0.03, 0.10))
        coarse = rng.random((dims[0] // bs + 1, dims[1] // bs + 1, dims[2] // bs + 1)) < dens
        occ = np.kron(coarse, np.ones((bs, bs, bs), bool))[:dims[0], :dims[1], :dims[2]]
        occ[:, :, :2] = False
        idx = np.argwhere(occ)
        if len(idx) == 0:
            continue
        leaves = np.concatenate([idx + kmin, np.ones((len(idx), 1), int)], 1).astype(np.int32)
        dm = oracle.DistMap(leaves, res, wmin, wmax)
        n = int(rng.integers(2, 14))
        planar = trial % 4 == 3
        box = (wmin[0], wmin[1], 0.2, wmax[0], wmax[1], 1.2) if planar else wmin + wmax      # a slab that shrinks to z = 0.7
        ms = L.random_swarm(n, world=box, seed=int(rng.integers(1, 1 << 30)), edt=dm.dist, edt_key_min=dm.key_min, min_clearance=0.5)
        ms.world_min, ms.world_max = np.asarray(wmin, np.float32), np.asarray(wmax, np.float32)
        gm, gr = float(rng.choice([0.05, 0.1, 0.2])), float(rng.choice([0.25, 0.3, 0.5]))
        kw = dict(world_dimension=2, world_z_2d=0.7) if planar else {}
        pl = L.SwarmPlanner(ms, PlannerConfig(use_octomap=True, goal_mode="prior_based", grid_margin=gm, grid_resolution=gr, **kw))
        pl.set_distmap(dm.dist, dm.key_min, res)
        prm = oracle.make_params(world_min=ms.world_min, world_max=ms.world_max, use_sfc=True, obs_f32=True, **kw)
        sw = oracle.Swarm(prm, ms.radius, ms.downwash, ms.max_vel, ms.max_acc, ms.nominal_velocity)
        sw.set_distmap(dm)
        state = np.zeros((n, 9), np.float32)
        state[:, :3] = ms.start
        traj = np.zeros((n, 3, 30), np.float32)
        stale = np.zeros_like(traj)
        for tick in range(1, 21):
            goals_ref = oracle.goal_prior_based_map(prm, dm, state, ms.goal, traj, tick, ms.radius, ms.downwash, grid_margin=gm, grid_res=gr)
            g = pl.plan(state, ms.goal, traj)
            where = (trial, n, planar, gm, gr, tick)
            assert (g["status"] < 4).all(), where
            assert np.array_equal(pl.last_goals(), goals_ref), where
            sw.stale[:] = stale
            o = sw.tick(state, goals_ref, traj, tick, want_lsc=False, nthreads=8)
            assert np.array_equal(g["sfc"], o["sfc"]), where
            ok = _check(g, o, where)
            agent_ticks += n
            stale = np.where(ok[:, None, None], g["traj"], stale).astype(np.float32)
            traj = g["traj"]
            state = next_state_host(traj)
        pl.close()
    assert agent_ticks > 2500, agent_ticks


def test_fuzz_configuration_values(oracle):
    """Parameters the shipped launch files never vary: segment time dt (horizon = 5 dt), the two cost weights, and the three
    thresholds of the goal rule (goal_threshold, priority_dist_threshold, goal_radius) -- statuses and plans against the oracle,
    goals bit for bit."""
    import lsc_planner_amd as L
    from lsc_planner_amd.planner import PlannerConfig, next_state_host
    for dt in (0.1, 0.16, 0.25, 0.3):
        for wc, wt in ((0.01, 1.0), (0.1, 5.0)):
            ms = L.circle_swap(8, 2.5, world=(-5, -5, 0, 5, 5, 2.5))
            pl = L.SwarmPlanner(ms, PlannerConfig(goal_mode="prior_based", dt=dt, horizon=5 * dt, control_input_weight=wc, terminal_weight=wt))
            prm = oracle.make_params(dt=dt, w_control=wc, w_terminal=wt, world_min=ms.world_min, world_max=ms.world_max, obs_f32=True)
            sw = oracle.Swarm(prm, ms.radius, ms.downwash, ms.max_vel, ms.max_acc, ms.nominal_velocity)
            state = np.zeros((8, 9), np.float32)
            state[:, :3] = ms.start
            traj = np.zeros((8, 3, 30), np.float32)
            stale = np.zeros_like(traj)
            for tick in range(1, 21):
                g = pl.plan(state, ms.goal, traj)
                goals = oracle.goal_prior_based(state, ms.goal, traj, tick, dt=dt)
                assert np.array_equal(pl.last_goals(), goals), (dt, tick)
                sw.stale[:] = stale
                o = sw.tick(state, goals, traj, tick, want_lsc=False, nthreads=8)
                ok = _check(g, o, (dt, wc, wt, tick))
                stale = np.where(ok[:, None, None], g["traj"], stale).astype(np.float32)
                traj = g["traj"]
                state = next_state_host(traj, dt=dt)
            pl.close()
    rng = np.random.default_rng(3)
    for trial in range(30):
        gt, pd, gr = float(rng.choice([0.05, 0.1, 0.3])), float(rng.choice([0.2, 0.4, 0.8, 1.5])), float(rng.choice([0.5, 1.0, 2.0, 4.0]))
        n = int(rng.integers(3, 16))
        ms = L.random_swarm(n, world=(-3, -3, 0, 3, 3, 2.5), seed=int(rng.integers(1, 1 << 30)), min_sep=0.5, shrink=0.3)
        pl = L.SwarmPlanner(ms, PlannerConfig(goal_mode="prior_based", goal_threshold=gt, priority_dist_threshold=pd, goal_radius=gr))
        state = np.zeros((n, 9), np.float32)
        state[:, :3] = ms.start
        traj = np.zeros((n, 3, 30), np.float32)
        for tick in range(1, 21):
            ref = oracle.goal_prior_based(state, ms.goal, traj, tick, goal_threshold=gt, priority_dist_threshold=pd, goal_radius=gr)
            g = pl.plan(state, ms.goal, traj)
            assert np.array_equal(pl.last_goals(), ref), (trial, tick, gt, pd, gr)
            traj = g["traj"]
            state = next_state_host(traj)
        pl.close()


def test_fuzz_device_resident_chain_equals_host_buffer_chain():
    """lsc_tick_device_fused (what bench.py times: everything stays on the device, one launch per tick, the next ideal state
    computed in the kernel) chained over ten ticks against lsc_replan_tick + host propagation on the same random missions, with
    and without the disturbance checks and with a three-row LDS capacity: plans, states, statuses and costs bit for bit."""
    import torch
    import lsc_planner_amd as L
    from lsc_planner_amd.mission import Mission
    from lsc_planner_amd.planner import PlannerConfig, next_state_host
    dev = torch.device("cuda", 0)
    agent_ticks = 0
    for trial in range(40):
        rng = np.random.default_rng(90000 + trial)
        n = int(rng.integers(1, 40))
        side, top = float(rng.uniform(1.0, 6.0)), float(rng.uniform(0.8, 3.0))
        wmin, wmax = np.array([-side, -side, 0], np.float32), np.array([side, side, top], np.float32)
        start = rng.uniform(wmin + 0.05, wmax - 0.05, (n, 3)).astype(np.float32)
        goal = rng.uniform(wmin - 0.2, wmax + 0.2, (n, 3)).astype(np.float32)
        radius, dw = rng.uniform(0.05, 0.3, n), rng.uniform(1.0, 3.0, n)
        vmax, amax = np.repeat(rng.uniform(0.3, 2.5, (n, 1)), 3, 1), np.repeat(rng.uniform(0.5, 5.0, (n, 1)), 3, 1)
        vnom = rng.uniform(0.3, 2.0, n)
        ms = Mission(start, goal, wmin, wmax, radius, dw, vmax, amax, vnom, name="fuzz")
        cfg = dict(goal_mode="prior_based" if trial % 2 else "static", reset_threshold=0.15 if trial % 3 == 0 else 0.0,
                   max_rows_per_cp=3 if trial % 5 == 0 else 0)
        host, devp = L.SwarmPlanner(ms, PlannerConfig(**cfg)), L.SwarmPlanner(ms, PlannerConfig(**cfg))
        state = np.zeros((n, 9), np.float32)
        state[:, :3] = start
        traj = np.zeros((n, 3, 30), np.float32)
        f32 = dict(dtype=torch.float32, device=dev)
        st_d = [torch.tensor(state, device=dev), torch.zeros((n, 9), **f32)]
        tj_d = [torch.zeros((n, 90), **f32), torch.zeros((n, 90), **f32)]
        goal_d = torch.tensor(goal, device=dev)
        cost = torch.zeros(n, dtype=torch.float64, device=dev)
        status = torch.zeros(n, dtype=torch.int32, device=dev)
        iters = torch.zeros(n, dtype=torch.int32, device=dev)
        stream = torch.cuda.current_stream().cuda_stream
        for tick in range(1, 11):
            g = host.plan(state, goal, traj)
            devp.tick_device_fused(st_d[0], goal_d, tj_d[0], tj_d[1], st_d[1], cost, status, iters, tick, stream)
            torch.cuda.synchronize()
            where = (trial, n, cfg, tick)
            state = next_state_host(g["traj"])
            assert np.array_equal(tj_d[1].cpu().numpy().reshape(n, 3, 30), g["traj"]), where
            assert np.array_equal(status.cpu().numpy(), g["status"]), where
            assert np.array_equal(st_d[1].cpu().numpy(), state), where
            ok = g["status"] == 0
            assert np.array_equal(cost.cpu().numpy()[ok], g["cost"][ok]), where
            traj = g["traj"]
            st_d.reverse()
            tj_d.reverse()
            agent_ticks += n
        host.close()
        devp.close()
    assert agent_ticks > 5000


@pytest.mark.parametrize("variant,seed0", [("planar", 81000), ("m4", 82000), ("planar_m4", 83000)])
def test_fuzz_planar_worlds_and_four_segments(oracle, variant, seed0):
    """The round-4 builds under the same generator (tests/fuzz_variants.py): the planar 60-variable QP (world_dimension 2), the M = 4
    library against the M = 4 oracle, and both at once; statuses, goals, cost and plans as in the LSC-mode fuzz above."""
    import lsc_planner_amd as L
    from fuzz_variants import run_variant
    agent_ticks, failures, bad = run_variant(L, oracle, seed0, 50, variant)
    assert not bad, bad[:3]
    # (until round 5 a planar trial ended when an agent whose first QP failed was left out of the plane; its stale plan now starts in the plane)
    assert agent_ticks > 45 * 50 and failures > 40, (agent_ticks, failures)
