"""-m gpu: the HIP path (through the C ABI) against the oracle and the committed golden fixtures.

Tolerances (stated once): LSC normals / margins are integer-like geometry -> bit-exact.  QP optimum: the oracle
and the kernel are two different interior-point codes converging to the same unique optimum, so
  cost: |dcost| <= 1e-6 * |cost|   (north_star allows 1e-3),   control points: |dx| <= 2e-5 m (float32 storage).
"""
import numpy as np
import pytest

from conftest import golden_mission, oracle_swarm

pytestmark = pytest.mark.gpu

from tolerances import COST_ATOL, COST_RTOL, TRAJ_ATOL


@pytest.fixture(scope="module")
def L():
    import lsc_planner_amd as L
    L.load_library()          # fails loudly if liblsc_hip.so is missing on a GPU box
    return L


def test_gjk_kernel_bitwise_vs_reference_golden(L, gjk_golden):
    pl = L.SwarmPlanner(L.circle_swap(4, 1.0))
    v, d = pl.gjk_batch(gjk_golden["pts"].astype(np.float64))
    assert np.array_equal(d, gjk_golden["dist"])
    assert np.array_equal(v, gjk_golden["v"])
    pl.close()


def test_gjk_kernel_bitwise_vs_oracle_random(L, oracle):
    rng = np.random.default_rng(99)
    pts = (rng.normal(size=(20000, 6, 3)) * rng.uniform(0.05, 3.0, size=(20000, 1, 1))
           + rng.normal(size=(20000, 1, 3))).astype(np.float32).astype(np.float64)
    pl = L.SwarmPlanner(L.circle_swap(4, 1.0))
    v, d = pl.gjk_batch(pts)
    for i in range(0, 20000, 7):
        do, vo, _, _ = oracle.gjk_origin(pts[i])
        assert do == d[i] and np.array_equal(vo, v[i])
    pl.close()


@pytest.mark.parametrize("name,keep", [("multi_simple4", (1, 2, 3, 20)), ("multi_circle20", (1, 15))])
@pytest.mark.parametrize("prune", [True, False])
def test_tick_matches_golden_snapshots(L, ticks, name, keep, prune):
    ms = golden_mission(ticks, name)
    for tick in keep:
        pl = L.SwarmPlanner(ms, L.PlannerConfig(prune=prune))
        pl.planner_seq = tick - 1
        g = pl.plan(ticks[f"{name}/tick{tick}/state"], ms.goal, ticks[f"{name}/tick{tick}/prev"], want_constraints=True)
        assert (g["status"] == 0).all()
        assert np.array_equal(g["normal"], ticks[f"{name}/tick{tick}/normal"])
        assert np.array_equal(g["d"], ticks[f"{name}/tick{tick}/d"])
        ref_cost = ticks[f"{name}/tick{tick}/cost"]
        assert (np.abs(g["cost"] - ref_cost) <= COST_RTOL * np.abs(ref_cost)).all()
        assert np.abs(g["traj"] - ticks[f"{name}/tick{tick}/traj"]).max() <= TRAJ_ATOL
        pl.close()


def _run_vs_oracle(L, O, ms, n_ticks, prune=True, every=1):
    from lsc_planner_amd.planner import next_state_host
    N = ms.qn
    pl = L.SwarmPlanner(ms, L.PlannerConfig(prune=prune))
    sw = oracle_swarm(O, ms)
    state = np.zeros((N, 9), np.float32); state[:, :3] = ms.start
    traj = np.zeros((N, 3, 30), np.float32)
    for tick in range(1, n_ticks + 1):
        check = tick % every == 0 or tick <= 2
        g = pl.plan(state, ms.goal, traj, want_constraints=check)
        if check:
            sw.stale[:] = traj
            o = sw.tick(state, ms.goal, traj, tick, want_lsc=True, nthreads=8)
            assert np.array_equal(g["normal"], o["normal"]), tick
            assert np.array_equal(g["d"], o["d"]), tick
            assert np.array_equal(g["status"], o["status"]), tick
            ok = o["status"] == 0
            assert (np.abs(g["cost"] - o["cost"])[ok] <= COST_RTOL * np.abs(o["cost"])[ok]).all(), tick
            assert np.abs(g["traj"] - o["traj"]).max() <= TRAJ_ATOL, tick
        traj = g["traj"]
        state = next_state_host(traj)
    pl.close()


def test_circle20_mission_ticks_vs_oracle(L, oracle):
    _run_vs_oracle(L, oracle, L.circle_swap(20, 8.0), 40, prune=True, every=4)


def test_circle64_headline_config_vs_oracle(L, oracle):
    """BASELINE configs[2]: the 64-agent circle swap; a few ticks checked against the oracle at full size."""
    _run_vs_oracle(L, oracle, L.circle_swap(64, 8.0), 24, prune=True, every=8)


def test_pruned_and_unpruned_rows_give_the_same_plan(L):
    """Dropping provably redundant LSC rows must not move the optimum."""
    from lsc_planner_amd.planner import next_state_host
    ms = L.circle_swap(32, 4.0)
    a, b = L.SwarmPlanner(ms, L.PlannerConfig(prune=True)), L.SwarmPlanner(ms, L.PlannerConfig(prune=False))
    state = np.zeros((32, 9), np.float32); state[:, :3] = ms.start
    traj = np.zeros((32, 3, 30), np.float32)
    for tick in range(1, 31):
        ga, gb = a.plan(state, ms.goal, traj), b.plan(state, ms.goal, traj)
        assert (ga["status"] == 0).all() and (gb["status"] == 0).all()
        assert (np.abs(ga["cost"] - gb["cost"]) <= COST_RTOL * np.abs(gb["cost"])).all()
        assert np.abs(ga["traj"] - gb["traj"]).max() <= TRAJ_ATOL
        assert (b.row_counts() == 27 * 31).all() and (a.row_counts() <= 27 * 31).all()
        traj = gb["traj"]
        state = next_state_host(traj)
    assert a.row_counts().mean() < 0.7 * 27 * 31
    a.close(); b.close()


def test_infeasible_qp_keeps_stale_trajectory_like_the_reference(L, oracle):
    """The scene of log/QPmodel.lp: agent 3's QP is infeasible; the reference swallows the failure and keeps the
    optimiser's previous trajectory (zeros on the first tick), src/traj_planner.cpp:1553-1584."""
    import json, os
    from conftest import GOLDEN
    sc = json.load(open(os.path.join(GOLDEN, "qpmodel_lp.json")))["scene"]
    starts = np.array(sc["starts_xy_z07"], np.float32)
    N = len(starts)
    goal = starts.copy(); goal[:, :2] *= -0.5
    ms = L.Mission(starts, goal, np.asarray(sc["world"][:3], np.float32), np.asarray(sc["world"][3:], np.float32),
                   np.full(N, sc["radius"]), np.full(N, sc["downwash"]), np.tile(sc["max_vel"], (N, 1)).astype(float),
                   np.tile(sc["max_acc"], (N, 1)).astype(float), np.full(N, sc["nominal_velocity"]))
    pl = L.SwarmPlanner(ms)
    sw = oracle_swarm(oracle, ms)
    state = np.zeros((N, 9), np.float32); state[:, :3] = starts
    g = pl.plan(state, goal, np.zeros((N, 3, 30), np.float32))
    o = sw.tick(state, goal, np.zeros((N, 3, 30), np.float32), 1)
    assert np.array_equal(g["status"], o["status"]) and g["status"][sc["agent"]] == 1
    bad = g["status"] != 0
    assert (g["traj"][bad] == 0).all()
    assert np.abs(g["traj"] - o["traj"]).max() <= TRAJ_ATOL
    pl.close()


def test_first_tick_with_moving_agents_uses_float32_constant_velocity_model(L, oracle):
    """planner_seq < 2: prediction = pos + vel * m_intp * dt in float32 (src/traj_planner.cpp:699-712) -- needs
    unfused float32 arithmetic on the device to stay bit-exact."""
    rng = np.random.default_rng(8)
    ms = L.circle_swap(16, 3.0)
    state = np.zeros((16, 9), np.float32)
    state[:, :3] = ms.start
    state[:, 3:6] = rng.normal(size=(16, 3)).astype(np.float32) * np.float32(0.3)
    state[:, 6:9] = rng.normal(size=(16, 3)).astype(np.float32) * np.float32(0.2)
    pl = L.SwarmPlanner(ms)
    sw = oracle_swarm(oracle, ms)
    g = pl.plan(state, ms.goal, np.zeros((16, 3, 30), np.float32), want_constraints=True)
    o = sw.tick(state, ms.goal, np.zeros((16, 3, 30), np.float32), 1, want_lsc=True)
    assert np.array_equal(g["normal"], o["normal"]) and np.array_equal(g["d"], o["d"])
    assert np.array_equal(g["status"], o["status"])
    ok = o["status"] == 0
    assert (np.abs(g["cost"] - o["cost"])[ok] <= COST_RTOL * np.abs(o["cost"])[ok]).all()
    assert np.abs(g["traj"] - o["traj"]).max() <= TRAJ_ATOL
    pl.close()


def test_sfc_boxes_and_ticks_match_oracle_on_forest(L, oracle):
    """use_octomap: corridor boxes bit-exact vs the oracle's literal restatement of corridor_constructor.hpp, and the
    QP with SFC rows within the stated tolerances (BASELINE configs[3] map, smaller swarm)."""
    from maputil import forest_leaves
    from lsc_planner_amd.planner import next_state_host
    leaves, res = forest_leaves()
    dm = oracle.DistMap(leaves, res, [-5, -5, 0], [5, 5, 2.5])
    ms = L.random_swarm(48, world=(-5, -5, 0, 5, 5, 2.5), seed=11, edt=dm.dist, edt_key_min=dm.key_min)
    N = ms.qn
    pl = L.SwarmPlanner(ms, L.PlannerConfig(use_octomap=True))
    pl.set_distmap(dm.dist, dm.key_min, res)
    prm = oracle.make_params(world_min=ms.world_min, world_max=ms.world_max, use_sfc=True, obs_f32=True)
    sw = oracle.Swarm(prm, ms.radius, ms.downwash, ms.max_vel, ms.max_acc, ms.nominal_velocity)
    sw.set_distmap(dm)
    state = np.zeros((N, 9), np.float32); state[:, :3] = ms.start
    traj = np.zeros((N, 3, 30), np.float32)
    for tick in range(1, 13):
        g = pl.plan(state, ms.goal, traj, want_constraints=True)
        sw.stale[:] = traj if tick > 1 else 0
        o = sw.tick(state, ms.goal, traj, tick, want_lsc=True, nthreads=8)
        assert np.array_equal(g["sfc"], o["sfc"]), tick
        assert np.array_equal(g["normal"], o["normal"]) and np.array_equal(g["d"], o["d"])
        assert np.array_equal(g["status"], o["status"]), (tick, g["status"], o["status"])
        ok = o["status"] == 0
        assert ok.sum() >= N - 4
        assert (np.abs(g["cost"] - o["cost"])[ok] <= COST_RTOL * np.abs(o["cost"])[ok]).all()
        assert np.abs(g["traj"][ok] - o["traj"][ok]).max() <= TRAJ_ATOL
        traj = g["traj"]
        state = next_state_host(traj)
    pl.close()


def test_sfc_box_growth_bitwise_on_random_seeds(L, oracle):
    """Many independent expandBoxFromPoint calls (first-tick boxes) on the forest map: GPU integral-image growth ==
    oracle's per-lattice-point loops, including seeds that are rejected (status 4)."""
    from maputil import forest_leaves
    leaves, res = forest_leaves()
    dm = oracle.DistMap(leaves, res, [-5, -5, 0], [5, 5, 2.5])
    rng = np.random.default_rng(5)
    N = 256
    start = rng.uniform([-4.8, -4.8, 0.1], [4.8, 4.8, 2.4], size=(N, 3)).astype(np.float32)
    goal = rng.uniform([-4.8, -4.8, 0.1], [4.8, 4.8, 2.4], size=(N, 3)).astype(np.float32)
    ms = L.Mission(start, goal, np.array([-5, -5, 0], np.float32), np.array([5, 5, 2.5], np.float32), np.full(N, 0.15),
                   np.full(N, 2.0), np.ones((N, 3)), np.full((N, 3), 2.0), np.ones(N))
    pl = L.SwarmPlanner(ms, L.PlannerConfig(use_octomap=True))
    pl.set_distmap(dm.dist, dm.key_min, res)
    state = np.zeros((N, 9), np.float32); state[:, :3] = start
    g = pl.plan(state, goal, np.zeros((N, 3, 30), np.float32))
    prm = oracle.make_params(world_min=ms.world_min, world_max=ms.world_max, use_sfc=True, obs_f32=True)
    n_blocked = 0
    for q in range(N):
        rc, box = dm.expand_box(prm, start[q], goal[q], 0.15)
        if rc:
            n_blocked += 1
            assert g["status"][q] == 4
        else:
            assert np.array_equal(g["sfc"][q, 0], box.astype(np.float32)), (q, g["sfc"][q, 0], box)
            assert np.array_equal(g["sfc"][q, 4], g["sfc"][q, 0])
    assert 0 < n_blocked < N // 2
    pl.close()


def test_heterogeneous_agents_match_oracle(L, oracle):
    """Per-agent radius / downwash / velocity and acceleration limits / nominal speed (Mission::agents, src/mission.cpp:60-130):
    the pair downwash, the float32-rounded obstacle radius and the per-axis limits must all follow the agent."""
    from lsc_planner_amd.planner import next_state_host
    rng = np.random.default_rng(21)
    N = 18
    base = L.circle_swap(N, 3.0, world=(-6, -6, 0, 6, 6, 3.0))
    ms = L.Mission(base.start, base.goal, base.world_min, base.world_max,
                   rng.uniform(0.10, 0.22, N), rng.uniform(1.5, 2.5, N), rng.uniform(0.8, 1.5, (N, 3)), rng.uniform(1.0, 2.5, (N, 3)),
                   rng.uniform(0.7, 1.3, N))
    pl = L.SwarmPlanner(ms)
    sw = oracle_swarm(oracle, ms)
    state = np.zeros((N, 9), np.float32); state[:, :3] = ms.start
    traj = np.zeros((N, 3, 30), np.float32)
    for tick in range(1, 21):
        g = pl.plan(state, ms.goal, traj, want_constraints=True)
        sw.stale[:] = traj if tick > 1 else 0
        o = sw.tick(state, ms.goal, traj, tick, want_lsc=True, nthreads=8)
        assert np.array_equal(g["normal"], o["normal"]) and np.array_equal(g["d"], o["d"]), tick
        assert np.array_equal(g["status"], o["status"]), tick
        ok = o["status"] == 0
        assert (np.abs(g["cost"] - o["cost"])[ok] <= COST_RTOL * np.abs(o["cost"])[ok]).all(), tick
        assert np.abs(g["traj"][ok] - o["traj"][ok]).max() <= TRAJ_ATOL, tick
        traj = g["traj"]
        state = next_state_host(traj)
    pl.close()


def test_heterogeneous_radii_use_their_own_corridor_margin(L, oracle):
    """SFC margin = r + res/2 - 1e-5 per agent: two radius classes -> two blocked-cell integral images."""
    from maputil import forest_leaves
    leaves, res = forest_leaves()
    dm = oracle.DistMap(leaves, res, [-5, -5, 0], [5, 5, 2.5])
    rng = np.random.default_rng(9)
    N = 64
    start = rng.uniform([-4.5, -4.5, 0.3], [4.5, 4.5, 2.2], size=(N, 3)).astype(np.float32)
    goal = rng.uniform([-4.5, -4.5, 0.3], [4.5, 4.5, 2.2], size=(N, 3)).astype(np.float32)
    radius = np.where(np.arange(N) % 2 == 0, 0.15, 0.25)
    ms = L.Mission(start, goal, np.array([-5, -5, 0], np.float32), np.array([5, 5, 2.5], np.float32), radius, np.full(N, 2.0),
                   np.ones((N, 3)), np.full((N, 3), 2.0), np.ones(N))
    pl = L.SwarmPlanner(ms, L.PlannerConfig(use_octomap=True))
    pl.set_distmap(dm.dist, dm.key_min, res)
    state = np.zeros((N, 9), np.float32); state[:, :3] = start
    g = pl.plan(state, goal, np.zeros((N, 3, 30), np.float32))
    prm = oracle.make_params(world_min=ms.world_min, world_max=ms.world_max, use_sfc=True, obs_f32=True)
    differ = 0
    for q in range(N):
        rc, box = dm.expand_box(prm, start[q], goal[q], radius[q])
        if rc:
            assert g["status"][q] == 4
            continue
        assert np.array_equal(g["sfc"][q, 0], box.astype(np.float32)), q
        rc2, box2 = dm.expand_box(prm, start[q], goal[q], 0.4 - radius[q])
        differ += int(rc2 != 0 or not np.array_equal(box, box2))
    assert differ > 5           # the two margins really give different corridors on this map
    pl.close()


@pytest.mark.parametrize("solver,noise", [("interior_point", 0.0), ("active_set", 0.02)])
def test_prior_based_goal_planning_bitwise_and_mission_completes(L, oracle, solver, noise):
    """mode/goal = prior_based (the reference's default) on the empty map: device goals == oracle restatement of
    goalPlanningWithPriority bit for bit, ticks match, and the 20-agent circle swap reaches its goals without collision
    (with static goals it deadlocks in the centre).  The unperturbed circle is perfectly symmetric: it is flown by the interior point,
    whose 1e-6 m of agent-dependent noise breaks the ties of the priority rule; the active-set solve (exact optimum: the symmetry would
    survive, and two agents end in an infeasible QP on which the oracle agrees) flies it with the reference's goal noise
    (multisim/max_noise, src/mission.cpp:386-395)."""
    from lsc_planner_amd.planner import next_state_host
    ms = L.circle_swap(20, 8.0)
    N = 20
    if noise > 0:
        ms.goal[:, :3] += np.random.default_rng(20).uniform(0, noise, (N, 3)).astype(np.float32)
    pl = L.SwarmPlanner(ms, L.PlannerConfig(goal_mode="prior_based", solver=solver))
    sw = oracle_swarm(oracle, ms)
    state = np.zeros((N, 9), np.float32); state[:, :3] = ms.start
    traj = np.zeros((N, 3, 30), np.float32)
    saw_retreat = False
    for tick in range(1, 260):
        g = pl.plan(state, ms.goal, traj)
        goals = pl.last_goals()
        og = oracle.goal_prior_based(state, ms.goal, traj, tick)
        assert np.array_equal(goals, og), tick
        saw_retreat |= bool((np.linalg.norm(goals - state[:, :3], axis=1) < 0.6).any() and (np.linalg.norm(state[:, :3] - ms.goal, axis=1) > 1).all())
        if tick % 10 == 0 or tick < 4:
            sw.stale[:] = traj if tick > 1 else 0
            o = sw.tick(state, og, traj, tick, nthreads=8)
            assert np.array_equal(g["status"], o["status"])
            assert (np.abs(g["cost"] - o["cost"]) <= COST_RTOL * np.abs(o["cost"]) + COST_ATOL).all(), (tick, g["cost"], o["cost"])   # costs -> 0 near the goal: absolute floor
            assert np.abs(g["traj"] - o["traj"]).max() <= TRAJ_ATOL, tick
        assert (g["status"] == 0).all()
        traj = g["traj"]
        state = next_state_host(traj)
        p = state[:, :3].astype(np.float64).copy(); p[:, 2] /= 2.0
        D = np.linalg.norm(p[:, None] - p[None], axis=2) + np.eye(N) * 9
        assert D.min() >= 0.3 - 1e-4
        if np.linalg.norm(state[:, :3] - ms.goal, axis=1).max() < 0.1:
            break
    assert tick < 220, "mission did not finish"
    pl.close()


@pytest.mark.parametrize("name,n_ticks", [("multi_simple4", 120), ("multi_circle20", 240)])
def test_symmetric_missions_under_the_default_solver_follow_the_oracle_tick_by_tick(L, oracle, ticks, name, n_ticks):
    """BASELINE configs[0] / configs[1] in closed loop under the DEFAULT solver (the exact active-set solve) with multisim/max_noise = 0 as
    launch/testall_empty.launch:47 sets it, held to the (exact) oracle at EVERY tick: goals bit for bit (goalPlanningWithPriority,
    src/traj_planner.cpp:540-608), statuses equal, costs and plans within the tolerance table.  Both missions are PERFECTLY symmetric
    (agent i and its partner are mirror images), the exact optimum keeps them so, and the strict comparisons of the priority rule
    (:566-577) tie for good.  The documented outcome -- the oracle's as much as the kernel's, so a property of the reference's rule under
    an exact QP solver, not a kernel artefact:
      multi_simple4 : agents 0 and 1 -- head-on along the x axis -- stop 1.15 m from their goals (tick ~19) while the diagonal pair arrives,
                      and every plan's end point stands still from tick ~50 on;
      multi_circle20: agents' QPs turn infeasible in the crowd (tick ~205), two of them (2 and 18, mirror images) never recover --
                      an agent keeps its stale plan on a failure (src/traj_planner.cpp:1548-1585) -- and whoever has to pass them waits.
    lsc_sim reports both and applies the reference's remedy, multisim/max_noise, to the first (tests/test_gpu_sim.py); profiles/
    r06_closed_loop_default_solver.log is this loop's print-out (tests/closed_loop_default.py)."""
    from lsc_planner_amd.planner import next_state_host
    ms = golden_mission(ticks, name)
    N = ms.qn
    pl = L.SwarmPlanner(ms, L.PlannerConfig(goal_mode="prior_based", solver="active_set"))   # the library's default solver (named: LSC_SOLVER re-runs this file under the other one)
    sw = oracle_swarm(oracle, ms)
    state = np.zeros((N, 9), np.float32); state[:, :3] = ms.start
    traj = np.zeros((N, 3, 30), np.float32)
    still = 0
    failed_ticks = np.zeros(N, int)
    for tick in range(1, n_ticks + 1):
        g = pl.plan(state, ms.goal, traj)
        og = oracle.goal_prior_based(state, ms.goal, traj, tick)
        assert np.array_equal(pl.last_goals(), og), tick
        sw.stale[:] = traj if tick > 1 else 0
        o = sw.tick(state, og, traj, tick, nthreads=8)
        assert np.array_equal(g["status"], o["status"]), (tick, g["status"], o["status"])
        ok = g["status"] == 0
        assert (np.abs(g["cost"] - o["cost"])[ok] <= (COST_RTOL * np.abs(o["cost"]) + COST_ATOL)[ok]).all(), tick
        assert np.abs(g["traj"] - o["traj"]).max() <= TRAJ_ATOL, tick              # (a failed agent keeps its stale plan in both)
        still = still + 1 if np.abs(g["traj"][:, :, 29] - traj[:, :, 29]).max() < 1e-5 else 0
        failed_ticks = np.where(g["status"] == 1, failed_ticks + 1, 0)
        traj = g["traj"]
        state = next_state_host(traj)
        p = state[:, :3].astype(np.float64).copy(); p[:, 2] /= 2.0
        D = np.linalg.norm(p[:, None] - p[None], axis=2) + np.eye(N) * 9
        assert D.min() >= 0.3 - 1e-4, tick                                          # nobody collides while waiting
    dist = np.linalg.norm(state[:, :3] - ms.goal, axis=1)
    assert dist.max() > 0.1, "the mission finished: the symmetry was broken somewhere"
    if name == "multi_simple4":
        assert (g["status"] == 0).all() and still >= 50 and np.abs(dist[:2] - 1.15).max() < 1e-3 and dist[2:].max() < 1e-3, (still, dist)
    else:
        assert np.nonzero(failed_ticks >= 20)[0].tolist() == [2, 18], failed_ticks
        assert (dist[[2, 18]] > 8.0).all(), dist
    pl.close()
