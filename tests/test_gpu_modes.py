"""-m gpu: the reference's alternate planner modes through the C ABI against the oracle
(oracle/lsc_oracle_modes.c, whose QPs are pinned to HiGHS in tests/test_oracle_pins.py):

  * mode/planner = bvc      TrajPlanner::generateBVC, position prediction, no stop-at-horizon rows, N_constraint_segments
  * SlackMode               collision_constraint / dynamical_limit slack variables in populatebyrow
  * disturbance reset       obstaclePredictionCheck / initialTrajPlanningCheck, the persistent slack set, corridor re-init

Constraint dumps (normals, margins) bit-exact; statuses equal; cost / control points within the tolerances of
tests/test_gpu_parity.py (these QPs are solved by the dense general kernel, csrc/lsc_general.hip).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tolerances import COST_ATOL, COST_RTOL, TRAJ_ATOL


@pytest.fixture(scope="module")
def L():
    import lsc_planner_amd as L
    L.load_library()
    return L


def _swarm_ex(O, ms, modes, **prm_kw):
    prm = O.make_params(world_min=ms.world_min, world_max=ms.world_max, obs_f32=True, **prm_kw)
    return prm, O.SwarmEx(prm, modes, ms.radius, ms.downwash, ms.max_vel, ms.max_acc, ms.nominal_velocity)


def _compare(g, o, tick, lsc=True):
    if lsc:
        assert np.array_equal(g["normal"], o["normal"]), tick
        assert np.array_equal(g["d"], o["d"]), tick
    assert np.array_equal(g["status"], o["status"]), (tick, g["status"], o["status"])
    ok = o["status"] == 0
    assert (np.abs(g["cost"] - o["cost"])[ok] <= COST_RTOL * np.abs(o["cost"])[ok] + COST_ATOL).all(), (tick, g["cost"], o["cost"])
    assert np.abs(g["traj"] - o["traj"]).max() <= TRAJ_ATOL, tick


def _run(L, O, ms, cfg_kw, modes, n_ticks, gust=None, goal_mode="static"):
    """Chained ticks GPU vs oracle; gust = {tick: (agent, offset)} moves an agent off its plan before that tick."""
    from lsc_planner_amd.planner import next_state_host
    N = ms.qn
    pl = L.SwarmPlanner(ms, L.PlannerConfig(goal_mode=goal_mode, **cfg_kw))
    prm, sw = _swarm_ex(O, ms, modes)
    state = np.zeros((N, 9), np.float32)
    state[:, :3] = ms.start
    traj = np.zeros((N, 3, 30), np.float32)
    stale = np.zeros_like(traj)
    for tick in range(1, n_ticks + 1):
        if gust and tick in gust:
            q, off = gust[tick]
            state[q, :3] += np.asarray(off, np.float32)
        own = sw.disturbance_update(state, traj, tick)
        goals = sw.goal_prior_based(state, ms.goal, traj, tick, own_reset=own) if goal_mode == "prior_based" else ms.goal
        g = pl.plan(state, ms.goal, traj, want_constraints=True)
        if goal_mode == "prior_based":
            assert np.array_equal(pl.last_goals(), goals), tick
        sw.stale[:] = stale
        o = sw.tick(state, goals, traj, tick, want_lsc=True, nthreads=8)
        _compare(g, o, tick)
        ok = o["status"] == 0
        stale = np.where(ok[:, None, None], g["traj"], stale).astype(np.float32)
        traj = g["traj"]
        state = next_state_host(traj)
    pl.close()
    return state, sw


def test_bvc_mode_matches_the_oracle(L, oracle):
    ms = L.circle_swap(10, 1.6, world=(-5, -5, 0, 5, 5, 2.5))
    state, _ = _run(L, oracle, ms, dict(planner_mode="bvc"), oracle.make_modes(planner="bvc"), 25)
    p = state[:, :3].astype(np.float64).copy(); p[:, 2] /= 2
    D = np.linalg.norm(p[:, None] - p[None], axis=2) + np.eye(10) * 9
    assert D.min() >= 0.3 - 1e-4                                         # buffered Voronoi cells keep the agents apart


def test_bvc_with_fewer_constraint_segments_and_heterogeneous_agents(L, oracle):
    rng = np.random.default_rng(3)
    ms = L.random_swarm(9, world=(-3, -3, 0, 3, 3, 2.5), seed=12)
    ms.radius[:] = rng.uniform(0.1, 0.25, 9)
    ms.downwash[:] = rng.uniform(1.0, 2.5, 9)
    _run(L, oracle, ms, dict(planner_mode="bvc", n_constraint_segments=2), oracle.make_modes(planner="bvc", n_constraint_segments=2), 15)


@pytest.mark.parametrize("slack", ["collision_constraint", "dynamical_limit"])
def test_slack_modes_match_the_oracle(L, oracle, slack):
    """SlackMode is a BVC matter: in LSC mode the reference fixes it to none (TrajPlanner::checkPlannerMode,
    src/traj_planner.cpp:445-448), so the combination the reference can reach is mode/planner = bvc with slack variables."""
    ms = L.circle_swap(8, 1.2, world=(-5, -5, 0, 5, 5, 2.5))
    _run(L, oracle, ms, dict(planner_mode="bvc", slack_mode=slack), oracle.make_modes(planner="bvc", slack=slack), 20)


def test_lsc_mode_fixes_the_slack_mode_to_none(L):
    """`LSC does not need slack variables, fix to none` (src/traj_planner.cpp:445-448): asking for a slack mode in LSC mode
    must plan exactly what LSC without slack plans -- on the fast path -- and say so."""
    from lsc_planner_amd.planner import next_state_host
    ms = L.circle_swap(8, 1.2, world=(-5, -5, 0, 5, 5, 2.5))
    a = L.SwarmPlanner(ms, L.PlannerConfig(slack_mode="collision_constraint"))
    assert b"slack_mode fixed to none" in a.L.lsc_last_note(a.ctx) and a.L.lsc_last_error(a.ctx) == b""      # a remark, not an error
    b = L.SwarmPlanner(ms)
    state = np.zeros((8, 9), np.float32); state[:, :3] = ms.start
    traj = np.zeros((8, 3, 30), np.float32)
    for tick in range(1, 9):
        ga, gb = a.plan(state, ms.goal, traj), b.plan(state, ms.goal, traj)
        for k in ("traj", "cost", "status", "iters"):
            assert np.array_equal(ga[k], gb[k]), (tick, k)
        traj = gb["traj"]; state = next_state_host(traj)
    a.close(); b.close()


def test_slack_variables_keep_an_otherwise_infeasible_swarm_planning(L, oracle):
    """The scene of the reference's log/QPmodel.lp (agent 3's QP is infeasible under the hard LSC rows -- CPLEX failed
    on it, HiGHS certifies it).  Where the reference puts slack variables on collision rows (mode/planner = bvc,
    SlackMode::COLLISIONCONSTRAINT) the same swarm keeps planning tick after tick, and the GPU agrees with the oracle."""
    import json, os
    from conftest import GOLDEN
    sc = json.load(open(os.path.join(GOLDEN, "qpmodel_lp.json")))["scene"]
    starts = np.array(sc["starts_xy_z07"], np.float32)
    N = len(starts)
    goal = starts.copy(); goal[:, :2] *= -0.5
    ms = L.Mission(starts, goal, np.asarray(sc["world"][:3], np.float32), np.asarray(sc["world"][3:], np.float32),
                   np.full(N, sc["radius"]), np.full(N, sc["downwash"]), np.tile(sc["max_vel"], (N, 1)).astype(float),
                   np.tile(sc["max_acc"], (N, 1)).astype(float), np.full(N, sc["nominal_velocity"]))
    hard = L.SwarmPlanner(ms)
    state = np.zeros((N, 9), np.float32); state[:, :3] = starts
    g = hard.plan(state, goal, np.zeros((N, 3, 30), np.float32))
    hard.close()
    assert g["status"][sc["agent"]] == 1
    from lsc_planner_amd.planner import next_state_host
    pl = L.SwarmPlanner(ms, L.PlannerConfig(planner_mode="bvc", slack_mode="collision_constraint"))
    prm, sw = _swarm_ex(oracle, ms, oracle.make_modes(planner="bvc", slack="collision_constraint"))
    traj = np.zeros((N, 3, 30), np.float32)
    for tick in range(1, 7):
        g = pl.plan(state, goal, traj, want_constraints=True)
        o = sw.tick(state, goal, traj, tick, want_lsc=True, nthreads=8)
        assert (o["status"] == 0).all()
        _compare(g, o, tick)
        traj = g["traj"]; state = next_state_host(traj)
    pl.close()


@pytest.mark.parametrize("goal_mode", ["static", "prior_based"])
def test_disturbance_reset_and_its_persistent_slack_rows(L, oracle, goal_mode):
    """Two gusts push agents off their plans (0.3 m > reset_threshold 0.15): prediction / initial trajectory are reset to
    the current position, the disturbed agents enter everybody's slack set and stay there; before the first gust the swarm
    runs on the fast path, afterwards on the general kernel -- every tick against the oracle."""
    ms = L.circle_swap(8, 1.5, world=(-5, -5, 0, 5, 5, 2.5))
    gust = {6: (2, (0.3, 0.1, 0.0)), 11: (5, (-0.2, 0.25, 0.05))}
    state, sw = _run(L, oracle, ms, dict(reset_threshold=0.15), oracle.make_modes(reset_threshold=0.15), 18, gust=gust,
                     goal_mode=goal_mode)
    assert sw.slack_set.sum() == 2 * 7 + 2 * 7 - 2              # rows of agents 2 and 5 (all others) + their columns, overlap counted once


def test_checks_on_but_nobody_disturbed_is_the_fast_path_bit_for_bit(L):
    """reset_threshold > 0 without disturbances must not change a single bit (and must not leave status 6 behind)."""
    from lsc_planner_amd.planner import next_state_host
    ms = L.circle_swap(16, 2.0)
    a, b = L.SwarmPlanner(ms, L.PlannerConfig(goal_mode="prior_based")), L.SwarmPlanner(ms, L.PlannerConfig(goal_mode="prior_based", reset_threshold=0.15))
    state = np.zeros((16, 9), np.float32); state[:, :3] = ms.start
    traj = np.zeros((16, 3, 30), np.float32)
    for _ in range(20):
        ga, gb = a.plan(state, ms.goal, traj), b.plan(state, ms.goal, traj)
        for k in ("traj", "cost", "status", "iters"):
            assert np.array_equal(ga[k], gb[k]), k
        traj = ga["traj"]; state = next_state_host(traj)
    a.close(); b.close()


def test_device_resident_ticks_take_the_general_kernel_after_a_disturbance(L, oracle):
    """lsc_tick_device does not see the states on the host: with the checks on it launches the general kernel every tick
    (its workgroups leave at once when nobody is flagged).  Same gust, same plans as the host-buffer ticks."""
    import torch
    from lsc_planner_amd.planner import next_state_host
    ms = L.circle_swap(8, 1.5, world=(-5, -5, 0, 5, 5, 2.5))
    N = 8
    cfg = L.PlannerConfig(reset_threshold=0.15)
    h, d = L.SwarmPlanner(ms, cfg), L.SwarmPlanner(ms, cfg)
    dev = torch.device("cuda", 0)
    state = np.zeros((N, 9), np.float32); state[:, :3] = ms.start
    traj = np.zeros((N, 3, 30), np.float32)
    goal = torch.from_numpy(ms.goal).to(dev)
    cost = torch.zeros(N, dtype=torch.float64, device=dev)
    status = torch.zeros(N, dtype=torch.int32, device=dev)
    iters = torch.zeros(N, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for tick in range(1, 13):
        if tick == 5:
            state[3, :3] += np.float32([0.25, -0.2, 0.0])
        g = h.plan(state, ms.goal, traj)
        s_d = torch.from_numpy(state).to(dev)
        a_d = torch.from_numpy(traj.reshape(N, 90)).to(dev)
        b_d = torch.zeros((N, 90), device=dev)
        d.tick_device(s_d, goal, a_d, b_d, cost, status, iters, tick, st)
        torch.cuda.synchronize()
        assert np.array_equal(b_d.cpu().numpy().reshape(N, 3, 30), g["traj"]), tick
        assert np.array_equal(status.cpu().numpy(), g["status"]) and (g["status"] == 0).all(), tick
        traj = g["traj"]; state = next_state_host(traj)
    h.close(); d.close()


def test_host_tick_after_device_ticks_that_saw_a_disturbance(L):
    """A context may mix the two kinds of tick.  The device-resident ticks flag an off-plan agent on the device only; a later
    host-buffer tick on the same context must still launch the alternate-mode kernel for the agents whose rows carry slack
    variables (the host mirror of the flags is refreshed), and must never hand internal status 6 to the caller.  Reference
    for the result: a context that ran every tick through the host-buffer entry point."""
    import torch
    from lsc_planner_amd.planner import next_state_host
    ms = L.circle_swap(8, 1.5, world=(-5, -5, 0, 5, 5, 2.5))
    N = 8
    cfg = L.PlannerConfig(reset_threshold=0.15)
    h, m = L.SwarmPlanner(ms, cfg), L.SwarmPlanner(ms, cfg)
    dev = torch.device("cuda", 0)
    state = np.zeros((N, 9), np.float32); state[:, :3] = ms.start
    traj = np.zeros((N, 3, 30), np.float32)
    goal = torch.from_numpy(ms.goal).to(dev)
    cost = torch.zeros(N, dtype=torch.float64, device=dev)
    status = torch.zeros(N, dtype=torch.int32, device=dev)
    iters = torch.zeros(N, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for tick in range(1, 13):
        if tick == 5:
            state[3, :3] += np.float32([0.25, -0.2, 0.0])      # the gust arrives during the device-resident stretch
        g = h.plan(state, ms.goal, traj)
        if tick < 8:
            b_d = torch.zeros((N, 90), device=dev)
            m.planner_seq += 1
            m.tick_device(torch.from_numpy(state).to(dev), goal, torch.from_numpy(traj.reshape(N, 90)).to(dev), b_d, cost, status, iters, tick, st)
            torch.cuda.synchronize()
            got, gst = b_d.cpu().numpy().reshape(N, 3, 30), status.cpu().numpy()
        else:
            r = m.plan(state, ms.goal, traj)                    # no disturbance in THIS tick's inputs: the mirror must know about tick 5
            got, gst = r["traj"], r["status"]
        assert np.array_equal(gst, g["status"]) and (gst == 0).all(), (tick, gst, g["status"])
        assert np.array_equal(got, g["traj"]), tick
        traj = g["traj"]; state = next_state_host(traj)
    h.close(); m.close()


def test_disturbance_reinitialises_the_corridor_on_octomap_worlds(L, oracle):
    """initialTrajPlanningCheck sets flag_initialize_sfc again (src/traj_planner.cpp:1059): all five boxes restart from the
    current position.  Static goals (the grid search with slack obstacles is covered on the empty map)."""
    from maputil import forest_leaves
    from lsc_planner_amd.planner import next_state_host
    leaves, res = forest_leaves()
    world = (-5, -5, 0, 5, 5, 2.5)
    dm = oracle.DistMap(leaves, res, world[:3], world[3:])
    ms = L.random_swarm(10, world=world, seed=31, edt=dm.dist, edt_key_min=dm.key_min)
    N = ms.qn
    pl = L.SwarmPlanner(ms, L.PlannerConfig(use_octomap=True, reset_threshold=0.15))
    pl.set_distmap(dm.dist, dm.key_min, res)
    modes = oracle.make_modes(reset_threshold=0.15)
    prm, sw = _swarm_ex(oracle, ms, modes, use_sfc=True)
    sw.set_distmap(dm)
    state = np.zeros((N, 9), np.float32); state[:, :3] = ms.start
    traj = np.zeros((N, 3, 30), np.float32)
    stale = np.zeros_like(traj)
    for tick in range(1, 13):
        if tick == 6:
            state[4, :3] += np.float32([0.2, 0.2, 0.0])
        g = pl.plan(state, ms.goal, traj, want_constraints=True)
        sw.stale[:] = stale
        o = sw.tick(state, ms.goal, traj, tick, want_lsc=True, nthreads=8)
        assert np.array_equal(g["sfc"], o["sfc"]), tick
        if tick == 6:
            assert (g["sfc"][4] == g["sfc"][4][0]).all()              # five copies of the fresh box
        _compare(g, o, tick)
        ok = o["status"] == 0
        stale = np.where(ok[:, None, None], g["traj"], stale).astype(np.float32)
        traj = g["traj"]; state = next_state_host(traj)
    pl.close()


def test_disturbance_on_an_octomap_world_in_the_default_goal_mode(L, oracle):
    """Everything at once: octomap world, mode/goal = prior_based (grid A*: slack obstacles are stamped into the grid as
    higher priority by decree and take no part in the retreat rule, src/traj_planner.cpp:548-551; the line-of-sight goal
    starts from the reset initial trajectory), corridor re-initialisation, slack rows -- goals and boxes bit-exact."""
    from maputil import forest_leaves
    from lsc_planner_amd.planner import next_state_host
    leaves, res = forest_leaves()
    world = (-5, -5, 0, 5, 5, 2.5)
    dm = oracle.DistMap(leaves, res, world[:3], world[3:])
    ms = L.random_swarm(10, world=world, seed=8, edt=dm.dist, edt_key_min=dm.key_min)
    N = ms.qn
    pl = L.SwarmPlanner(ms, L.PlannerConfig(use_octomap=True, goal_mode="prior_based", reset_threshold=0.15))
    pl.set_distmap(dm.dist, dm.key_min, res)
    prm, sw = _swarm_ex(oracle, ms, oracle.make_modes(reset_threshold=0.15), use_sfc=True)
    sw.set_distmap(dm)
    state = np.zeros((N, 9), np.float32); state[:, :3] = ms.start
    traj = np.zeros((N, 3, 30), np.float32)
    stale = np.zeros_like(traj)
    for tick in range(1, 15):
        if tick in (5, 9):
            state[tick % N, :3] += np.float32([0.2, -0.15, 0.0])
        own = sw.disturbance_update(state, traj, tick)
        goals = oracle.goal_prior_based_map(prm, dm, state, ms.goal, traj, tick, ms.radius, ms.downwash, slack_set=sw.slack_set,
                                            own_reset=own)
        g = pl.plan(state, ms.goal, traj, want_constraints=True)
        assert np.array_equal(pl.last_goals(), goals), tick
        sw.stale[:] = stale
        o = sw.tick(state, goals, traj, tick, want_lsc=True, nthreads=8)
        assert np.array_equal(g["sfc"], o["sfc"]), tick
        _compare(g, o, tick)
        ok = o["status"] == 0
        stale = np.where(ok[:, None, None], g["traj"], stale).astype(np.float32)
        traj = g["traj"]; state = next_state_host(traj)
    assert sw.slack_set.any()
    pl.close()
