"""-m gpu: goal planning with a distance field (mode/goal = prior_based on an octomap world) through the C ABI against
the oracle: the grid path (cell by cell -- this is where the reference's hash-order tie-breaking shows), the flags and
the resulting current_goal_position, bit for bit."""
import numpy as np
import pytest

from conftest import oracle_swarm

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    import lsc_planner_amd as L
    L.load_library()
    return L


def _compare_goal_stage(L, O, ms, dm, dist, key_min, res, state, traj, tick, pl, grid_margin=0.2, **world):
    prm = O.make_params(world_min=ms.world_min, world_max=ms.world_max, obs_f32=True, **world)
    ref, paths, flags = O.goal_prior_based_map(prm, dm, state, ms.goal, traj, tick, ms.radius, ms.downwash, grid_margin=grid_margin,
                                               want_paths=True)
    g = pl.plan(state, ms.goal, traj)
    tr = pl.goal_trace()
    assert (g["status"] != 5).all(), "goal planner capacity"
    for qi in range(ms.qn):
        assert tr["flags"][qi] == flags[qi], (tick, qi, tr["flags"][qi], flags[qi])
        if not (flags[qi] & 1):
            assert np.array_equal(tr["paths"][qi], paths[qi]), (tick, qi, len(tr["paths"][qi]), len(paths[qi]))
    assert np.array_equal(pl.last_goals(), ref), tick
    return g


def test_forest_goal_planning_bitwise_over_a_mission(L, oracle):
    from maputil import forest_leaves
    from lsc_planner_amd.planner import next_state_host
    leaves, res = forest_leaves()
    wmin, wmax = (-5, -5, 0), (5, 5, 2.5)
    dm = oracle.DistMap(leaves, res, wmin, wmax)
    ms = L.random_swarm(24, world=wmin + wmax, seed=4, edt=dm.dist, edt_key_min=dm.key_min)
    pl = L.SwarmPlanner(ms, L.PlannerConfig(use_octomap=True, goal_mode="prior_based"))
    pl.set_distmap(dm.dist, dm.key_min, res)
    pl.set_goal_trace(512)
    N = ms.qn
    state = np.zeros((N, 9), np.float32)
    state[:, :3] = ms.start
    traj = np.zeros((N, 3, 30), np.float32)
    longest, most = 0, 0
    for tick in range(1, 41):
        g = _compare_goal_stage(L, oracle, ms, dm, dm.dist, dm.key_min, res, state, traj, tick, pl)
        tr = pl.goal_trace()
        longest = max(longest, int(tr["path_len"].max()))
        most = max(most, int(tr["expansions"].max()))
        traj = g["traj"]
        state = next_state_host(traj)
    pl.close()
    assert longest >= 20 and most >= 500, (longest, most)      # real searches happened
    assert np.linalg.norm(state[:, :3] - ms.goal, axis=1).mean() < np.linalg.norm(ms.start - ms.goal, axis=1).mean()


def test_planar_world_goal_planning(L, oracle):
    """world/dimension = 2 (src/grid_based_planner.cpp:82-85, 127-133, 199-215; src/mission.cpp:88-112): the planning grid is
    the single layer z = world/z_2d, starts, goals and the stamped higher-priority agents sit in it whatever their height is,
    and the search has no vertical moves.  Paths, flags and goals against the oracle, bit for bit.  (The QP of a planar world has 60 variables: tests/test_gpu_round4.py.)"""
    from maputil import forest_leaves
    from lsc_planner_amd.planner import next_state_host
    leaves, res = forest_leaves()
    wmin, wmax = (-5, -5, 0), (5, 5, 2.5)
    z2d = 0.7
    dm = oracle.DistMap(leaves, res, wmin, wmax)
    ms = L.random_swarm(20, world=wmin + wmax, seed=12, edt=dm.dist, edt_key_min=dm.key_min)
    ms.start[:, 2] = ms.goal[:, 2] = np.float32(z2d)          # what Mission::initialize does with world/dimension = 2
    pl = L.SwarmPlanner(ms, L.PlannerConfig(use_octomap=True, goal_mode="prior_based", world_dimension=2, world_z_2d=z2d))
    pl.set_distmap(dm.dist, dm.key_min, res)
    pl.set_goal_trace(512)
    N = ms.qn
    state = np.zeros((N, 9), np.float32)
    state[:, :3] = ms.start
    traj = np.zeros((N, 3, 30), np.float32)
    longest = 0
    for tick in range(1, 31):
        g = _compare_goal_stage(L, oracle, ms, dm, dm.dist, dm.key_min, res, state, traj, tick, pl, world_dimension=2, world_z_2d=z2d)
        tr = pl.goal_trace()
        assert tr["grid_dims"][2] == 1 and tr["grid_min"][2] == z2d
        for path in tr["paths"]:
            assert (path[:, 2] == 0).all()
        longest = max(longest, int(tr["path_len"].max()))
        traj = g["traj"]
        state = next_state_host(traj)
    pl.close()
    assert longest >= 15, longest
    assert np.linalg.norm(state[:, :3] - ms.goal, axis=1).mean() < np.linalg.norm(ms.start - ms.goal, axis=1).mean()


def test_random_mazes_tie_breaking(L, oracle):
    """Synthetic 'distance fields' (0 inside random blocks, 1 m elsewhere) make dense mazes with many equal-cost paths:
    the path must still be the reference's, cell for cell."""
    rng = np.random.default_rng(11)
    wmin, wmax = (-4, -4, 0), (4, 4, 2.0)
    res = 0.1
    for trial in range(6):
        kmin = np.array([np.floor(wmin[k] / res) + 32768 for k in range(3)], np.int32)
        dims = [int(np.floor(wmax[k] / res) + 32768 - kmin[k] + 1) for k in range(3)]
        coarse = rng.random((dims[0] // 3 + 1, dims[1] // 3 + 1, dims[2] // 3 + 1)) < (0.12 + 0.05 * (trial % 3))
        dist = np.where(np.kron(coarse, np.ones((3, 3, 3), bool))[:dims[0], :dims[1], :dims[2]], 0.0, 1.0).astype(np.float32)
        dm = oracle.DistMap.from_array(dist, kmin, res)
        n = 12
        ms = L.random_swarm(n, world=wmin + wmax, seed=100 + trial, edt=dist, edt_key_min=kmin, min_clearance=0.5)
        pl = L.SwarmPlanner(ms, L.PlannerConfig(use_octomap=True, goal_mode="prior_based", grid_margin=0.05))
        pl.set_distmap(dist, kmin, res)
        pl.set_goal_trace(1024)
        state = np.zeros((n, 9), np.float32)
        state[:, :3] = ms.start
        traj = np.zeros((n, 3, 30), np.float32)
        _compare_goal_stage(L, oracle, ms, dm, dist, kmin, res, state, traj, 1, pl, grid_margin=0.05)
        pl.close()


def test_open_row_overflow_is_reported_as_status_5(L):
    """An OPEN row that outgrows its LDS capacity must surface as LSC_STATUS_GOAL_CAPACITY (stale trajectory kept), never
    as a silently different goal.  The capacity is forced down through lsc_config.goal_row_cap."""
    from maputil import forest_leaves, write_bt
    import tempfile, os
    leaves, res = forest_leaves()
    bt = os.path.join(tempfile.mkdtemp(), "f.bt")
    write_bt(bt, leaves, res)
    world = (-5, -5, 0, 5, 5, 2.5)
    dist, kmin, r = L.edt_from_bt(bt, np.asarray(world[:3], np.float32), np.asarray(world[3:], np.float32))
    ms = L.random_swarm(16, world=world, seed=4, edt=dist, edt_key_min=kmin, edt_res=r)
    state = np.zeros((16, 9), np.float32)
    state[:, :3] = ms.start
    traj = np.zeros((16, 3, 30), np.float32)
    ref = L.SwarmPlanner(ms, L.PlannerConfig(use_octomap=True, goal_mode="prior_based"))
    ref.load_octomap(bt)
    g_ref = ref.plan(state, ms.goal, traj)
    goals_ref = ref.last_goals()
    ref.close()
    pl = L.SwarmPlanner(ms, L.PlannerConfig(use_octomap=True, goal_mode="prior_based", goal_row_cap=30))
    pl.load_octomap(bt)
    g = pl.plan(state, ms.goal, traj)
    goals = pl.last_goals()
    pl.close()
    over = g["status"] == 5
    assert over.any() and (~over).any()
    assert (g_ref["status"] == 0).all()
    # agents whose search fitted are unaffected; the others keep their stale trajectory (zeros on the first tick)
    assert np.array_equal(goals[~over], goals_ref[~over])
    assert np.array_equal(g["traj"][~over], g_ref["traj"][~over])
    assert (g["traj"][over] == 0).all()


@pytest.mark.parametrize("search", ["auto", "key64"])
def test_register_resident_search_returns_the_general_search(L, search):
    """The register-resident search emulates the same containers as the general search: goals, paths, flags and the number of expanded nodes must be identical, tick after tick -- on a
    3-D forest (rows up to ~60 entries), on the 2 x 2 tiled forest (67 rows: two bookkeeping slots per lane, rows beyond 64
    entries, rehashes) and with a tiny row capacity (the capacity error must surface in the same agents)."""
    from maputil import forest_leaves, write_bt
    from lsc_planner_amd.planner import next_state_host
    import os
    import sys
    import tempfile
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from config_runs import forest_tiles
    cases = []
    leaves, res = forest_leaves()
    bt1 = os.path.join(tempfile.mkdtemp(), "f.bt")
    write_bt(bt1, leaves, res)
    cases.append((bt1, (-5, -5, 0, 5, 5, 2.5), 40, 4, 12, {}))
    bt2, world2 = forest_tiles(2)
    cases.append((bt2, world2, 96, 7, 6, {}))
    cases.append((bt1, (-5, -5, 0, 5, 5, 2.5), 16, 4, 1, {"goal_row_cap": 30}))
    for bt, world, n, seed, ticks, extra in cases:
        dist, kmin, r = L.edt_from_bt(bt, np.asarray(world[:3], np.float32), np.asarray(world[3:], np.float32))
        ms = L.random_swarm(n, world=world, seed=seed, edt=dist, edt_key_min=kmin, edt_res=r)
        pls = [L.SwarmPlanner(ms, L.PlannerConfig(use_octomap=True, goal_mode="prior_based", goal_search=s, **extra)) for s in ("general", search)]
        for pl in pls:
            pl.load_octomap(bt)
            pl.set_goal_trace(1024)
        state = np.zeros((n, 9), np.float32)
        state[:, :3] = ms.start
        traj = np.zeros((n, 3, 30), np.float32)
        most = 0
        for tick in range(1, ticks + 1):
            out = [pl.plan(state, ms.goal, traj) for pl in pls]
            tr = [pl.goal_trace() for pl in pls]
            assert np.array_equal(out[0]["status"], out[1]["status"]), (search, tick)
            ok = out[0]["status"] != 5
            if extra and tick == 1:
                assert (~ok).any() and ok.any()
            assert np.array_equal(tr[0]["flags"][ok], tr[1]["flags"][ok]), (search, tick)
            assert np.array_equal(tr[0]["expansions"][ok], tr[1]["expansions"][ok]), (search, tick)
            for q in np.nonzero(ok)[0]:
                assert np.array_equal(tr[0]["paths"][q], tr[1]["paths"][q]), (search, tick, q)
            assert np.array_equal(pls[0].last_goals()[ok], pls[1].last_goals()[ok]), (search, tick)
            assert np.array_equal(out[0]["traj"], out[1]["traj"]), (search, tick)
            most = max(most, int(tr[0]["expansions"].max()))
            traj = out[0]["traj"]
            state = next_state_host(traj)
        for pl in pls:
            pl.close()
        assert extra or most >= 1000, most
