"""CPU: the goal planner's search (oracle/lsc_oracle_goal.cpp, which borrows libstdc++'s std::unordered_map) against an
independent model that keeps every OPEN row as an explicit array in the container's iteration order
(tests/astar_model.py -- the same model lsc_goal.hip implements with wave-parallel scans).  Agreement on dense random
grids, where equal-cost ties are everywhere, pins the model of the hash-table order: insertion at the head of the bucket
or of the list, re-insertion in list order on a rehash, bucket counts 13, 29, 59, ..."""
import numpy as np
import pytest

import astar_model


def _random_case(rng):
    dims = (rng.integers(3, 36), rng.integers(3, 36), rng.integers(1, 9))
    occ = (rng.random(dims) < rng.choice([0.0, 0.1, 0.25, 0.35])).astype(np.uint8)
    s = [int(rng.integers(0, d)) for d in dims]
    g = [int(rng.integers(0, d)) for d in dims]
    occ[tuple(s)] = 0
    return occ, s, g


def test_row_order_model_equals_std_unordered_map(oracle):
    rng = np.random.default_rng(20260928)
    big = 0
    for _ in range(60):
        occ, s, g = _random_case(rng)
        ref = oracle.astar(occ, s, g)
        got, nexp = astar_model.astar(occ, s, g)
        assert ref.shape == got.shape and np.array_equal(ref, got)
        big += nexp > 500
    assert big >= 5          # searches large enough to go through several rehashes


def test_the_comparison_is_sensitive_to_the_order_rule(oracle):
    """Same searches with a deliberately wrong rule (always insert at the front): paths must differ somewhere,
    otherwise the test above would not be checking the tie-breaking at all."""
    rng = np.random.default_rng(20260928)
    orig = astar_model.Row._place
    astar_model.Row._place = lambda self, lst, nb, e: lst.insert(0, e)
    try:
        diff = 0
        for _ in range(40):
            occ, s, g = _random_case(rng)
            ref = oracle.astar(occ, s, g)
            got, _ = astar_model.astar(occ, s, g)
            diff += not (ref.shape == got.shape and np.array_equal(ref, got))
    finally:
        astar_model.Row._place = orig
    assert diff > 0


def test_search_properties(oracle):
    rng = np.random.default_rng(5)
    for _ in range(40):
        occ, s, g = _random_case(rng)
        p = oracle.astar(occ, s, g)
        if len(p) == 0:
            continue
        assert tuple(p[0]) == tuple(s)
        assert p[-1][0] == g[0] and p[-1][1] == g[1]            # the goal test ignores the altitude (isearch.cpp:74)
        assert (np.abs(np.diff(p, axis=0)).sum(1) == 1).all()    # axis moves only
        assert not occ[p[:, 0], p[:, 1], p[:, 2]].any()
        assert len({tuple(c) for c in p}) == len(p)


def test_goal_on_the_forest_map_is_visible_and_within_radius(oracle):
    from maputil import forest_leaves
    import lsc_planner_amd as L
    leaves, res = forest_leaves()
    wmin, wmax = (-5, -5, 0), (5, 5, 2.5)
    dm = oracle.DistMap(leaves, res, wmin, wmax)
    prm = oracle.make_params(world_min=wmin, world_max=wmax, obs_f32=True)
    ms = L.random_swarm(16, world=wmin + wmax, seed=3, edt=dm.dist, edt_key_min=dm.key_min)
    state = np.zeros((16, 9), np.float32)
    state[:, :3] = ms.start
    traj = np.zeros((16, 3, 30), np.float32)
    goals, paths, flags = oracle.goal_prior_based_map(prm, dm, state, ms.goal, traj, 1, ms.radius, ms.downwash, want_paths=True)
    d = np.linalg.norm(goals - ms.start, axis=1)
    assert (d <= 2.0 + 1e-4).all()                              # plan/goal_radius
    assert (d > 0.05).any()
    assert max(len(p) for p in paths) >= 20
    dims, gmin = oracle.grid_dims(prm)
    assert tuple(dims) == (33, 33, 9) and abs(gmin[0] + 4.8) < 1e-9


def test_planar_world_has_one_grid_layer_and_no_vertical_moves(oracle):
    """world/dimension = 2: grid_min[2] = grid_max[2] = world/z_2d, one layer (src/grid_based_planner.cpp:82-88); start and
    goal cells are forced into it (:199-202) even when the agent has drifted off that height."""
    from maputil import forest_leaves
    import lsc_planner_amd as L
    leaves, res = forest_leaves()
    wmin, wmax = (-5, -5, 0), (5, 5, 2.5)
    dm = oracle.DistMap(leaves, res, wmin, wmax)
    prm = oracle.make_params(world_min=wmin, world_max=wmax, obs_f32=True, world_dimension=2, world_z_2d=0.7)
    dims, gmin = oracle.grid_dims(prm)
    assert tuple(dims) == (33, 33, 1) and gmin[2] == 0.7
    ms = L.random_swarm(12, world=wmin + wmax, seed=5, edt=dm.dist, edt_key_min=dm.key_min)
    ms.start[:, 2] = ms.goal[:, 2] = np.float32(0.7)
    state = np.zeros((12, 9), np.float32)
    state[:, :3] = ms.start
    state[::2, 2] += np.float32(0.4)                            # half of the agents are 0.4 m above the plane: same cells
    traj = np.zeros((12, 3, 30), np.float32)
    goals, paths, flags = oracle.goal_prior_based_map(prm, dm, state, ms.goal, traj, 1, ms.radius, ms.downwash, want_paths=True)
    assert max(len(p) for p in paths) >= 10
    for p in paths:
        assert len(p) == 0 or (p[:, 2] == 0).all()


@pytest.mark.parametrize("dims", [(34, 34, 9), (67, 67, 9), (128, 100, 5)])
def test_32_bit_search_keys_order_and_tie_like_the_reference_doubles(dims):
    """The goal search's Key32 (csrc/lsc_abi.cpp: goal_key_table): (steps << rb) + table[d2] must order AND tie exactly
    like F = g + H = 10 steps + 10 sqrt(d2) in the reference's double arithmetic (src/Astar-3D/isearch.cpp) for every step count and
    every squared distance of the grid.  Checked exhaustively: all (steps, d2) sorted by key, F must be non-decreasing along that
    order and equal exactly where the keys are equal.  Host-only hook of the product library: no GPU involved."""
    import ctypes
    from lsc_planner_amd import _lib
    L = _lib.load_library()
    H, W, A = dims
    words = (H - 1) ** 2 + (W - 1) ** 2 + (A - 1) ** 2 + 1
    tab = np.zeros(words, np.uint32)
    rb = ctypes.c_int()
    assert L.lsc_goal_key_table(words, tab.ctypes.data_as(ctypes.POINTER(ctypes.c_uint)), ctypes.byref(rb)) == 0
    steps = np.arange(0, H + W + A + 400, dtype=np.uint64)                  # longer than any path of these grids
    key = ((steps[:, None] << np.uint64(rb.value)) + tab[None, :].astype(np.uint64)).ravel()
    assert key.max() < 2 ** 32 - 1
    F = (np.float64(10.0) * steps[:, None].astype(np.float64) + np.float64(10.0) * np.sqrt(np.arange(words, dtype=np.float64))[None, :]).ravel()
    order = np.argsort(key, kind="stable")
    k, f = key[order], F[order]
    assert (np.diff(f) >= 0).all()
    assert np.array_equal(np.diff(k.astype(np.int64)) == 0, np.diff(f) == 0)
