"""-m gpu: edge cases of the tick through the C ABI -- smallest swarms, agents at their goal, agents on the world
boundary, ragged / empty shards, argument errors.  Same tolerances as test_gpu_parity.py."""
import ctypes

import numpy as np
import pytest

from conftest import oracle_swarm

pytestmark = pytest.mark.gpu

from tolerances import COST_ATOL, COST_RTOL, TRAJ_ATOL


@pytest.fixture(scope="module")
def L():
    import lsc_planner_amd as L
    L.load_library()
    return L


def _mission(L, start, goal, world=(-10, -10, 0, 10, 10, 2.5)):
    n = len(start)
    base = L.circle_swap(max(n, 2), 4.0, world=world)
    return L.Mission(np.asarray(start, np.float32), np.asarray(goal, np.float32), base.world_min, base.world_max,
                     base.radius[:n], base.downwash[:n], base.max_vel[:n], base.max_acc[:n], base.nominal_velocity[:n],
                     name=f"edge{n}")


def _ticks_vs_oracle(L, O, ms, n_ticks, cfg=None):
    from lsc_planner_amd.planner import next_state_host
    N = ms.qn
    pl = L.SwarmPlanner(ms, cfg or L.PlannerConfig())
    sw = oracle_swarm(O, ms)
    state = np.zeros((N, 9), np.float32)
    state[:, :3] = ms.start
    traj = np.zeros((N, 3, 30), np.float32)
    for tick in range(1, n_ticks + 1):
        g = pl.plan(state, ms.goal, traj, want_constraints=N > 1)
        sw.stale[:] = traj
        o = sw.tick(state, ms.goal, traj, tick, want_lsc=N > 1, nthreads=1)
        if N > 1:
            assert np.array_equal(g["normal"], o["normal"]), tick
            assert np.array_equal(g["d"], o["d"]), tick
        assert np.array_equal(g["status"], o["status"]), tick
        ok = o["status"] == 0
        assert (np.abs(g["cost"] - o["cost"])[ok] <= COST_RTOL * np.abs(o["cost"])[ok] + COST_ATOL).all(), tick
        assert np.abs(g["traj"] - o["traj"]).max() <= TRAJ_ATOL, tick
        traj = g["traj"]
        state = next_state_host(traj)
    pl.close()
    return state


def test_single_agent_has_no_lsc_rows(L, oracle):
    """N = 1: zero obstacles, the QP has only the dynamic-limit rows and bounds (the reference's N_obs = 0 loops)."""
    ms = _mission(L, [[-3.0, 0.5, 1.0]], [[3.0, -0.5, 1.2]])
    state = _ticks_vs_oracle(L, oracle, ms, 12)
    assert np.linalg.norm(state[0, :3] - ms.goal[0]) < np.linalg.norm(ms.start[0] - ms.goal[0])


def test_two_agents_head_on(L, oracle):
    ms = _mission(L, [[-2.0, 0.0, 1.0], [2.0, 0.0, 1.0]], [[2.0, 0.0, 1.0], [-2.0, 0.0, 1.0]])
    _ticks_vs_oracle(L, oracle, ms, 25)


def test_agent_already_at_its_goal_keeps_hovering(L, oracle):
    """cost -> 0: the relative cost tolerance needs its absolute floor; the plan must stay put."""
    ms = _mission(L, [[1.0, 1.0, 1.0], [-4.0, 0.0, 1.0], [4.0, 3.0, 1.5]], [[1.0, 1.0, 1.0], [4.0, 0.0, 1.0], [-4.0, -3.0, 1.0]])
    state = _ticks_vs_oracle(L, oracle, ms, 10)
    assert np.abs(state[0, :3] - ms.goal[0]).max() < 1e-3


def test_start_on_the_world_boundary_and_goal_outside(L, oracle):
    """Bounds active from the first tick; a desired goal outside the world box is only approached up to the bound."""
    ms = _mission(L, [[-10.0, 0.0, 0.0], [0.0, 9.5, 2.5]], [[-12.0, 0.0, 1.0], [0.0, -9.0, 1.0]])
    state = _ticks_vs_oracle(L, oracle, ms, 15)
    assert state[0, 0] >= -10.0 - 1e-5


def test_prior_based_goal_mode_small_swarms(L, oracle):
    from lsc_planner_amd.planner import next_state_host
    for n in (1, 2, 3):
        ms = L.circle_swap(max(n, 2), 1.0)
        if n == 1:
            ms = _mission(L, [[-1.0, 0.0, 1.0]], [[1.0, 0.0, 1.0]])
        pl = L.SwarmPlanner(ms, L.PlannerConfig(goal_mode="prior_based"))
        N = ms.qn
        state = np.zeros((N, 9), np.float32)
        state[:, :3] = ms.start
        traj = np.zeros((N, 3, 30), np.float32)
        for tick in range(1, 9):
            g = pl.plan(state, ms.goal, traj)
            goals = pl.last_goals()
            ref = oracle.goal_prior_based(state, ms.goal, traj, tick)
            assert np.array_equal(goals, ref), (n, tick)
            traj = g["traj"]
            state = next_state_host(traj)
        pl.close()


def test_empty_and_ragged_shards(L):
    """A rank may own zero agents (more GPUs than agents): the tick is a no-op, not an error."""
    ms = L.circle_swap(5, 3.0)
    N = ms.qn
    state = np.zeros((N, 9), np.float32)
    state[:, :3] = ms.start
    traj = np.zeros((N, 3, 30), np.float32)
    full = L.SwarmPlanner(ms)
    ref = full.plan(state, ms.goal, traj)
    full.close()
    got = np.zeros_like(ref["traj"])
    for first, count in ((0, 2), (2, 0), (2, 3)):
        pl = L.SwarmPlanner(ms)
        pl.set_shard(first, count)
        g = pl.plan(state, ms.goal, traj)
        assert g["traj"].shape[0] == count
        got[first:first + count] = g["traj"]
        pl.close()
    assert np.array_equal(got, ref["traj"])


def test_argument_errors_are_codes_not_crashes(L):
    lib = L.load_library()
    ms = L.circle_swap(4, 3.0)
    pl = L.SwarmPlanner(ms)
    assert lib.lsc_set_shard(pl.ctx, 3, 5) != 0                 # beyond N
    assert lib.lsc_set_shard(pl.ctx, -1, 2) != 0
    assert lib.lsc_replan_tick(pl.ctx, None, None, None, 1, None, None, None, None, None, None, None) != 0
    assert lib.lsc_kernel_time_ms(pl.ctx, 7, ctypes.byref(ctypes.c_double()), None) != 0
    assert len(lib.lsc_last_error(pl.ctx)) >= 0
    pl.close()
    with pytest.raises(Exception):
        L.SwarmPlanner(ms, L.PlannerConfig(use_octomap=True)).plan(np.zeros((4, 9), np.float32), ms.goal,
                                                                  np.zeros((4, 3, 30), np.float32))   # no distance map set


def test_per_launch_times_are_reported(L):
    ms = L.circle_swap(8, 3.0)
    pl = L.SwarmPlanner(ms)
    state = np.zeros((8, 9), np.float32)
    state[:, :3] = ms.start
    traj = np.zeros((8, 3, 30), np.float32)
    pl.set_timing(True)
    for _ in range(3):
        pl.plan(state, ms.goal, traj)
    t = pl.kernel_times_ms(0)
    avg, n = pl.kernel_time_ms(0)
    assert n == 3 and len(t) == 3 and (t > 0).all() and abs(t.mean() - avg) < 1e-9
    pl.close()
