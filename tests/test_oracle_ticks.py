"""The oracle reproduces the committed tick snapshots; float32 state propagation; mission-level behaviour."""
import numpy as np

from conftest import golden_mission, oracle_swarm


def test_oracle_reproduces_golden_ticks(oracle, ticks):
    for name, keep in (("multi_simple4", (1, 2, 3, 20)), ("multi_circle20", (1, 15))):
        ms = golden_mission(ticks, name)
        for tick in keep:
            sw = oracle_swarm(oracle, ms)
            sw.stale[:] = ticks[f"{name}/tick{tick}/stale"]
            r = sw.tick(ticks[f"{name}/tick{tick}/state"], ms.goal, ticks[f"{name}/tick{tick}/prev"], tick, want_lsc=True, nthreads=4)
            assert np.array_equal(r["normal"], ticks[f"{name}/tick{tick}/normal"])
            assert np.array_equal(r["d"], ticks[f"{name}/tick{tick}/d"])
            assert np.abs(r["traj"] - ticks[f"{name}/tick{tick}/traj"]).max() <= 1e-6
            assert np.allclose(r["cost"], ticks[f"{name}/tick{tick}/cost"], rtol=1e-9)
            assert (r["status"] == 0).all()


def test_first_tick_normals_are_constant_over_segments(oracle, ticks):
    """Tick 1 uses the current-velocity model with v = 0: all trajectories are points, so the 5 normals of a pair
    coincide and d is constant -- exactly what log/QPmodel.lp rows c46-c288 show."""
    n = ticks["multi_circle20/tick1/normal"]
    d = ticks["multi_circle20/tick1/d"]
    assert np.array_equal(n, np.repeat(n[:, :, :1], 5, axis=2))
    assert np.array_equal(d, np.repeat(d[:, :, :1, :1], 5, axis=2).repeat(6, axis=3))


def test_next_state_is_float32_derivative_of_segment_one(oracle):
    from lsc_planner_amd.planner import next_state_host
    rng = np.random.default_rng(3)
    traj = rng.normal(size=(7, 3, 30)).astype(np.float32)
    host = next_state_host(traj)
    for q in range(7):
        assert np.array_equal(oracle.next_state(traj[q]), host[q])
    c = traj[0, 1, 6:9]
    v0 = np.float32(np.float32((c[1] - c[0]) * np.float32(5)) * np.float32(5))
    assert host[0, 4] == v0 and host[0, 1] == c[0]


def test_mission_simple4_reaches_goals_without_collision(oracle, ticks):
    from lsc_planner_amd.planner import next_state_host
    ms = golden_mission(ticks, "multi_simple4")
    sw = oracle_swarm(oracle, ms)
    N = ms.qn
    state = np.zeros((N, 9), np.float32); state[:, :3] = ms.start
    traj = np.zeros((N, 3, 30), np.float32)
    for tick in range(1, 120):
        r = sw.tick(state, ms.goal, traj, tick, nthreads=4)
        assert (r["status"] == 0).all()
        traj = r["traj"]
        state = next_state_host(traj)
        p = state[:, :3].astype(np.float64).copy(); p[:, 2] /= 2.0
        D = np.linalg.norm(p[:, None] - p[None], axis=2) + np.eye(N) * 9
        assert D.min() >= 0.3 - 1e-4                     # safety ratio >= 1 (multi_sync_simulator.cpp:446-503)
        if np.linalg.norm(state[:, :3] - ms.goal, axis=1).max() < 0.1:
            break
    assert tick < 80


def test_prior_based_goal_rules(oracle):
    """goalPlanningWithPriority on an empty map: LOS goal = desired goal clamped to goal_radius from the end of the
    initial trajectory; retreat by priority_dist_threshold + 0.1 from a closer higher-priority agent."""
    state = np.zeros((2, 9), np.float32)
    state[0, :3] = [0, 0, 1]; state[1, :3] = [5, 0, 1]
    goal = np.array([[10, 0, 1], [5.5, 0, 1]], np.float32)
    prev = np.zeros((2, 3, 30), np.float32)
    g = oracle.goal_prior_based(state, goal, prev, 1)
    assert np.allclose(g[0], [2, 0, 1]) and np.allclose(g[1], [5.5, 0, 1])          # clamp to 2 m / goal within reach
    # agent 1 (closer to its goal -> higher priority) sits 0.3 m in front of agent 0: agent 0 retreats to 0.5 m
    state[1, :3] = [0.3, 0, 1]; goal[1] = [0.9, 0, 1]
    prev[:, 0, :] = state[:, 0:1]; prev[:, 2, :] = 1
    g = oracle.goal_prior_based(state, goal, prev, 2)
    assert np.allclose(g[0], [-0.5, 0, 1], atol=1e-6)
    assert np.allclose(g[1], [0.9, 0, 1])
