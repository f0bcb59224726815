"""-m gpu, round 5: several independent missions in flight on one GPU (the reference's mission-list outer loop as a batch axis)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    import lsc_planner_amd as L
    return L


def test_four_concurrent_missions_plan_the_same_bits_as_four_sequential_ones(L):
    """Four contexts on four streams, ticks enqueued round-robin, against the same four missions flown one after the other: every
    trajectory, ideal state, cost, status and iteration count of every tick must be the same bits (the contexts share nothing but
    the chip).  The reference flies its mission list back to back (src/multi_sync_simulator_node.cpp:43-70, src/param.cpp:106-122)."""
    import torch
    from bench import MissionRun, rotated_mission
    dev = torch.device("cuda", 0)
    base = L.circle_swap(64, circle_radius=8.0, z=1.0, world=(-10, -10, 0, 10, 10, 2.5))
    missions = [base] + [rotated_mission(L, base, 0.3 * m, f"rot{m}") for m in range(1, 4)]
    ticks = 70          # into the crossing: hundreds of rows per agent, 8-12 iterations for the slowest

    def cfg():
        return L.PlannerConfig(goal_mode="prior_based", reset_threshold=0.15)

    def snapshot(r):
        torch.cuda.synchronize()
        return [t.cpu().numpy().copy() for t in (r.prev, r.states[0], r.cost, r.status, r.iters)]

    seq_out = []
    for ms in missions:
        r = MissionRun(L, torch, ms, cfg(), dev, torch.cuda.current_stream())
        snaps = []
        for t in range(ticks):
            r.tick()
            if t % 10 == 9 or t == ticks - 1:
                snaps.append(snapshot(r))
        seq_out.append(snaps)
        r.close()
    streams = [torch.cuda.Stream(device=dev) for _ in missions]
    runs = [MissionRun(L, torch, ms, cfg(), dev, st) for ms, st in zip(missions, streams)]
    torch.cuda.synchronize()
    con_out = [[] for _ in missions]
    for t in range(ticks):
        for r in runs:
            r.tick()
        if t % 10 == 9 or t == ticks - 1:
            for m, r in enumerate(runs):
                con_out[m].append(snapshot(r))
    for r in runs:
        r.close()
    # ... and through ONE launch per tick (lsc_tick_device_fused_batch: blockIdx.y = mission)
    runs = [MissionRun(L, torch, ms, cfg(), dev, torch.cuda.current_stream()) for ms in missions]
    bat_out = [[] for _ in missions]
    for t in range(ticks):
        for r in runs:
            r.seq += 1
        L.tick_device_fused_batch([r.pl for r in runs], [r.states[0] for r in runs], [r.goal for r in runs], [r.prev for r in runs],
                                  [r.nxt for r in runs], [r.states[1] for r in runs], [r.cost for r in runs], [r.status for r in runs],
                                  [r.iters for r in runs], [r.seq for r in runs], runs[0].stream)
        for r in runs:
            r.states.reverse()
            r.prev, r.nxt = r.nxt, r.prev
        if t % 10 == 9 or t == ticks - 1:
            for m, r in enumerate(runs):
                bat_out[m].append(snapshot(r))
    for r in runs:
        r.close()
    for m in range(len(missions)):
        assert len(seq_out[m]) == len(con_out[m]) == len(bat_out[m])
        for a, b, c in zip(seq_out[m], con_out[m], bat_out[m]):
            for x, y, z in zip(a, b, c):
                assert np.array_equal(x, y), f"mission {m} (streams)"
                assert np.array_equal(x, z), f"mission {m} (batch launch)"
        assert (seq_out[m][-1][3] == 0).all()
    # the missions are not copies of each other
    assert not np.array_equal(seq_out[0][-1][0], seq_out[1][-1][0])


def test_planar_world_keeps_planning_after_a_first_solve_that_fails(L):
    """in a planar world with z_2d != 0 an agent whose FIRST solve fails kept the optimiser's zero trajectory (z = 0), and
    every later host-buffer tick of the whole swarm was refused (LSC_EINVAL: not in the plane).  The stale plan's z block now starts at
    z_2d (the reference overrides the agent's own z on its next state callback, src/traj_planner.cpp:304-314), so the run goes on."""
    from lsc_planner_amd.planner import next_state_host
    # agent 1 starts inside agent 0's collision model -> its first QP is infeasible (status 1), agent 2 is far away
    ms = L.circle_swap(3, circle_radius=2.0, z=0.7, world=(-5, -5, 0, 5, 5, 2.5))
    ms.start[1] = ms.start[0] + np.array([0.05, 0.0, 0.0], np.float32)
    pl = L.SwarmPlanner(ms, L.PlannerConfig(world_dimension=2, world_z_2d=0.7))
    state = np.zeros((3, 9), np.float32)
    state[:, :3] = ms.start
    traj = np.zeros((3, 3, 30), np.float32)
    failed_first = False
    for tick in range(4):
        g = pl.plan(state, ms.goal, traj)           # raised LscError(LSC_EINVAL) at tick 2 before the fix
        if tick == 0:
            failed_first = bool((g["status"] != 0).any())
        assert np.all(g["traj"][:, 2, :] == np.float32(0.7)), "every plan of a planar world lies in the plane, failed solves included"
        traj = g["traj"]
        state = next_state_host(traj)
        assert np.all(state[:, 2] == np.float32(0.7))
    pl.close()
    assert failed_first, "the scene is meant to make the first solve of an agent fail"


def test_batch_launch_with_ragged_swarms_modes_and_refusals(L):
    """lsc_tick_device_fused_batch beyond the bench's four equal swarms: swarms of different sizes in one launch (workgroups beyond a
    swarm's size leave at once), a planar batch, a batch after a gust (the hand-over launch of the alternate-mode kernel is batched too),
    and the combinations it refuses (mixed alternate-mode classes, an octomap context) with the reason in lsc_last_error."""
    import torch
    from bench import MissionRun
    dev = torch.device("cuda", 0)

    def fly(missions, cfgs, ticks, batch, gust_at=None):
        runs = [MissionRun(L, torch, ms, cfg, dev, torch.cuda.current_stream()) for ms, cfg in zip(missions, cfgs)]
        for t in range(ticks):
            if gust_at is not None and t == gust_at:
                for r in runs:                       # push agent 0 of every swarm 0.3 m off its plan (> reset_threshold)
                    r.states[0][0, 0] += 0.3
            if batch:
                for r in runs:
                    r.seq += 1
                L.tick_device_fused_batch([r.pl for r in runs], [r.states[0] for r in runs], [r.goal for r in runs], [r.prev for r in runs],
                                          [r.nxt for r in runs], [r.states[1] for r in runs], [r.cost for r in runs], [r.status for r in runs],
                                          [r.iters for r in runs], [r.seq for r in runs], runs[0].stream)
                for r in runs:
                    r.states.reverse()
                    r.prev, r.nxt = r.nxt, r.prev
            else:
                for r in runs:
                    r.tick()
        torch.cuda.synchronize()
        out = [[x.cpu().numpy().copy() for x in (r.prev, r.states[0], r.cost, r.status, r.iters)] for r in runs]
        for r in runs:
            r.close()
        return out

    def same(a, b):
        return all(np.array_equal(x, y) for ra, rb in zip(a, b) for x, y in zip(ra, rb))

    # ragged: 5, 20 and 64 agents in one launch
    ms = [L.circle_swap(n, circle_radius=max(1.2, 8.0 * n / 64.0), z=1.0, world=(-10, -10, 0, 10, 10, 2.5)) for n in (5, 20, 64)]
    cfgs = lambda **kw: [L.PlannerConfig(goal_mode="prior_based", **kw) for _ in ms]
    assert same(fly(ms, cfgs(), 25, False), fly(ms, cfgs(), 25, True))
    # disturbance checks on, a gust at tick 6: every swarm hands agents to the alternate-mode kernel, batched
    a, b = fly(ms, cfgs(reset_threshold=0.15), 12, False, gust_at=6), fly(ms, cfgs(reset_threshold=0.15), 12, True, gust_at=6)
    assert same(a, b)
    # planar worlds
    mp = [L.circle_swap(n, circle_radius=max(1.2, 8.0 * n / 64.0), z=0.8, world=(-10, -10, 0, 10, 10, 2.5)) for n in (8, 16)]
    pc = lambda: [L.PlannerConfig(world_dimension=2, world_z_2d=0.8) for _ in mp]
    assert same(fly(mp, pc(), 15, False), fly(mp, pc(), 15, True))
    # refusals: mixed classes
    with pytest.raises(L.LscError, match="alternate-mode"):
        fly(ms[:2], [L.PlannerConfig(reset_threshold=0.15), L.PlannerConfig()], 1, True)
    with pytest.raises(L.LscError, match="planar|classes|launch failed"):
        fly(mp, [L.PlannerConfig(world_dimension=2, world_z_2d=0.8), L.PlannerConfig()], 1, True)


def test_the_interior_point_alone_still_passes_the_parity_files():
    """The library's default solver is the active-set solve, so the rest of the suite holds THAT to the oracle, the HiGHS pins and the
    fuzzers.  The interior point stays in the product -- it takes every agent the active-set solve hands over (infeasible QPs, working
    sets beyond its capacity), planar worlds and the second pass -- so the parity, pin, soak and edge files run once more with
    LSC_SOLVER=interior_point (the harness's default solver)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = [os.path.join(root, "tests", f) for f in ("test_gpu_parity.py", "test_gpu_edges.py", "test_gpu_round3.py", "test_gpu_soak.py")]
    env = dict(os.environ, LSC_SOLVER="interior_point")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-k", "not lds_held_before"] + files, env=env, cwd=root,
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:]


def test_active_set_solver_statistics_and_hand_over(L):
    """lsc_solver_stats: on the bench mission every agent-replan is finished by the active-set solve (nothing handed over); a scene with an
    infeasible QP (an agent inside another's collision model) ends with the infeasible verdict and the stale plan under both solvers -- the
    active-set solve proves it itself (no admissible step), marginal cases go to the interior point."""
    from lsc_planner_amd.planner import next_state_host
    ms = L.circle_swap(64, 8.0)
    pl = L.SwarmPlanner(ms, L.PlannerConfig(goal_mode="prior_based"))
    state = np.zeros((64, 9), np.float32); state[:, :3] = ms.start
    traj = np.zeros((64, 3, 30), np.float32)
    pl.iterations_total(reset=True)
    for _ in range(70):
        g = pl.plan(state, ms.goal, traj)
        assert (g["status"] == 0).all()
        traj = g["traj"]; state = next_state_host(traj)
    st = pl.solver_stats()
    assert st["solved"] == 64 * 70 and st["handed_over"] == 0 and st["ip_iterations"] == 0 and 0 < st["changes"] < 64 * 70 * 12
    pl.close()
    ms = L.circle_swap(3, circle_radius=2.0, z=1.0, world=(-5, -5, 0, 5, 5, 2.5))
    ms.start[1] = ms.start[0] + np.array([0.05, 0.0, 0.0], np.float32)
    out = {}
    for solver in ("active_set", "interior_point"):
        pl = L.SwarmPlanner(ms, L.PlannerConfig(solver=solver))
        state = np.zeros((3, 9), np.float32); state[:, :3] = ms.start
        pl.iterations_total(reset=True)
        out[solver] = pl.plan(state, ms.goal, np.zeros((3, 3, 30), np.float32))
        if solver == "active_set":
            st = pl.solver_stats()      # (an infeasibility that is not round-off is the active-set solve's own verdict: no admissible step; marginal ones go to the interior point)
            assert st["solved"] + st["handed_over"] == 3
        pl.close()
    assert np.array_equal(out["active_set"]["status"], out["interior_point"]["status"]) and (out["active_set"]["status"] != 0).any()


def test_hand_over_path_returns_the_interior_points_plans(L):
    """Everything only the interior point needs (assembly tables, Hessian constants, slot tables of the row reduction, its zeroed arrays and
    y-tables) is built where the active-set solve hands an agent over -- once in ~2 000 agent-replans in the field.  solver = "hand_over"
    (lsc_config.solver 2) is the test mode that makes EVERY agent take that path: the active-set solve runs, then the interior point starts
    from the state the late set-up leaves.  Its plans must be the interior point's own (solver = "interior_point": the same algorithm from
    the same start, another instantiation of the kernel) -- over the crossing of the bench mission, a corridor world and a planar world."""
    from lsc_planner_amd.planner import next_state_host
    cases = [(L.circle_swap(64, 8.0), dict(goal_mode="prior_based"), 45, 25), (L.circle_swap(12, 3.0), dict(world_dimension=2, world_z_2d=1.0), 12, 1)]
    for ms, kw, ticks, first in cases:
        # (a handed-over agent takes the interior point's COLD start: the reference run is the interior point without its warm start)
        pls = {s: L.SwarmPlanner(ms, L.PlannerConfig(solver=s, warm_start_mu=0.0, **kw)) for s in ("interior_point", "hand_over")}
        N = ms.qn
        state = np.zeros((N, 9), np.float32); state[:, :3] = ms.start
        traj = np.zeros((N, 3, 30), np.float32)
        pls["hand_over"].iterations_total(reset=True)
        worst = 0.0
        for tick in range(1, ticks + 1):
            g = {s: p.plan(state, ms.goal, traj) for s, p in pls.items()}
            assert np.array_equal(g["interior_point"]["status"], g["hand_over"]["status"]), tick
            if tick >= first:
                worst = max(worst, float(np.abs(g["interior_point"]["traj"] - g["hand_over"]["traj"]).max()))
                ok = g["interior_point"]["status"] == 0
                assert np.allclose(g["interior_point"]["cost"][ok], g["hand_over"]["cost"][ok], rtol=1e-9, atol=1e-12), tick
            traj = g["interior_point"]["traj"]; state = next_state_host(traj)
        st = pls["hand_over"].solver_stats()
        assert st["solved"] == 0 and st["handed_over"] == N * ticks and st["ip_iterations"] > 0, st
        assert worst <= 1e-6, worst          # (float32 plans of two instantiations of one algorithm: equal up to the last bit of a rounding)
        for p in pls.values():
            p.close()


def test_instance_where_a_handed_over_agent_carries_the_interior_points_tolerance():
    """Found by the M = 4 fuzzer (seed 8500018, tick 4): ten agents in a 12 m world, half-second segments.  Agent 9 needs more than the 12
    rows the active-set solve's working set holds, is handed to the interior point, and its plan is 5.8e-5 m from the oracle's exact optimum
    at 4.2e-10 relative higher cost -- the interior point's stopping tolerance along a flat direction, not the active-set solve's doing: the
    plan is bit-identical to solver = interior_point.  Pins (a) that agreement, (b) statuses and costs within the table, (c) every agent the
    active-set solve finished itself within TRAJ_ATOL of the oracle, the handed-over ones within the interior point's 1e-4 m."""
    import os
    import lsc_planner_amd as L
    from lsc_planner_amd.mission import Mission
    from lsc_planner_amd.planner import PlannerConfig
    from tolerances import COST_ATOL, COST_RTOL, FUZZ_TRAJ_ATOL_HALF_SECOND
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fuzz_found_m4_handover_8500018.npz"))
    L.load_library(4)
    ms = Mission(d["state"][:, :3].copy(), d["goal"].copy(), d["wmin"], d["wmax"], d["radius"], d["dw"], d["vmax"], d["amax"], d["vnom"], name="replay")
    res = {}
    for solver in ("active_set", "interior_point"):
        pl = L.SwarmPlanner(ms, PlannerConfig(goal_mode="static", dt=0.5, horizon=2.0, solver=solver, warm_start_mu=0.0))     # (hand-overs start cold)
        assert pl.M == 4
        pl.plan(d["state"], d["goal"], d["traj"])           # (sequence number 1 takes the current-velocity model: only to move it on)
        pl.iterations_total(reset=True)
        res[solver] = pl.plan(d["state"], d["goal"], d["traj"])
        if solver == "active_set":
            st = pl.solver_stats()
            assert st["solved"] + st["handed_over"] == 10 and 1 <= st["handed_over"] <= 5, st
        pl.close()
    g, ip = res["active_set"], res["interior_point"]
    assert np.array_equal(g["status"], d["ostatus"]) and (g["status"] == 0).all()
    assert (np.abs(g["cost"] - d["ocost"]) <= COST_RTOL * np.abs(d["ocost"]) + COST_ATOL).all()
    diff = np.abs(g["traj"].astype(np.float64) - d["otraj"]).reshape(10, -1).max(1)
    assert diff.max() <= FUZZ_TRAJ_ATOL_HALF_SECOND, diff
    assert np.array_equal(g["traj"][9], ip["traj"][9]) and g["cost"][9] == ip["cost"][9]      # the handed-over agent: the interior point's plan
    assert g["cost"][9] >= d["ocost"][9]                                                         # ... on the costlier side of the exact optimum


def test_shards_of_a_large_swarm_cull_obstacles_and_plan_the_same_bits(L):
    """A shard of at most one agent per CU out of a swarm of >= 512 agents runs the 512-lane latency build with the obstacle-level sphere
    cull in front of its unit-level cull (lsc_prep_kernel ahead of the tick: launch_plan).  Same rows in the same order as without any
    cull (prune = 3), hence bit-identical plans, costs, statuses, iteration and row counts -- 1024 agents as two of the eight shards a
    node would hold (the first and one in the middle), six ticks."""
    from lsc_planner_amd.planner import next_state_host
    n = 1024
    ms = L.random_swarm(n, seed=20260929)
    shards = [(0, 128), (512, 128)]
    with_cull = [L.SwarmPlanner(ms, L.PlannerConfig(prune=1, goal_mode="prior_based")) for _ in shards]
    without = [L.SwarmPlanner(ms, L.PlannerConfig(prune=3, goal_mode="prior_based")) for _ in shards]
    for (first, count), a, b in zip(shards, with_cull, without):
        a.set_shard(first, count); b.set_shard(first, count)
    whole = L.SwarmPlanner(ms, L.PlannerConfig(goal_mode="prior_based"))          # (the throughput build: flies the mission)
    state = np.zeros((n, 9), np.float32); state[:, :3] = ms.start
    traj = np.zeros((n, 3, 30), np.float32)
    for tick in range(1, 7):
        g = whole.plan(state, ms.goal, traj)
        for (first, count), a, b in zip(shards, with_cull, without):
            ga, gb = a.plan(state, ms.goal, traj), b.plan(state, ms.goal, traj)
            for k in ("traj", "cost", "status", "iters"):
                assert np.array_equal(ga[k], gb[k]), (tick, first, k)
            assert np.array_equal(a.row_counts()[first:first + count], b.row_counts()[first:first + count]), (tick, first)
            assert np.array_equal(ga["status"], g["status"][first:first + count]), (tick, first)
        traj = g["traj"]; state = next_state_host(traj)
    assert 0 < with_cull[0].row_counts()[:128].mean() < 27 * (n - 1) / 10
    for p in with_cull + without + [whole]:
        p.close()
