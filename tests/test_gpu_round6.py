"""Round 6: neighbour lists of large swarms (lsc_neigh.hip) -- the obstacle loop of TrajPlanner::generateLSC
(src/traj_planner.cpp:1335-1407) through a uniform grid instead of a walk over all N - 1 obstacles per agent."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    import lsc_planner_amd as L
    return L


def _start(ms):
    state = np.zeros((ms.qn, 9), np.float32)
    state[:, :3] = ms.start
    return state, np.zeros((ms.qn, 3, 30), np.float32)


class _Env:
    """Environment variables lsc_set_agents reads (capacities / cell size of the neighbour lists) for the planners created inside."""

    def __init__(self, **kv):
        self.kv = {k: str(v) for k, v in kv.items()}

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        os.environ.update(self.kv)

    def __exit__(self, *exc):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _fly(L, planners, ms, ticks, check):
    from lsc_planner_amd.planner import next_state_host
    state, traj = _start(ms)
    for tick in range(1, ticks + 1):
        g = [p.plan(state, ms.goal, traj) for p in planners]
        check(tick, g)
        traj = g[0]["traj"]
        state = next_state_host(traj)


@pytest.mark.parametrize("n,world,ticks,cfg", [(1024, (-20, -20, 0, 20, 20, 5), 40, dict(goal_mode="prior_based", reset_threshold=0.15)),
                                               (1024, (-20, -20, 0, 20, 20, 5), 12, dict()),
                                               (640, (-9, -9, 0, 9, 9, 3.0), 25, dict(goal_mode="prior_based")),
                                               (640, (-9, -9, 0, 9, 9, 3.0), 12, dict(reset_threshold=0.15))])
def test_neighbour_lists_change_nothing_but_the_time(L, n, world, ticks, cfg):
    """Swarms of >= 512 agents get their (obstacle, segment) units as a sorted list from the grid kernels instead of walking all obstacles:
    a superset of what the in-kernel cull keeps, in the same order, and the exact per-row test decides in both cases -- so plans, costs,
    statuses, goals, iteration and row counts are bit-identical to the context without lists (LSC_NO_NEIGHBOUR_LISTS: the round-5 cull)
    and to prune = 3 (no cull at all), tick after tick of a mission in which the swarm mixes (the second world is crowded: 640 agents in
    18 x 18 x 3 m).  With prior_based goals the candidates of the priority rule (src/traj_planner.cpp:540-608) come from the same grid,
    and with reset_threshold the disturbance checks (:866-878, 1047-1061) are made by the build kernel: at tick 9 agent 7 is pushed
    0.3 m off its plan and the whole swarm switches to the slack-variable QPs, in all three contexts alike."""
    ms = L.random_swarm(n, world=world, seed=20260930, min_sep=0.5)
    with _Env(LSC_NEIGH_ALWAYS=1):      # (lists are used only where they pay -- two rounds of workgroups or >= 2048 agents --: the tests force them)
        a = L.SwarmPlanner(ms, L.PlannerConfig(prune=1, **cfg))
    with _Env(LSC_NO_NEIGHBOUR_LISTS=1):
        b = L.SwarmPlanner(ms, L.PlannerConfig(prune=1, **cfg))
    c = L.SwarmPlanner(ms, L.PlannerConfig(prune=3, **cfg))
    assert a.neighbour_counts() is not None and b.neighbour_counts() is None and c.neighbour_counts() is None
    seen, cands, retreats = [], [], 0
    from lsc_planner_amd.planner import next_state_host
    state, traj = _start(ms)
    for tick in range(1, ticks + 1):
        if tick == 9 and cfg.get("reset_threshold"):
            state[7, 0] += 0.3
        g = [p.plan(state, ms.goal, traj) for p in (a, b, c)]
        for k in ("traj", "cost", "status", "iters"):
            assert np.array_equal(g[0][k], g[1][k]), (tick, k)
            if tick <= 10 or tick % 8 == 0:
                assert np.array_equal(g[0][k], g[2][k]), (tick, k)
        assert np.array_equal(a.row_counts(), b.row_counts()), tick
        assert np.array_equal(a.last_goals(), b.last_goals()) and np.array_equal(a.last_goals(), c.last_goals()), tick
        units, prio = a.neighbour_counts(priority=True)
        if not (cfg.get("reset_threshold") and tick >= 9):                   # (a disturbed swarm plans in lsc_general_kernel: no LSC units at all)
            assert (units >= 0).mean() > (0.999 if n == 1024 else 0.5), tick    # (nearly) every agent had a list: in the crowded world some exceed its 1024 units
            units = np.where(units < 0, 5 * (n - 1), units)
            assert (units * 6 >= a.row_counts()).all(), tick                  # ... that held every unit its rows came from
            seen.append(units.mean())
        assert (prio >= 0).all(), tick
        cands.append(int(prio.sum()))
        if cfg.get("goal_mode") == "prior_based":
            retreats += int((np.linalg.norm(a.last_goals() - state[:, :3], axis=1) < 0.55).sum())
        traj = g[0]["traj"]
        state = next_state_host(traj)
    assert max(seen) < 5 * (n - 1) / (4 if n == 1024 else 1), seen            # a list is a fraction of the 5 (N - 1) units there are
    if cfg.get("goal_mode") == "prior_based":
        assert sum(cands) > 0 and (n == 1024 or retreats > 0), (cands, retreats)   # the rule had candidates, and in the crowd it fired
    else:
        assert sum(cands) == 0
    for p in (a, b, c):
        p.close()


@pytest.mark.parametrize("env", [dict(LSC_NEIGH_CELL=1000.0),                       # one bucket for everybody: 12 slots + the overflow list
                                 dict(LSC_NEIGH_CELL=1000.0, LSC_NEIGH_OVF_CAP=16),  # ... which overflows too: nobody gets a list
                                 dict(LSC_NEIGH_LIST_CAP=40),                        # lists too short for the crowded agents: those cull by themselves
                                 dict(LSC_NEIGH_CELL=0.02),                          # queries of more cells than a workgroup visits: no lists
                                 dict(LSC_NEIGH_CELL=0.7)])                          # many cells, several buckets met twice through the hash
def test_neighbour_list_overflow_paths_plan_the_same_bits(L, env):
    """Every capacity of the neighbour lists has a way out that changes no result: a full bucket spills into an overflow list every query
    walks; a full overflow list, a query over too many cells, too many candidates or a list beyond its capacity leave the agent without a
    list (count -1) and its own phase B culls as it did in round 5."""
    ms = L.random_swarm(600, world=(-12, -12, 0, 12, 12, 3.0), seed=11)
    cfg = dict(goal_mode="prior_based", reset_threshold=0.15, priority_dist_threshold=0.8)      # (candidates of the priority rule in every tick)
    with _Env(LSC_NEIGH_ALWAYS=1, **env):
        a = L.SwarmPlanner(ms, L.PlannerConfig(prune=1, **cfg))
    b = L.SwarmPlanner(ms, L.PlannerConfig(prune=3, **cfg))
    none, some = [], []

    def check(tick, g):
        for k in ("traj", "cost", "status", "iters"):
            assert np.array_equal(g[0][k], g[1][k]), (tick, k)
        assert np.array_equal(a.row_counts(), b.row_counts()), tick
        assert np.array_equal(a.last_goals(), b.last_goals()), tick
        u, pr = a.neighbour_counts(priority=True)
        assert ((pr < 0) == (u < 0)).all() or "LSC_NEIGH_LIST_CAP" in env      # (a unit list beyond its capacity does not take the priority candidates with it)
        none.append(int((u < 0).sum())); some.append(int((u >= 0).sum()))

    _fly(L, [a, b], ms, 8, check)
    if "LSC_NEIGH_OVF_CAP" in env or env.get("LSC_NEIGH_CELL") == 0.02:
        assert sum(some) == 0, (none, some)
    elif "LSC_NEIGH_LIST_CAP" in env:
        assert sum(none) > 0 and sum(some) > 0, (none, some)
    else:
        assert sum(none) == 0, (none, some)
    a.close(); b.close()


def test_neighbour_lists_of_a_shard_and_of_the_four_segment_build(L):
    """A shard (lsc_set_shard) gets lists for its own agents only -- the grid holds the whole swarm --, and the M = 4 library builds its
    lists with four segments per obstacle: both plan the bits of the context without lists."""
    import lsc_planner_amd as LL
    from lsc_planner_amd.planner import next_state_host
    ms = L.random_swarm(768, world=(-16, -16, 0, 16, 16, 4.0), seed=5)
    whole = L.SwarmPlanner(ms, L.PlannerConfig(prune=3))
    with _Env(LSC_NEIGH_ALWAYS=1):
        part = L.SwarmPlanner(ms, L.PlannerConfig(prune=1))
    whole.set_shard(300, 200); part.set_shard(300, 200)        # (the same build of the plan kernel: 200 agents take the 512-lane one)
    state, traj = _start(ms)
    full = L.SwarmPlanner(ms, L.PlannerConfig(prune=1))
    for tick in range(1, 6):
        g, h = whole.plan(state, ms.goal, traj), part.plan(state, ms.goal, traj)
        for k in ("traj", "cost", "status", "iters"):
            assert np.array_equal(g[k], h[k]), (tick, k)
        assert (part.neighbour_counts()[300:500] >= 0).all()
        traj = full.plan(state, ms.goal, traj)["traj"]; state = next_state_host(traj)
    whole.close(); part.close(); full.close()
    cfg = dict(dt=0.5, horizon=2.0)
    with _Env(LSC_NEIGH_ALWAYS=1):
        a = LL.SwarmPlanner(ms, LL.PlannerConfig(prune=1, **cfg))
    b = LL.SwarmPlanner(ms, LL.PlannerConfig(prune=3, **cfg))
    state, traj = _start(ms)
    traj = np.zeros((ms.qn, 3, 24), np.float32)
    for tick in range(1, 6):
        g, h = a.plan(state, ms.goal, traj), b.plan(state, ms.goal, traj)
        for k in ("traj", "cost", "status"):
            assert np.array_equal(g[k], h[k]), (tick, k)
        assert (a.neighbour_counts() >= 0).all()
        traj = g["traj"]; state = next_state_host(traj, dt=0.5)
    a.close(); b.close()


def test_neighbour_lists_are_used_where_they_pay(L):
    """The lists cost two launches in front of the tick: a context uses them when its shard takes more than one round of throughput
    workgroups or the swarm has >= 2048 agents (profiles/r06_neighbour_lists.log), not for a one-round shard of a smaller swarm."""
    ms = L.random_swarm(1024, seed=20260929)
    whole, part = L.SwarmPlanner(ms, L.PlannerConfig()), L.SwarmPlanner(ms, L.PlannerConfig())
    part.set_shard(0, 128)
    state, traj = _start(ms)
    whole.plan(state, ms.goal, traj); part.plan(state, ms.goal, traj)
    assert (whole.neighbour_counts() >= 0).all()
    assert (part.neighbour_counts()[:128] == 0).all()         # (never written: the buffer starts zeroed)
    whole.close(); part.close()
    ms = L.random_swarm(2048, world=(-28, -28, 0, 28, 28, 5), seed=3)
    part = L.SwarmPlanner(ms, L.PlannerConfig())
    part.set_shard(256, 256)
    state, traj = _start(ms)
    part.plan(state, ms.goal, traj)
    assert (part.neighbour_counts()[256:512] > 0).all()
    part.close()


@pytest.mark.parametrize("planar", [False, True])
def test_neighbour_lists_with_mixed_agents_and_in_a_planar_world(L, planar):
    """Agents of different radius, downwash and speed limits -- the grid's query radius uses the swarm's extremes (largest obstacle-side
    radius, smallest / largest downwash), each sphere test the pair's own numbers -- and a planar world (world/dimension 2: every agent at
    z = z_2d, rows without their z term): bit-identical to the contexts without lists / without any cull."""
    from lsc_planner_amd.planner import next_state_host
    rng = np.random.default_rng(12)
    n = 600
    ms = L.random_swarm(n, world=(-14, -14, 0, 14, 14, 4.0), seed=21, min_sep=0.9)
    ms.radius[:] = rng.choice([0.1, 0.15, 0.25], n)
    ms.downwash[:] = rng.choice([1.0, 2.0, 3.0], n) if not planar else 2.0
    ms.max_vel[:] = rng.choice([0.6, 1.0, 1.4], n)[:, None]
    ms.max_acc[:] = rng.choice([1.5, 2.0, 3.0], n)[:, None]
    cfg = dict(goal_mode="prior_based", priority_dist_threshold=0.7)
    if planar:
        cfg.update(world_dimension=2, world_z_2d=1.0)
        ms.start[:, 2] = 1.0; ms.goal[:, 2] = 1.0
    with _Env(LSC_NEIGH_ALWAYS=1):
        a = L.SwarmPlanner(ms, L.PlannerConfig(prune=1, **cfg))
    with _Env(LSC_NO_NEIGHBOUR_LISTS=1):
        b = L.SwarmPlanner(ms, L.PlannerConfig(prune=1, **cfg))
    c = L.SwarmPlanner(ms, L.PlannerConfig(prune=3, **cfg))
    state, traj = _start(ms)
    for tick in range(1, 16):
        g = [p.plan(state, ms.goal, traj) for p in (a, b, c)]
        for k in ("traj", "cost", "status", "iters"):
            assert np.array_equal(g[0][k], g[1][k]) and np.array_equal(g[0][k], g[2][k]), (tick, k)
        assert np.array_equal(a.row_counts(), c.row_counts()), tick
        assert np.array_equal(a.last_goals(), c.last_goals()), tick
        units, prio = a.neighbour_counts(priority=True)
        assert (units >= 0).mean() > 0.9 and (prio >= 0).all(), tick
        traj = g[0]["traj"]
        state = next_state_host(traj)
        if planar:
            state[:, 2] = 1.0; state[:, 5] = 0.0; state[:, 8] = 0.0
    for p in (a, b, c):
        p.close()


def test_neighbour_list_fuzzer(L):
    """Three seeds of tests/fuzz_neighbours.py (random sizes, densities, agent constants, goal modes, disturbances, cell sizes): lists against
    no cull at all, every tick the same bits.  The large-count run: profiles/r06_neighbour_fuzz.log."""
    import fuzz_neighbours
    for seed in (101, 102, 103):
        r = fuzz_neighbours.one_seed(L, seed, 8)
        assert not isinstance(r, str), r


def test_crowded_swarm_with_more_surviving_units_than_the_old_list_held(L):
    """1024 agents of mixed size in 14 x 14 x 9 m (seed 50 of tests/fuzz_neighbours.py): nearly every obstacle is near, an agent keeps up to
    ~2 500 rows and thousands of units survive the in-kernel cull.  Until round 6 that list was indexed linearly from rrhs over 12 R entries
    "in rrhs, rn and cmap" -- arrays that are not contiguous --, so entries beyond 4 R lay where the GJK passes put their temporary rows, and
    the kernel read overwritten unit indices: a memory fault at tick 3, with or without the neighbour lists.  Both cull paths against no
    cull at all, through the tick that faulted."""
    from lsc_planner_amd.planner import next_state_host
    rng = np.random.default_rng(50)
    n = 1024
    for _ in range(4):
        rng.random()                                        # (not the fuzzer's draws: only the world matters)
    half = float(np.sqrt(n / 0.6 / 9.0) / 2.0)
    ms = L.random_swarm(n, world=(-half, -half, 0, half, half, 9.0), seed=50, min_sep=0.7)
    ms.radius[:] = rng.choice([0.1, 0.15, 0.2], n)
    ms.downwash[:] = rng.choice([1.0, 1.5, 2.0, 3.0], n)
    ms.max_vel[:] = rng.choice([0.5, 1.0, 1.5], n)[:, None]
    ms.max_acc[:] = rng.choice([1.0, 2.0, 4.0], n)[:, None]
    with _Env(LSC_NEIGH_ALWAYS=1):
        a = L.SwarmPlanner(ms, L.PlannerConfig(prune=1))
    with _Env(LSC_NO_NEIGHBOUR_LISTS=1):
        b = L.SwarmPlanner(ms, L.PlannerConfig(prune=1))
    c = L.SwarmPlanner(ms, L.PlannerConfig(prune=3))
    state, traj = _start(ms)
    most = 0
    for tick in range(1, 6):
        g = [p.plan(state, ms.goal, traj) for p in (a, b, c)]
        for k in ("traj", "cost", "status", "iters"):
            assert np.array_equal(g[0][k], g[2][k]) and np.array_equal(g[1][k], g[2][k]), (tick, k)
        assert np.array_equal(a.row_counts(), c.row_counts()) and np.array_equal(b.row_counts(), c.row_counts()), tick
        most = max(most, int(c.row_counts().max()))
        traj = g[2]["traj"]
        state = next_state_host(traj)
    assert most > 1500                                      # (rows of the fullest agent; its surviving units outnumber the first 4 R = 1836 entries of the old list)
    for p in (a, b, c):
        p.close()


def test_neighbour_lists_far_from_the_origin(L):
    """The grid kernels form their bounds in float32 (rounded up); at coordinates of kilometres a float32 ulp is 0.5 mm, so the bounds carry a
    term proportional to the coordinates.  A 600-agent swarm 5 km from the origin plans the same bits with the lists as without any cull."""
    from lsc_planner_amd.planner import next_state_host
    off = np.array([5000.0, -3000.0, 200.0], np.float32)
    ms = L.random_swarm(600, world=(-12, -12, 0, 12, 12, 3.0), seed=31)
    ms.start[:] = ms.start + off; ms.goal[:] = ms.goal + off
    ms.world_min[:] = ms.world_min + off; ms.world_max[:] = ms.world_max + off
    cfg = dict(goal_mode="prior_based", priority_dist_threshold=0.8)
    with _Env(LSC_NEIGH_ALWAYS=1):
        a = L.SwarmPlanner(ms, L.PlannerConfig(prune=1, **cfg))
    b = L.SwarmPlanner(ms, L.PlannerConfig(prune=3, **cfg))
    state, traj = _start(ms)
    for tick in range(1, 13):
        ga, gb = a.plan(state, ms.goal, traj), b.plan(state, ms.goal, traj)
        for k in ("traj", "cost", "status", "iters"):
            assert np.array_equal(ga[k], gb[k]), (tick, k)
        assert np.array_equal(a.row_counts(), b.row_counts()) and np.array_equal(a.last_goals(), b.last_goals()), tick
        assert (a.neighbour_counts() >= 0).all()
        traj = ga["traj"]; state = next_state_host(traj)
    a.close(); b.close()


def test_instance_where_the_cold_start_alone_called_a_feasible_qp_infeasible():
    """Found by the M = 4 fuzzer in round 6 (seed 9500131, tick 3): twelve agents in a 10 m world, half-second segments.  Agent 10 is handed
    to the interior point, whose COLD start (the only one a handed-over agent got since round 5) converges to gap 1e-11 on a feasible point
    and then loses its factorisation before the step-length test passes -- status 1 where the oracle, and the warm start, find the optimum
    1.27613.  A handed-over agent whose cold start gives up on a feasible point with a closed gap now gets the warm start's second opinion;
    agents whose cold start diverges (infeasible QPs) still pay one start.  Pins statuses, costs and plans against the oracle's."""
    import lsc_planner_amd as L
    from lsc_planner_amd.mission import Mission
    from lsc_planner_amd.planner import PlannerConfig
    from tolerances import COST_ATOL, COST_RTOL, FUZZ_TRAJ_ATOL_HALF_SECOND
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fuzz_found_m4_cold_start_9500131.npz"))
    L.load_library(4)
    ms = Mission(d["state"][:, :3].copy(), d["goal"].copy(), d["wmin"], d["wmax"], d["radius"], d["dw"], d["vmax"], d["amax"], d["vnom"], name="replay")
    pl = L.SwarmPlanner(ms, PlannerConfig(goal_mode="static", dt=0.5, horizon=2.0, solver="active_set"))
    assert pl.M == 4
    pl.plan(d["state"], d["goal"], d["traj"])               # (sequence number 1 takes the current-velocity model: only to move it on)
    pl.iterations_total(reset=True)
    g = pl.plan(d["state"], d["goal"], d["traj"])
    st = pl.solver_stats()
    pl.close()
    assert st["handed_over"] >= 1, st
    assert np.array_equal(g["status"], d["ostatus"]) and (g["status"] == 0).all(), g["status"]
    assert (np.abs(g["cost"] - d["ocost"]) <= COST_RTOL * np.abs(d["ocost"]) + COST_ATOL).all(), np.abs(g["cost"] - d["ocost"])
    assert np.abs(g["traj"].astype(np.float64) - d["otraj"]).max() <= FUZZ_TRAJ_ATOL_HALF_SECOND


def test_simulator_flies_a_large_swarm_the_same_with_and_without_the_lists(L, tmp_path):
    """The C++ host side (lsc_sim: device-resident fused ticks, safety ratio on the device, result CSV) with a 640-agent swarm: 25 ticks with
    the neighbour lists forced on write the same result CSV as 25 ticks with round 5's walks over all agents."""
    import csv
    import subprocess
    from test_gpu_sim import SIM, _write_mission
    ms = L.random_swarm(640, world=(-14, -14, 0, 14, 14, 4.0), seed=8)
    mp = tmp_path / "random640.json"
    _write_mission(str(mp), ms)
    outs = []
    for name, env in (("lists", dict(LSC_NEIGH_ALWAYS="1")), ("walks", dict(LSC_NO_NEIGHBOUR_LISTS="1"))):
        d = tmp_path / name
        d.mkdir()
        r = subprocess.run([SIM, "--mission", str(mp), "--csv", str(d), "--quiet", "--max-iter", "25", "--reset-threshold", "0.15"],
                           capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
        assert r.returncode in (0, 1, 2), r.stdout + r.stderr            # (25 ticks do not finish the mission: the run ends at --max-iter)
        assert "total flight time:" in r.stdout or "max" in (r.stdout + r.stderr).lower(), r.stdout + r.stderr
        rows = list(csv.reader(open(d / "result_LSC_640agents.csv")))
        assert rows[0][:15] == "id,t,px,py,pz,vx,vy,vz,ax,ay,az,planning_time,qp_cost,planning_report,size".split(",")
        # (everything but the wall-clock column planning_time, as strings: positions, velocities, accelerations, costs, reports)
        outs.append([[v for j, v in enumerate(r_) if j % 15 != 11] for r_ in rows[1:]])
    same = outs[0] == outs[1]                                # (a plain bool: pytest's diff of two 300 KB tables takes minutes)
    assert len(outs[0]) >= 48 and len(outs[0][0]) == 640 * 14 and same
