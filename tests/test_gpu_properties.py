"""-m gpu: size-independent properties at BASELINE sizes (64 / 256 / 1024 agents) and the device-resident loop."""
import numpy as np
import pytest

from tolerances import COST_RTOL, LARGE_SWARM_COST_ATOL, TRAJ_ATOL

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    import lsc_planner_amd as L
    L.load_library()
    return L


def _feasibility(ms, state, traj, prev_state=None):
    """Every planned trajectory satisfies the reference's rows: world box, velocity / acceleration limits (rows
    skipped exactly where the reference skips them), C2 continuity, stop at the horizon, initial state."""
    t = traj.astype(np.float64).reshape(len(traj), 3, 5, 6)
    lo, hi = ms.world_min.astype(np.float64), ms.world_max.astype(np.float64)
    flat = t.reshape(len(traj), 3, 30)
    assert (flat[:, :, 3:] >= lo[None, :, None] - 1e-5).all() and (flat[:, :, 3:] <= hi[None, :, None] + 1e-5).all()
    v = 25.0 * np.diff(t, axis=3)
    a = 500.0 * np.diff(t, n=2, axis=3)
    vm, am = ms.max_vel[:, :, None, None], ms.max_acc[:, :, None, None]
    vmask = np.ones((5, 5), bool); vmask[0, :2] = False
    amask = np.ones((5, 4), bool); amask[0, 0] = False
    assert (np.abs(v)[:, :, vmask] <= vm.reshape(len(traj), 3, 1) + 2e-3).all()      # float32 storage of c ~ 1e-6 * 25
    assert (np.abs(a)[:, :, amask] <= am.reshape(len(traj), 3, 1) + 5e-2).all()      # ... * 500
    assert np.abs(t[:, :, 1:, 0] - t[:, :, :-1, 5]).max() <= 1e-6                     # position continuity
    assert np.abs((t[:, :, 1:, 1] - t[:, :, 1:, 0]) - (t[:, :, :-1, 5] - t[:, :, :-1, 4])).max() <= 2e-6
    assert np.abs(t[:, :, 4, 5] - t[:, :, 4, 4]).max() <= 1e-6 and np.abs(t[:, :, 4, 5] - t[:, :, 4, 3]).max() <= 1e-6
    assert np.abs(t[:, :, 0, 0] - state[:, :3]).max() <= 1e-6


def _min_separation(traj, downwash=2.0):
    """Bernstein control polygons of every pair stay >= 2r apart at the shared control points is NOT required; what the
    LSC guarantees is separation of the trajectories: sample them."""
    from math import comb
    N = len(traj)
    t = traj.astype(np.float64).reshape(N, 3, 5, 6)
    s = np.linspace(0, 1, 6)
    B = np.stack([comb(5, i) * s ** i * (1 - s) ** (5 - i) for i in range(6)], 0)       # [6 ctrl][6 samples]
    p = np.einsum("nkmi,is->nkms", t, B).reshape(N, 3, -1)
    p[:, 2] /= downwash
    worst = 9.0
    for j in range(p.shape[2]):
        q = p[:, :, j]
        D = np.linalg.norm(q[:, None] - q[None], axis=2) + np.eye(N) * 9
        worst = min(worst, D.min())
    return worst


@pytest.mark.parametrize("n,radius,ticks", [(64, 8.0, 60), (256, 16.0, 12)])
def test_circle_swap_stays_feasible_and_collision_free(L, n, radius, ticks):
    from lsc_planner_amd.planner import next_state_host
    ms = L.circle_swap(n, radius, world=(-radius - 2, -radius - 2, 0, radius + 2, radius + 2, 2.5))
    pl = L.SwarmPlanner(ms)
    state = np.zeros((n, 9), np.float32); state[:, :3] = ms.start
    traj = np.zeros((n, 3, 30), np.float32)
    for tick in range(1, ticks + 1):
        g = pl.plan(state, ms.goal, traj)
        assert (g["status"] == 0).all(), (tick, np.nonzero(g["status"])[0])
        assert np.isfinite(g["traj"]).all() and np.isfinite(g["cost"]).all()
        _feasibility(ms, state, g["traj"])
        assert _min_separation(g["traj"]) >= 0.3 - 2e-4
        traj = g["traj"]
        state = next_state_host(traj)
    pl.close()


def test_1024_agent_stress_one_tick(L):
    """BASELINE configs[4]: 1024-agent empty-map swarm (seeded sampler of SURVEY 8(d)), two ticks."""
    from lsc_planner_amd.planner import next_state_host
    ms = L.random_swarm(1024, seed=20260929)
    pl = L.SwarmPlanner(ms)
    state = np.zeros((1024, 9), np.float32); state[:, :3] = ms.start
    traj = np.zeros((1024, 3, 30), np.float32)
    for tick in (1, 2, 3):
        g = pl.plan(state, ms.goal, traj)
        assert (g["status"] == 0).all()
        _feasibility(ms, state, g["traj"])
        assert _min_separation(g["traj"]) >= 0.3 - 2e-4
        traj = g["traj"]
        state = next_state_host(traj)
    assert pl.row_counts().max() < 27 * 1023
    pl.close()


def test_relabelling_agents_permutes_the_plans(L):
    """Agents are interchangeable: a permutation of the inputs permutes costs exactly up to solver tolerance
    (bucket order changes the floating-point summation order, nothing else)."""
    rng = np.random.default_rng(4)
    ms = L.circle_swap(24, 3.0)
    perm = rng.permutation(24)
    ms2 = L.Mission(ms.start[perm], ms.goal[perm], ms.world_min, ms.world_max, ms.radius[perm], ms.downwash[perm],
                    ms.max_vel[perm], ms.max_acc[perm], ms.nominal_velocity[perm])
    a, b = L.SwarmPlanner(ms), L.SwarmPlanner(ms2)
    sa = np.zeros((24, 9), np.float32); sa[:, :3] = ms.start
    ta = np.zeros((24, 3, 30), np.float32)
    from lsc_planner_amd.planner import next_state_host
    for _ in range(6):
        ga = a.plan(sa, ms.goal, ta)
        gb = b.plan(sa[perm], ms.goal[perm], ta[perm])
        assert (np.abs(ga["cost"][perm] - gb["cost"]) <= COST_RTOL * np.abs(gb["cost"])).all()
        assert np.abs(ga["traj"][perm] - gb["traj"]).max() <= TRAJ_ATOL
        ta = ga["traj"]; sa = next_state_host(ta)
    a.close(); b.close()


def test_device_resident_loop_equals_host_buffer_ticks(L):
    """lsc_tick_device + lsc_propagate_device (what bench.py times) == lsc_replan_tick + host propagation, bitwise."""
    import torch
    from lsc_planner_amd.planner import next_state_host
    ms = L.circle_swap(64, 8.0)
    N = 64
    h, d = L.SwarmPlanner(ms), L.SwarmPlanner(ms)
    dev = torch.device("cuda", 0)
    state_h = np.zeros((N, 9), np.float32); state_h[:, :3] = ms.start
    traj_h = np.zeros((N, 3, 30), np.float32)
    state = torch.from_numpy(state_h.copy()).to(dev)
    goal = torch.from_numpy(ms.goal).to(dev)
    a, b = torch.zeros((N, 90), device=dev), torch.zeros((N, 90), device=dev)
    cost = torch.zeros(N, dtype=torch.float64, device=dev)
    status = torch.zeros(N, dtype=torch.int32, device=dev)
    iters = torch.zeros(N, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for seq in range(1, 16):
        g = h.plan(state_h, ms.goal, traj_h)
        traj_h = g["traj"]; state_h = next_state_host(traj_h)
        d.tick_device(state, goal, a, b, cost, status, iters, seq, st)
        d.propagate_device(b, state, st)
        a, b = b, a
        torch.cuda.synchronize()
        assert np.array_equal(a.cpu().numpy().reshape(N, 3, 30), traj_h)
        assert np.array_equal(state.cpu().numpy(), state_h)
        assert np.array_equal(cost.cpu().numpy(), g["cost"])
    h.close(); d.close()


def test_dense_sweep_kernel_bitwise_vs_plan_kernel_dump(L):
    import torch
    from lsc_planner_amd.planner import next_state_host
    ms = L.circle_swap(40, 5.0)
    N = 40
    pl = L.SwarmPlanner(ms)
    state = np.zeros((N, 9), np.float32); state[:, :3] = ms.start
    traj = np.zeros((N, 3, 30), np.float32)
    dev = torch.device("cuda", 0)
    for seq in (1, 2, 3):
        nrm = torch.zeros((N, N - 1, 5, 3), device=dev)
        dd = torch.zeros((N, N - 1, 5, 6), dtype=torch.float64, device=dev)
        pl.sweep_device(torch.from_numpy(state).to(dev), torch.from_numpy(traj.reshape(N, 90)).to(dev), seq, nrm, dd,
                        torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        g = pl.plan(state, ms.goal, traj, want_constraints=True)
        assert np.array_equal(nrm.cpu().numpy(), g["normal"]) and np.array_equal(dd.cpu().numpy(), g["d"])
        traj = g["traj"]; state = next_state_host(traj)
    pl.close()


def test_sharded_contexts_cover_the_swarm(L):
    """Two shard contexts (as two ranks would hold) reproduce the unsharded tick."""
    from lsc_planner_amd.planner import next_state_host
    ms = L.circle_swap(30, 4.0)
    full, s0, s1 = L.SwarmPlanner(ms), L.SwarmPlanner(ms), L.SwarmPlanner(ms)
    s0.set_shard(0, 17); s1.set_shard(17, 13)
    state = np.zeros((30, 9), np.float32); state[:, :3] = ms.start
    traj = np.zeros((30, 3, 30), np.float32)
    for _ in range(5):
        g = full.plan(state, ms.goal, traj)
        g0, g1 = s0.plan(state, ms.goal, traj), s1.plan(state, ms.goal, traj)
        assert np.array_equal(np.concatenate([g0["traj"], g1["traj"]]), g["traj"])
        assert np.array_equal(np.concatenate([g0["cost"], g1["cost"]]), g["cost"])
        traj = g["traj"]; state = next_state_host(traj)
    full.close(); s0.close(); s1.close()


def test_fused_single_launch_tick_equals_host_buffer_ticks(L):
    """lsc_tick_device_fused (goal planning + plan + next ideal state in ONE launch, what bench.py times on one GPU)
    == lsc_replan_tick + host-side propagation, bitwise, in mode/goal = prior_based."""
    import torch
    from lsc_planner_amd.planner import next_state_host
    ms = L.circle_swap(64, 8.0)
    N = 64
    cfg = L.PlannerConfig(goal_mode="prior_based")
    h, d = L.SwarmPlanner(ms, cfg), L.SwarmPlanner(ms, cfg)
    dev = torch.device("cuda", 0)
    state_h = np.zeros((N, 9), np.float32); state_h[:, :3] = ms.start
    traj_h = np.zeros((N, 3, 30), np.float32)
    s0 = torch.from_numpy(state_h.copy()).to(dev); s1 = torch.zeros_like(s0)
    goal = torch.from_numpy(ms.goal).to(dev)
    a, b = torch.zeros((N, 90), device=dev), torch.zeros((N, 90), device=dev)
    cost = torch.zeros(N, dtype=torch.float64, device=dev)
    status = torch.zeros(N, dtype=torch.int32, device=dev)
    iters = torch.zeros(N, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for seq in range(1, 41):
        g = h.plan(state_h, ms.goal, traj_h)
        goals_h = h.last_goals()
        traj_h = g["traj"]; state_h = next_state_host(traj_h)
        d.tick_device_fused(s0, goal, a, b, s1, cost, status, iters, seq, st)
        a, b = b, a
        s0, s1 = s1, s0
        torch.cuda.synchronize()
        assert np.array_equal(d.last_goals(), goals_h), seq
        assert np.array_equal(a.cpu().numpy().reshape(N, 3, 30), traj_h), seq
        assert np.array_equal(s0.cpu().numpy(), state_h), seq
        assert np.array_equal(cost.cpu().numpy(), g["cost"]), seq
    h.close(); d.close()


def test_ticks_are_deterministic_across_contexts_and_runs(L):
    """Same inputs -> bit-identical plans, costs and iteration counts, from a fresh context and on a repeated call (no
    atomics or launch-order dependence anywhere on the path; the tie rules of the reductions are fixed)."""
    from lsc_planner_amd.planner import next_state_host
    ms = L.circle_swap(48, 5.0)
    runs = []
    for rep in range(2):
        pl = L.SwarmPlanner(ms, L.PlannerConfig(goal_mode="prior_based"))
        state = np.zeros((48, 9), np.float32); state[:, :3] = ms.start
        traj = np.zeros((48, 3, 30), np.float32)
        out = []
        for tick in range(30):
            g = pl.plan(state, ms.goal, traj)
            if tick == 10:
                pl.planner_seq -= 1
                g2 = pl.plan(state, ms.goal, traj)             # the same tick again on the same context
                assert np.array_equal(g["traj"], g2["traj"]) and np.array_equal(g["cost"], g2["cost"])
                assert np.array_equal(g["iters"], g2["iters"])
            out.append((g["traj"].copy(), g["cost"].copy(), g["iters"].copy()))
            traj = g["traj"]; state = next_state_host(traj)
        pl.close()
        runs.append(out)
    for a, b in zip(*runs):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


def test_1024_agent_mission_against_the_oracle(L, oracle):
    """BASELINE configs[4] at full size over a stretch of the mission (mode/goal prior_based): statuses equal, plans within
    tolerance at checkpoints (the oracle runs the 1024 QPs of a tick on all host cores in ~0.2 s)."""
    from lsc_planner_amd.planner import next_state_host
    ms = L.random_swarm(1024, seed=20260929)
    pl = L.SwarmPlanner(ms, L.PlannerConfig(goal_mode="prior_based"))
    prm = oracle.make_params(world_min=ms.world_min, world_max=ms.world_max, obs_f32=True)
    sw = oracle.Swarm(prm, ms.radius, ms.downwash, ms.max_vel, ms.max_acc, ms.nominal_velocity)
    state = np.zeros((1024, 9), np.float32); state[:, :3] = ms.start
    traj = np.zeros((1024, 3, 30), np.float32)
    stale = np.zeros_like(traj)
    for tick in range(1, 61):
        g = pl.plan(state, ms.goal, traj)
        if tick in (1, 2, 15, 30, 60):
            sw.stale[:] = stale
            goals = oracle.goal_prior_based(state, ms.goal, traj, tick)     # the oracle's own goals; the GPU's must equal them
            assert np.array_equal(pl.last_goals(), goals), tick
            o = sw.tick(state, goals, traj, tick, want_lsc=False, nthreads=32)
            assert np.array_equal(g["status"], o["status"]), tick
            ok = o["status"] == 0
            # absolute floor: both solvers stop at a primal residual of 1e-9 * (world extent = 20 m), which moves a
            # near-zero cost of an agent that has almost arrived by a few 1e-8
            dc = np.abs(g["cost"] - o["cost"])[ok]
            bad = dc > COST_RTOL * np.abs(o["cost"])[ok] + LARGE_SWARM_COST_ATOL
            assert not bad.any(), (tick, dc[bad], o["cost"][ok][bad], g["iters"][ok][bad])
            assert np.abs(g["traj"] - o["traj"]).max() <= TRAJ_ATOL, tick
        stale = np.where((g["status"] == 0)[:, None, None], g["traj"], stale).astype(np.float32)
        traj = g["traj"]
        state = next_state_host(traj)
    pl.close()
