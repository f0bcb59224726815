"""-m gpu: what round 2 added to the path, through the C ABI.

  * second pass (rows in HBM) for agents whose LSC rows outgrow the LDS capacity: the reference never drops a row
    (src/traj_optimizer.cpp:437-466), so "overflow" must be solved, not reported;
  * the native agent-sharded tick (lsc_comm_init / lsc_tick_device_sharded / lsc_replan_tick_all): the exchange the
    reference performs in MultiSyncSimulator::update (src/multi_sync_simulator.cpp:297-303) as one in-place RCCL
    all-gather.  One GPU here, so the communicator has world size 1 -- the collective code path (librccl bound at run
    time, ncclCommInitRank, ncclAllGather on the tick's stream) still executes on hardware.
Same tolerances as test_gpu_parity.py.
"""
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


def _start(ms):
    N = ms.qn
    state = np.zeros((N, 9), np.float32)
    state[:, :3] = ms.start
    return state, np.zeros((N, 3, 30), np.float32)


def test_row_capacity_overflow_is_solved_by_the_second_pass(L, oracle):
    """A dense ring with the LDS capacity forced down to 2 rows per control point and pruning off: every agent
    overflows the first pass; the second pass (rows in HBM) must return what an unconstrained context and the oracle
    return -- status 3 no longer exists."""
    from lsc_planner_amd.planner import next_state_host
    ms = L.circle_swap(12, 0.9)
    small = L.SwarmPlanner(ms, L.PlannerConfig(max_rows_per_cp=2, prune=False))
    roomy = L.SwarmPlanner(ms, L.PlannerConfig(prune=False))
    sw = oracle_swarm(oracle, ms)
    state, traj = _start(ms)
    for tick in range(1, 9):
        g, r = small.plan(state, ms.goal, traj, want_constraints=True), roomy.plan(state, ms.goal, traj)
        sw.stale[:] = traj
        o = sw.tick(state, ms.goal, traj, tick, want_lsc=True)
        assert (g["status"] != 3).all(), tick
        assert np.array_equal(g["status"], o["status"]) and np.array_equal(g["status"], r["status"]), tick
        assert np.array_equal(g["normal"], o["normal"]) and np.array_equal(g["d"], o["d"]), tick
        assert (small.row_counts() == 27 * 11).all(), tick          # all 27 (N-1) rows were kept
        ok = o["status"] == 0
        assert (np.abs(g["cost"] - o["cost"])[ok] <= COST_RTOL * np.abs(o["cost"])[ok] + COST_ATOL).all(), tick
        assert np.abs(g["traj"] - o["traj"]).max() <= TRAJ_ATOL, tick
        assert np.abs(g["traj"] - r["traj"]).max() <= TRAJ_ATOL, tick
        traj = g["traj"]
        state = next_state_host(traj)
    small.close(); roomy.close()


@pytest.mark.parametrize("n", [96, 256])
def test_unpruned_swarms_beyond_the_lds_capacity(L, n):
    """prune = 0 with N - 1 > 64 rows per control point: every agent takes the second pass and must agree with the pruned
    solve (dropping redundant rows does not move the optimum)."""
    from lsc_planner_amd.planner import next_state_host
    R = 8.0 * n / 64
    ms = L.circle_swap(n, R, world=(-R - 2, -R - 2, 0, R + 2, R + 2, 2.5))
    a, b = L.SwarmPlanner(ms, L.PlannerConfig(prune=True)), L.SwarmPlanner(ms, L.PlannerConfig(prune=False))
    state, traj = _start(ms)
    for tick in range(1, 5):
        ga, gb = a.plan(state, ms.goal, traj), b.plan(state, ms.goal, traj)
        assert (ga["status"] == 0).all() and (gb["status"] == 0).all(), tick
        assert (b.row_counts() == 27 * (n - 1)).all()
        assert (np.abs(ga["cost"] - gb["cost"]) <= COST_RTOL * np.abs(gb["cost"]) + COST_ATOL).all(), tick
        assert np.abs(ga["traj"] - gb["traj"]).max() <= TRAJ_ATOL, tick
        traj = ga["traj"]
        state = next_state_host(traj)
    a.close(); b.close()


def test_mixed_passes_in_one_tick(L, oracle):
    """Capacity 6 with pruning on: some agents of a tight ring fit the LDS pass, others spill -- both kinds in one
    launch pair, all against the oracle."""
    from lsc_planner_amd.planner import next_state_host
    ms = L.circle_swap(16, 1.3)
    pl = L.SwarmPlanner(ms, L.PlannerConfig(max_rows_per_cp=6, prune=True))
    sw = oracle_swarm(oracle, ms)
    state, traj = _start(ms)
    seen_small = seen_big = False
    for tick in range(1, 13):
        g = pl.plan(state, ms.goal, traj)
        sw.stale[:] = traj
        o = sw.tick(state, ms.goal, traj, tick)
        assert np.array_equal(g["status"], o["status"]), tick
        ok = o["status"] == 0
        assert (np.abs(g["cost"] - o["cost"])[ok] <= COST_RTOL * np.abs(o["cost"])[ok] + COST_ATOL).all(), tick
        assert np.abs(g["traj"] - o["traj"]).max() <= TRAJ_ATOL, tick
        rows = pl.row_counts()
        seen_small |= bool((rows <= 6).any())
        seen_big |= bool((rows > 27 * 6).any())       # more rows than 27 buckets x 6 can hold: certainly spilled
        traj = g["traj"]
        state = next_state_host(traj)
    assert seen_big
    pl.close()


def test_native_rccl_sharded_tick_equals_the_fused_tick(L):
    """World-size-1 RCCL communicator: lsc_tick_device_sharded (plan -> ncclAllGather in place -> propagate) must be
    bit-identical to the fused single-launch tick, in the default goal mode, on a padded table."""
    import torch
    ms = L.circle_swap(64, 8.0)
    N = 64
    dev = torch.device("cuda", 0)
    f = L.SwarmPlanner(ms, L.PlannerConfig(goal_mode="prior_based"))
    s = L.SwarmPlanner(ms, L.PlannerConfig(goal_mode="prior_based", comm=(1, 0, L.comm_unique_id())))
    assert (s.world, s.rank, s.shard_rows, s.table_rows, s.first, s.count) == (1, 0, N, N, 0, N)
    st0 = np.zeros((N, 9), np.float32); st0[:, :3] = ms.start
    f0 = torch.from_numpy(st0.copy()).to(dev); f1 = torch.zeros_like(f0)
    s0 = torch.from_numpy(st0.copy()).to(dev)
    goal = torch.from_numpy(ms.goal).to(dev)
    fa, fb = torch.zeros((N, 90), device=dev), torch.zeros((N, 90), device=dev)
    sa, sb = torch.zeros((N, 90), device=dev), torch.zeros((N, 90), device=dev)
    mk = lambda dt: torch.zeros(N, dtype=dt, device=dev)
    fc, fs, fi = mk(torch.float64), mk(torch.int32), mk(torch.int32)
    sc, ss, si = mk(torch.float64), mk(torch.int32), mk(torch.int32)
    stream = torch.cuda.current_stream().cuda_stream
    s.set_timing(True)
    for seq in range(1, 31):
        f.tick_device_fused(f0, goal, fa, fb, f1, fc, fs, fi, seq, stream)
        fa, fb = fb, fa
        f0, f1 = f1, f0
        s.tick_device_sharded(s0, goal, sa, sb, sc, ss, si, seq, stream)
        sa, sb = sb, sa
        torch.cuda.synchronize()
        assert torch.equal(fa, sa), seq
        assert torch.equal(f0, s0), seq
        assert torch.equal(fc, sc) and torch.equal(fs, ss), seq
    x = s.kernel_times_ms(2)
    assert len(x) == 30 and (x >= 0).all()
    f.close(); s.close()


def test_replan_tick_all_returns_every_agents_outputs(L):
    """lsc_replan_tick_all (host buffers, what a replicated simulator per rank calls) == lsc_replan_tick, incl. the planned goals."""
    from lsc_planner_amd.planner import next_state_host
    ms = L.circle_swap(10, 2.0)
    a = L.SwarmPlanner(ms, L.PlannerConfig(goal_mode="prior_based"))
    b = L.SwarmPlanner(ms, L.PlannerConfig(goal_mode="prior_based", comm=(1, 0, L.comm_unique_id())))
    state, traj = _start(ms)
    for _ in range(6):
        ga, gb = a.plan(state, ms.goal, traj), b.plan_all(state, ms.goal, traj)
        for k in ("traj", "cost", "status", "iters"):
            assert np.array_equal(ga[k], gb[k]), k
        assert np.array_equal(a.last_goals(), gb["goal"])
        traj = ga["traj"]
        state = next_state_host(traj)
    a.close(); b.close()


def test_comm_call_order_is_checked(L):
    """lsc_comm_init after lsc_set_agents would leave unpadded tables: refused with LSC_ESTATE, not undefined behaviour."""
    import ctypes
    from lsc_planner_amd import _lib
    ms = L.circle_swap(4, 1.0)
    pl = L.SwarmPlanner(ms)
    buf = (ctypes.c_ubyte * _lib.COMM_ID_BYTES).from_buffer_copy(L.comm_unique_id())
    assert pl.L.lsc_comm_init(pl.ctx, 1, 0, buf) == -5
    assert pl.L.lsc_comm_init(pl.ctx, 2, 2, buf) == -1
    pl.close()


def test_config3_256_agents_in_simple_forest_as_written(L, oracle):
    """BASELINE configs[3] exactly as SURVEY 8(d)#4 states it: 256 agents in world/simple_forest.bt, world
    [-5,-5,0,5,5,2.5], starts/goals from numpy default_rng(20260928) (box shrunk by 0.5 m, EDT >= 0.45 m, pairwise
    downwash-scaled distance >= 0.6 m, goals = permutation of a second such set -- L.random_swarm is that sampler; the
    256 agents DO fit the 10 x 10 m world, no tiling needed), reference-default goal mode (grid A* + line-of-sight goal),
    EDT + SFC path.  24 chained ticks; at checkpoints goals and corridor boxes bit-exact and statuses / costs / plans
    within tolerance against the oracle; every tick the swarm planned through two shard contexts (as two ranks hold
    them) equals the unsharded tick bit for bit."""
    from maputil import forest_leaves
    from lsc_planner_amd.planner import next_state_host
    leaves, res = forest_leaves()
    world = (-5, -5, 0, 5, 5, 2.5)
    dm = oracle.DistMap(leaves, res, world[:3], world[3:])
    N = 256
    ms = L.random_swarm(N, world=world, seed=20260928, edt=dm.dist, edt_key_min=dm.key_min, edt_res=res)
    q = ms.start.astype(np.float64).copy(); q[:, 2] /= 2
    D = np.linalg.norm(q[:, None] - q[None], axis=2) + np.eye(N) * 9
    assert D.min() >= 0.6 - 1e-6 and np.abs(ms.start[:, :2]).max() <= 4.5          # the sampler's contract
    cfg = dict(use_octomap=True, goal_mode="prior_based")
    full, s0, s1 = (L.SwarmPlanner(ms, L.PlannerConfig(**cfg)) for _ in range(3))
    for p in (full, s0, s1):
        p.set_distmap(dm.dist, dm.key_min, res)
    s0.set_shard(0, 128); s1.set_shard(128, 128)
    prm = oracle.make_params(world_min=ms.world_min, world_max=ms.world_max, use_sfc=True, obs_f32=True)
    sw = oracle.Swarm(prm, ms.radius, ms.downwash, ms.max_vel, ms.max_acc, ms.nominal_velocity)
    sw.set_distmap(dm)
    state, traj = _start(ms)
    stale = np.zeros_like(traj)
    sfc_prev = np.zeros((N, 5, 6), np.float32)
    checked = 0
    for tick in range(1, 25):
        g = full.plan(state, ms.goal, traj)
        g0, g1 = s0.plan(state, ms.goal, traj), s1.plan(state, ms.goal, traj)
        for k in ("traj", "cost", "status", "sfc"):
            assert np.array_equal(np.concatenate([g0[k], g1[k]]), g[k]), (tick, k)
        assert np.array_equal(np.concatenate([s0.last_goals()[:128], s1.last_goals()[128:]]), full.last_goals()), tick
        assert (g["status"] != 5).all() and (g["status"] != 4).all(), tick
        if tick in (1, 2, 12, 24):
            goals_ref = oracle.goal_prior_based_map(prm, dm, state, ms.goal, traj, tick, ms.radius, ms.downwash)
            assert np.array_equal(full.last_goals(), goals_ref), tick
            # the oracle's persistent per-agent state (optimiser's last trajectory, corridor history) = what the previous tick left
            sw.stale[:] = stale
            sw.sfc[:] = sfc_prev
            sw.sfc_init[:] = 1 if tick == 1 else 0
            o = sw.tick(state, goals_ref, traj, tick, nthreads=16)
            assert np.array_equal(g["sfc"], o["sfc"]), tick
            assert np.array_equal(g["status"], o["status"]), (tick, np.flatnonzero(g["status"] != o["status"]))
            ok = o["status"] == 0
            assert ok.sum() >= N - 16, (tick, int(ok.sum()))
            assert (np.abs(g["cost"] - o["cost"])[ok] <= COST_RTOL * np.abs(o["cost"])[ok] + COST_ATOL).all(), tick
            assert np.abs(g["traj"][ok] - o["traj"][ok]).max() <= TRAJ_ATOL, tick
            checked += 1
        ok = g["status"] == 0
        stale = np.where(ok[:, None, None], g["traj"], stale).astype(np.float32)
        sfc_prev = g["sfc"].copy()
        traj = g["traj"]
        state = next_state_host(traj)
    assert checked == 4
    # the swarm is making progress towards its goals
    assert np.linalg.norm(state[:, :3] - ms.goal, axis=1).mean() < np.linalg.norm(ms.start - ms.goal, axis=1).mean() - 1.0
    full.close(); s0.close(); s1.close()


def test_corridor_history_shift_is_not_reordered(L, oracle):
    """Regression: the shift of the corridor history used to be done by one lane through wave-uniform addresses; the
    compiler issued those reads as scalar loads, which are not ordered against the vector stores that followed, and
    now and then a store of the new box overtook the read of the element it replaced (one float of one box wrong,
    depending on timing, i.e. on the context).  Four contexts on the same inputs must all equal the oracle."""
    from maputil import forest_leaves
    from lsc_planner_amd.planner import next_state_host
    leaves, res = forest_leaves()
    world = (-5, -5, 0, 5, 5, 2.5)
    dm = oracle.DistMap(leaves, res, world[:3], world[3:])
    ms = L.random_swarm(10, world=world, seed=31, edt=dm.dist, edt_key_min=dm.key_min)
    pls = []
    for _ in range(4):
        p = L.SwarmPlanner(ms, L.PlannerConfig(use_octomap=True))
        p.set_distmap(dm.dist, dm.key_min, res)
        pls.append(p)
    prm = oracle.make_params(world_min=ms.world_min, world_max=ms.world_max, obs_f32=True, use_sfc=True)
    sw = oracle.Swarm(prm, ms.radius, ms.downwash, ms.max_vel, ms.max_acc, ms.nominal_velocity)
    sw.set_distmap(dm)
    state, traj = _start(ms)
    for tick in range(1, 13):
        gs = [p.plan(state, ms.goal, traj) for p in pls]
        o = sw.tick(state, ms.goal, traj, tick, nthreads=4)
        for g in gs:
            assert np.array_equal(g["sfc"], o["sfc"]), tick
        traj = gs[0]["traj"]
        state = next_state_host(traj)
    for p in pls:
        p.close()


@pytest.mark.parametrize("n", [160, 1024])
def test_spatial_pre_cull_changes_nothing_but_the_time(L, n):
    """Large swarms drop far obstacles before the GJK (a sphere bound that implies the exact per-row redundancy test) and
    compact the survivors: same rows in the same order, hence bit-identical plans, costs, iteration counts and row counts
    (prune = 3 is prune = 1 without the pre-cull)."""
    from lsc_planner_amd.planner import next_state_host
    ms = L.random_swarm(n, seed=20260929) if n >= 512 else L.random_swarm(n, world=(-8, -8, 0, 8, 8, 2.5), seed=4)
    a, b = L.SwarmPlanner(ms, L.PlannerConfig(prune=1)), L.SwarmPlanner(ms, L.PlannerConfig(prune=3))
    state, traj = _start(ms)
    for tick in range(1, 7):
        ga, gb = a.plan(state, ms.goal, traj), b.plan(state, ms.goal, traj)
        for k in ("traj", "cost", "status", "iters"):
            assert np.array_equal(ga[k], gb[k]), (tick, k)
        assert np.array_equal(a.row_counts(), b.row_counts()), tick
        traj = ga["traj"]
        state = next_state_host(traj)
    assert 0 < a.row_counts().mean() < 27 * (n - 1) / 10
    a.close(); b.close()


@pytest.mark.parametrize("solver", ["active_set", "interior_point"])
@pytest.mark.parametrize("n,world,reset_threshold", [(320, (-12, -12, 0, 12, 12, 3.0), 0.0),
                                                     (600, (-16, -16, 0, 16, 16, 3.0), 0.15)])
def test_throughput_build_agrees_with_the_latency_build(L, n, world, reset_threshold, solver):
    """A shard with more agents than the GPU has CUs runs the 256-lane throughput build (two workgroups per CU, smaller LDS
    row capacity, assembly tables read from L2); max_rows_per_cp = 64 pins the 512-lane latency build.  Same rows, same
    statuses; costs and plans within the parity tolerances (block reductions combine 4 instead of 8 partial sums).
    The second case has more than two workgroups per CU, so the launch is ordered longest-agent-first by the previous
    tick's iterations and rows (lsc_prep_kernel), and runs the build with the disturbance checks compiled in."""
    from lsc_planner_amd.planner import next_state_host
    ms = L.random_swarm(n, world=world, seed=9)
    tp = L.SwarmPlanner(ms, L.PlannerConfig(reset_threshold=reset_threshold, solver=solver))
    lat = L.SwarmPlanner(ms, L.PlannerConfig(max_rows_per_cp=64, reset_threshold=reset_threshold, solver=solver))
    lds, thr = tp.row_capacity()
    assert 0 < thr < lds and lat.row_capacity()[1] == 0
    state, traj = _start(ms)
    for tick in range(1, 11):
        ga, gb = tp.plan(state, ms.goal, traj), lat.plan(state, ms.goal, traj)
        assert np.array_equal(ga["status"], gb["status"]) and (ga["status"] == 0).mean() > 0.98, tick
        assert np.array_equal(tp.row_counts(), lat.row_counts()), tick
        ok = ga["status"] == 0        # an infeasible agent keeps its stale plan in both builds; its cost is not defined
        assert (np.abs(ga["cost"] - gb["cost"])[ok] <= (COST_RTOL * np.abs(gb["cost"]) + COST_ATOL)[ok]).all(), tick
        assert np.abs(ga["traj"] - gb["traj"]).max() <= TRAJ_ATOL, tick
        traj = gb["traj"]
        state = next_state_host(traj)
    tp.close(); lat.close()


def test_throughput_build_through_two_shard_contexts(L):
    """1 200 agents as two ranks would hold them (600 each: more than two workgroups per CU, so each shard runs the
    throughput build with its own launch order and the bounding spheres of all 1 200 agents) against one unsharded context:
    bit for bit."""
    from lsc_planner_amd.planner import next_state_host
    n = 1200
    ms = L.random_swarm(n, world=(-22, -22, 0, 22, 22, 4.0), seed=5)
    cfg = dict(goal_mode="prior_based", reset_threshold=0.15)
    full, s0, s1 = (L.SwarmPlanner(ms, L.PlannerConfig(**cfg)) for _ in range(3))
    s0.set_shard(0, 600)
    s1.set_shard(600, 600)
    state, traj = _start(ms)
    for tick in range(1, 7):
        g, a, b = full.plan(state, ms.goal, traj), s0.plan(state, ms.goal, traj), s1.plan(state, ms.goal, traj)
        assert np.array_equal(np.concatenate([a["traj"], b["traj"]]), g["traj"]), tick
        assert np.array_equal(np.concatenate([a["cost"], b["cost"]]), g["cost"]), tick
        assert np.array_equal(np.concatenate([a["status"], b["status"]]), g["status"]), tick
        traj = g["traj"]
        state = next_state_host(traj)
    for p in (full, s0, s1):
        p.close()


def test_bench_line_contract_and_exchange_paths(tmp_path):
    """bench.py as the driver runs it (short): exactly one JSON line on stdout carrying the contract's fields, `roofline` and
    -- at N = 1 -- `cpu_baseline`; the sharded sequence on one GPU through the native communicator and through the
    torch.distributed fallback both report themselves in `rccl`."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    bench = os.path.join(ROOT, "bench.py")

    def run(*flags):
        r = subprocess.run([sys.executable, bench, "--steps", "12", "--warmup", "4", "--sweep-agents", "0", *flags],
                           capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
        assert len(lines) == 1, r.stdout[:500]
        return json.loads(lines[0])

    d = run()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 12 and d["warmup"] == 4 and d["value"] > 1e4 and d["vs_baseline"] is None
    assert abs(d["value"] - 64 * 12 / (d["ms_per_step"] * 12e-3)) < 0.01 * d["value"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(d["roofline"]) and d["roofline"]["avg_launch_ms"] <= d["ms_per_step"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(d["cpu_baseline"]) and d["cpu_baseline"]["kind"] == "port"
    assert "workload" in d["config"] and "64-agent generated circle swap" in d["config"]["workload"]
    native = run("--unfused", "--no-cpu-baseline", "--no-latency-leg")
    assert native["rccl"]["native"] is True and native["rccl"]["world_size"] == 1 and native["rccl"]["exchange_us_per_tick"]["mean"] > 0
    fallback = run("--unfused", "--torch-exchange", "--no-cpu-baseline", "--no-latency-leg")
    assert fallback["rccl"]["native"] is False and "torch.distributed" in fallback["rccl"]["collective"]
    # (same kernels, another exchange path: the two lines must be of one size.  Not a tight bound -- 12 steps at a 0.04 ms tick, where the
    #  exchange's launch overhead is a fifth of the tick: observed 995 k vs 1 208 k)
    assert abs(fallback["value"] - native["value"]) < 0.5 * native["value"]
