"""-m gpu: the C++ host side (TrajPlanner facade + headless MultiSyncSimulator, csrc/host) driving the C ABI."""
import csv
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, golden_mission

pytestmark = pytest.mark.gpu

SIM = os.path.join(ROOT, "lsc_planner_amd", "lsc_sim")
# The golden mission multi_simple4 is PERFECTLY symmetric.  The active-set solve (lsc_sim's default) returns the exact optimum, the swarm
# stays symmetric to the last float32 bit and ties in the priority rule for good (the exact oracle does the same, tick by tick:
# test_gpu_parity.py::test_symmetric_missions_under_the_default_solver_follow_the_oracle_tick_by_tick); an interior point leaves 1e-6 m of
# agent-dependent noise that breaks the tie.  lsc_sim notices the deadlock (the reference's own bookkeeping, src/traj_planner.cpp:396-409) and
# applies the reference's remedy, multisim/max_noise (launch/simulation.launch:47: 0.02), so the DEFAULT run ends.  Only the tests whose
# subject IS the interior point (its 55-tick mission, the instrumented kernel) name it.
INTERIOR_POINT = ["--solver", "interior_point"]


def _write_mission(path, ms):
    doc = {"quadrotors": {"crazyflie": {"max_vel": list(map(float, ms.max_vel[0])), "max_acc": list(map(float, ms.max_acc[0])),
                                        "radius": float(ms.radius[0]), "nominal_velocity": float(ms.nominal_velocity[0]),
                                        "downwash": float(ms.downwash[0])}},
           "world": [{"dimension": [float(v) for v in list(ms.world_min) + list(ms.world_max)]}],
           "agents": [{"type": "crazyflie", "cid": i + 1, "start": [float(v) for v in ms.start[i]], "goal": [float(v) for v in ms.goal[i]]}
                      for i in range(ms.qn)], "obstacles": []}
    json.dump(doc, open(path, "w"))


def test_headless_simulator_runs_the_reference_mission(ticks, tmp_path):
    assert os.path.exists(SIM), "lsc_sim not built (python -c 'import __graft_entry__ as g; g.build()')"
    ms = golden_mission(ticks, "multi_simple4")
    mp = tmp_path / "multi_simple4.json"
    _write_mission(str(mp), ms)
    r = subprocess.run([SIM, "--mission", str(mp), "--csv", str(tmp_path), "--quiet"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr                 # 0 = finished without collision
    out = r.stdout
    # default solver, max_noise 0 (launch/testall_empty.launch:47): the swarm gridlocks in the middle (tick ~50), the simulator says so
    # 20 ticks later, applies the goal noise once, and the mission ends
    assert "deadlock: no agent's horizon end point has moved for 20 ticks" in r.stderr and "applying it now" in r.stderr, r.stderr
    t = float(out.split("total flight time:")[1].split()[0])
    assert 14.0 < t < 32.0, t                                      # 10.8 s of flight (oracle mission, interior point) + the wait for the verdict
    ratio = float(out.split("safety ratio between agent:")[1].split()[0])
    assert ratio >= 1.0 - 1e-3
    # result CSV in the reference's schema: 15 columns per agent, 2 record steps per tick
    rows = list(csv.reader(open(tmp_path / "result_LSC_4agents.csv")))
    assert rows[0][:15] == "id,t,px,py,pz,vx,vy,vz,ax,ay,az,planning_time,qp_cost,planning_report,size".split(",")
    assert all(len(r_) == 60 for r_ in rows)
    assert abs(float(rows[1][2]) - 1.0) < 1e-6 and float(rows[1][14]) == 0.15
    summ = list(csv.reader(open(tmp_path / "summary_LSC_4agents.csv")))
    assert len(summ[0]) == 25 and summ[1][3] == "0"


def test_result_csv_round_trip_through_the_reader(ticks, tmp_path):
    """SURVEY 8(f)#3 writer AND reader: a run's result CSV read back by the restatement of MultiSyncReplayer::readCSVFile
    (src/multi_sync_replayer.cpp:53-114, `lsc_sim --replay`) holds the run: agent count, two records per tick, the make span
    and the flown distance the simulator printed, every agent's last record at its goal."""
    ms = golden_mission(ticks, "multi_simple4")
    mp = tmp_path / "multi_simple4.json"
    _write_mission(str(mp), ms)
    r = subprocess.run([SIM, "--mission", str(mp), "--csv", str(tmp_path), "--quiet"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    flight = float(r.stdout.split("total flight time:")[1].split()[0])
    dist = float(r.stdout.split("total distance:")[1].split()[0])
    rp = subprocess.run([SIM, "--replay", str(tmp_path / "result_LSC_4agents.csv")], capture_output=True, text=True, timeout=60)
    assert rp.returncode == 0, rp.stdout + rp.stderr
    lines = rp.stdout.strip().splitlines()
    head = lines[0].split()
    n_rec = int(head[6])
    assert head[2] == "4" and head[4] == "0" and n_rec % 2 == 0
    assert abs(float(head[8]) - (flight - 0.1)) < 0.25           # last record: one record step before the time isFinished() saw
    assert abs(float(head[10]) - dist) <= 1e-3 * dist              # same points, printed with 6 significant digits
    for q in range(4):
        w = lines[1 + q].split()
        assert w[3] == "0.15"
        assert np.linalg.norm(np.array(w[5:8], float) - ms.goal[q]) < 0.15, (q, w)      # (goal threshold 0.1 + the remedy's noise, <= 0.02 per axis)


def test_symmetric_mission_gridlocks_under_the_exact_optimum_and_the_simulator_says_so(ticks, tmp_path):
    """configs[0] as testall_empty.launch flies it (max_noise 0), default solver, remedy OFF: the two head-on agents stop 1.15 m from their
    goals and stay there until --max-iter -- the exact oracle's behaviour too (test_gpu_parity.py) --, lsc_sim prints the deadlock line ONCE and
    names the reference's remedy; --on-deadlock ignore is the reference's silence.  The interior point's 1e-6 m of noise breaks the tie:
    that run is the oracle's 55-tick mission."""
    ms = golden_mission(ticks, "multi_simple4")
    mp = tmp_path / "multi_simple4.json"
    _write_mission(str(mp), ms)
    r = subprocess.run([SIM, "--mission", str(mp), "--max-iter", "150", "--on-deadlock", "report"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr                 # nobody collides while waiting
    assert r.stderr.count("deadlock: no agent's horizon end point has moved for 20 ticks") == 1 and "--max-noise 0.02" in r.stderr
    assert "applying it now" not in r.stderr
    assert float(r.stdout.split("total flight time:")[1].split()[0]) == 0.0          # isFinished() never held
    last = [ln for ln in r.stdout.splitlines() if ln.startswith("[MultiSyncSimulator] iter ")][-1]
    assert "max dist to goal 1.150" in last and "qp failures 0" in last, last
    q = subprocess.run([SIM, "--mission", str(mp), "--max-iter", "150", "--on-deadlock", "ignore", "--quiet"], capture_output=True, text=True, timeout=300)
    assert q.returncode == 0 and "deadlock" not in q.stderr and float(q.stdout.split("total flight time:")[1].split()[0]) == 0.0
    ip = subprocess.run([SIM, "--mission", str(mp), "--quiet"] + INTERIOR_POINT, capture_output=True, text=True, timeout=300)
    assert ip.returncode == 0 and "deadlock" not in ip.stderr, ip.stderr
    assert 5.0 < float(ip.stdout.split("total flight time:")[1].split()[0]) < 16.0   # oracle mission: 55 ticks = 10.8 s


def test_default_solver_flies_the_reference_mission_with_the_launch_file_s_noise(ticks, tmp_path):
    """lsc_sim's default solver -- the active-set solve -- on the reference's 4-agent mission with multisim/max_noise = 0.02 as
    launch/simulation.launch sets it (the mission itself is perfectly symmetric, see INTERIOR_POINT): finishes without collision in about the
    oracle's time, for three seeds."""
    ms = golden_mission(ticks, "multi_simple4")
    mp = tmp_path / "multi_simple4.json"
    _write_mission(str(mp), ms)
    for seed in ("3", "7", "11"):
        r = subprocess.run([SIM, "--mission", str(mp), "--quiet", "--max-noise", "0.02", "--noise-seed", seed], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        t = float(r.stdout.split("total flight time:")[1].split()[0])
        assert 5.0 < t < 16.0, (seed, t)
        assert float(r.stdout.split("safety ratio between agent:")[1].split()[0]) >= 1.0 - 1e-3


def test_mission_list_is_flown_back_to_back_like_the_reference_s_node(ticks, tmp_path):
    """lsc_sim --mission a --mission b / --mission-dir DIR: the outer loop of src/multi_sync_simulator_node.cpp:43-70 over
    Param::mission_file_names (src/param.cpp:106-122) -- one summary line per mission, each equal to the mission flown alone."""
    ms = golden_mission(ticks, "multi_simple4")
    d = tmp_path / "missions"
    d.mkdir()
    _write_mission(str(d / "a_simple4.json"), ms)
    ms2 = golden_mission(ticks, "multi_simple4")
    ms2.goal[:, :2] *= 0.8                                        # a second, different mission of the same swarm size
    _write_mission(str(d / "b_simple4_short.json"), ms2)
    noise = ["--max-noise", "0.02", "--noise-seed", "5"]
    alone = []
    for f in ("a_simple4.json", "b_simple4_short.json"):
        o = tmp_path / ("alone_" + f)
        o.mkdir()
        r = subprocess.run([SIM, "--mission", str(d / f), "--csv", str(o), "--quiet"] + noise, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        alone.append(list(csv.DictReader(open(o / "summary_LSC_4agents.csv")))[0])
    o = tmp_path / "list"
    o.mkdir()
    r = subprocess.run([SIM, "--mission-dir", str(d), "--csv", str(o), "--quiet"] + noise, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "mission 1 of 2" in r.stdout and "mission 2 of 2" in r.stdout
    rows = list(csv.DictReader(open(o / "summary_LSC_4agents.csv")))
    assert len(rows) == 2
    for a, b in zip(alone, rows):
        for col in ("total_flight_time", "total_flight_distance", "is_collided", "safety_ratio_agent"):
            assert a[col] == b[col], (col, a[col], b[col])
    assert rows[0]["total_flight_time"] != rows[1]["total_flight_time"]


def test_simulator_trajectory_equals_python_host_layer(ticks, tmp_path):
    """Same mission through the Python harness: the C++ and Python host layers feed the ABI identically."""
    import lsc_planner_amd as L
    from lsc_planner_amd.planner import next_state_host
    ms = golden_mission(ticks, "multi_simple4")
    mp = tmp_path / "m.json"
    _write_mission(str(mp), ms)
    subprocess.run([SIM, "--mission", str(mp), "--csv", str(tmp_path), "--quiet", "--max-iter", "12"], check=False, timeout=300)
    rows = list(csv.reader(open(tmp_path / "result_LSC_4agents.csv")))[1:]
    pl = L.SwarmPlanner(ms, L.PlannerConfig(goal_mode="prior_based"))   # lsc_sim defaults to mode/goal = prior_based
    state = np.zeros((4, 9), np.float32); state[:, :3] = ms.start
    traj = np.zeros((4, 3, 30), np.float32)
    for tick in range(10):
        g = pl.plan(state, ms.goal, traj)
        traj = g["traj"]
        row = rows[2 * tick]                                      # record step at future_time = 0 of this tick
        for q in range(4):
            p = [float(row[15 * q + 2 + k]) for k in range(3)]
            assert np.allclose(p, traj[q, :, 0], atol=2e-6), (tick, q)
        state = next_state_host(traj)
    pl.close()


def test_forest_mission_in_the_default_goal_mode(tmp_path):
    """Octomap world + mode/goal = prior_based (every shipped launch file): .bt -> EDT -> SFC boxes, grid A* goals, QP --
    the 8-agent mission must finish collision-free, and the first ticks equal the Python host layer."""
    import lsc_planner_amd as L
    from lsc_planner_amd.planner import next_state_host
    from maputil import forest_leaves, write_bt
    leaves, res = forest_leaves()
    bt = tmp_path / "forest.bt"
    write_bt(str(bt), leaves, res)
    world = (-5, -5, 0, 5, 5, 2.5)
    dist, kmin, r = L.edt_from_bt(str(bt), np.asarray(world[:3], np.float32), np.asarray(world[3:], np.float32))
    ms = L.random_swarm(8, world=world, seed=21, edt=dist, edt_key_min=kmin, edt_res=r)
    mp = tmp_path / "forest8.json"
    _write_mission(str(mp), ms)
    rr = subprocess.run([SIM, "--mission", str(mp), "--world", str(bt), "--csv", str(tmp_path), "--quiet", "--max-iter", "200"],
                        capture_output=True, text=True, timeout=600)
    assert rr.returncode == 0, rr.stdout + rr.stderr
    ratio = float(rr.stdout.split("safety ratio between agent:")[1].split()[0])
    assert ratio >= 1.0 - 1e-3
    summ = list(csv.reader(open(tmp_path / "summary_LSC_8agents.csv")))
    goal_t, sfc_t, opt_t = float(summ[1][10]), float(summ[1][12]), float(summ[1][13])   # per-phase columns (seconds per agent-plan)
    assert 0 < sfc_t < opt_t < 1e-3 and goal_t > 0
    rows = list(csv.reader(open(tmp_path / "result_LSC_8agents.csv")))[1:]
    pl = L.SwarmPlanner(ms, L.PlannerConfig(goal_mode="prior_based", use_octomap=True))
    pl.load_octomap(str(bt))
    state = np.zeros((8, 9), np.float32); state[:, :3] = ms.start
    traj = np.zeros((8, 3, 30), np.float32)
    for tick in range(8):
        g = pl.plan(state, ms.goal, traj)
        assert (g["status"] == 0).all()
        traj = g["traj"]
        row = rows[2 * tick]
        for q in range(8):
            p = [float(row[15 * q + 2 + k]) for k in range(3)]
            assert np.allclose(p, traj[q, :, 0], atol=2e-6), (tick, q)
        state = next_state_host(traj)
    pl.close()


def test_forest_mission_in_a_planar_world(tmp_path):
    """`--dimension 2 --z-2d 0.7` (world/dimension, world/z_2d): the mission's heights are replaced by z_2d, the goal planner
    searches one layer; the run finishes collision-free, nobody leaves the plane, and the summary says dimension 2."""
    import lsc_planner_amd as L
    from maputil import forest_leaves, write_bt
    leaves, res = forest_leaves()
    bt = tmp_path / "forest.bt"
    write_bt(str(bt), leaves, res)
    world = (-5, -5, 0, 5, 5, 2.5)
    dist, kmin, r = L.edt_from_bt(str(bt), np.asarray(world[:3], np.float32), np.asarray(world[3:], np.float32))
    # starts / goals sampled in the plane itself (a slab that shrinks to z = 0.7), free of the trees there; the mission file
    # then carries other heights, which --dimension 2 must override
    ms = L.random_swarm(8, world=(-5, -5, 0.2, 5, 5, 1.2), seed=21, edt=dist, edt_key_min=kmin, edt_res=r)
    assert (ms.start[:, 2] == np.float32(0.7)).all()
    ms.world_min, ms.world_max = np.asarray(world[:3], np.float32), np.asarray(world[3:], np.float32)
    ms.start[:, 2] = 1.9
    ms.goal[:, 2] = 0.3
    mp = tmp_path / "forest8.json"
    _write_mission(str(mp), ms)
    rr = subprocess.run([SIM, "--mission", str(mp), "--world", str(bt), "--csv", str(tmp_path), "--quiet", "--max-iter", "200",
                         "--dimension", "2", "--z-2d", "0.7"], capture_output=True, text=True, timeout=600)
    assert rr.returncode == 0, rr.stdout + rr.stderr
    summ = list(csv.reader(open(tmp_path / "summary_LSC_8agents.csv")))
    assert summ[1][21] == "2"
    rows = list(csv.reader(open(tmp_path / "result_LSC_8agents.csv")))[1:]
    z = np.array([[float(row[15 * q + 4]) for q in range(8)] for row in rows])
    assert np.abs(z - 0.7).max() < 1e-3, np.abs(z - 0.7).max()


def test_goal_noise_like_the_published_runs(ticks, tmp_path):
    """multisim/max_noise (0.02 in the reference's test launch files, src/mission.cpp:386-395): the desired goals move by at most
    that much per axis, reproducibly with --noise-seed, and the mission still finishes."""
    ms = golden_mission(ticks, "multi_simple4")
    mp = tmp_path / "m.json"
    _write_mission(str(mp), ms)
    outs = []
    for d in ("a", "b", "c"):
        (tmp_path / d).mkdir()
        args = [SIM, "--mission", str(mp), "--csv", str(tmp_path / d), "--quiet"] + ([] if d == "c" else ["--max-noise", "0.02", "--noise-seed", "7"])
        r = subprocess.run(args, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        rows = list(csv.reader(open(tmp_path / d / "result_LSC_4agents.csv")))[1:]
        outs.append(np.array([[float(rows[-1][15 * q + 2 + k]) for k in range(3)] for q in range(4)]))
    assert np.array_equal(outs[0], outs[1])                       # same seed, same run
    shift = outs[0] - outs[2]                                     # a run ends within the goal threshold of the (noisy) goals
    assert np.abs(shift).max() > 1e-4 and np.abs(shift).max() < 0.02 + 0.2, shift


def test_simulator_over_the_native_communicator_writes_the_same_run(ticks, tmp_path):
    """lsc_sim --ranks 1 --comm-file: the multi-GPU form (rendezvous file, lsc_comm_init, lsc_replan_tick_all with its
    RCCL all-gather group) must produce the same result CSV as the plain run.  One GPU here, hence world size 1."""
    ms = golden_mission(ticks, "multi_simple4")
    mp = tmp_path / "m.json"
    _write_mission(str(mp), ms)
    a, b = tmp_path / "plain", tmp_path / "comm"
    a.mkdir(); b.mkdir()
    r1 = subprocess.run([SIM, "--mission", str(mp), "--csv", str(a), "--quiet", "--max-iter", "40"], capture_output=True, text=True, timeout=300)
    r2 = subprocess.run([SIM, "--mission", str(mp), "--csv", str(b), "--quiet", "--max-iter", "40", "--ranks", "1", "--rank", "0",
                         "--comm-file", str(tmp_path / "token")], capture_output=True, text=True, timeout=300)
    assert r1.returncode == 0 and r2.returncode == 0, r1.stderr + r2.stderr
    ra = list(csv.reader(open(a / "result_LSC_4agents.csv")))
    rb = list(csv.reader(open(b / "result_LSC_4agents.csv")))
    assert len(ra) == len(rb) > 40
    skip = {15 * q + 11 for q in range(4)}                          # planning_time column (wall clock)
    for x, y in zip(ra, rb):
        assert [v for i, v in enumerate(x) if i not in skip] == [v for i, v in enumerate(y) if i not in skip]


@pytest.mark.parametrize("world", [2, 8])
def test_simulator_on_several_ranks_writes_the_single_gpu_run(world, ticks, tmp_path):
    """lsc_sim --ranks W, one process per GPU (rendezvous file, lsc_comm_init, lsc_replan_tick_all's RCCL group, the safety accounting's
    all-reduce): rank 0's result CSV must be the plain single-GPU run's.  8 = the target node; the 4-agent mission then leaves ranks
    4-7 without agents.  Skips on a box with fewer GPUs (exchange point: MultiSyncSimulator::update, src/multi_sync_simulator.cpp:249-304)."""
    import torch
    if torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs on this box (has {torch.cuda.device_count()})")
    ms = golden_mission(ticks, "multi_simple4")
    mp = tmp_path / "m.json"
    _write_mission(str(mp), ms)
    a, b = tmp_path / "plain", tmp_path / "ranks"
    a.mkdir(); b.mkdir()
    r1 = subprocess.run([SIM, "--mission", str(mp), "--csv", str(a), "--quiet", "--max-iter", "40"], capture_output=True, text=True, timeout=300)
    assert r1.returncode == 0, r1.stderr
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = [subprocess.Popen([SIM, "--mission", str(mp), "--csv", str(b), "--quiet", "--max-iter", "40", "--ranks", str(world), "--rank", str(r),
                               "--device", str(r), "--comm-file", str(tmp_path / "token")], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    ra = list(csv.reader(open(a / "result_LSC_4agents.csv")))
    rb = list(csv.reader(open(b / "result_LSC_4agents.csv")))
    assert len(ra) == len(rb) > 40
    skip = {15 * q + 11 for q in range(4)}                          # planning_time column (wall clock)
    for x, y in zip(ra, rb):
        assert [v for i, v in enumerate(x) if i not in skip] == [v for i, v in enumerate(y) if i not in skip]


def test_blocked_corridor_seed_stops_the_run_like_the_reference(tmp_path):
    """An agent that starts inside an obstacle's margin: the reference's expandBoxFromPoint throws std::invalid_argument
    out of plan() (include/corridor_constructor.hpp:35-38) and the simulator dies; lsc_sim must fail as loudly instead
    of flying a stale trajectory under planning_report = SUCCESS."""
    import lsc_planner_amd as L
    from maputil import forest_leaves, write_bt
    leaves, res = forest_leaves()
    bt = tmp_path / "forest.bt"
    write_bt(str(bt), leaves, res)
    world = (-5, -5, 0, 5, 5, 2.5)
    dist, kmin, r = L.edt_from_bt(str(bt), np.asarray(world[:3], np.float32), np.asarray(world[3:], np.float32))
    ms = L.random_swarm(4, world=world, seed=5, edt=dist, edt_key_min=kmin, edt_res=r)
    # move agent 2 onto the centre of an occupied voxel
    occ = np.argwhere(dist == 0)
    c = occ[len(occ) // 2]
    ms.start[2] = ((c + kmin - 32768) + 0.5) * r
    mp = tmp_path / "blocked.json"
    _write_mission(str(mp), ms)
    rr = subprocess.run([SIM, "--mission", str(mp), "--world", str(bt), "--quiet", "--max-iter", "5"], capture_output=True, text=True, timeout=300)
    assert rr.returncode == 3, rr.stdout + rr.stderr
    assert "CorridorConstructor" in rr.stderr and "agent 2" in rr.stderr


def test_simulator_in_bvc_mode_finishes_the_mission(ticks, tmp_path):
    """lsc_sim --planner bvc: the alternate planner mode end to end through the C++ host (mode/planner = bvc of
    src/param.cpp:40-45), file name and summary columns as the reference writes them."""
    ms = golden_mission(ticks, "multi_simple4")
    mp = tmp_path / "m.json"
    _write_mission(str(mp), ms)
    r = subprocess.run([SIM, "--mission", str(mp), "--csv", str(tmp_path), "--quiet", "--planner", "bvc"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    ratio = float(r.stdout.split("safety ratio between agent:")[1].split()[0])
    assert ratio >= 1.0 - 1e-3
    summ = list(csv.reader(open(tmp_path / "summary_BVC_4agents.csv")))
    assert summ[1][16] == "BVC" and summ[1][17] == "current_position" and summ[1][18] == "current_posiotion" and summ[1][20] == "orca"


def test_the_one_published_mission_outcome_of_the_reference(tmp_path):
    """log/summary_LSC_16agents.csv is the only outcome the reference publishes for this path: multi_square16.json in
    simple_forest.bt, no collision, minimal safety ratio 1.004-1.010, flight time 21.8 / 22.8 s, flown distance 169.0 /
    169.5 m (two runs with multisim/max_noise 0.02, unseeded -- a soft target).  The same mission file (reference schema)
    and map through lsc_sim in the reference's default goal mode must land in that neighbourhood."""
    from conftest import GOLDEN
    from maputil import forest_leaves, write_bt
    fix = json.load(open(os.path.join(GOLDEN, "multi_square16.json")))
    leaves, res = forest_leaves()
    bt = tmp_path / "simple_forest.bt"
    write_bt(str(bt), leaves, res)
    rr = subprocess.run([SIM, "--mission", os.path.join(GOLDEN, "multi_square16.json"), "--world", str(bt), "--csv", str(tmp_path), "--quiet"],
                        capture_output=True, text=True, timeout=900)
    assert rr.returncode == 0, rr.stdout + rr.stderr                      # finished, no collision
    summ = list(csv.DictReader(open(tmp_path / "summary_LSC_16agents.csv")))[0]
    pub = fix["published_outcome"]
    t_pub = np.mean([p["total_flight_time"] for p in pub]); d_pub = np.mean([p["total_flight_distance"] for p in pub])
    assert float(summ["is_collided"]) == 0 and float(summ["safety_ratio_agent"]) >= 1.0 - 1e-3
    assert abs(float(summ["total_flight_time"]) - t_pub) <= 0.25 * t_pub, summ["total_flight_time"]
    assert abs(float(summ["total_flight_distance"]) - d_pub) <= 0.10 * d_pub, summ["total_flight_distance"]
    print("lsc_sim:", summ["total_flight_time"], "s,", summ["total_flight_distance"], "m; published:", t_pub, "s,", d_pub, "m")


def test_phase_stats_fill_the_planning_time_columns(ticks, tmp_path):
    """lsc_sim --phase-stats: initial_traj_planning_time / lsc_generation_time / traj_optimization_time of the summary
    (PlanningTimeStatistics, include/sp_const.hpp:89-128) come from the instrumented plan kernel; the run itself is the same."""
    ms = golden_mission(ticks, "multi_simple4")
    mp = tmp_path / "m.json"
    _write_mission(str(mp), ms)
    outs = {}
    for tag, extra in (("plain", []), ("stats", ["--phase-stats"])):
        d = tmp_path / tag
        d.mkdir()
        # (the instrumented kernel is the interior point's: both runs name it, so that "the run itself is the same" can be held to the digit)
        r = subprocess.run([SIM, "--mission", str(mp), "--csv", str(d), "--quiet", "--reset-threshold", "0"] + INTERIOR_POINT + extra,
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        outs[tag] = list(csv.DictReader(open(d / "summary_LSC_4agents.csv")))[0]
    plain, stats = outs["plain"], outs["stats"]
    assert plain["total_flight_time"] == stats["total_flight_time"] and plain["total_flight_distance"] == stats["total_flight_distance"]
    assert float(plain["lsc_generation_time"]) == 0.0 and float(plain["initial_traj_planning_time"]) == 0.0
    for col in ("initial_traj_planning_time", "lsc_generation_time", "traj_optimization_time"):
        assert 1e-7 < float(stats[col]) < 1e-3, (col, stats[col])               # microseconds per agent-plan, in seconds
