"""Host-side logic that needs no GPU: ABI surface, device-GJK header arithmetic on the host, missions, sharding."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT


def test_library_exports_every_declared_symbol():
    from lsc_planner_amd import _lib
    L = _lib.load_library()                       # loads without a GPU; compute calls are not made here
    hdr = open(os.path.join(ROOT, "include", "lsc_planner_amd.h")).read()
    declared = set(re.findall(r"\b(lsc_[a-z0-9_]+)\s*\(", hdr)) - {"lsc_ctx"}
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(L, name)


def test_create_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import lsc_planner_amd as L
    with pytest.raises(L.LscError):
        L.SwarmPlanner(L.circle_swap(4, 1.0))


def test_product_never_imports_the_oracle():
    """Nothing under lsc_planner_amd/ may import, include, link or execute anything under oracle/."""
    forbidden = ("from oracle", "import oracle", "lsc_oracle.h", "liblsc_oracle", "oracle/", "orc_")
    for dirpath, _, files in os.walk(os.path.join(ROOT, "lsc_planner_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h", "Makefile")):
                txt = open(os.path.join(dirpath, f)).read()
                for pat in forbidden:
                    assert pat not in txt, (f, pat)
    # the helper scripts outside tests/ do not use it either (the fuzzers that do live under tests/)
    for f in os.listdir(os.path.join(ROOT, "tools")):
        if f.endswith((".py", ".sh")):
            txt = open(os.path.join(ROOT, "tools", f)).read()
            for pat in ("from oracle", "import oracle", "liblsc_oracle", "lsc_oracle.h"):
                assert pat not in txt, (f, pat)


def test_device_gjk_header_matches_oracle_on_host(oracle, gjk_golden):
    so = os.path.join(ROOT, "tests", "native", "libgjk_host_check.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.dirname(so)])
    H = ctypes.CDLL(so)
    dp, ip, fp = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_float)
    pts = np.ascontiguousarray(gjk_golden["pts"].astype(np.float64))
    n = len(pts)
    v = np.zeros((n, 3)); d = np.zeros(n); nv = np.zeros(n, np.int32)
    H.gjkhdr_batch(pts.ctypes.data_as(dp), n, v.ctypes.data_as(dp), d.ctypes.data_as(dp), nv.ctypes.data_as(ip))
    assert np.array_equal(d, gjk_golden["dist"]) and np.array_equal(v, gjk_golden["v"]) and np.array_equal(nv, gjk_golden["nvrtx"])
    rng = np.random.default_rng(2)
    for t in range(300):
        a = rng.normal(size=3) * 3
        init = (a[:, None] + np.outer(rng.normal(size=3), np.linspace(0, 1, 30)) + rng.normal(size=(3, 30)) * 0.02).astype(np.float32)
        obs = ((a + rng.normal(size=3) * 1.5)[:, None] + np.outer(rng.normal(size=3), np.linspace(0, 1, 30))).astype(np.float32)
        ra, ro = 0.15, float(np.float32(0.15))
        dw = (2.0 * ra + 2.0 * ro) / (ra + ro)
        nrm, dd = oracle.lsc_pair(init, obs, ra, ro, 2.0, 2.0)
        for m in range(5):
            pa = np.ascontiguousarray(init[:, m * 6:(m + 1) * 6].T); po = np.ascontiguousarray(obs[:, m * 6:(m + 1) * 6].T)
            n2 = np.zeros(3, np.float32); d2 = np.zeros(6)
            H.lschdr_segment(pa.ctypes.data_as(fp), po.ctypes.data_as(fp), ctypes.c_double(dw), ctypes.c_double(ro + ra),
                             n2.ctypes.data_as(fp), d2.ctypes.data_as(dp))
            assert np.array_equal(n2, nrm[m]) and np.array_equal(d2, dd[m])


def test_mission_loader_and_generators(tmp_path):
    import json
    import lsc_planner_amd as L
    doc = {"quadrotors": {"crazyflie": {"max_vel": [1, 1, 1], "max_acc": [2, 2, 1], "radius": 0.15, "nominal_velocity": 1.0, "downwash": 2.0}},
           "world": [{"dimension": [-5, -5, 0, 5, 5, 2.5]}],
           "agents": [{"type": "crazyflie", "cid": 1, "start": [1.0, 0.1, 0.4], "goal": [-1, 0, 0.4]},
                      {"type": "crazyflie", "cid": 2, "start": [-1, 0, 0.4], "goal": [1, 0, 0.4]}], "obstacles": []}
    p = tmp_path / "m.json"
    p.write_text(json.dumps(doc))
    ms = L.load_mission(str(p))
    assert ms.qn == 2 and ms.start.dtype == np.float32 and ms.start[0, 1] == np.float32(0.1)
    assert ms.max_acc[0, 2] == 1.0 and ms.world_max[2] == np.float32(2.5)
    flat = L.load_mission(str(p), world_dimension=2, world_z_2d=0.7)      # src/mission.cpp:88-112
    assert (flat.start[:, 2] == np.float32(0.7)).all() and (flat.goal[:, 2] == np.float32(0.7)).all()
    assert np.array_equal(flat.start[:, :2], ms.start[:, :2])
    doc["world"].append(doc["world"][0])
    p.write_text(json.dumps(doc))
    with pytest.raises(ValueError):
        L.load_mission(str(p))
    c = L.circle_swap(64)
    assert c.qn == 64 and np.allclose(c.goal[:, :2], -c.start[:, :2]) and (c.goal[:, 2] == 1).all()
    assert abs(np.linalg.norm(c.start[5, :2]) - 8.0) < 1e-5
    r1, r2 = L.random_swarm(40, seed=7), L.random_swarm(40, seed=7)
    assert np.array_equal(r1.start, r2.start) and np.array_equal(r1.goal, r2.goal)
    q = r1.start.astype(np.float64).copy(); q[:, 2] /= 2
    D = np.linalg.norm(q[:, None] - q[None], axis=2) + np.eye(40) * 9
    assert D.min() >= 0.6 - 1e-6


def test_shard_bounds_cover_all_agents():
    from lsc_planner_amd.sharded import shard_bounds, shard_rows, table_rows
    for n in (1, 4, 5, 20, 64, 65, 1024):
        for w in (1, 2, 3, 4, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == n
            for (f0, c0), (f1, _) in zip(spans, spans[1:]):
                assert f0 + c0 == f1 or (c0 == 0 and f0 == n)
            # every block starts at a multiple of the padded block size: that is what makes the exchange one in-place
            # equal-sized all-gather
            rows = shard_rows(n, w)
            assert table_rows(n, w) == rows * w >= n
            assert all(f == min(r * rows, n) and c <= rows for r, (f, c) in enumerate(spans))


_WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from oracle import oracle as O
import lsc_planner_amd as L
from lsc_planner_amd.sharded import ShardedSwarm, table_rows
from lsc_planner_amd.planner import next_state_host
rank, world = int(sys.argv[2]), int(sys.argv[3])
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = sys.argv[4]
dist.init_process_group("gloo", rank=rank, world_size=world)
ms = L.circle_swap(5, circle_radius=1.5, world=(-5, -5, 0, 5, 5, 2.5))       # 5 agents: ragged shards (world 4: one rank owns nobody)
prm = O.make_params(world_min=ms.world_min, world_max=ms.world_max, obs_f32=True)
sw = O.Swarm(prm, ms.radius, ms.downwash, ms.max_vel, ms.max_acc, ms.nominal_velocity)
def tick_fn(state, goal, prev, nxt, seq, first, count):
    r = sw.tick(state.numpy(), goal.numpy(), prev.numpy().reshape(-1, 3, 30), seq)      # CPU stand-in for the HIP tick
    nxt[first:first + count] = torch.from_numpy(r["traj"].reshape(-1, 90))[first:first + count]
def prop_fn(traj, state):
    state.copy_(torch.from_numpy(next_state_host(traj.numpy())))
sh = ShardedSwarm(dist, ms.qn, tick_fn, prop_fn)
state = torch.zeros((ms.qn, 9)); state[:, :3] = torch.from_numpy(ms.start)
goal = torch.from_numpy(ms.goal)
rows = table_rows(ms.qn, world)                                               # padded once, at allocation
a, b = torch.zeros((rows, 90)), torch.zeros((rows, 90))
for _ in range(4):
    b.zero_()
    sh.step(state, goal, a, b)
    a, b = b, a
np.save(sys.argv[5] + f"/rank{rank}.npy", a[:ms.qn].numpy())
dist.destroy_process_group()
'''


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_stepping_gloo_equals_single_process(oracle, tmp_path, world):
    """gloo runs of the sharded loop (5 agents: 3+2 over two ranks, 2+2+1+0 over four) == the unsharded oracle run."""
    import socket
    import lsc_planner_amd as L
    from lsc_planner_amd.planner import next_state_host
    from conftest import oracle_swarm
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    w = tmp_path / "worker.py"
    w.write_text(_WORKER)
    procs = [subprocess.Popen([sys.executable, str(w), ROOT, str(r), str(world), str(port), str(tmp_path)]) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=240) == 0
    ms = L.circle_swap(5, circle_radius=1.5, world=(-5, -5, 0, 5, 5, 2.5))
    sw = oracle_swarm(oracle, ms)
    state = np.zeros((5, 9), np.float32); state[:, :3] = ms.start
    traj = np.zeros((5, 3, 30), np.float32)
    for tick in range(1, 5):
        traj = sw.tick(state, ms.goal, traj, tick)["traj"]
        state = next_state_host(traj)
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"rank{r}.npy").reshape(5, 3, 30), traj), r


_ML_WORKER = r'''
import json, os, sys
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from lsc_planner_amd.sharded import mission_list_ids, mission_list_summary
rank, world = int(sys.argv[2]), int(sys.argv[3])
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = sys.argv[4]
dist.init_process_group("gloo", rank=rank, world_size=world)
K = 4
ids = mission_list_ids(world, rank, K)
# stand-in numbers for what bench.py's mission_list_leg measures on a GPU: K missions of 64 agents, 20 steps, rank 1 is the slow one
same_device = len(sys.argv) > 6 and sys.argv[6] == "same-device"
try:
    out = mission_list_summary(dist, rank, world, 64 * len(ids), 20, 0.001 * (1 + rank), 0.05 + 0.01 * rank, 0 if same_device else rank, f"uuid-{rank}", failed=rank)
    res = {"ids": ids, "summary": out}
except RuntimeError as e:
    res = {"ids": ids, "error": str(e)}
json.dump(res, open(sys.argv[5] + f"/ml_rank{rank}.json", "w"))
dist.barrier()
dist.destroy_process_group()
'''


def test_mission_list_rank_bookkeeping_gloo(tmp_path):
    """bench.py --gpus 2 --mission-list, the part that needs no GPU: rank r flies missions [r K, (r + 1) K) of the list; rank 0's line is the
    sum of the agent-replans over the SLOWEST rank's time, carries the per-rank values and the largest per-rank tick p99, and refuses a run
    in which two ranks drove the same device (outer loop: src/multi_sync_simulator_node.cpp:43-70)."""
    import json
    import socket
    w = tmp_path / "ml_worker.py"
    w.write_text(_ML_WORKER)
    for mode in ("", "same-device"):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        procs = [subprocess.Popen([sys.executable, str(w), ROOT, str(r), "2", str(port), str(tmp_path)] + ([mode] if mode else [])) for r in range(2)]
        for p in procs:
            assert p.wait(timeout=240) == 0
        r0, r1 = (json.load(open(tmp_path / f"ml_rank{r}.json")) for r in range(2))
        assert r0["ids"] == [0, 1, 2, 3] and r1["ids"] == [4, 5, 6, 7]
        if mode:
            assert "ranks share a device" in r0["error"]
            continue
        assert r1["summary"] is None
        sm = r0["summary"]
        assert sm["agents_in_flight"] == 512 and sm["elapsed_s_max_over_ranks"] == 0.002
        assert abs(sm["value"] - 2 * 256 * 20 / 0.002) < 1e-6
        assert sm["per_rank_values"] == [round(256 * 20 / 0.001, 1), round(256 * 20 / 0.002, 1)] and sm["per_rank_devices"] == [0, 1]
        assert abs(sm["tick_p99_ms_max_over_ranks"] - 0.06) < 1e-12 and sm["failed_agents_last_tick"] == 1


def test_mission_list_of_the_bench_is_the_concurrent_missions_family():
    """The missions of `bench.py --mission-list` on one GPU are the ones `concurrent_missions` flies (mission 0 = the headline swarm, the
    others that swarm turned about the vertical axis); with more ranks the list is longer and every mission distinct."""
    import importlib.util
    import lsc_planner_amd as L
    from lsc_planner_amd.sharded import mission_list_ids
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    ms, _ = bench.weak_scaling_mission(L, 1)
    one = bench.mission_list_missions(L, ms, mission_list_ids(1, 0, 4), 4)
    assert one[0] is ms and len(one) == 4
    for m in range(1, 4):
        ref = bench.rotated_mission(L, ms, 2.0 * np.pi * (m / (7.0 * 4) + 0.013 * m), "x")
        assert np.array_equal(one[m].start, ref.start) and np.array_equal(one[m].goal, ref.goal)
    seen = []
    for r in range(2):
        for m in bench.mission_list_missions(L, ms, mission_list_ids(2, r, 4), 8):
            assert m.qn == 64 and not any(np.array_equal(m.start, o) for o in seen)
            seen.append(m.start)
    assert len(seen) == 8


def test_weak_scaling_workload_of_the_bench_is_one_circle_per_rank():
    """bench.py --gpus G: G circles of 64 agents in one world, rank r owning circle r (lsc_comm_info's partitioning), each an
    exact translate of the single-GPU mission and far enough from the others that no row between circles can be active."""
    import importlib.util
    import lsc_planner_amd as L
    from lsc_planner_amd.sharded import shard_bounds
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    one, text1 = bench.weak_scaling_mission(L, 1)
    assert one.qn == 64 and "64-agent generated circle swap (R=8 m" in text1
    for G in (2, 4, 8):
        ms, text = bench.weak_scaling_mission(L, G)
        assert ms.qn == 64 * G and f"{G} x 64-agent" in text
        assert (ms.start >= ms.world_min).all() and (ms.start <= ms.world_max).all()
        assert (ms.goal >= ms.world_min).all() and (ms.goal <= ms.world_max).all()
        for r in range(G):
            first, count = shard_bounds(ms.qn, G, r)
            assert count == 64
            shift = ms.start[first] - one.start[0]
            assert np.allclose(ms.start[first:first + 64] - shift, one.start, atol=1e-5)
            assert np.allclose(ms.goal[first:first + 64] - shift, one.goal, atol=1e-5)
            others = np.delete(np.arange(ms.qn), np.arange(first, first + 64))
            d = np.linalg.norm(ms.start[first:first + 64, None, :2] - ms.start[None, others, :2], axis=2).min()
            assert d >= 14.0 - 1e-3
    big, text8 = bench.weak_scaling_mission(L, 8, single_circle=True)
    assert big.qn == 512 and "512-agent generated circle swap (R=64 m" in text8


def test_the_header_is_plain_c(tmp_path):
    """include/lsc_planner_amd.h is the C ABI a cgo / JNI / ctypes / C++ caller binds: it must compile as C99 and as C++11
    with nothing but itself, and a C program must link against the library's exported names."""
    src = tmp_path / "abi.c"
    src.write_text('#include "lsc_planner_amd.h"\n'
                   'int main(void) { lsc_config c; lsc_default_config(&c); return c.device == 0 && c.dt > 0.19 ? 0 : 1; }\n')
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", inc, str(src)])
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", inc, "-x", "c++", str(src)])
    lib = os.path.join(ROOT, "lsc_planner_amd")
    exe = tmp_path / "abi"
    subprocess.check_call(["gcc", "-std=c99", "-I", inc, str(src), "-o", str(exe), "-L", lib, "-llsc_hip", "-Wl,-rpath," + lib])
    assert subprocess.run([str(exe)], timeout=120).returncode == 0     # lsc_default_config needs no GPU


def test_ctypes_config_struct_matches_the_header(tmp_path):
    """lsc_planner_amd/_lib.py::LscConfig mirrors `lsc_config` field by field: same size, same offset of every field."""
    import ctypes
    from lsc_planner_amd._lib import LscConfig
    names = [f[0] for f in LscConfig._fields_]
    src = tmp_path / "layout.c"
    body = "".join('    printf("%s %%zu\\n", offsetof(lsc_config, %s));\n' % (n, n) for n in names)
    src.write_text('#include <stddef.h>\n#include <stdio.h>\n#include "lsc_planner_amd.h"\n'
                   'int main(void) {\n    printf("sizeof %zu\\n", sizeof(lsc_config));\n' + body + '    return 0;\n}\n')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    assert int(out["sizeof"]) == ctypes.sizeof(LscConfig)
    for n in names:
        assert int(out[n]) == getattr(LscConfig, n).offset, n


def test_the_four_segment_library_exports_the_same_abi():
    """liblsc_hip_m4.so (same sources, -DLSC_SEGMENTS=4: dt 0.5 / horizon 2.0 of src/param.cpp:66-67) loads and says so."""
    from lsc_planner_amd import _lib
    L4 = _lib.load_library(4)
    assert L4.lsc_segments() == 4 and _lib.load_library(5).lsc_segments() == 5
    for name in _lib.EXPORTS:
        assert hasattr(L4, name), name


def test_result_csv_reader_reads_the_reference_s_layout(tmp_path):
    """MultiSyncReplayer::readCSVFile (src/multi_sync_replayer.cpp:53-114): 15 columns per agent, 6 per obstacle, counts from the
    header's "id" / "obs_id" cells, radius from the `size` column, make span = time of the last record.  `lsc_sim --replay` needs
    no GPU."""
    import subprocess
    sim = os.path.join(ROOT, "lsc_planner_amd", "lsc_sim")
    assert os.path.exists(sim), "lsc_sim not built"
    hdr_a = "id,t,px,py,pz,vx,vy,vz,ax,ay,az,planning_time,qp_cost,planning_report,size"
    hdr_o = "obs_id,t,px,py,pz,size"
    rows = [",".join([hdr_a, hdr_a, hdr_o])]
    for i in range(4):
        t = 0.1 * i
        a0 = [0, t, 1.0 + t, 2.0, 0.5, 1, 0, 0, 0, 0, 0, 0.001, 3.5, 0, 0.15]
        a1 = [1, t, -1.0, 2.0 - 2 * t, 0.5, 0, -2, 0, 0, 0, 0, 0.001, 1.5, 0, 0.2]
        ob = [0, t, 5.0, 5.0, 1.0, 0.3]
        rows.append(",".join(repr(float(v)) if isinstance(v, float) else str(v) for v in a0 + a1 + ob))
    f = tmp_path / "result_LSC_2agents.csv"
    f.write_text("\n".join(rows) + "\n")
    r = subprocess.run([sim, "--replay", str(f)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.strip().splitlines()
    head = lines[0].split()
    assert head[:9] == ["replay:", "agents", "2", "obstacles", "1", "records", "4", "makeSpan", "0.3"], head
    assert abs(float(head[10]) - (0.3 + 0.6)) < 1e-6                        # |dp| summed over both agents (float32 points)
    a0, a1 = lines[1].split(), lines[2].split()
    assert a0[:4] == ["agent", "0", "radius", "0.15"] and abs(float(a0[5]) - 1.3) < 1e-6
    assert a1[:4] == ["agent", "1", "radius", "0.2"] and abs(float(a1[6]) - 1.4) < 1e-6
    bad = subprocess.run([sim, "--replay", str(tmp_path / "missing.csv")], capture_output=True, text=True, timeout=60)
    assert bad.returncode == 3 and "invalid csv file" in bad.stderr
