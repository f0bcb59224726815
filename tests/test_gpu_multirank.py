"""-m gpu, and only on a box with at least two GPUs: the product's NATIVE multi-rank path with more than one rank.

One process per GPU (tests/multirank_worker.py), each with its own context: lsc_comm_init -> lsc_tick_device_sharded must equal
the fused single-GPU tick bit for bit on every rank (64 agents on 2 and 4 ranks; 5 agents on 4 ranks: ragged shards, one rank
owning nobody; 64, 5 and 1024 agents on the target node's 8 ranks), lsc_replan_tick_all must hand every rank all N outputs, and lsc_safety_ratio's all-reduce the swarm's minimum.
On a 1-GPU lease every test here skips; on the driver's 8-GPU box it is the first evidence that RCCL saw N ranks
(until now world sizes > 1 only ran through gloo with the torch fallback)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _n_gpus():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 0


def _single_gpu_reference(N, ticks_dev=12, ticks_host=5):
    import torch
    import lsc_planner_amd as L
    from lsc_planner_amd.planner import next_state_host
    from multirank_worker import swarm_and_config
    ms, cfg = swarm_and_config(L, N)
    dev = torch.device("cuda", 0)
    pl = L.SwarmPlanner(ms, L.PlannerConfig(**cfg))
    st = np.zeros((N, 9), np.float32); st[:, :3] = ms.start
    s0 = torch.from_numpy(st.copy()).to(dev); s1 = torch.zeros_like(s0)
    goal = torch.from_numpy(ms.goal).to(dev).contiguous()
    a, b = torch.zeros((N, 90), device=dev), torch.zeros((N, 90), device=dev)
    cost = torch.zeros(N, dtype=torch.float64, device=dev)
    status = torch.zeros(N, dtype=torch.int32, device=dev)
    iters = torch.zeros(N, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    out = dict(trajs=[], states=[], costs=[], stats=[], h_traj=[], h_cost=[], h_status=[], h_goal=[], h_min=[])
    for seq in range(1, ticks_dev + 1):
        pl.tick_device_fused(s0, goal, a, b, s1, cost, status, iters, seq, stream)
        a, b = b, a
        s0, s1 = s1, s0
        torch.cuda.synchronize()
        out["trajs"].append(a.cpu().numpy().copy()); out["states"].append(s0.cpu().numpy().copy())
        out["costs"].append(cost.cpu().numpy().copy()); out["stats"].append(status.cpu().numpy().copy())
    pl.close()
    pl = L.SwarmPlanner(ms, L.PlannerConfig(**cfg))
    state, traj = st.copy(), np.zeros((N, 3, 30), np.float32)
    for _ in range(ticks_host):
        g = pl.plan(state, ms.goal, traj)
        _, _, mn = pl.safety_ratio([0.0, 0.1])
        out["h_traj"].append(g["traj"].copy()); out["h_cost"].append(g["cost"].copy()); out["h_status"].append(g["status"].copy())
        out["h_goal"].append(pl.last_goals().copy()); out["h_min"].append(mn)
        traj = g["traj"]
        state = next_state_host(traj)
    pl.close()
    return {k: np.array(v) for k, v in out.items()}


# (world 1: the worker itself, on any box.  8 ranks = the target node: 64 agents (8 per rank), 5 agents (three ranks own nobody) and the
#  1024 agents of BASELINE configs[4], where a whole-swarm context takes the throughput build and a rank's 128 agents the latency
#  build: both sides pin the latency build there -- multirank_worker.swarm_and_config -- so that the comparison stays bit for bit;
#  that the two builds agree within the parity tolerances is test_throughput_build_agrees_with_the_latency_build's job)
@pytest.mark.parametrize("world,N", [(1, 64), (2, 64), (4, 64), (4, 5), (2, 5), (8, 64), (8, 5), (1, 1024), (8, 1024)])
def test_native_rccl_with_more_than_one_rank(world, N, tmp_path):
    if _n_gpus() < world:
        pytest.skip(f"needs {world} GPUs on this box (has {_n_gpus()})")
    ref = _single_gpu_reference(N)
    token = str(tmp_path / "token")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "multirank_worker.py"), str(r), str(world), str(N), str(tmp_path), token],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
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
    shard = -(-N // world)
    for r in range(world):
        Z = np.load(tmp_path / f"rank{r}.npz")
        first = min(r * shard, N)
        count = min(shard, N - first)
        assert list(Z["info"]) == [world, r, shard, shard * world, first, count], (r, Z["info"])
        # after the in-place all-gather every rank holds ALL N new trajectories and ideal states: the fused single-GPU tick's
        assert np.array_equal(Z["trajs"], ref["trajs"]), r
        assert np.array_equal(Z["states"], ref["states"]), r
        sl = slice(first, first + count)          # costs / statuses of the device-resident tick: the rank's entries only
        assert np.array_equal(Z["costs"][:, sl], ref["costs"][:, sl]) and np.array_equal(Z["stats"][:, sl], ref["stats"][:, sl]), r
        # host-buffer form: all N outputs on every rank, the safety accounting's all-reduce(min)
        for k in ("h_traj", "h_cost", "h_status", "h_goal"):
            assert np.array_equal(Z[k], ref[k]), (r, k)
        assert np.array_equal(Z["h_min"], ref["h_min"]), r
    if N == 5 and world >= 4:
        # the ranks that own nobody took part in every collective (4 ranks: rank 3; 8 ranks: shards of one agent, ranks 5-7 empty)
        for r in range(-(-N // shard), world):
            assert np.load(tmp_path / f"rank{r}.npz")["info"][5] == 0
