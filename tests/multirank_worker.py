"""One rank of tests/test_gpu_multirank.py: python multirank_worker.py RANK WORLD N OUT_DIR TOKEN_FILE

Plans its shard of an N-agent circle swap through the NATIVE multi-rank path of the C ABI -- lsc_comm_init, then
lsc_tick_device_sharded (plan kernel -> in-place ncclAllGather -> ideal states), lsc_replan_tick_all and lsc_safety_ratio's
all-reduce -- on GPU `RANK`, and writes what it saw to OUT_DIR/rank<R>.npz.  The rendezvous token travels through a file
(rank 0 writes it): no torch.distributed anywhere, the only communicator is the library's own RCCL one."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def swarm_and_config(L, N):
    """The swarm and the planner configuration of an N-agent case -- shared with the single-GPU reference of the test.
    N > 256: a whole-swarm context would take the 256-lane throughput build (more agents than CUs) while a rank's shard takes the 512-lane
    latency build; an explicit row capacity selects the latency build on both sides (lsc_row_capacity: throughput_rows = 0), so that
    the sharded ticks can be held to the single-GPU ones bit for bit."""
    R = 8.0 * N / 64.0 if N >= 16 else 1.2
    ms = L.circle_swap(N, circle_radius=R, z=1.0, world=(-R - 2, -R - 2, 0, R + 2, R + 2, 2.5))
    cfg = dict(goal_mode="prior_based", reset_threshold=0.15)
    if N > 256:
        cfg["max_rows_per_cp"] = 48
    return ms, cfg


def main():
    rank, world, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    out_dir, token_file = sys.argv[4], sys.argv[5]
    ticks_dev, ticks_host = 12, 5
    import torch
    import lsc_planner_amd as L
    from lsc_planner_amd.planner import next_state_host
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    if rank == 0:
        tok = L.comm_unique_id()
        with open(token_file + ".tmp", "wb") as f:
            f.write(tok)
        os.replace(token_file + ".tmp", token_file)
    else:
        t0 = time.time()
        while not os.path.exists(token_file):
            if time.time() - t0 > 120:
                raise SystemExit("no rendezvous token")
            time.sleep(0.05)
        tok = open(token_file, "rb").read()
    ms, cfg = swarm_and_config(L, N)
    cfg["device"] = rank
    pl = L.SwarmPlanner(ms, L.PlannerConfig(comm=(world, rank, tok), **cfg))
    rows = pl.table_rows
    info = np.array([pl.world, pl.rank, pl.shard_rows, pl.table_rows, pl.first, pl.count])
    # ---- device-resident sharded ticks
    st = np.zeros((N, 9), np.float32); st[:, :3] = ms.start
    s0 = torch.from_numpy(st.copy()).to(dev)
    goal = torch.from_numpy(ms.goal).to(dev).contiguous()
    a, b = torch.zeros((rows, 90), device=dev), torch.zeros((rows, 90), device=dev)
    cost = torch.zeros(N, dtype=torch.float64, device=dev)
    status = torch.full((N,), -7, dtype=torch.int32, device=dev)
    iters = torch.zeros(N, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    trajs, states, costs, stats = [], [], [], []
    for seq in range(1, ticks_dev + 1):
        pl.tick_device_sharded(s0, goal, a, b, cost, status, iters, seq, stream)
        a, b = b, a
        torch.cuda.synchronize()
        trajs.append(a[:N].cpu().numpy().copy()); states.append(s0.cpu().numpy().copy())
        costs.append(cost.cpu().numpy().copy()); stats.append(status.cpu().numpy().copy())
    pl.close()
    # ---- host-buffer form: every rank receives all N outputs; safety accounting with its all-reduce
    # (one communicator per token: the host-buffer leg gets a token of its own)
    tok2 = token_file + ".2"
    if rank == 0:
        t2 = L.comm_unique_id()
        with open(tok2 + ".tmp", "wb") as f:
            f.write(t2)
        os.replace(tok2 + ".tmp", tok2)
    else:
        t0 = time.time()
        while not os.path.exists(tok2):
            if time.time() - t0 > 120:
                raise SystemExit("no second rendezvous token")
            time.sleep(0.05)
        t2 = open(tok2, "rb").read()
    pl = L.SwarmPlanner(ms, L.PlannerConfig(comm=(world, rank, t2), **cfg))
    state, traj = st.copy(), np.zeros((N, 3, 30), np.float32)
    h_traj, h_cost, h_status, h_goal, h_min = [], [], [], [], []
    for _ in range(ticks_host):
        g = pl.plan_all(state, ms.goal, traj)
        ratio, partner, mn = pl.safety_ratio([0.0, 0.1])
        h_traj.append(g["traj"].copy()); h_cost.append(g["cost"].copy()); h_status.append(g["status"].copy())
        h_goal.append(g["goal"].copy()); h_min.append(mn)
        traj = g["traj"]
        state = next_state_host(traj)
    pl.close()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), info=info, trajs=np.array(trajs), states=np.array(states), costs=np.array(costs),
             stats=np.array(stats), h_traj=np.array(h_traj), h_cost=np.array(h_cost), h_status=np.array(h_status), h_goal=np.array(h_goal),
             h_min=np.array(h_min))


if __name__ == "__main__":
    main()
