#!/usr/bin/env python3
"""Strong-scaling emulation on ONE GPU: the random1024 / forest256 swarm of bench.py --workload planned through G shard
contexts (G = 1, 2, 4, 8; contiguous blocks of ceil(N / G) agents, each seeing the full replicated trajectory table --
exactly what rank r of a G-GPU run executes), one after the other on the same device.  Per shard the device time of
its launches is taken from HIP events; a G-GPU tick lasts as long as its slowest rank plus the all-gather (5-6 us on a
world-size-1 communicator, `python bench.py --unfused`; the xGMI figure is the driver's to measure), so

    projected tick(G) = max over shards (goal + corridor + plan launches) + propagate launch [+ all-gather]

What this shows without an 8-GPU box: where the per-tick latency floor of one workgroup per agent caps the speed-up.
Needs a GPU; nothing here touches oracle/ or /root/reference.

    python tools/shard_emulation.py [--workload random1024|forest256] [--ticks 60] [--static-goal] [--agents 8192 --shards 1,8]

--agents N (random workload only): a seeded random swarm of N agents in a world of random1024's density (40 m x sqrt(N / 1024) square,
5 m high): what weak scaling beyond one GPU's 1024 agents plans per rank.
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="random1024", choices=["random1024", "forest256"])
    ap.add_argument("--ticks", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--static-goal", action="store_true")
    ap.add_argument("--shards", default="1,2,4,8")
    ap.add_argument("--agents", type=int, default=1024)
    a = ap.parse_args()
    import torch
    import lsc_planner_amd as L
    import bench
    dev = torch.device("cuda", 0)
    bt = None
    if a.workload == "random1024":
        half = 20.0 * (a.agents / 1024.0) ** 0.5
        ms = L.random_swarm(a.agents, seed=20260929) if a.agents == 1024 else L.random_swarm(a.agents, world=(-half, -half, 0, half, half, 5), seed=20260929)
    else:
        ms, bt = bench.forest256_mission(L)
    N = ms.qn
    goal_mode = "static" if a.static_goal else "prior_based"
    f32 = dict(dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    for G in [int(g) for g in a.shards.split(",")]:
        rows = -(-N // G)
        pls = []
        for r in range(G):
            p = L.SwarmPlanner(ms, L.PlannerConfig(goal_mode=goal_mode, reset_threshold=0.15, use_octomap=bt is not None))
            if bt:
                p.load_octomap(bt)
            first = min(r * rows, N)
            p.set_shard(first, min(rows, N - first))
            pls.append(p)
        state = torch.zeros((N, 9), **f32)
        state[:, :3] = torch.from_numpy(ms.start).to(dev)
        goal = torch.from_numpy(ms.goal).to(dev).contiguous()
        trajs = [torch.zeros((N, 90), **f32), torch.zeros((N, 90), **f32)]
        cost = torch.zeros(N, dtype=torch.float64, device=dev)
        status = torch.zeros(N, dtype=torch.int32, device=dev)
        iters = torch.zeros(N, dtype=torch.int32, device=dev)
        seq = 0
        for t in range(a.warmup + a.ticks):
            if t == a.warmup:
                torch.cuda.synchronize()
                for p in pls:
                    p.set_timing(True)
            seq += 1
            for p in pls:
                p.tick_device(state, goal, trajs[0], trajs[1], cost, status, iters, seq, stream)
            pls[0].propagate_device(trajs[1], state, stream)
            trajs.reverse()
        torch.cuda.synchronize()
        per = []
        for p in pls:
            k = p.kernel_times_ms(0)
            g = p.kernel_times_ms(3) if bt and goal_mode == "prior_based" else np.zeros(len(k))
            c = p.kernel_times_ms(4) if bt else np.zeros(len(k))
            n = min(len(k), len(g), len(c)) if bt else len(k)
            per.append(k[:n] + (g[:n] if len(g) >= n else 0) + (c[:n] if len(c) >= n else 0))
        n = min(len(x) for x in per)
        tick = np.max(np.stack([x[:n] for x in per]), axis=0)              # slowest shard of every tick
        plan_mean = [round(float(p.kernel_times_ms(0).mean()), 4) for p in pls]
        st = status.cpu().numpy()
        nl = pls[0].neighbour_counts()
        line = {"workload": a.workload if N == 1024 or bt else f"random{N}", "neighbour_lists": nl is not None, "goal_mode": goal_mode, "agents": N, "shards": G, "agents_per_shard": rows, "ticks": a.ticks,
                "slowest_shard_ms": {"mean": round(float(tick.mean()), 4), "p99": round(float(np.percentile(tick, 99)), 4)},
                "plan_kernel_ms_mean_per_shard": plan_mean,
                "agent_replans_per_s_projected_without_collective": round(N / (float(tick.mean()) * 1e-3), 0),
                "speedup_vs_one_shard": None, "failed_agents_last_tick": int((st != 0).sum())}
        if G == 1:
            base = float(tick.mean())
        line["speedup_vs_one_shard"] = round(base / float(tick.mean()), 3)
        print(json.dumps(line), flush=True)
        for p in pls:
            p.close()


if __name__ == "__main__":
    main()
