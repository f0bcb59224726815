#!/usr/bin/env python3
"""Single-GPU runs of the BASELINE.json configurations other than the bench headline (those are parity-test cases, not
bench lines): device-resident tick loop, one JSON line per configuration.

    python tools/config_runs.py [--ticks 100] [--only circle20,circle64,forest256,random1024]

  circle20    20-agent circle swap, empty map, mode/goal prior_based           (BASELINE configs[1] geometry)
  circle64    the bench headline workload                                      (configs[2])
  forest256   BASELINE configs[3] as SURVEY 8(d)#4 writes it: 256 agents in world/simple_forest.bt, world [-5,5]^2 x [0,2.5],
              seed 20260928; EDT + SFC path, mode/goal static
  forest256p  the same with mode/goal prior_based (the reference's default): priority rule + grid A* + line-of-sight goal
  forest256x4 / forest256x4p   a roomier variant, NOT a BASELINE config: the same 256 agents' worth of swarm (seed 7) in the forest
              tiled 2 x 2 (20 x 20 x 2.5 m; 67 x 67 x 9 search grid: the goal planner's long-search stress case)
  random1024  1024-agent random swarm, empty 40 x 40 x 5 m world               (configs[4] on one GPU)
Needs a GPU; nothing here touches oracle/ or /root/reference.  The forest occupancy comes from the committed leaf
fixture (lsc_planner_amd/data/simple_forest_leaves.npz) written out as a .bt file and read back by the product's own reader.
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def forest_tiles(tiles):
    from lsc_planner_amd.maps import forest_leaves, write_bt
    leaves, res = forest_leaves()
    per = int(round(10.0 / res))           # the fixture covers [-5, 5]^2
    out = []
    for tx in range(tiles):
        for ty in range(tiles):
            sh = leaves.copy()
            sh[:, 0] += tx * per
            sh[:, 1] += ty * per
            out.append(sh)
    path = os.path.join(tempfile.mkdtemp(prefix="lsc_forest_"), f"forest_{tiles}x{tiles}.bt")
    write_bt(path, np.concatenate(out), res)
    lo = (-5.0, -5.0, 0.0)
    hi = (-5.0 + 10.0 * tiles, -5.0 + 10.0 * tiles, 2.5)
    return path, lo + hi


def run(name, ms, cfg, ticks, warmup, bt=None):
    import torch
    import lsc_planner_amd as L
    dev = torch.device("cuda", 0)
    pl = L.SwarmPlanner(ms, cfg)
    if bt:
        pl.load_octomap(bt)
    N = ms.qn
    f32 = dict(dtype=torch.float32, device=dev)
    states = [torch.zeros((N, 9), **f32), torch.zeros((N, 9), **f32)]
    states[0][:, :3] = torch.from_numpy(ms.start).to(dev)
    goal = torch.from_numpy(ms.goal).to(dev).contiguous()
    trajs = [torch.zeros((N, 90), **f32), torch.zeros((N, 90), **f32)]
    cost = torch.zeros(N, dtype=torch.float64, device=dev)
    status = torch.zeros(N, dtype=torch.int32, device=dev)
    iters = torch.zeros(N, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    seq = 0
    failed = 0

    def tick():
        nonlocal seq
        seq += 1
        pl.tick_device_fused(states[0], goal, trajs[0], trajs[1], states[1], cost, status, iters, seq, stream)
        states.reverse()
        trajs.reverse()

    for _ in range(warmup):
        tick()
    torch.cuda.synchronize()
    pl.iterations_total(reset=True)
    pl.set_timing(True)
    t0 = time.perf_counter()
    for _ in range(ticks):
        tick()
        failed += 0   # statuses are read once at the end: no host sync inside the timed loop
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    k = pl.kernel_times_ms(0)
    kg = pl.kernel_times_ms(3) if bt and cfg.goal_mode == "prior_based" else np.zeros(0)
    kc = pl.kernel_times_ms(4) if bt else np.zeros(0)
    it = pl.iterations_total(reset=False)
    st = status.cpu().numpy()
    dist = float(np.linalg.norm(states[0][:, :3].cpu().numpy() - ms.goal, axis=1).mean())
    line = {"config": name, "agents": N, "ticks": ticks, "warmup": warmup, "agent_replans_per_s": round(N * ticks / el, 1),
            "ms_per_tick": round(1e3 * el / ticks, 4), "plan_kernel_ms": {"mean": round(float(k.mean()), 4),
                                                                          "p99": round(float(np.percentile(k, 99)), 4)},
            "goal_kernel_ms": {"mean": round(float(kg.mean()), 4), "p99": round(float(np.percentile(kg, 99)), 4)} if len(kg) else None,
            "corridor_kernel_ms": {"mean": round(float(kc.mean()), 4)} if len(kc) else None,
            "mean_ip_iterations": round(it / (N * ticks), 2),
            "status_last_tick": {int(s): int((st == s).sum()) for s in np.unique(st)},
            "active_lsc_rows_mean": float(pl.row_counts().mean()), "reference_rows_per_agent": 27 * (N - 1),
            "mean_distance_to_goal_m": round(dist, 3)}
    pl.close()
    print(json.dumps(line), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ticks", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--only", default="circle20,circle64,forest256,forest256p,forest256x4,forest256x4p,random1024")
    a = ap.parse_args()
    import lsc_planner_amd as L
    from lsc_planner_amd.planner import PlannerConfig
    want = a.only.split(",")
    if "circle20" in want:
        run("circle20", L.circle_swap(20, 8.0), PlannerConfig(goal_mode="prior_based"), a.ticks, a.warmup)
    if "circle64" in want:
        run("circle64", L.circle_swap(64, 8.0), PlannerConfig(goal_mode="prior_based"), a.ticks, a.warmup)
    for name, mode in (("forest256", "static"), ("forest256p", "prior_based")):
        if name in want:
            sys.path.insert(0, ROOT)
            import bench
            ms, bt = bench.forest256_mission(L)
            run(name, ms, PlannerConfig(goal_mode=mode, use_octomap=True, reset_threshold=0.15), a.ticks, a.warmup, bt=bt)
    for name, mode in (("forest256x4", "static"), ("forest256x4p", "prior_based")):
        if name in want:
            bt, world = forest_tiles(2)
            wmin, wmax = np.asarray(world[:3], np.float32), np.asarray(world[3:], np.float32)
            dist, kmin, res = L.edt_from_bt(bt, wmin, wmax)
            ms = L.random_swarm(256, world=world, seed=7, edt=dist, edt_key_min=kmin, edt_res=res)
            run(name, ms, PlannerConfig(goal_mode=mode, use_octomap=True), a.ticks, a.warmup, bt=bt)
    if "random1024" in want:
        run("random1024", L.random_swarm(1024), PlannerConfig(goal_mode="prior_based"), a.ticks, a.warmup)


if __name__ == "__main__":
    main()
