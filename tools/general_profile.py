#!/usr/bin/env python3
"""Section profile of lsc_general_kernel (lsc_general_profile): where an interior-point iteration of the alternate planner
modes goes.  64-agent circle swap, device-resident ticks.

    python tools/general_profile.py [--ticks 40] [--modes bvc,collision_constraint,dynamical_limit,gust]

Prints one JSON line per mode: kernel time per tick, iterations, and the sections in microseconds per iteration of the agent
that spent most (a tick lasts as long as its slowest agent).  Needs a GPU; nothing here touches oracle/ or /root/reference.
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHADER_MHZ = 2400.0


def run(name, cfg, ticks, gust):
    import torch
    import lsc_planner_amd as L
    dev = torch.device("cuda", 0)
    ms = L.circle_swap(64, 8.0)
    pl = L.SwarmPlanner(ms, cfg)
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
    bad = 0

    def tick():
        nonlocal seq, bad
        seq += 1
        pl.tick_device_fused(states[0], goal, trajs[0], trajs[1], states[1], cost, status, iters, seq, stream)
        states.reverse()
        trajs.reverse()

    for _ in range(5):
        tick()
    if gust:                                     # push a quarter of the swarm off its plan once: they and whoever sees them
        states[0][::4, :3] += 0.25               # stay on the general path for the rest of the mission
    torch.cuda.synchronize()
    pl.set_timing(True)
    pl.phase_profile(1)
    for _ in range(ticks):
        tick()
        bad += int((status != 0).sum().item())
    torch.cuda.synchronize()
    kp = pl.kernel_times_ms(0)                   # plan kernel + lsc_general_kernel, one timed region per tick
    g = pl.general_profile().astype(np.float64)
    pl.phase_profile(0)
    tot = g[:, :12].sum(axis=1)
    worst = int(np.argmax(tot))
    it = max(g[worst, 12], 1.0)
    sec = {n: round(float(g[worst, k]) / it / SHADER_MHZ, 2) for k, n in enumerate(pl.GENERAL_SECTIONS[:12])}
    line = {"mode": name, "ticks": ticks, "plan_and_general_kernel_ms": {"mean": round(float(kp.mean()), 4), "p99": round(float(np.percentile(kp, 99)), 4)},
            "agents_on_general_path_per_tick": round(float(g[:, 13].sum()) / ticks, 1),
            "iterations_per_solve": round(float(g[:, 12].sum() / max(g[:, 13].sum(), 1.0)), 2),
            "cold_starts_per_solve": round(float(g[:, 15].sum() / max(g[:, 13].sum(), 1.0)), 3),
            "busiest_agent": {"us_per_iteration": round(float(tot[worst]) / it / SHADER_MHZ, 2), "iterations": int(g[worst, 12]),
                              "sections_us_per_iteration": sec},
            "busiest_agent_triangular_solves_alone_us_per_iteration": round(float(g[worst, 14]) / it / SHADER_MHZ, 2),
            "failed_plans": bad}
    pl.close()
    print(json.dumps(line), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ticks", type=int, default=40)
    ap.add_argument("--modes", default="bvc,collision_constraint,dynamical_limit,gust")
    a = ap.parse_args()
    from lsc_planner_amd.planner import PlannerConfig
    want = a.modes.split(",")
    if "bvc" in want:
        run("bvc", PlannerConfig(planner_mode="bvc", goal_mode="prior_based"), a.ticks, False)
    if "collision_constraint" in want:
        run("bvc+collision_constraint", PlannerConfig(planner_mode="bvc", slack_mode="collision_constraint", goal_mode="prior_based"), a.ticks, False)
    if "dynamical_limit" in want:
        run("bvc+dynamical_limit", PlannerConfig(planner_mode="bvc", slack_mode="dynamical_limit", goal_mode="prior_based"), a.ticks, False)
    if "gust" in want:
        run("lsc+gust", PlannerConfig(goal_mode="prior_based", reset_threshold=0.15), a.ticks, True)


if __name__ == "__main__":
    main()
