#!/usr/bin/env python3
"""Closed-loop comparison of the two QP solvers of the fast path (lsc_config.solver): perturbed circle swaps and random swarms flown to
completion, device-resident, under the active-set solve (+ interior point as fallback) and under the interior point alone.  Per solver:
failed agent-ticks (a QP the solver -- in the end always the interior point -- found infeasible: the stale plan is kept), agents that
did not arrive, smallest downwash-scaled distance, plan-kernel time, and the counters of lsc_solver_stats.

    python tools/solver_compare.py [--reps 4] [--ticks 300]

Needs a GPU; nothing here touches oracle/ or /root/reference."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lsc_planner_amd as L                                   # noqa: E402
from lsc_planner_amd.planner import PlannerConfig              # noqa: E402


def fly(ms, solver, ticks, dev):
    n = ms.qn
    pl = L.SwarmPlanner(ms, PlannerConfig(goal_mode="prior_based", reset_threshold=0.15, solver=solver))
    f32 = dict(dtype=torch.float32, device=dev)
    st = [torch.zeros((n, 9), **f32), torch.zeros((n, 9), **f32)]
    st[0][:, :3] = torch.from_numpy(ms.start).to(dev)
    tj = [torch.zeros((n, 90), **f32), torch.zeros((n, 90), **f32)]
    goal = torch.from_numpy(ms.goal).to(dev)
    cost = torch.zeros(n, dtype=torch.float64, device=dev)
    status = torch.zeros(n, dtype=torch.int32, device=dev)
    iters = torch.zeros(n, dtype=torch.int32, device=dev)
    fails = torch.zeros((), dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    radius2 = torch.tensor(ms.radius[:, None] + ms.radius[None, :], device=dev)
    mind = torch.full((), 9.0, dtype=torch.float64, device=dev)
    pl.iterations_total(reset=True)
    pl.set_timing(True)
    done_at = None
    for tick in range(1, ticks + 1):
        pl.tick_device_fused(st[0], goal, tj[0], tj[1], st[1], cost, status, iters, tick, stream)
        st.reverse(); tj.reverse()
        fails += (status != 0).sum()
        if tick % 5 == 0:
            p = st[0][:, :3].double().clone()
            p[:, 2] /= 2.0
            D = torch.cdist(p, p) / radius2 + torch.eye(n, device=dev, dtype=torch.float64) * 9
            mind = torch.minimum(mind, D.min())
            if done_at is None and float((st[0][:, :3] - goal).norm(dim=1).max()) < 0.1:
                done_at = tick
                break
    torch.cuda.synchronize()
    k = pl.kernel_times_ms(0)
    fin = st[0].cpu().numpy()
    out = dict(failed=int(fails.item()), not_arrived=int((np.linalg.norm(fin[:, :3] - ms.goal, axis=1) >= 0.1).sum()), done_at=done_at,
               min_ratio=round(float(mind.item()), 4), kernel_ms=round(float(k.sum()), 2), p99_tick_us=round(1e3 * float(np.percentile(k, 99)), 1),
               ticks=len(k), stats=pl.solver_stats())
    pl.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--ticks", type=int, default=300)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(7)
    missions = []
    for rep in range(a.reps):
        for n, R in ((20, 8.0), (48, 6.0), (64, 8.0), (80, 10.0)):
            ms = L.circle_swap(n, R, world=(-R - 2, -R - 2, 0, R + 2, R + 2, 2.5))
            ms.goal[:, :3] += rng.uniform(0, 0.05, (n, 3)).astype(np.float32)
            ms.start[:, :2] += rng.uniform(-0.05, 0.05, (n, 2)).astype(np.float32)
            missions.append((f"circle{n}#{rep}", ms))
        missions.append((f"random64#{rep}", L.random_swarm(64, world=(-5, -5, 0, 5, 5, 2.5), seed=100 + rep)))
        missions.append((f"random128#{rep}", L.random_swarm(128, world=(-8, -8, 0, 8, 8, 2.5), seed=200 + rep)))
    tot = {s: dict(failed=0, not_arrived=0, kernel_ms=0.0, missions_with_failures=0, min_ratio=9.0) for s in ("active_set", "interior_point")}
    for name, ms in missions:
        row = {}
        for s in tot:
            r = fly(ms, s, a.ticks, dev)
            row[s] = r
            tot[s]["failed"] += r["failed"]; tot[s]["not_arrived"] += r["not_arrived"]; tot[s]["kernel_ms"] += r["kernel_ms"]
            tot[s]["missions_with_failures"] += int(r["failed"] > 0)
            tot[s]["min_ratio"] = min(tot[s]["min_ratio"], r["min_ratio"])
        print(json.dumps({"mission": name, "agents": ms.qn, **row}), flush=True)
    print(json.dumps({"missions": len(missions), "total": tot}), flush=True)


if __name__ == "__main__":
    main()
