#!/usr/bin/env python3
"""Phase profile of lsc_plan_kernel (instrumented kernel variant, lsc_phase_profile): where a tick's device time goes.

    python tools/phase_profile.py [--agents 64] [--ticks 60] [--from-tick 21] [--static-goal]

Runs the bench mission (generated circle swap, empty map) through the host-buffer ABI, switches the instrumented
variant on at --from-tick and prints the mean per-agent time per phase (100 MHz wall clock, s_memrealtime).
Needs a GPU; output is what profiles/r01_phase_profile_*.log hold.
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lsc_planner_amd as L  # noqa: E402
from lsc_planner_amd.planner import PlannerConfig, next_state_host  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--agents", type=int, default=64)
    ap.add_argument("--radius", type=float, default=0.0, help="circle radius (default 8 m x sqrt(agents/64))")
    ap.add_argument("--ticks", type=int, default=60)
    ap.add_argument("--from-tick", type=int, default=21)
    ap.add_argument("--static-goal", action="store_true")
    ap.add_argument("--random", action="store_true", help="seeded random swarm in a 40 x 40 x 5 m world (BASELINE configs[4]) instead of the circle")
    ap.add_argument("--max-rows-per-cp", type=int, default=0,
                    help="explicit LDS row capacity: a swarm larger than the chip then takes the 512-lane latency build instead of the 256-lane throughput build")
    ap.add_argument("--solver", default="active_set", choices=["active_set", "interior_point"])
    a = ap.parse_args()
    N = a.agents
    R = a.radius or 8.0 * max(1.0, (N / 64.0) ** 0.5)
    ms = L.random_swarm(N, seed=20260929) if a.random else L.circle_swap(N, R, world=(-R - 2, -R - 2, 0, R + 2, R + 2, 2.5))
    pl = L.SwarmPlanner(ms, PlannerConfig(goal_mode="static" if a.static_goal else "prior_based", max_rows_per_cp=a.max_rows_per_cp, solver=a.solver))
    state = np.zeros((N, 9), np.float32)
    state[:, :3] = ms.start
    traj = np.zeros((N, 3, 30), np.float32)
    it_sum = 0
    for tick in range(1, a.ticks + 1):
        if tick == a.from_tick:
            pl.phase_profile(enable=1)
        g = pl.plan(state, ms.goal, traj)
        if tick >= a.from_tick:
            it_sum += int(g["iters"].sum())
        if (g["status"] != 0).any():
            print("tick", tick, "failed agents", np.nonzero(g["status"])[0])
        traj = g["traj"]
        state = next_state_host(traj)
    p = pl.phase_profile(enable=0).astype(np.float64)
    nt = a.ticks - a.from_tick + 1
    tot = p[:, :12].sum(1).mean()                                  # (the entries behind the first twelve are parts of those)
    print(f"{N} agents, ticks {a.from_tick}..{a.ticks}: {tot / nt / 100:.1f} us per tick per agent (instrumented), "
          f"{it_sum / (nt * N):.2f} IP iterations per agent-replan")
    for name, v in zip(pl.PHASES, p.mean(0)):
        print(f"   {name:15s} {v / nt / 100:9.2f} us/tick  {100 * v / tot:5.1f}%   {v / max(it_sum / N, 1) / 100:7.2f} us/iteration")
    pl.close()


if __name__ == "__main__":
    main()
