#!/usr/bin/env python3
"""Plan-kernel time over six missions (three circle swaps, three random swarms; 48-128 agents, 200 ticks each through the host-buffer
ABI): total kernel time of ticks 21..200, the sum over ticks of the slowest agent's interior-point iterations, and their ratio --
microseconds per iteration of the agent a tick waits for.  The solver's path is sensitive to rounding, so one mission is a noisy A/B;
six are not.  LSC_HIP_LIB=<other build> python tools/multi_eval.py  runs the same missions through another build of the library.
Needs a GPU; nothing here touches oracle/ or /root/reference."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lsc_planner_amd as L                                              # noqa: E402
from lsc_planner_amd.planner import PlannerConfig, next_state_host       # noqa: E402


SOLVER = os.environ.get("LSC_SOLVER", "active_set")      # LSC_SOLVER=interior_point: the interior point alone (rounds 1-4)


def run(ms, ticks=200):
    N = ms.qn
    pl = L.SwarmPlanner(ms, PlannerConfig(goal_mode="prior_based", reset_threshold=0.15, solver=SOLVER))
    state = np.zeros((N, 9), np.float32)
    state[:, :3] = ms.start
    traj = np.zeros((N, 3, 30), np.float32)
    pl.set_timing(True)
    worst = []
    global FAILED, ARRIVED
    for _ in range(ticks):
        g = pl.plan(state, ms.goal, traj)
        worst.append(int(g["iters"].max()))
        FAILED += int((g["status"] != 0).sum())
        traj = g["traj"]
        state = next_state_host(traj)
    T = pl.kernel_times_ms(0)[:ticks] * 1e3
    st = pl.solver_stats()
    for k in st:
        STATS[k] = STATS.get(k, 0) + st[k]
    ARRIVED += int((np.linalg.norm(state[:, :3] - ms.goal, axis=1) < 0.1).sum())
    pl.close()
    return T[20:].sum(), float(np.sum(worst[20:])), float(np.percentile(T[20:], 99))


FAILED = ARRIVED = 0
STATS = {}


def main():
    tot_t = tot_i = 0.0
    out = []
    for name, ms in [("circle64", L.circle_swap(64, 8.0)), ("circle48", L.circle_swap(48, 6.0)),
                     ("circle80", L.circle_swap(80, 10.0, world=(-12, -12, 0, 12, 12, 2.5))),
                     ("random64a", L.random_swarm(64, world=(-6, -6, 0, 6, 6, 2.5), seed=11)),
                     ("random64b", L.random_swarm(64, world=(-5, -5, 0, 5, 5, 2.5), seed=12)),
                     ("random128", L.random_swarm(128, world=(-8, -8, 0, 8, 8, 2.5), seed=13))]:
        f0 = FAILED
        t, i, p99 = run(ms)
        tot_t += t
        tot_i += i
        out.append("%s %.0f us / %d = %.2f (p99 tick %.0f us, failed %d)" % (name, t, i, t / i, p99, FAILED - f0))
    lib = os.path.basename(os.environ.get("LSC_HIP_LIB", "liblsc_hip.so"))
    print("%s solver %s: kernel time %.1f ms, sum of tick-max iterations %d, %.2f us per tick-max iteration, failed agent-ticks %d, agents at their goal after 200 ticks %d | %s"
          % (lib, SOLVER, tot_t / 1e3, tot_i, tot_t / tot_i, FAILED, ARRIVED, " ; ".join(out)) + " | " + str(STATS))


if __name__ == "__main__":
    main()
