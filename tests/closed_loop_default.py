#!/usr/bin/env python3
"""Closed loop of a PERFECTLY symmetric mission under the default solver (the exact active-set solve), held to the oracle at EVERY tick.

    python tests/closed_loop_default.py [--mission multi_simple4|circle20] [--ticks 120] [--solver active_set]

Prints, per tick: largest distance to the goal, the statuses that are not 0, whether goals are bit-identical to the oracle's
(goalPlanningWithPriority restated, src/traj_planner.cpp:540-608), the largest plan / relative cost difference to the oracle's tick on the
SAME inputs, and how far the swarm is from its own mirror image.  What profiles/r06_closed_loop_default_solver.log holds; the
assertions live in tests/test_gpu_parity.py::test_symmetric_missions_under_the_default_solver_follow_the_oracle_tick_by_tick.
Needs a GPU.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))      # (this script lives with the tests: it runs the oracle, which tools/ must not)
import lsc_planner_amd as L  # noqa: E402
from lsc_planner_amd.planner import PlannerConfig, next_state_host  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mission", default="multi_simple4")
    ap.add_argument("--ticks", type=int, default=120)
    ap.add_argument("--solver", default="active_set")
    a = ap.parse_args()
    from oracle import oracle as O
    from conftest import golden_mission, oracle_swarm
    if a.mission == "circle20":
        ms = L.circle_swap(20, 8.0)
    else:
        ms = golden_mission(np.load(os.path.join(ROOT, "tests", "golden", "ticks.npz")), a.mission)
    N = ms.qn
    pl = L.SwarmPlanner(ms, PlannerConfig(goal_mode="prior_based", solver=a.solver))
    sw = oracle_swarm(O, ms)
    state = np.zeros((N, 9), np.float32)
    state[:, :3] = ms.start
    traj = np.zeros((N, 3, 30), np.float32)
    still = 0
    for tick in range(1, a.ticks + 1):
        g = pl.plan(state, ms.goal, traj)
        goals = pl.last_goals()
        og = O.goal_prior_based(state, ms.goal, traj, tick)
        sw.stale[:] = traj if tick > 1 else 0
        o = sw.tick(state, og, traj, tick, nthreads=8)
        dtraj = float(np.abs(g["traj"] - o["traj"]).max())
        ok = (g["status"] == 0) & (o["status"] == 0)
        dcost = float((np.abs(g["cost"] - o["cost"]) / np.maximum(1e-12, np.abs(o["cost"])))[ok].max()) if ok.any() else 0.0
        moved = float(np.abs(g["traj"][:, :, 29] - traj[:, :, 29]).max())
        still = still + 1 if moved < 1e-5 else 0
        dist = np.linalg.norm(state[:, :3] - ms.goal, axis=1)
        print(f"tick {tick:3d} max dist to goal {dist.max():7.4f} status!=0 {np.nonzero(g['status'])[0].tolist()} oracle status!=0 "
              f"{np.nonzero(o['status'])[0].tolist()} goals bitwise {bool(np.array_equal(goals, og))} |dtraj| {dtraj:.2e} dcost {dcost:.1e} "
              f"end point moved {moved:.2e} still {still}")
        traj = g["traj"]
        state = next_state_host(traj)
        if np.linalg.norm(state[:, :3] - ms.goal, axis=1).max() < 0.1:
            print("mission finished at tick", tick)
            break
    pl.close()


if __name__ == "__main__":
    main()
