#!/usr/bin/env python3
"""Which grid searches does a tick of the goal planner wait for?  (study behind the reachability pre-pass of lsc_goal_kernel)

    python tools/goal_attempts.py [--config forest256p|forest256x4p] [--ticks 40]

goalPlanningWithPriority (src/traj_planner.cpp:590-600) searches twice: with the higher-priority agents stamped into the grid and,
if that finds nothing, without them.  A search that FAILS pops every cell it can reach -- the order in which it does so has no
observable effect.  This tool flies the configuration and prints, per tick, the expansion counts of the longest searches together with
their flag word (bit 1: the second search ran, i.e. the first one failed) -- how much of a tick's longest search is a failed attempt.
Needs a GPU; nothing here touches oracle/ or /root/reference."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="forest256p")
    ap.add_argument("--ticks", type=int, default=40)
    a = ap.parse_args()
    import lsc_planner_amd as L
    from lsc_planner_amd.planner import PlannerConfig, next_state_host
    if a.config == "forest256p":
        import bench
        ms, bt = bench.forest256_mission(L)
        cfg = PlannerConfig(goal_mode="prior_based", use_octomap=True, reset_threshold=0.15)
    else:
        from config_runs import forest_tiles
        bt, world = forest_tiles(2)
        wmin, wmax = np.asarray(world[:3], np.float32), np.asarray(world[3:], np.float32)
        dist, kmin, res = L.edt_from_bt(bt, wmin, wmax)
        ms = L.random_swarm(256, world=world, seed=7, edt=dist, edt_key_min=kmin, edt_res=res)
        cfg = PlannerConfig(goal_mode="prior_based", use_octomap=True)
    pl = L.SwarmPlanner(ms, cfg)
    pl.load_octomap(bt)
    pl.set_goal_trace(64)
    N = ms.qn
    state = np.zeros((N, 9), np.float32)
    state[:, :3] = ms.start
    traj = np.zeros((N, 3, 30), np.float32)
    tot_max = tot_max_failed = 0
    for t in range(1, a.ticks + 1):
        g = pl.plan(state, ms.goal, traj)
        tr = pl.goal_trace()
        e = tr["expansions"].astype(np.int64)
        fl = tr["flags"].astype(np.int64)
        traj = g["traj"]
        state = next_state_host(traj)
        top = np.argsort(-e)[:6]
        second = (fl & 2) != 0
        tot_max += int(e.max())
        tot_max_failed += int(e.max()) if second[int(np.argmax(e))] else 0
        print(json.dumps({"tick": t, "grid": [int(v) for v in tr["grid_dims"]], "searched": int((e > 0).sum()), "second_attempts": int(second.sum()),
                          "expansions_sum": int(e.sum()), "expansions_sum_of_second_attempts": int(e[second].sum()),
                          "top6": [[int(e[i]), int(fl[i])] for i in top],
                          "longest_without_a_failed_attempt": int(e[~second].max()) if (~second).any() else 0}), flush=True)
    print(json.dumps({"sum_over_ticks_of_the_longest_search": tot_max, "of_which_the_longest_had_a_failed_first_attempt": tot_max_failed}))
    pl.close()


if __name__ == "__main__":
    main()
