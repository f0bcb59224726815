#!/usr/bin/env python3
"""Section profile of the goal planner's grid search (lsc_goal_profile) on the 256-agent forest configurations.

    python tools/goal_profile.py [--tiles 2] [--ticks 20] [--seed 7]

Prints, per tick window, the expansions and the shader cycles per expanded node of the slowest searches (the launch lasts as
long as its longest search) and of all searches together, split into findMin / deleteMin / screening / insertions.
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
    ap.add_argument("--tiles", type=int, default=2)
    ap.add_argument("--ticks", type=int, default=20)
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--search", default="auto")
    a = ap.parse_args()
    import lsc_planner_amd as L
    from lsc_planner_amd.planner import PlannerConfig, next_state_host
    from config_runs import forest_tiles
    bt, world = forest_tiles(a.tiles)
    wmin, wmax = np.asarray(world[:3], np.float32), np.asarray(world[3:], np.float32)
    dist, kmin, res = L.edt_from_bt(bt, wmin, wmax)
    ms = L.random_swarm(256, world=world, seed=a.seed, edt=dist, edt_key_min=kmin, edt_res=res)
    pl = L.SwarmPlanner(ms, PlannerConfig(goal_mode="prior_based", use_octomap=True, goal_search=a.search))
    pl.load_octomap(bt)
    N = ms.qn
    state = np.zeros((N, 9), np.float32)
    state[:, :3] = ms.start
    traj = np.zeros((N, 3, 30), np.float32)
    pl.set_timing(True)
    for t in range(1, a.ticks + 1):
        pl.goal_profile(1)
        g = pl.plan(state, ms.goal, traj)
        prof = pl.goal_profile(-1).astype(float)
        tr = pl.goal_trace()
        e = tr["expansions"].astype(float)
        traj = g["traj"]
        state = next_state_host(traj)
        if t <= 3 or t % 5 == 0:
            top = np.argsort(-e)[:8]
            tot = prof[:, :4].sum(1)
            def split(idx):
                n = max(e[idx].sum(), 1.0)
                return {k: round(float(prof[idx, 4 + i].sum() / n), 1) for i, k in enumerate(("find_min", "delete_min", "screening", "insertions"))}
            line = {"tick": t, "grid": [int(v) for v in tr["grid_dims"]], "expansions_mean": round(float(e.mean()), 1), "expansions_max": int(e.max()),
                    "kernel_cycles_max": int(tot.max()), "cycles_per_node_top8": round(float(prof[top, 2].sum() / max(e[top].sum(), 1)), 1),
                    "split_top8": split(top), "split_all": split(np.arange(N)),
                    "general_top8": {"pops_per_node": round(float(prof[top, 9].sum() / max(e[top].sum(), 1)), 4), "cycles_per_pop": round(float(prof[top, 8].sum() / max(prof[top, 9].sum(), 1)), 1),
                                     "inserts_per_node": round(float(prof[top, 11].sum() / max(e[top].sum(), 1)), 4), "cycles_per_insert": round(float(prof[top, 10].sum() / max(prof[top, 11].sum(), 1)), 1)},
                    "phases_mean_cycles": {k: int(prof[:, i].mean()) for i, k in enumerate(("prologue", "grid_setup", "search", "path_los"))},
                    "phases_of_slowest": {k: int(prof[int(np.argmax(tot)), i]) for i, k in enumerate(("prologue", "grid_setup", "search", "path_los"))}}
            print(json.dumps(line), flush=True)
    pl.close()


if __name__ == "__main__":
    main()
