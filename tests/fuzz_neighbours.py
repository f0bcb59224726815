#!/usr/bin/env python3
"""Seeded fuzzer of the neighbour lists of large swarms (lsc_planner_amd/csrc/lsc_neigh.hip): random swarm sizes, world densities, agent
radii / downwash / speed limits, priority thresholds, goal modes, disturbance checks with agents pushed off their plans, grid cell sizes --
the context WITH lists against the context without any cull (prune = 3), every tick: trajectories, costs, statuses, iteration counts, row
counts and goals must be the same bits.

    python tests/fuzz_neighbours.py [--seeds 40] [--first-seed 1] [--ticks 12] [--small]

Needs a GPU.  (Lives with the tests: the three-seed version is tests/test_gpu_round6.py::test_neighbour_list_fuzzer.)"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


SIZES = [512, 600, 777, 1024, 1500]
SMALL = False


def one_seed(L, seed, ticks):
    from lsc_planner_amd.planner import next_state_host
    rng = np.random.default_rng(seed)
    n = int(rng.choice(SIZES))
    density = rng.choice([0.05, 0.13, 0.3, 0.6])                          # agents per cubic metre (random1024: 0.13)
    height = float(rng.choice([2.5, 5.0, 9.0]))
    half = float(np.sqrt(n / density / height) / 2.0)
    sep = float(rng.choice([0.5, 0.7]))
    ms = L.random_swarm(n, world=(-half, -half, 0, half, half, height), seed=int(seed), min_sep=sep)
    if rng.random() < 0.6:
        ms.radius[:] = rng.choice([0.1, 0.15, 0.2], n)
        ms.downwash[:] = rng.choice([1.0, 1.5, 2.0, 3.0], n)
        ms.max_vel[:] = rng.choice([0.5, 1.0, 1.5], n)[:, None]
        ms.max_acc[:] = rng.choice([1.0, 2.0, 4.0], n)[:, None]
    cfg = dict(goal_mode=str(rng.choice(["static", "prior_based"])), priority_dist_threshold=float(rng.choice([0.4, 0.8, 1.5])),
               reset_threshold=float(rng.choice([0.0, 0.15])))
    if SMALL:
        # the in-kernel cull of swarms below the lists' 512 agents, small LDS row capacities (unit list beyond its slots, second pass), M = 4
        cfg["max_rows_per_cp"] = int(rng.choice([0, 0, 3, 12]))
        if rng.random() < 0.25:
            cfg.update(dt=0.5, horizon=2.0)
    env = {"LSC_NEIGH_ALWAYS": "1"}
    if rng.random() < 0.5:
        env["LSC_NEIGH_CELL"] = str(rng.choice([0.4, 0.9, 1.6, 3.0, 7.0, 40.0]))
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    a = L.SwarmPlanner(ms, L.PlannerConfig(prune=1, **cfg))
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    b = L.SwarmPlanner(ms, L.PlannerConfig(prune=3, **cfg))
    state = np.zeros((n, 9), np.float32); state[:, :3] = ms.start
    traj = np.zeros((n, 3, a.SEGV), np.float32)
    push_tick = int(rng.integers(3, ticks)) if cfg["reset_threshold"] > 0 and rng.random() < 0.5 else 0
    units_seen, without = [], 0
    for tick in range(1, ticks + 1):
        if tick == push_tick:
            state[rng.integers(0, n, 3), :3] += np.float32(0.25)
        ga, gb = a.plan(state, ms.goal, traj), b.plan(state, ms.goal, traj)
        for k in ("traj", "cost", "status", "iters"):
            if not np.array_equal(ga[k], gb[k]):
                return f"seed {seed} tick {tick}: {k} differs (n {n}, {cfg}, {env})"
        if not np.array_equal(a.row_counts(), b.row_counts()) or not np.array_equal(a.last_goals(), b.last_goals()):
            return f"seed {seed} tick {tick}: rows / goals differ (n {n}, {cfg}, {env})"
        u = a.neighbour_counts()
        if u is not None:
            units_seen.append(float(np.where(u < 0, 5 * (n - 1), u).mean())); without += int((u < 0).sum())
        traj = ga["traj"]
        state = next_state_host(traj, dt=cfg.get("dt", 0.2))
    a.close(); b.close()
    return (n, float(density), cfg["goal_mode"], cfg["reset_threshold"], env.get("LSC_NEIGH_CELL", "default"),
            round(float(np.mean(units_seen)), 1) if units_seen else f"in-kernel cull, max_rows_per_cp {cfg.get('max_rows_per_cp', 0)}, M {a.M}", without)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=40)
    ap.add_argument("--first-seed", type=int, default=1)
    ap.add_argument("--ticks", type=int, default=12)
    ap.add_argument("--small", action="store_true", help="swarms of 110 .. 500 agents (no lists: the in-kernel cull), small LDS row capacities, M = 4 now and then")
    a = ap.parse_args()
    global SIZES, SMALL
    if a.small:
        SIZES, SMALL = [110, 160, 256, 330, 500], True
    import lsc_planner_amd as L
    bad = 0
    for seed in range(a.first_seed, a.first_seed + a.seeds):
        r = one_seed(L, seed, a.ticks)
        if isinstance(r, str):
            print("MISMATCH", r, flush=True); bad += 1
        else:
            what = f"{r[5]} units per agent listed on average, {r[6]} agent-ticks without a list" if not isinstance(r[5], str) else r[5]
            print(f"seed {seed}: {r[0]} agents, {r[1]} per m^3, goals {r[2]}, reset_threshold {r[3]}, cell {r[4]}: {what}", flush=True)
    print(f"{a.seeds} seeds x {a.ticks} ticks: {bad} mismatching", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
