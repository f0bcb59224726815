#!/usr/bin/env python3
"""Duality-gap tolerance of the plan kernel's interior point: what each setting buys and what it costs.

For gap_tolerance in {1e-9 (the pinned strict mode), 1e-8, 1e-7, 1e-6}:
  * the six missions of tools/multi_eval.py, 200 ticks each through the host-buffer ABI, flown by the STRICT planner; every other
    setting plans the same tick inputs from a context of its own (same warm start: the shifted previous plan is an input), so the
    columns compare solves of identical QPs: kernel time per tick (p50 / p99 / total), tick-max iterations, status agreement,
    max |cost - cost_strict| / (1 + |cost_strict|), max |cost - cost_strict| / |cost_strict|, max |control point - strict|;
  * the 520 agent-ticks with HiGHS verdicts (tests/golden/qp_pin_ticks.npz): status agreement and cost against HiGHS itself,
    in the form of tests/tolerances.py (|d| <= 1e-6 |c| + 1e-8) -- how many instances fall outside it.
Needs a GPU; nothing here touches oracle/ or /root/reference.   python tools/gap_sweep.py [--ticks 200] > profiles/r04_gap_sweep.log
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lsc_planner_amd as L                                              # noqa: E402
from lsc_planner_amd.planner import PlannerConfig, next_state_host       # noqa: E402

TOLS = (1e-9, 1e-8, 1e-7, 1e-6)


def missions():
    return [("circle64", L.circle_swap(64, 8.0)), ("circle48", L.circle_swap(48, 6.0)),
            ("circle80", L.circle_swap(80, 10.0, world=(-12, -12, 0, 12, 12, 2.5))),
            ("random64a", L.random_swarm(64, world=(-6, -6, 0, 6, 6, 2.5), seed=11)),
            ("random64b", L.random_swarm(64, world=(-5, -5, 0, 5, 5, 2.5), seed=12)),
            ("random128", L.random_swarm(128, world=(-8, -8, 0, 8, 8, 2.5), seed=13))]


def fly(ms, ticks, extra):
    N = ms.qn
    pls = [L.SwarmPlanner(ms, PlannerConfig(goal_mode="prior_based", reset_threshold=0.15, gap_tolerance=t, **extra)) for t in TOLS]
    for p in pls:
        p.set_timing(True)
    state = np.zeros((N, 9), np.float32)
    state[:, :3] = ms.start
    traj = np.zeros((N, 3, 30), np.float32)
    acc = [dict(worst=[], dabs=0.0, drel=0.0, dcp=0.0, status_diff=0, iters=0) for _ in TOLS]
    for tick in range(ticks):
        res = [p.plan(state, ms.goal, traj) for p in pls]
        ref = res[0]
        for a, r in zip(acc, res):
            a["worst"].append(int(r["iters"].max()))
            a["iters"] += int(r["iters"].sum())
            a["status_diff"] += int((r["status"] != ref["status"]).sum())
            ok = (r["status"] == 0) & (ref["status"] == 0)
            if ok.any():
                d = np.abs(r["cost"] - ref["cost"])[ok]
                a["dabs"] = max(a["dabs"], float((d / (1.0 + np.abs(ref["cost"][ok]))).max()))
                big = np.abs(ref["cost"][ok]) > 1e-3
                if big.any():
                    a["drel"] = max(a["drel"], float((d[big] / np.abs(ref["cost"][ok][big])).max()))
                a["dcp"] = max(a["dcp"], float(np.abs(r["traj"] - ref["traj"])[ok].max()))
        traj = ref["traj"]
        state = next_state_host(traj)
    out = []
    for a, p in zip(acc, pls):
        T = p.kernel_times_ms(0)[:ticks] * 1e3
        p.close()
        out.append(dict(total_us=float(T[20:].sum()), p50=float(np.percentile(T[20:], 50)), p99=float(np.percentile(T[20:], 99)),
                        tickmax_iters=int(np.sum(a["worst"][20:])), iters=a["iters"], status_diff=a["status_diff"],
                        dcost_over_1pf=a["dabs"], dcost_rel=a["drel"], dcp=a["dcp"]))
    return out


def pins(extra):
    Z = np.load(os.path.join(ROOT, "tests", "golden", "qp_pin_ticks.npz"))
    out = []
    for tol in TOLS:
        bad_status = outside = n = 0
        worst = 0.0
        for t in range(int(Z["count"])):
            g = lambda k: Z[f"t{t}_{k}"]
            ms = L.Mission(g("start"), g("goal"), g("world_min"), g("world_max"), g("radius"), g("downwash"), g("max_vel"), g("max_acc"),
                           g("nominal_velocity"))
            pl = L.SwarmPlanner(ms, PlannerConfig(gap_tolerance=tol, **extra))
            pl.planner_seq = int(g("tick")) - 1
            r = pl.plan(g("state"), g("goal"), g("traj"))
            pl.close()
            v, c = g("verdict"), g("cost")
            known = v >= 0
            bad_status += int((r["status"][known] != (v[known] == 1)).sum())
            opt = (v == 0) & (r["status"] == 0)
            d = np.abs(r["cost"][opt] - c[opt])
            outside += int((d > 1e-6 * np.abs(c[opt]) + 1e-8).sum())
            if opt.any():
                worst = max(worst, float((d / (1.0 + np.abs(c[opt]))).max()))
            n += int(known.sum())
        out.append(dict(instances=n, status_mismatch=bad_status, outside_tolerance_table=outside, worst_dcost_over_1pf=worst))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ticks", type=int, default=200)
    args = ap.parse_args()
    extra = {}
    total = [dict(total_us=0.0, tickmax_iters=0, iters=0, status_diff=0, dcost_over_1pf=0.0, dcost_rel=0.0, dcp=0.0, p99=[], p50=[]) for _ in TOLS]
    for name, ms in missions():
        rows = fly(ms, args.ticks, extra)
        print(json.dumps({"mission": name, "per_tolerance": {f"{t:g}": r for t, r in zip(TOLS, rows)}}), flush=True)
        for T, r in zip(total, rows):
            T["total_us"] += r["total_us"]; T["tickmax_iters"] += r["tickmax_iters"]; T["iters"] += r["iters"]
            T["status_diff"] += r["status_diff"]
            for k in ("dcost_over_1pf", "dcost_rel", "dcp"):
                T[k] = max(T[k], r[k])
            T["p99"].append(r["p99"]); T["p50"].append(r["p50"])
    P = pins(extra)
    print("\ngap_tolerance | kernel time of the six missions, ms | vs 1e-9 | tick-max iterations | tick p50 / p99 us (mean over missions) | "
          "status diff | max dcost/(1+|f|) | max dcost/|f| (|f| > 1e-3) | max dcp, m | HiGHS pins: status mismatches, outside the table, worst dcost/(1+|f|)")
    for tol, T, p in zip(TOLS, total, P):
        print("%8g | %8.1f | %5.1f %% | %6d | %6.1f / %6.1f | %d | %.2e | %.2e | %.2e | %d, %d of %d, %.2e"
              % (tol, T["total_us"] / 1e3, 100.0 * (T["total_us"] / total[0]["total_us"] - 1.0), T["tickmax_iters"], np.mean(T["p50"]), np.mean(T["p99"]),
                 T["status_diff"], T["dcost_over_1pf"], T["dcost_rel"], T["dcp"], p["status_mismatch"], p["outside_tolerance_table"], p["instances"],
                 p["worst_dcost_over_1pf"]))


if __name__ == "__main__":
    main()
