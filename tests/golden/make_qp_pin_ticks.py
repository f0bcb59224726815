"""HiGHS verdicts on whole ticks, for replay THROUGH THE KERNEL on the GPU box (tests/test_gpu_round3.py): the tick's inputs
(mission, states, previous plans) and, per agent, what HiGHS says about its QP -- certified infeasible (phase-1 LP: strictly
positive minimal uniform violation of the rows), optimal with this cost (0), or optimal with at most this cost (2: HiGHS stopped
short of a feasible point with a lower objective) -- independent of the oracle's solver.

Run in the build container (SciPy's bundled HiGHS):  python tests/golden/make_qp_pin_ticks.py
Writes tests/golden/qp_pin_ticks.npz.  The missions are dense, seeded soak swarms like those of tests/test_oracle_pins.py (eight of them; packed so
tightly that many QPs are infeasible); of every mission the first tick that holds an infeasible QP and ticks 8, 14 and 20 are kept.  The QP rows come from the oracle's assembly, which log/QPmodel.lp pins.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", ".."))

import highs_qp as H  # noqa: E402


def main():
    import lsc_planner_amd as L
    from lsc_planner_amd.planner import next_state_host
    from oracle import oracle as O
    rng = np.random.default_rng(778)
    out = {}
    n_inf = n_opt = n_none = kept = 0
    missions = 0
    for trial in range(60):
        if missions >= 8:
            break
        n = int(rng.integers(8, 22))
        side = float(rng.uniform(0.7, 1.2))
        seed = int(rng.integers(1, 1 << 30))
        try:
            ms = L.random_swarm(n, world=(-side, -side, 0, side, side, 2.5), seed=seed, min_sep=0.31, shrink=0.15)
        except ValueError:
            continue
        if trial % 3 == 0:
            ms.radius[:] = rng.uniform(0.1, 0.25, n)
            ms.downwash[:] = rng.uniform(1.0, 2.5, n)
            ms.max_vel[:] = rng.uniform(0.6, 1.5, (n, 1))
            ms.max_acc[:] = rng.uniform(1.0, 3.0, (n, 1))
        prm = O.make_params(world_min=ms.world_min, world_max=ms.world_max, obs_f32=True)
        sw = O.Swarm(prm, ms.radius, ms.downwash, ms.max_vel, ms.max_acc, ms.nominal_velocity)
        state = np.zeros((n, 9), np.float32)
        state[:, :3] = ms.start
        traj = np.zeros((n, 3, 30), np.float32)
        seen = False
        for tick in range(1, 21):
            o = sw.tick(state, ms.goal, traj, tick, want_lsc=True, nthreads=8)
            if (o["status"] == 1).any() and (not seen or tick in (8, 14, 20)):
                seen = True
                obs = np.array([O.shift_traj(p) for p in traj]) if tick >= 2 else \
                    np.array([O.const_vel_traj(state[j, :3], state[j, 3:6]) for j in range(n)])
                verdict = np.full(n, -1, np.int32)
                cost = np.zeros(n)
                for a in range(n):
                    others = [j for j in range(n) if j != a]
                    qp = O.qp_assemble(prm, state[a], ms.goal[a], float(ms.nominal_velocity[a]), ms.max_vel[a], ms.max_acc[a],
                                       obs[others], o["normal"][a], o["d"][a])
                    A, lo, hi = H.rows_of(qp)
                    st, t = H.min_violation(A, lo, hi, qp.lo, qp.hi)
                    if st == "Optimal" and t > 1e-7:
                        verdict[a] = 1                                   # certificate: the rows cannot all hold
                        continue
                    if not (st == "Optimal" and t <= 1e-9):
                        continue                                          # too close to call: no verdict
                    ms_, x, obj, viol = H.solve_oracle_qp(qp)
                    if ms_ == "Optimal" and viol <= 1e-7:
                        verdict[a], cost[a] = 0, obj
                        # HiGHS's active-set code stops ~1e-6 short on a few instances (tests/test_oracle_pins.py: "highs_short"):
                        # when a point that satisfies the ORIGINAL rows to 1e-9 has a lower objective, HiGHS's number is only an
                        # upper bound of the optimum -- recorded as verdict 2 with that bound
                        st_o, xo, co, _, _ = qp.solve()
                        if st_o == 0 and co < obj - (1e-7 * abs(obj) + 1e-9):
                            vo = max(np.max(lo - A @ xo), np.max(A @ xo - hi), np.max(qp.lo - xo), np.max(xo - qp.hi))
                            if vo <= 1e-9:
                                verdict[a] = 2
                k = f"t{kept}"
                out[k + "_state"], out[k + "_traj"], out[k + "_tick"] = state.copy(), traj.copy(), np.int32(tick)
                out[k + "_verdict"], out[k + "_cost"] = verdict, cost
                for name in ("start", "goal", "world_min", "world_max", "radius", "downwash", "max_vel", "max_acc", "nominal_velocity"):
                    out[k + "_" + name] = getattr(ms, name)
                kept += 1
                n_inf += int((verdict == 1).sum()); n_opt += int((verdict == 0).sum() + (verdict == 2).sum()); n_none += int((verdict < 0).sum())
                print(f"trial {trial} tick {tick}: n {n}  infeasible {int((verdict == 1).sum())} optimal {int((verdict == 0).sum())} none {int((verdict < 0).sum())}"
                      f"  upper bounds {int((verdict == 2).sum())}  oracle agrees {bool(((verdict < 0) | (np.minimum(verdict, 1) == o['status']) | (verdict == 2)).all())}", flush=True)
            traj = o["traj"]
            state = next_state_host(traj)
        missions += int(seen)
    out["count"] = np.int32(kept)
    out["solver"] = np.array("HiGHS " + H.version() + " (scipy.optimize._highspy)")
    np.savez_compressed(os.path.join(HERE, "qp_pin_ticks.npz"), **out)
    print(f"{kept} ticks: {n_inf} certified infeasible, {n_opt} optimal, {n_none} without a verdict")


if __name__ == "__main__":
    main()
