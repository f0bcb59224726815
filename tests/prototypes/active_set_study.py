#!/usr/bin/env python3
"""would a warm-started DUAL ACTIVE-SET method shorten the slowest agent's solve?  Measured on the CPU, no GPU minutes.

The reference picked CPLEX's dual simplex-type root algorithm (src/traj_optimizer.cpp:42-56: RootAlgorithm Dual, "fastest of five").
The kernel's interior point needs 9-12 iterations (~17-20 us each) for the agent a tick waits for.  A dual active-set method
(Goldfarb-Idnani, 1983) on the 39-unknown reduced problem costs one O(n^2) update plus one pass over the rows per CHANGE of its
working set, so what decides is the number of changes per agent-tick when the solve starts from the previous tick's active set
shifted by one segment.  This script flies missions with the oracle, poses every agent's QP (the reference's rows, nothing pruned),
solves it with a dense-algebra Goldfarb-Idnani restatement -- cold, and warm from the shifted previous active set -- and reports the
changes: per agent-tick (median, p99), per tick the maximum over agents (what a tick would wait for), infeasible / degenerate cases,
and the agreement of the optimum with the oracle's interior point.

DECISION RULE (written before the run): build a kernel only if the tick-max number of changes is <= 25 on >= 99 % of the
crossing ticks; otherwise record the negative result and stop touching the solver.

    python tests/prototypes/active_set_study.py [--missions circle64,random64b,random128] [--from-tick 40] [--to-tick 140] [--threads 16]

It imports the oracle, so it lives under tests/ (test infrastructure), like the other prototype."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O                      # noqa: E402
import lsc_planner_amd as L                         # noqa: E402  (mission generators only: no GPU call)
from lsc_planner_amd.planner import next_state_host  # noqa: E402

M, NC, SEGV, NV = 5, 6, 30, 90
ROW_DT = np.dtype({"names": ["nnz", "idx", "val", "rhs", "sense"], "formats": ["<i4", ("<i4", 9), ("<f8", 9), "<f8", "<i4"],
                   "offsets": [0, 4, 40, 112, 120], "itemsize": 128})


def rows_dense(qp):
    """Inequalities G x <= h and equalities of an assembled QP as dense arrays, plus an identity per inequality row:
    ('lsc', obstacle, m, i) / ('ax', kind-and-order-in-the-row-list) / ('bnd', variable, side)."""
    import ctypes
    assert ctypes.sizeof(O.OrcRow) == ROW_DT.itemsize
    r = np.frombuffer(qp._rows, dtype=ROW_DT, count=qp.nrows)
    nv = qp.nv
    A = np.zeros((qp.nrows, nv))
    k = np.arange(9)[None, :] < r["nnz"][:, None]
    rr = np.repeat(np.arange(qp.nrows), 9).reshape(-1, 9)
    A[rr[k], r["idx"][k]] = r["val"][k]
    sense, rhs = r["sense"], r["rhs"]
    eq = sense == 0
    ge = sense == 1
    le = ~eq & ~ge
    G = np.concatenate([-A[ge], A[le]])
    h = np.concatenate([-rhs[ge], rhs[le]])
    order = np.concatenate([np.nonzero(ge)[0], np.nonzero(le)[0]])
    fin_hi, fin_lo = np.isfinite(qp.hi), np.isfinite(qp.lo)
    Ib = np.eye(nv)
    G = np.concatenate([G, Ib[fin_hi], -Ib[fin_lo]])
    h = np.concatenate([h, qp.hi[fin_hi], -qp.lo[fin_lo]])
    return A[eq], rhs[eq], G, h, order, np.nonzero(fin_hi)[0], np.nonzero(fin_lo)[0]


def row_keys(order, n_obs, hi_idx, lo_idx):
    """Identity of every inequality row that survives a shift of the plan by one segment.  populatebyrow's order
    (src/traj_optimizer.cpp:394-536): 45 + ... equalities first, then per obstacle 27 LSC rows (m, i without m = 0, i < 3), then the
    velocity / acceleration rows per axis, then the stop-at-horizon equalities; bounds are not rows (appended here)."""
    keys = []
    lsc_mi = [(m, i) for m in range(M) for i in range(NC) if not (m == 0 and i < 3)]
    # inequality rows in populatebyrow's order = all rows with sense != 0
    ineq_sorted = np.sort(order)
    pos_of = {int(r): p for p, r in enumerate(ineq_sorted)}
    n_lsc = 27 * n_obs
    for r in order:
        p = pos_of[int(r)]
        if p < n_lsc:
            j, q = divmod(p, 27)
            m, i = lsc_mi[q]
            keys.append(("lsc", j, m, i))
        else:
            keys.append(("ax", p - n_lsc))            # velocity / acceleration rows: their order is (axis, m, i, sign): see shift_key
    for v in hi_idx:
        keys.append(("hi", int(v)))
    for v in lo_idx:
        keys.append(("lo", int(v)))
    return keys


def shift_keys(active_keys, ax_shift):
    """Previous tick's active rows -> the rows they become when the plan moves on by one segment (control point (m, i) -> (m - 1, i));
    the last segment is predicted from the previous last segment as well."""
    out = set()
    for k in active_keys:
        if k[0] == "lsc":
            _, j, m, i = k
            if m >= 1 and not (m - 1 == 0 and i < 3):
                out.add(("lsc", j, m - 1, i))
            if m == M - 1:
                out.add(k)
        elif k[0] in ("hi", "lo"):
            v = k[1]
            kx, t = divmod(v, SEGV)
            if t >= NC and t - NC >= 3:
                out.add((k[0], kx * SEGV + t - NC))
            if t >= (M - 1) * NC:
                out.add(k)
        else:
            s = ax_shift.get(k[1])
            if s is not None:
                out.add(("ax", s))
            if k[1] in ax_shift.get("last", ()):
                out.add(k)
    return out


class GI:
    """Goldfarb-Idnani dual active set for  min 1/2 y'Hy + g'y  s.t.  G y <= h  (H positive definite), dense algebra, no factor
    updates: this counts working-set changes, it is not a fast solver."""

    def __init__(self, H, g, G, h, tol=1e-9, rule="rhs"):
        self.H, self.g, self.G, self.h, self.tol = H, g, G, h, tol
        self.L = np.linalg.cholesky(H)
        self.n = len(g)
        # selection rule of the row that joins: violation over 1 + |h| ("rhs", what the kernel does), the raw violation ("raw"), or the
        # violation over sqrt(n'H^-1 n) = the distance of the row's plane in the metric of the cost ("metric")
        if rule == "raw":
            self.scale = np.ones(len(h))
        elif rule == "metric":
            Y = np.linalg.solve(self.L, G.T)
            self.scale = np.sqrt(np.maximum((Y * Y).sum(0), 1e-300))
        else:
            self.scale = 1.0 + np.abs(h)

    def _hsolve(self, B):
        return np.linalg.solve(self.L.T, np.linalg.solve(self.L, B))

    def eqp(self, W):
        """min on the working set as equalities: returns (y, u) with u the multipliers of G_W y = h_W (u >= 0 <=> dual feasible)."""
        y0 = -self._hsolve(self.g)
        if not W:
            return y0, np.zeros(0)
        Gw = self.G[W]
        HiGt = self._hsolve(Gw.T)
        S = Gw @ HiGt
        rhs = Gw @ y0 - self.h[W]
        u = np.linalg.solve(S + 1e-14 * np.eye(len(W)), rhs)
        return y0 - HiGt @ u, u

    def solve(self, W0=(), max_changes=400):
        """Returns status (0 optimal, 1 infeasible, 2 gave up), y, cost, counts dict."""
        G, h, tol = self.G, self.h, self.tol
        W = self._independent(list(W0))
        cnt = {"repair_drops": len(W0) - len(W), "adds": 0, "drops": 0, "start_size": len(W)}
        y, u = self.eqp(W)
        # repair: drop rows with negative multipliers until the start is dual feasible
        while len(W) and u.min() < -1e-10:
            j = int(np.argmin(u))
            W.pop(j)
            cnt["repair_drops"] += 1
            y, u = self.eqp(W)
        u = np.maximum(u, 0.0)
        scale = self.scale
        while True:
            if cnt["adds"] + cnt["drops"] > max_changes:
                return 2, y, None, cnt
            s = h - G @ y
            viol = s / scale
            viol[W] = 0.0
            p = int(np.argmin(viol))
            if (s / (1.0 + np.abs(h)))[p] >= -tol if False else (np.where(np.isin(np.arange(len(h)), W), 0.0, s / (1.0 + np.abs(h))).min() >= -tol):
                cost = 0.5 * y @ self.H @ y + self.g @ y
                cnt["final_size"] = len(W)
                self.W = list(W)
                return 0, y, cost, cnt
            npv = G[p]
            up = 0.0
            while True:
                # direction: z in primal space (keeps W active), r = rate of the multipliers of W
                if W:
                    Gw = G[W]
                    HiGt = self._hsolve(Gw.T)
                    S = Gw @ HiGt
                    Hin = self._hsolve(npv)
                    r = np.linalg.solve(S + 1e-14 * np.eye(len(W)), Gw @ Hin)
                    z = Hin - HiGt @ r
                else:
                    z = self._hsolve(npv)
                    r = np.zeros(0)
                zn = float(npv @ z)
                # the constraint is  npv.y <= h_p  and is violated: y moves by -t z while u_p grows by t; stationarity then asks the
                # multipliers of W to change by -t r (they must stay >= 0: the first to reach zero leaves the working set)
                t1, jdrop = np.inf, -1
                if len(W):
                    pos = r > 1e-12
                    if pos.any():
                        ratios = np.where(pos, u / np.where(pos, r, 1.0), np.inf)
                        jdrop = int(np.argmin(ratios))
                        t1 = float(ratios[jdrop])
                sp = float(h[p] - npv @ y)                      # negative: violation
                t2 = (-sp / zn) if zn > 1e-13 else np.inf
                t = min(t1, t2)
                if not np.isfinite(t):
                    return 1, y, None, cnt                      # infeasible
                if np.isfinite(t2) or zn > 1e-13:
                    y = y - t * z
                u = u - t * r if len(W) else u
                up += t
                if t == t2:
                    W.append(p)
                    u = np.append(np.maximum(u, 0.0), up)
                    cnt["adds"] += 1
                    break
                W.pop(jdrop)
                u = np.delete(u, jdrop)
                u = np.maximum(u, 0.0)
                cnt["drops"] += 1

    def _independent(self, W):
        """Drops rows of the proposed working set that are linearly dependent on the ones before them (and beyond n)."""
        keep, Q = [], np.zeros((0, self.n))
        for w in W:
            v = self.G[w].copy()
            nv0 = np.linalg.norm(v)
            if len(Q):
                v -= Q.T @ (Q @ v)
            if np.linalg.norm(v) > 1e-8 * max(nv0, 1e-300) and len(keep) < self.n:
                keep.append(w)
                Q = np.vstack([Q, v / np.linalg.norm(v)])
        return keep


def ax_shift_table():
    """Velocity / acceleration rows in populatebyrow's order (src/traj_optimizer.cpp:468-525): per axis k, per segment m, per i: two
    signs.  Returns {row position -> position of the same row one segment earlier} and the set of rows of the last segment."""
    pos, table = {}, {}
    p = 0
    for kind, imax in (("v", 5), ("a", 4)):
        for k in range(3):
            for m in range(M):
                for i in range(imax):
                    if m == 0 and ((kind == "v" and i < 2) or (kind == "a" and i < 1)):
                        continue
                    for sgn in (0, 1):
                        pos[(kind, k, m, i, sgn)] = p
                        p += 1
    last = set()
    for (kind, k, m, i, sgn), q in pos.items():
        prev = pos.get((kind, k, m - 1, i, sgn))
        if prev is not None:
            table[q] = prev
        if m == M - 1:
            last.add(q)
    table["last"] = last
    return table, p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--missions", default="circle64,random64b,random128")
    ap.add_argument("--from-tick", type=int, default=40)
    ap.add_argument("--to-tick", type=int, default=140)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--rule", default="rhs", choices=["rhs", "raw", "metric"])
    ap.add_argument("--agents-stride", type=int, default=1, help="study every k-th agent (all agents fly)")
    a = ap.parse_args()
    catalog = {"circle64": lambda: L.circle_swap(64, 8.0), "circle48": lambda: L.circle_swap(48, 6.0),
               "circle80": lambda: L.circle_swap(80, 10.0, world=(-12, -12, 0, 12, 12, 2.5)),
               "random64a": lambda: L.random_swarm(64, world=(-6, -6, 0, 6, 6, 2.5), seed=11),
               "random64b": lambda: L.random_swarm(64, world=(-5, -5, 0, 5, 5, 2.5), seed=12),
               "random128": lambda: L.random_swarm(128, world=(-8, -8, 0, 8, 8, 2.5), seed=13), "circle20": lambda: L.circle_swap(20, 8.0)}
    ax_shift, n_ax_rows = ax_shift_table()
    print("decision rule (written before the run): build a dual active-set kernel only if the tick-max number of working-set changes is <= 25 "
          "on >= 99 % of the crossing ticks", flush=True)
    grand = []
    for name in a.missions.split(","):
        ms = catalog[name]()
        N = ms.qn
        prm = O.make_params(world_min=ms.world_min, world_max=ms.world_max, obs_f32=True)
        sw = O.Swarm(prm, ms.radius, ms.downwash, ms.max_vel, ms.max_acc, ms.nominal_velocity)
        state = np.zeros((N, 9), np.float32)
        state[:, :3] = ms.start
        traj = np.zeros((N, 3, SEGV), np.float32)
        prev_active = [None] * N
        Z = None
        t0 = time.time()
        per_tick = []
        for tick in range(1, a.to_tick + 1):
            goals = O.goal_prior_based(state, ms.goal, traj, tick)
            study = tick >= a.from_tick - 1
            o = sw.tick(state, goals, traj, tick, want_lsc=study, nthreads=a.threads)
            if study:
                rec = []
                init = np.stack([O.shift_traj(traj[q]) if tick >= 2 else O.const_vel_traj(state[q, :3], state[q, 3:6]) for q in range(N)])
                for qi in range(0, N, a.agents_stride):
                    others = [q for q in range(N) if q != qi]
                    qp = O.qp_assemble(prm, state[qi], goals[qi], ms.nominal_velocity[qi], ms.max_vel[qi], ms.max_acc[qi], init[others],
                                       o["normal"][qi], o["d"][qi])
                    Aeq, beq, G, h, order, hi_idx, lo_idx = rows_dense(qp)
                    assert len(order) == 27 * (N - 1) + n_ax_rows, (len(order), n_ax_rows)
                    if Z is None:
                        _, sv, Vt = np.linalg.svd(Aeq)
                        Z = Vt[len(sv):].T                     # the equality rows' coefficients are the same for every agent and tick
                        Apinv = np.linalg.pinv(Aeq)
                    xp = Apinv @ beq
                    Hy = Z.T @ qp.P @ Z
                    Hy = 0.5 * (Hy + Hy.T)
                    gy = Z.T @ (qp.P @ xp + qp.c)
                    Gy, hy = G @ Z, h - G @ xp
                    keys = row_keys(order, N - 1, hi_idx, lo_idx)
                    gi = GI(Hy, gy, Gy, hy, rule=a.rule)
                    st_c, y_c, cost_c, cnt_c = gi.solve()
                    r = {"agent": qi, "status": st_c, "oracle_status": int(o["status"][qi]), "rows": len(hy)}
                    if st_c == 0:
                        Wc = list(gi.W)
                        x = xp + Z @ y_c
                        cost = 0.5 * x @ qp.P @ x + qp.c @ x + qp.cst
                        r.update(cold_changes=cnt_c["adds"] + cnt_c["drops"], active=cnt_c["final_size"], cost=cost,
                                 plan_dev=float(np.abs(x.reshape(3, SEGV) - o["traj"][qi].astype(np.float64)).max()) if o["status"][qi] == 0 else None,
                                 cost_rel_err=abs(cost - o["cost"][qi]) / (abs(o["cost"][qi]) + 1e-2) if o["status"][qi] == 0 else None)
                        act_keys = {keys[w] for w in Wc}
                        if prev_active[qi] is not None and tick >= a.from_tick:
                            pred = shift_keys(prev_active[qi], ax_shift)
                            kpos = {k: p for p, k in enumerate(keys)}
                            W0 = [kpos[k] for k in pred if k in kpos]
                            st_w, y_w, cost_w, cnt_w = gi.solve(W0)
                            r.update(warm_status=st_w, warm_changes=cnt_w["repair_drops"] + cnt_w["adds"] + cnt_w["drops"],
                                     warm_repairs=cnt_w["repair_drops"], warm_adds=cnt_w["adds"], warm_drops=cnt_w["drops"], predicted=len(W0),
                                     symdiff=len(pred.symmetric_difference(act_keys)),
                                     warm_cost_dev=abs(cost_w - cost_c) / max(1e-12, abs(cost_c)) if st_w == 0 else None)
                        prev_active[qi] = act_keys
                    else:
                        prev_active[qi] = None
                    rec.append(r)
                if tick >= a.from_tick:
                    per_tick.append(rec)
                    w = [x["warm_changes"] for x in rec if "warm_changes" in x]
                    c = [x["cold_changes"] for x in rec if "cold_changes" in x]
                    if tick % 10 == 0:
                        print(f"  {name} tick {tick}: warm changes max {max(w, default=-1)} median {np.median(w) if w else -1:.0f} | cold max {max(c, default=-1)} | "
                              f"active max {max((x['active'] for x in rec if 'active' in x), default=-1)} | {time.time() - t0:.0f} s", flush=True)
            traj = o["traj"]
            sw.stale[:] = np.where((o["status"] == 0)[:, None, None], traj, sw.stale)
            state = next_state_host(traj)
        # ---- mission summary
        allr = [x for rec in per_tick for x in rec]
        warm = np.array([x["warm_changes"] for x in allr if "warm_changes" in x])
        cold = np.array([x["cold_changes"] for x in allr if "cold_changes" in x])
        act = np.array([x["active"] for x in allr if "active" in x])
        sym = np.array([x["symdiff"] for x in allr if "symdiff" in x])
        tmax_w = np.array([max((x["warm_changes"] for x in rec if "warm_changes" in x), default=0) for rec in per_tick])
        tmax_c = np.array([max((x["cold_changes"] for x in rec if "cold_changes" in x), default=0) for rec in per_tick])
        errs = np.array([x["cost_rel_err"] for x in allr if x.get("cost_rel_err") is not None])
        mism = sum(1 for x in allr if (x["status"] == 0) != (x["oracle_status"] == 0))
        gaveup = sum(1 for x in allr if x["status"] == 2 or x.get("warm_status") == 2)
        wdev = np.array([x["warm_cost_dev"] for x in allr if x.get("warm_cost_dev") is not None])
        pdev = np.array([x["plan_dev"] for x in allr if x.get("plan_dev") is not None])
        print(f"{name}: {N} agents, ticks {a.from_tick}..{a.to_tick}, {len(allr)} agent-ticks, rows per QP {allr[0]['rows']}\n"
              f"   optimal active set: median {np.median(act):.0f}, p99 {np.percentile(act, 99):.0f}, max {act.max()} (of 39 unknowns)\n"
              f"   cold start   changes per agent-tick: median {np.median(cold):.0f}, p99 {np.percentile(cold, 99):.0f}, max {cold.max()}; tick-max: median {np.median(tmax_c):.0f}, p99 {np.percentile(tmax_c, 99):.0f}\n"
              f"   warm start   changes per agent-tick: median {np.median(warm):.0f}, p99 {np.percentile(warm, 99):.0f}, max {warm.max()}; tick-max: median {np.median(tmax_w):.0f}, "
              f"p90 {np.percentile(tmax_w, 90):.0f}, p99 {np.percentile(tmax_w, 99):.0f}, max {tmax_w.max()}\n"
              f"   predicted-vs-optimal active set, symmetric difference: median {np.median(sym):.0f}, p99 {np.percentile(sym, 99):.0f}, max {sym.max()}\n"
              f"   ticks with tick-max warm changes <= 25: {100.0 * np.mean(tmax_w <= 25):.1f} %   (<= 15: {100.0 * np.mean(tmax_w <= 15):.1f} %, <= 40: {100.0 * np.mean(tmax_w <= 40):.1f} %)\n"
              f"   optimum vs the oracle's interior point: max |cost difference| / (|cost| + 0.01) {errs.max() if len(errs) else float('nan'):.2e} over {len(errs)} solved QPs; "
              f"plan (control points) vs the oracle's float32 plan: p99 {np.percentile(pdev, 99):.2e} m, max {pdev.max():.2e} m; feasibility verdicts that differ: {mism}; runs that gave up (cycling guard, 400 changes): {gaveup}; warm vs cold optimum: max {wdev.max() if len(wdev) else float('nan'):.2e}",
              flush=True)
        grand.append((name, tmax_w))
    allt = np.concatenate([t for _, t in grand])
    ok = 100.0 * np.mean(allt <= 25)
    print(f"ALL MISSIONS: {len(allt)} crossing ticks; tick-max warm-start changes <= 25 on {ok:.1f} % of them (median {np.median(allt):.0f}, p99 {np.percentile(allt, 99):.0f}) "
          f"-> decision: {'BUILD' if ok >= 99.0 else 'DO NOT BUILD (rule: >= 99 %)'}", flush=True)


if __name__ == "__main__":
    main()
