"""CPU prototype behind DESIGN 4.6's note on the linear tail of crowded ticks (not a test; it uses the oracle to produce the QPs of a
64-agent mission, hence it lives under tests/): a dense numpy Mehrotra interior point on those QPs, standard against a variant that
skips the predictor solve in the tail (sigma and the second-order term lagged from the previous iteration).
    python tests/prototypes/ipm_tail_prototype.py
Round-3 outcome: the lagged iterations reduce the gap 7-12x with one solve instead of two, but block earlier (alpha ~0.95) and cost
one more iteration: 13 iterations / 26 solves against 14 / 24 -- no gain in time; not built into the kernels."""
import sys, numpy as np, scipy.linalg as sla
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import oracle as O
import lsc_planner_amd as L
from lsc_planner_amd.planner import next_state_host

def collect(nticks=(42,), agents=(17, 5)):
    ms = L.circle_swap(64, 8.0)
    prm = O.make_params(world_min=ms.world_min, world_max=ms.world_max, obs_f32=True)
    sw = O.Swarm(prm, ms.radius, ms.downwash, ms.max_vel, ms.max_acc, ms.nominal_velocity)
    N = 64
    state = np.zeros((N, 9), np.float32); state[:, :3] = ms.start
    traj = np.zeros((N, 3, 30), np.float32)
    out = []
    for tick in range(1, max(nticks) + 1):
        o = sw.tick(state, ms.goal, traj, tick, want_lsc=True, nthreads=16)
        if tick in nticks:
            for a in agents:
                others = [j for j in range(N) if j != a]
                obs = np.array([O.shift_traj(traj[j]) if tick >= 2 else O.const_vel_traj(state[j, :3], state[j, 3:6]) for j in others], np.float32)
                qp = O.qp_assemble(prm, state[a], o["goal"][a] if "goal" in o else ms.goal[a], float(ms.nominal_velocity[a]), ms.max_vel[a], ms.max_acc[a], obs, o["normal"][a], o["d"][a])
                x0 = O.shift_traj(traj[a]).reshape(-1).astype(np.float64) if tick >= 2 else None
                out.append((tick, a, qp, x0, o["cost"][a], o["iters"][a] if "iters" in o else -1))
        sw.stale[:] = o["traj"]
        traj = o["traj"]; state = next_state_host(traj)
    return out

def dense(qp):
    return qp.dense()

def solve(qp, x0, lagged, mu0=0.03, verbose=False):
    Aeq, beq, G, h = dense(qp)
    P, c = qp.P, qp.c
    Z = sla.null_space(Aeq); xp = np.linalg.lstsq(Aeq, beq, rcond=None)[0]
    H = Z.T @ (2 * P if False else P) @ Z          # objective: 1/2 x'Px + c'x ?  (checked below by cost)
    g = Z.T @ (P @ xp + c)
    Gy = G @ Z; hy = h - G @ xp
    y = Z.T @ (x0 - xp) if x0 is not None else np.zeros(Z.shape[1])
    s = np.maximum(hy - Gy @ y, np.sqrt(mu0)); z = mu0 / s
    m = len(s)
    solves = 0; it = 0
    lag_ok = False; sigma_prev = 0.1; p_prev = None
    hist = []
    while it < 60:
        rp = Gy @ y + s - hy
        rd = H @ y + g + Gy.T @ z
        gap = s @ z; mu = gap / m
        xx = xp + Z @ y
        f = 0.5 * xx @ P @ xx + c @ xx + qp.cst
        if np.abs(rp).max() <= 1e-9 * max(1, np.abs(hy).max()) and gap <= 1e-11 * (1 + abs(f)) and np.abs(rd).max() <= 1e-5 * (1 + abs(f)):
            break
        w = z / s
        K = H + Gy.T @ (w[:, None] * Gy)
        try:
            cf = sla.cho_factor(K)
        except Exception:
            print('   chol failed at it', it); break
        def step(rc):      # rc: target for s*dz + z*ds  ( = -s z + ...)
            # ds = -rp - Gy dy ; dz = (rc - z ds)/s
            rhs = -rd - Gy.T @ ((rc + z * rp) / s)
            dy = sla.cho_solve(cf, rhs)
            ds = -rp - Gy @ dy
            dz = (rc - z * ds) / s
            return dy, ds, dz
        def maxstep(ds, dz):
            a = 1.0
            neg = ds < 0
            if neg.any(): a = min(a, (-s[neg] / ds[neg]).min())
            neg = dz < 0
            if neg.any(): a = min(a, (-z[neg] / dz[neg]).min())
            return a
        use_lag = lagged and lag_ok and p_prev is not None
        if use_lag:
            sigma = sigma_prev
            e2 = s * z * p_prev * (1 - p_prev)
            aaff = 1.0
            solves += 1
        else:
            dya, dsa, dza = step(-s * z)
            aaff = min(1.0, maxstep(dsa, dza))
            mu_aff = ((s + aaff * dsa) @ (z + aaff * dza)) / m
            sigma = (mu_aff / mu) ** 3
            e2 = dsa * dza
            solves += 2
        dy, ds, dz = step(sigma * mu - s * z - e2)
        tau = min(1 - 1e-5, max(0.99, aaff))
        alpha = min(1.0, tau * maxstep(ds, dz))
        p_prev = np.clip(-alpha * ds / s, 0.0, 1.0)
        y = y + alpha * dy; s = s + alpha * ds; z = z + alpha * dz
        gap_new = s @ z
        hist.append((gap, aaff, sigma, alpha, use_lag))
        # tail regime detection for the NEXT iteration
        lag_ok = (aaff >= 0.9 and alpha >= 0.95 and gap_new < 0.35 * gap)
        sigma_prev = sigma
        it += 1
    if verbose:
        for i, hh in enumerate(hist): print("   it %2d gap %.2e aaff %.3f sigma %.2e alpha %.4f lag %d" % ((i,) + hh))
    x = xp + Z @ y
    return it, solves, f

if __name__ == "__main__":
    cases = collect()
    tot = {False: [0, 0], True: [0, 0]}
    for tick, a, qp, x0, ocost, oit in cases:
        r = {}
        for lag in (False, True):
            it, sv, f = solve(qp, x0, lag, verbose=(a == 17 and tick == 42))
            r[lag] = (it, sv, f)
            tot[lag][0] += it; tot[lag][1] += sv
        print("tick %d agent %2d: standard %2d it / %2d solves, lagged %2d it / %2d solves; f %.9f vs %.9f oracle %.9f" % (tick, a, r[False][0], r[False][1], r[True][0], r[True][1], r[False][2], r[True][2], ocost))
    print("TOTAL standard: %d iterations %d solves ; lagged: %d iterations %d solves" % (tot[False][0], tot[False][1], tot[True][0], tot[True][1]))
