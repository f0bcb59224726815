"""The inequality the neighbour lists of large swarms rest on (lsc_planner_amd/csrc/lsc_neigh.hip), checked numerically on the CPU:
the sphere test of the grid query keeps every (obstacle, segment) unit that phase B's own pre-cull keeps (lsc_kernels.hip,
"spatial pre-cull"), whatever the control points, the downwash and the reach B_m are.  (That the pre-cull in turn keeps every unit with a
row the exact per-row test keeps is round 2's argument, tested on the GPU by prune = 1 against prune = 3.)"""
import numpy as np


def _unit_test_keeps(p, q, idw, B, r):
    """phase B's test on the six control points of both sides: keep unless |w_c| >= 2 s B + r + 2e-4 + R_w"""
    S = np.array([1.0, 1.0, idw])
    w = (p - q) * S
    wc = w.mean(axis=0)
    rw = np.sqrt(((w - wc) ** 2).sum(axis=1).max())
    need = 2.0 * max(1.0, idw) * B + r + 2e-4 + rw
    return not (wc @ wc >= need * need)


F = np.float32


def _sphere(points):
    """lsc_neigh_build_kernel: float32; centre = mean, radius around the stored centre, rounded up by 1e-5 r + 1e-5"""
    pts = points.astype(F)
    c = (pts.sum(axis=0, dtype=F) * F(1.0 / 6.0)).astype(F)
    d = pts - c
    rad = F(np.sqrt((d * d).sum(axis=1, dtype=F).max())) * F(1.0 + 1e-5) + F(1e-5)
    return c, F(rad)


def _sphere_test_keeps(ca, ra, co, ro, idw, B, r):
    """lsc_neigh_query_kernel: float32; keep unless |S (C_a - C_o)| >= (s (2 B + 3 (rho_a + rho_o)) + r) (1 + 2e-5) + 5e-4, with B as the
    build kernel stores it (rounded up by 1e-5 B + 1e-3)"""
    idw, r = F(idw), F(r)
    Bf = F(B) * F(1.0 + 1e-5) + F(1e-3)
    d = (ca - co) * np.array([1.0, 1.0, idw], F)
    need = (max(F(1.0), idw) * (F(2.0) * Bf + F(3.0) * (ra + ro)) + r) * F(1.0 + 2e-5) + F(5e-4)
    return not (F((d * d).sum(dtype=F)) >= need * need)


def test_sphere_test_of_the_grid_query_keeps_what_the_unit_test_keeps():
    rng = np.random.default_rng(6)
    kept_unit = kept_sphere = 0
    for trial in range(12000):
        scale = rng.choice([0.05, 0.3, 1.0, 3.0])
        p = (rng.normal(size=3) * 3 + rng.normal(size=(6, 3)) * scale * rng.uniform(0, 1)).astype(np.float32).astype(np.float64)
        off = rng.normal(size=3) * rng.choice([0.5, 2.0, 6.0])
        q = (p.mean(axis=0) + off + rng.normal(size=(6, 3)) * scale * rng.uniform(0, 1)).astype(np.float32).astype(np.float64)
        idw = 1.0 / rng.choice([0.5, 1.0, 2.0, 3.7])
        B = rng.uniform(0.0, 2.5)
        r = rng.uniform(0.1, 0.6)
        ku = _unit_test_keeps(p, q, idw, B, r)
        ca, ra = _sphere(p)
        co, ro = _sphere(q)
        ks = _sphere_test_keeps(ca, ra, co, ro, idw, B, r)
        assert ks or not ku, (trial, p, q, idw, B, r)
        kept_unit += ku
        kept_sphere += ks
    assert 0.2 < kept_unit / 12000 < 0.9 and kept_sphere >= kept_unit          # the trials straddle the bound
    assert kept_sphere < 1.4 * kept_unit                                       # ... which is not much looser than the test it replaces
