"""Oracle GJK restatement vs. the reference's own openGJK (golden vectors generated from oracle/_ref)."""
import ctypes
import os

import numpy as np


def test_oracle_gjk_matches_reference_golden_bitwise(oracle, gjk_golden):
    pts = gjk_golden["pts"].astype(np.float64)
    for i in range(len(pts)):
        d, v, nv, it = oracle.gjk_origin(pts[i])
        assert d == gjk_golden["dist"][i]
        assert np.array_equal(v, gjk_golden["v"][i])
        assert nv == gjk_golden["nvrtx"][i]


def test_oracle_gjk_live_against_reference_build(oracle):
    """Only where oracle/_ref was built: the build container (it is git- AND gpurun-ignored: the reference's GPLv3 openGJK does not travel)."""
    ref = oracle.ref_gjk_lib()
    if ref is None:
        import pytest
        pytest.skip("oracle/_ref not built here")
    rng = np.random.default_rng(5)
    zero = np.zeros(3)
    dp = ctypes.POINTER(ctypes.c_double)
    saved = os.dup(1); devnull = os.open(os.devnull, os.O_WRONLY); os.dup2(devnull, 1)
    try:
        for t in range(3000):
            p = (rng.normal(size=(6, 3)) * (0.2 + (t % 5))).astype(np.float32).astype(np.float64)
            v2 = np.zeros(3); n2 = ctypes.c_int()
            d2 = ref.ref_gjk(p.ctypes.data_as(dp), 6, zero.ctypes.data_as(dp), 1, v2.ctypes.data_as(dp), ctypes.byref(n2))
            d1, v1, n1, _ = oracle.gjk_origin(p)
            assert d1 == d2 and np.array_equal(v1, v2) and n1 == n2.value
    finally:
        os.dup2(saved, 1); os.close(devnull); os.close(saved)


def test_gjk_geometry_properties(oracle):
    """Witness is the closest hull point: inside-origin -> 0, single point -> the point, and v.(p - v) >= 0."""
    rng = np.random.default_rng(11)
    p = np.tile(np.array([0.3, -0.2, 0.1]), (6, 1))
    d, v, nv, _ = oracle.gjk_origin(p)
    assert np.allclose(v, p[0]) and nv == 1
    for _ in range(200):
        pts = rng.normal(size=(6, 3)) + np.array([2.5, 0, 0])
        d, v, nv, _ = oracle.gjk_origin(pts)
        assert abs(np.linalg.norm(v) - d) < 1e-12
        assert ((pts - v) @ v >= -1e-9).all()       # supporting-plane optimality condition
    cube = np.array([[1, 1, 1], [-1, -1, 1], [-1, 1, -1], [1, -1, -1], [1, 1, -1], [-1, -1, -1]], float)
    d, v, nv, _ = oracle.gjk_origin(cube)
    assert d < 1e-9
