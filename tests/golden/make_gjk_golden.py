"""Golden GJK vectors from the REFERENCE's own openGJK (built in-container into oracle/_ref by oracle/Makefile).

Run in the build container only:  python tests/golden/make_gjk_golden.py
Inputs are float32-valued 6-point hulls (what the hot path feeds GJK: relative control points), including the
degenerate families the planner produces (all points equal on the first tick, collinear, planar, origin inside,
ties between support points).  Outputs: distance, witness vector v, final simplex size.
"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle import oracle as O  # noqa: E402


def families(rng, n):
    out = []
    for t in range(n):
        kind = t % 10
        if kind == 0: pts = rng.normal(size=(6, 3)) + rng.normal(size=3) * 2
        elif kind == 1: pts = rng.normal(size=(6, 3)) * 0.3
        elif kind == 2:
            a = rng.normal(size=3) * 2; b = rng.normal(size=3)
            pts = a + np.outer(np.linspace(0, 1, 6), b) + rng.normal(size=(6, 3)) * 0.05
        elif kind == 3: pts = np.tile(rng.normal(size=3), (6, 1))
        elif kind == 4:
            a = rng.normal(size=3); b = rng.normal(size=3); pts = a + np.outer(np.linspace(-1, 1, 6), b)
        elif kind == 5: pts = rng.integers(-2, 3, size=(6, 3)).astype(float)
        elif kind == 6: pts = rng.normal(size=(6, 3)); pts[:, 2] = 0.5
        elif kind == 7: pts = rng.normal(size=(6, 3)); pts[3:] = pts[:3]
        elif kind == 8: pts = np.zeros((6, 3)); pts[:, 0] = np.linspace(0.3, 2.0, 6)       # origin on the extension
        else: pts = rng.normal(size=(6, 3)) * 1e-3 + np.array([0.3, 0.0, 0.0])           # thin hull near the origin
        out.append(pts.astype(np.float32).astype(np.float64))
    return np.asarray(out)


def main():
    ref = O.ref_gjk_lib()
    if ref is None:
        raise SystemExit("oracle/_ref/libref_opengjk.so missing: run `make -C oracle ref` in the build container")
    rng = np.random.default_rng(20260928)
    pts = families(rng, 4000)
    zero = np.zeros(3)
    dp = ctypes.POINTER(ctypes.c_double)
    v = np.zeros((len(pts), 3)); d = np.zeros(len(pts)); nv = np.zeros(len(pts), np.int32)
    # the reference prints a debug line on some degenerate inputs; silence fd 1 while it runs
    sys.stdout.flush()
    saved = os.dup(1); devnull = os.open(os.devnull, os.O_WRONLY); os.dup2(devnull, 1)
    try:
        for i, p in enumerate(pts):
            n = ctypes.c_int()
            d[i] = ref.ref_gjk(p.ctypes.data_as(dp), 6, zero.ctypes.data_as(dp), 1, v[i].ctypes.data_as(dp), ctypes.byref(n))
            nv[i] = n.value
    finally:
        os.dup2(saved, 1); os.close(devnull); os.close(saved)
    np.savez_compressed(os.path.join(os.path.dirname(__file__), "gjk_vectors.npz"), pts=pts.astype(np.float32), v=v, dist=d, nvrtx=nv)
    print("wrote", len(pts), "vectors; simplex sizes:", np.bincount(nv))


if __name__ == "__main__":
    main()
