"""Independent solver for pinning the oracle's QP optimum and infeasibility verdict: HiGHS (the copy SciPy bundles,
scipy.optimize._highspy -- HiGHS 1.8, active-set QP solver + dual simplex LP), driven through its own model API.

The reference solves these QPs with CPLEX 20.1 (src/traj_optimizer.cpp:31-154), which is proprietary and absent here.
A strictly convex QP has ONE optimum and infeasibility is a property of the constraint set, so agreement with a second,
unrelated, publicly audited solver pins both -- HiGHS shares no code, no algorithm (active set vs. the oracle's
interior point) and no author with the oracle or the kernel.  It also reads CPLEX LP files, so the reference's own
fixture log/QPmodel.lp can be handed to it verbatim (tests/test_oracle_pins.py).

TEST INFRASTRUCTURE ONLY.
"""
import numpy as np
import scipy.sparse as sp

try:
    import scipy.optimize._highspy._core as _hc
except Exception:  # pragma: no cover
    _hc = None


def available():
    return _hc is not None


def version():
    return f"{_hc.HIGHS_VERSION_MAJOR}.{_hc.HIGHS_VERSION_MINOR}.{_hc.HIGHS_VERSION_PATCH}"


def _new(time_limit=5.0):
    h = _hc._Highs()
    h.setOptionValue("output_flag", False)
    h.setOptionValue("time_limit", float(time_limit))
    # (no "threads" option: HiGHS sizes ONE process-wide task scheduler at the first run; an instance that later asks for another
    # thread count -- ours next to scipy.optimize.linprog's default -- refuses to run and reports "Not Set")
    h.setOptionValue("primal_feasibility_tolerance", 1e-9)
    h.setOptionValue("dual_feasibility_tolerance", 1e-9)
    return h


def _pass_lp(h, c, A, row_lo, row_hi, col_lo, col_hi, offset=0.0, P=None):
    inf = _hc.kHighsInf
    n = len(c)
    A = sp.csc_matrix(A) if A is not None and A.shape[0] else sp.csc_matrix((0, n))
    lp = _hc.HighsLp()
    lp.num_col_, lp.num_row_ = n, A.shape[0]
    lp.col_cost_ = np.asarray(c, float)
    lp.col_lower_ = np.where(np.isfinite(col_lo), col_lo, -inf)
    lp.col_upper_ = np.where(np.isfinite(col_hi), col_hi, inf)
    lp.row_lower_ = np.where(np.isfinite(row_lo), row_lo, -inf)
    lp.row_upper_ = np.where(np.isfinite(row_hi), row_hi, inf)
    lp.offset_ = float(offset)
    lp.a_matrix_.format_ = _hc.MatrixFormat.kColwise
    lp.a_matrix_.num_col_, lp.a_matrix_.num_row_ = n, A.shape[0]
    lp.a_matrix_.start_ = A.indptr.astype(np.int32)
    lp.a_matrix_.index_ = A.indices.astype(np.int32)
    lp.a_matrix_.value_ = A.data.astype(float)
    model = _hc.HighsModel()
    model.lp_ = lp
    if P is not None:
        L = sp.csc_matrix(np.tril(P))            # HiGHS wants the lower triangle of the Hessian of (1/2) x'Px, column-wise
        hs = _hc.HighsHessian()
        hs.dim_ = n
        hs.format_ = _hc.HessianFormat.kTriangular
        hs.start_ = L.indptr.astype(np.int32)
        hs.index_ = L.indices.astype(np.int32)
        hs.value_ = L.data.astype(float)
        model.hessian_ = hs
    st = h.passModel(model)
    assert st != _hc.HighsStatus.kError, "HiGHS rejected the model"


def rows_of(qp):
    """(A, lo, hi) of an oracle.QP: every constraint row as lo <= a.x <= hi (sense 0 '=', 1 '>=', 2 '<=')."""
    n = len(qp.c)
    A = np.zeros((qp.nrows, n))
    lo = np.full(qp.nrows, -np.inf)
    hi = np.full(qp.nrows, np.inf)
    for r in range(qp.nrows):
        idx, val, rhs, sense = qp.row(r)
        A[r, idx] = val
        if sense == 0:
            lo[r] = hi[r] = rhs
        elif sense == 1:
            lo[r] = rhs
        else:
            hi[r] = rhs
    return A, lo, hi


def solve_qp(P, c, cst, A, row_lo, row_hi, col_lo, col_hi):
    """min (1/2) x'Px + c'x + cst.  Returns (status string, x, objective)."""
    h = _new()
    _pass_lp(h, c, A, row_lo, row_hi, col_lo, col_hi, offset=cst, P=P)
    h.run()
    ms = h.modelStatusToString(h.getModelStatus())
    x = np.asarray(h.getSolution().col_value, float)
    return ms, x, float(h.getInfo().objective_function_value)


def solve_oracle_qp(qp):
    """HiGHS on an oracle.QP, through a small portfolio of EXACT reformulations of the same problem, because HiGHS 1.8's
    active-set QP code is not robust on every instance: (1) the 45/51 equality rows eliminated with an orthonormal
    null-space basis from an SVD (x = x0 + Z y; neither the oracle's Householder basis nor the kernel's analytic
    elimination) -- on the full problem it satisfies equalities with +-500/+-1000 coefficients only to ~1e-5 relative and then
    reports a slightly super-optimal "Optimal"; (2) the full problem as is; (3) the full problem with the slack variables
    (cost weight 1e5) rescaled by 1e-2; (4) formulation (1) with a 1e-9 proximal term (moves the optimum by < 1e-9
    relative, breaks degenerate ties).  The first formulation that ends "Optimal" with a point satisfying the ORIGINAL rows
    and bounds to 1e-7 is returned as (status, x in the original variables, objective, that violation); "Infeasible" from
    the first formulation is returned as such; otherwise the last status."""
    n = len(qp.c)
    A, lo, hi = rows_of(qp)

    def viol_of(x):
        if not np.isfinite(x).all():
            return np.inf
        return float(max(np.max(lo - A @ x), np.max(A @ x - hi), np.max(qp.lo - x), np.max(x - qp.hi)))

    def obj_of(x):
        return float(0.5 * x @ qp.P @ x + qp.c @ x + qp.cst)

    eq = lo == hi
    Ae, be = A[eq], lo[eq]
    pinv, Z = _null_space(Ae)
    x0 = pinv @ be
    fin = np.isfinite(qp.lo) | np.isfinite(qp.hi)
    Ai = np.vstack([A[~eq], np.eye(n)[fin]])
    li, ui = np.r_[lo[~eq], qp.lo[fin]], np.r_[hi[~eq], qp.hi[fin]]
    off = Ai @ x0
    P = Z.T @ qp.P @ Z
    P = 0.5 * (P + P.T)
    c = Z.T @ (qp.P @ x0 + qp.c)
    cst = 0.5 * x0 @ qp.P @ x0 + qp.c @ x0 + qp.cst
    ny = Z.shape[1]
    free = (np.full(ny, -np.inf), np.full(ny, np.inf))

    def reduced(prox):
        st, y, _ = solve_qp(P + prox * np.eye(ny), c, cst, Ai @ Z, li - off, ui - off, *free)
        return st, (x0 + Z @ y if len(y) == ny else np.full(n, np.nan))

    def full(scale):
        S = np.ones(n)
        S[90:] = scale
        st, xs, _ = solve_qp(qp.P * S[:, None] * S[None, :], qp.c * S, qp.cst, A * S[None, :], lo, hi, qp.lo / S, qp.hi / S)
        return st, (xs * S if len(xs) == n else np.full(n, np.nan))

    last = None
    for k, attempt in enumerate((lambda: reduced(0.0), lambda: full(1.0), lambda: full(1e-2), lambda: reduced(1e-9))):
        if k == 2 and n == 90:
            continue
        st, x = attempt()
        if k == 0 and st == "Infeasible":
            return st, x, np.nan, np.inf
        if st == "Optimal" and viol_of(x) <= 1e-7:
            return st, x, obj_of(x), viol_of(x)
        last = (st if st != "Optimal" else "Solve error", x, np.nan, viol_of(x))
    return last


_NS_CACHE = {}


def _null_space(Ae):
    """(pseudo-inverse, orthonormal null-space basis) of the equality block; it is the same matrix for every QP of a
    mission (only the right-hand side changes), so the SVD is done once per distinct matrix."""
    key = (Ae.shape, Ae.tobytes())
    if key not in _NS_CACHE:
        import scipy.linalg as sl
        if len(_NS_CACHE) > 8:
            _NS_CACHE.clear()
        _NS_CACHE[key] = (np.linalg.pinv(Ae), sl.null_space(Ae))
    return _NS_CACHE[key]


def min_violation(A, row_lo, row_hi, col_lo, col_hi):
    """Phase-1 LP: the smallest uniform violation t >= 0 such that  lo - t <= A x <= hi + t  (equalities and variable
    bounds kept exact).  t* > 0 certifies that the constraint set is empty; t* = 0 that it is not."""
    m, n = A.shape
    rows, lo, hi = [], [], []
    for r in range(m):
        if row_lo[r] == row_hi[r]:
            rows.append(np.r_[A[r], 0.0]); lo.append(row_lo[r]); hi.append(row_hi[r])
            continue
        if np.isfinite(row_lo[r]):
            rows.append(np.r_[A[r], 1.0]); lo.append(row_lo[r]); hi.append(np.inf)
        if np.isfinite(row_hi[r]):
            rows.append(np.r_[A[r], -1.0]); lo.append(-np.inf); hi.append(row_hi[r])
    h = _new()
    _pass_lp(h, np.r_[np.zeros(n), 1.0], np.asarray(rows), np.asarray(lo), np.asarray(hi), np.r_[col_lo, 0.0], np.r_[col_hi, np.inf])
    h.run()
    ms = h.modelStatusToString(h.getModelStatus())
    return ms, float(h.getInfo().objective_function_value)


def read_lp_file(path):
    """Hands a CPLEX-LP file to HiGHS as is (its own reader) and runs it: (model status, rows, cols, Hessian nnz)."""
    h = _new()
    st = h.readModel(path)
    assert st != _hc.HighsStatus.kError, f"HiGHS could not read {path}"
    h.run()
    return h.modelStatusToString(h.getModelStatus()), h.getNumRow(), h.getNumCol(), h.getHessianNumNz()
