"""Oracle QP assembly vs. the reference fixture log/QPmodel.lp, and the exact solve's optimality certificate."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN


@pytest.fixture(scope="module")
def lp():
    return json.load(open(os.path.join(GOLDEN, "qpmodel_lp.json")))


def _scene_qp(O, lp):
    sc = lp["scene"]
    starts = np.array(sc["starts_xy_z07"], np.float32)
    N, a = len(starts), sc["agent"]
    lin = np.zeros(90)
    for k, v in lp["lin"].items():
        lin[int(k)] = v
    goal = (-lin / 2)[[29, 59, 89]].astype(np.float32)
    prm = O.make_params(world_min=sc["world"][:3], world_max=sc["world"][3:], obs_f32=True)
    state = np.zeros(9, np.float32); state[:3] = starts[a]
    init = O.const_vel_traj(starts[a], np.zeros(3))
    others = [j for j in range(N) if j != a]
    obs = np.array([O.const_vel_traj(starts[j], np.zeros(3)) for j in others])
    r_o = float(np.float32(sc["radius"]))
    nd = [O.lsc_pair(init, o, sc["radius"], r_o, sc["downwash"], float(np.float32(sc["downwash"]))) for o in obs]
    qp = O.qp_assemble(prm, state, goal, sc["nominal_velocity"], sc["max_vel"], sc["max_acc"], obs,
                       np.array([x[0] for x in nd]), np.array([x[1] for x in nd]))
    return qp, lin


def test_rows_match_lp_fixture(oracle, lp):
    qp, lin = _scene_qp(oracle, lp)
    assert qp.nrows == len(lp["rows"]) == 546
    sense = {"=": 0, ">=": 1, "<=": 2}
    for r, R in enumerate(lp["rows"]):
        idx, val, rhs, s = qp.row(r)
        assert s == sense[R["sense"]]
        a1 = np.zeros(90); a1[idx] = val
        a2 = np.zeros(90); a2[R["idx"]] = R["val"]
        assert np.abs(a1 - a2).max() <= 1e-12 * max(1.0, np.abs(a2).max()), R["name"]
        assert abs(rhs - R["rhs"]) <= 1e-13 * max(1.0, abs(R["rhs"])), R["name"]     # LP prints 15 digits


def test_objective_and_bounds_match_lp_fixture(oracle, lp):
    qp, lin = _scene_qp(oracle, lp)
    P = np.zeros((90, 90))
    for i, j, v in lp["quad"]:          # "[ ... ] / 2" section: v x_i x_j
        if i == j: P[i, i] += v
        else: P[i, j] += v / 2; P[j, i] += v / 2
    assert np.abs(P - qp.P).max() <= 1e-12 * np.abs(P).max()
    assert np.abs(lin - qp.c).max() <= 1e-13
    for i in range(90):
        lo, hi = lp["bounds"][str(i)]
        if lo is None:
            assert not np.isfinite(qp.lo[i]) and not np.isfinite(qp.hi[i])
        else:
            assert abs(lo - qp.lo[i]) < 1e-14 and abs(hi - qp.hi[i]) < 1e-14


def test_lp_fixture_is_infeasible_like_the_reference_dump(oracle, lp):
    """The LP file is written on a CPLEX failure (src/traj_optimizer.cpp:100-102): the instance is infeasible."""
    qp, _ = _scene_qp(oracle, lp)
    st, x, cost, it, kkt = qp.solve()
    assert st == 1
    from scipy.optimize import linprog
    Aeq, beq, G, h = qp.dense()
    res = linprog(np.r_[np.zeros(90), 1.0], A_ub=np.c_[G, -np.ones(len(h))], b_ub=h,
                  A_eq=np.c_[Aeq, np.zeros(len(beq))], b_eq=beq, bounds=[(None, None)] * 91, method="highs")
    assert res.status == 0 and abs(res.fun - 0.0385092) < 1e-6      # minimal uniform violation


def test_constants_match_survey_appendix(oracle):
    Q = oracle.qbase(0.2)
    Qn = np.array([[720, -1800, 1200, 0, 0, -120], [-1800, 4800, -3600, 0, 600, 0], [1200, -3600, 3600, -1200, 0, 0],
                   [0, 0, -1200, 3600, -3600, 1200], [0, 600, 0, -3600, 4800, -1800], [-120, 0, 0, 1200, -1800, 720]], float)
    assert np.abs(Q - 3125 * Qn).max() < 1e-6
    A = oracle.aeq_base(0.2)
    assert np.allclose(A[1, :2], [-25, 25]) and np.allclose(A[2, :3], [500, -1000, 500])
    assert np.linalg.matrix_rank(A) == 15


def test_solver_optimality_certificate_and_scipy_cross_check(oracle, ticks):
    """KKT residuals certify the optimum of a convex QP independently of the solver; SLSQP agrees on the cost."""
    from conftest import golden_mission
    ms = golden_mission(ticks, "multi_simple4")
    prm = oracle.make_params(world_min=ms.world_min, world_max=ms.world_max, obs_f32=True)
    for tick in (2, 20):
        state, prev = ticks[f"multi_simple4/tick{tick}/state"], ticks[f"multi_simple4/tick{tick}/prev"]
        for a in range(ms.qn):
            others = [j for j in range(ms.qn) if j != a]
            obs = np.array([oracle.shift_traj(prev[j]) for j in others])
            nrm = ticks[f"multi_simple4/tick{tick}/normal"][a]
            dd = ticks[f"multi_simple4/tick{tick}/d"][a]
            qp = oracle.qp_assemble(prm, state[a], ms.goal[a], 1.0, ms.max_vel[a], ms.max_acc[a], obs, nrm, dd)
            st, x, cost, it, kkt = qp.solve()
            assert st == 0
            assert kkt[0] < 1e-5 * (1 + abs(cost)) and kkt[1] < 1e-8 and kkt[2] == 0 and kkt[3] < 1e-8 * (1 + abs(cost))
            assert abs(cost - ticks[f"multi_simple4/tick{tick}/cost"][a]) <= 1e-9 * abs(cost)
            if a == 0:
                from scipy.optimize import minimize
                Aeq, beq, G, h = qp.dense()
                f = lambda z: 0.5 * z @ qp.P @ z + qp.c @ z + qp.cst
                res = minimize(f, x + 0.01, jac=lambda z: qp.P @ z + qp.c, method="SLSQP",
                               constraints=[{"type": "eq", "fun": lambda z: Aeq @ z - beq, "jac": lambda z: Aeq},
                                            {"type": "ineq", "fun": lambda z: h - G @ z, "jac": lambda z: -G}],
                               options={"maxiter": 500, "ftol": 1e-14})
                assert res.fun >= cost - 1e-6 * abs(cost)          # nothing feasible beats the certified optimum
                assert abs(res.fun - cost) <= 1e-5 * abs(cost)


def test_ctypes_mirrors_of_the_oracle_structs_match_the_header(tmp_path):
    """oracle/oracle.py mirrors orc_params / orc_modes / orc_row of oracle/lsc_oracle.h: same size, same field offsets."""
    import ctypes
    import os
    import subprocess
    from conftest import ROOT
    from oracle import oracle as O
    structs = {"orc_params": O.OrcParams, "orc_modes": O.OrcModes, "orc_row": O.OrcRow}
    body = ""
    for cname, cls in structs.items():
        body += '    printf("%s.sizeof %%zu\\n", sizeof(%s));\n' % (cname, cname)
        for f in cls._fields_:
            body += '    printf("%s.%s %%zu\\n", offsetof(%s, %s));\n' % (cname, f[0], cname, f[0])
    src = tmp_path / "layout.c"
    src.write_text('#include <stddef.h>\n#include <stdio.h>\n#include "lsc_oracle.h"\nint main(void) {\n' + body + '    return 0;\n}\n')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "oracle"), str(src), "-o", str(exe)])
    out = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    for cname, cls in structs.items():
        assert int(out[cname + ".sizeof"]) == ctypes.sizeof(cls), cname
        for f in cls._fields_:
            assert int(out["%s.%s" % (cname, f[0])]) == getattr(cls, f[0]).offset, (cname, f[0])
