"""Planar worlds (world/dimension = 2) in the oracle: the QP is the reference's 60-variable model.

src/traj_optimizer.cpp: dim = param.world_dimension (:8); dim * M * (n + 1) variables (:264-266); cost, equalities, velocity and
acceleration rows over k < dim (:330, 394, 469); terminal cost, corridor rows and collision rows without their z term (:367, 423,
450); Box::convertToLSCs(dim) emits 2 dim half-spaces (src/collision_constraints.cpp:37-59); stop rows over k < dim (:529); the
stored control points get z = world/z_2d (:87-90).  Rounds 2-3 solved a 90-variable QP there -- oracle and product agreeing with
each other, not with the reference ("weak" #1).
"""
import numpy as np
import pytest

import highs_qp as H

Z2D = 0.7


def _planar_scene(O, n=8, ticks=9, use_sfc=False):
    """A planar circle swap flown for a few ticks with the oracle; returns what the QP of the last tick is made of."""
    import lsc_planner_amd as L
    ms = L.circle_swap(n, circle_radius=1.6, z=Z2D, world=(-5, -5, 0, 5, 5, 2.5))
    prm = O.make_params(world_min=ms.world_min, world_max=ms.world_max, obs_f32=True, world_dimension=2, world_z_2d=Z2D)
    sw = O.Swarm(prm, ms.radius, ms.downwash, ms.max_vel, ms.max_acc, ms.nominal_velocity)
    state = np.zeros((n, 9), np.float32)
    state[:, :3] = ms.start
    traj = np.zeros((n, 3, 30), np.float32)
    hist = []
    for tick in range(1, ticks + 1):
        goals = O.goal_prior_based(state, ms.goal, traj, tick, prm=prm)
        o = sw.tick(state, goals, traj, tick, want_lsc=True)
        hist.append((state.copy(), goals.copy(), traj.copy(), o))
        assert (o["status"] == 0).all()
        traj = o["traj"]
        state = np.array([O.next_state(traj[q]) for q in range(n)], np.float32)
    return ms, prm, hist


def _qp_of(O, prm, ms, a, state, goals, prev, o, tick, sfc=None):
    n = ms.qn
    others = [j for j in range(n) if j != a]
    shift = (lambda j: O.const_vel_traj(state[j, :3], state[j, 3:6])) if tick < 2 else (lambda j: O.shift_traj(prev[j]))
    obs = np.array([shift(j) for j in others])
    return O.qp_assemble(prm, state[a], goals[a], ms.nominal_velocity[a], ms.max_vel[a], ms.max_acc[a], obs, o["normal"][a],
                         o["d"][a], sfc=sfc)


def test_every_planned_control_point_sits_at_z_2d(oracle):
    ms, prm, hist = _planar_scene(oracle)
    for state, goals, prev, o in hist:
        assert (o["traj"][:, 2, :] == np.float32(Z2D)).all()
        assert (state[:, 2] == np.float32(Z2D)).all() and (state[:, 5] == 0).all() and (state[:, 8] == 0).all()
        assert (o["normal"][..., 2] == 0).all()          # the whole swarm is in the plane: no normal leaves it


def test_planar_qp_is_the_reference_s_60_variable_model(oracle):
    ms, prm, hist = _planar_scene(oracle)
    n = ms.qn
    tick = len(hist)
    state, goals, prev, o = hist[-1]
    prm3 = oracle.make_params(world_min=ms.world_min, world_max=ms.world_max, obs_f32=True)
    for a in (0, 3):
        qp = _qp_of(oracle, prm, ms, a, state, goals, prev, o, tick)
        qp3 = _qp_of(oracle, prm3, ms, a, state, goals, prev, o, tick)
        assert qp.nv == 60 and qp3.nv == 90
        # rows in populatebyrow's order: 15 equalities per axis, 27 rows per obstacle, 84 dynamic-limit rows per axis, 2 stop rows per axis
        assert qp.nrows == 2 * 15 + 27 * (n - 1) + 2 * 84 + 2 * 2
        assert qp3.nrows == 3 * 15 + 27 * (n - 1) + 3 * 84 + 3 * 2
        assert np.array_equal(qp.P, qp3.P[:60, :60]) and np.array_equal(qp.c, qp3.c[:60])
        assert np.array_equal(qp.lo, qp3.lo[:60]) and np.array_equal(qp.hi, qp3.hi[:60])
        T = oracle.lib().orc_terminal_segments(oracle._f(goals[a]), oracle._f(state[a]), ms.nominal_velocity[a], 0.2)
        assert qp3.cst - qp.cst == pytest.approx(T * float(goals[a, 2]) ** 2, rel=1e-12)
        # the planar rows are the 3-D rows of the x and y axes with the z coefficient cut off
        rows3 = [qp3.row(r) for r in range(qp3.nrows)]
        keep = [R for R in rows3 if min(R[0]) < 60]                       # rows that touch x or y at all
        assert len(keep) == qp.nrows
        for r, R3 in enumerate(keep):
            idx, val, rhs, sense = qp.row(r)
            assert max(idx) < 60 and sense == R3[3]
            cut = [(i, v) for i, v in zip(R3[0], R3[1]) if i < 60]
            assert idx == [i for i, _ in cut] and val == [v for _, v in cut]
            assert rhs == R3[2]                                           # n_z = 0 here, so dropping n_z q_z changes nothing
        # collision rows carry two coefficients, not three (:446-453)
        lsc_rows = range(30, 30 + 27 * (n - 1))
        assert all(len(qp.row(r)[0]) == 2 for r in lsc_rows)


def test_planar_optimum_vs_highs_and_vs_the_3d_solve(oracle):
    """HiGHS on the 60-variable model returns the oracle's optimum; and, since the z block decouples when the whole swarm sits
    at z_2d, the 90-variable solve of rounds 2-3 had the same x / y plan and cost -- which is why nothing was red."""
    ms, prm, hist = _planar_scene(oracle)
    tick = len(hist)
    state, goals, prev, o = hist[-1]
    prm3 = oracle.make_params(world_min=ms.world_min, world_max=ms.world_max, obs_f32=True)
    for a in range(ms.qn):
        qp = _qp_of(oracle, prm, ms, a, state, goals, prev, o, tick)
        st, x, cost, it, kkt = qp.solve()
        assert st == 0 and cost == pytest.approx(o["cost"][a], rel=1e-12)
        hs, hx, hcost, viol = H.solve_oracle_qp(qp)
        assert hs == "Optimal" and abs(hcost - cost) <= 1e-7 * abs(cost) + 1e-9, (a, hs, hcost, cost)
        qp3 = _qp_of(oracle, prm3, ms, a, state, goals, prev, o, tick)
        st3, x3, cost3, _, _ = qp3.solve()
        assert st3 == 0 and abs(cost3 - cost) <= 1e-8 * abs(cost) + 1e-10
        assert np.abs(x3[:60] - x).max() < 2e-6 and np.abs(x3[60:] - np.float32(Z2D)).max() < 1e-6


def test_corridor_rows_of_a_planar_world_are_four_half_spaces(oracle):
    """Box::convertToLSCs(param.world_dimension) (src/collision_constraints.cpp:37-59): x and y faces only."""
    ms, prm, hist = _planar_scene(oracle, ticks=3)
    prm_sfc = oracle.make_params(world_min=ms.world_min, world_max=ms.world_max, obs_f32=True, world_dimension=2, world_z_2d=Z2D,
                                 use_sfc=True)
    state, goals, prev, o = hist[-1]
    a = 1
    box = np.tile(np.r_[state[a, :3] - 0.6, state[a, :3] + 0.6].astype(np.float32), (5, 1))
    qp = _qp_of(oracle, prm_sfc, ms, a, state, goals, prev, o, 3, sfc=box)
    n = ms.qn
    assert qp.nrows == 30 + 4 * 27 + 27 * (n - 1) + 168 + 4
    for r in range(30, 30 + 4 * 27):
        idx, val, rhs, sense = qp.row(r)
        assert len(idx) == 1 and idx[0] < 60 and sense == 1 and abs(val[0]) == 1.0
    # face order of a segment: x_min, x_max, y_min, y_max, each over the control points that carry rows
    idx0, val0, rhs0, _ = qp.row(30)
    assert idx0 == [3] and val0 == [1.0] and rhs0 == float(box[0, 0])
    st, x, cost, _, _ = qp.solve()
    assert st == 0
