"""ctypes front-end of the CPU oracle (oracle/liblsc_oracle.so).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never from lsc_planner_amd/.
"""
import ctypes
import os
import subprocess

import numpy as np

import contextlib

_HERE = os.path.dirname(os.path.abspath(__file__))
M, NDEG, NC, SEGV, NV = 5, 5, 6, 30, 90


@contextlib.contextmanager
def segments(m):
    """Everything inside the block uses the oracle built for M = m segments (liblsc_oracle.so: 5, liblsc_oracle_m4.so: 4) --
    array shapes included.  Objects made inside (Swarm, QP) belong to that M."""
    global M, SEGV, NV
    old = M
    M, SEGV, NV = m, NC * m, 3 * NC * m
    try:
        yield
    finally:
        M, SEGV, NV = old, NC * old, 3 * NC * old

_dp = ctypes.POINTER(ctypes.c_double)
_fp = ctypes.POINTER(ctypes.c_float)
_ip = ctypes.POINTER(ctypes.c_int)


class OrcParams(ctypes.Structure):
    _fields_ = [
        ("dt", ctypes.c_double),
        ("w_control", ctypes.c_double),
        ("w_terminal", ctypes.c_double),
        ("world_min", ctypes.c_float * 3),
        ("world_max", ctypes.c_float * 3),
        ("reset_threshold", ctypes.c_double),
        ("use_sfc", ctypes.c_int),
        ("obs_f32", ctypes.c_int),
        ("world_dimension", ctypes.c_int),
        ("world_z_2d", ctypes.c_double),
    ]


class OrcRow(ctypes.Structure):
    _fields_ = [
        ("nnz", ctypes.c_int),
        ("idx", ctypes.c_int * 9),
        ("val", ctypes.c_double * 9),
        ("rhs", ctypes.c_double),
        ("sense", ctypes.c_int),
    ]


class OrcModes(ctypes.Structure):
    _fields_ = [
        ("planner_mode", ctypes.c_int),
        ("slack_mode", ctypes.c_int),
        ("slack_weight", ctypes.c_double),
        ("n_constraint_segments", ctypes.c_int),
        ("reset_threshold", ctypes.c_double),
    ]


def make_modes(planner="lsc", slack="none", slack_weight=100000.0, n_constraint_segments=-1, reset_threshold=0.0):
    """Alternate-mode switches (mode/planner, SlackMode, opt/slack_collision_weight, opt/N_constraint_segments,
    multisim/reset_threshold; reset_threshold <= 0 switches the disturbance checks off)."""
    m = OrcModes()
    m.planner_mode = {"lsc": 0, "bvc": 1}[planner]
    m.slack_mode = {"none": 0, "dynamical_limit": 1, "collision_constraint": 2}[slack]
    m.slack_weight, m.n_constraint_segments, m.reset_threshold = slack_weight, n_constraint_segments, reset_threshold
    return m


class OrcEdt(ctypes.Structure):
    _fields_ = [
        ("dist", _fp),
        ("nx", ctypes.c_int), ("ny", ctypes.c_int), ("nz", ctypes.c_int),
        ("key_min", ctypes.c_int * 3),
        ("res", ctypes.c_double),
    ]


def build(force=False, m=5):
    name = "liblsc_oracle.so" if m == 5 else f"liblsc_oracle_m{m}.so"
    so = os.path.join(_HERE, name)
    srcs = [os.path.join(_HERE, f) for f in ("lsc_oracle.c", "lsc_oracle_sfc.c", "lsc_oracle_modes.c", "lsc_oracle_goal.cpp", "lsc_oracle.h")]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, name], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference") and (force or not os.path.exists(os.path.join(_HERE, "_ref", "libref_opengjk.so"))):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
    return so


_libs = {}


def lib():
    if M not in _libs:
        L = ctypes.CDLL(build(m=M))
        L.orc_segments.restype = ctypes.c_int
        assert L.orc_segments() == M
        L.orc_gjk_origin.restype = ctypes.c_double
        L.orc_gjk_origin.argtypes = [_dp, ctypes.c_int, _dp, _ip, _ip]
        L.orc_qbase.argtypes = [ctypes.c_double, _dp]
        L.orc_aeq_base.argtypes = [ctypes.c_double, _dp]
        L.orc_lsc_pair.argtypes = [_fp, _fp, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, _fp, _dp]
        L.orc_qp_assemble.restype = ctypes.c_int
        L.orc_qp_assemble.argtypes = [ctypes.POINTER(OrcParams), _fp, _fp, ctypes.c_double, _dp, _dp, ctypes.c_int,
                                      _fp, _fp, _dp, _fp, _dp, _dp, _dp, _dp, _dp, ctypes.POINTER(OrcRow)]
        L.orc_qp_nvars.restype = ctypes.c_int
        L.orc_qp_nvars.argtypes = [ctypes.POINTER(OrcParams)]
        L.orc_qp_solve.restype = ctypes.c_int
        L.orc_qp_solve.argtypes = [_dp, _dp, ctypes.c_double, _dp, _dp, ctypes.POINTER(OrcRow), ctypes.c_int, _dp, _dp, _ip, _dp]
        L.orc_next_state.argtypes = [_fp, ctypes.c_double, _fp]
        L.orc_shift_traj.argtypes = [_fp, _fp]
        L.orc_const_vel_traj.argtypes = [_fp, _fp, ctypes.c_double, _fp]
        L.orc_terminal_segments.restype = ctypes.c_int
        L.orc_terminal_segments.argtypes = [_fp, _fp, ctypes.c_double, ctypes.c_double]
        L.orc_tick.restype = ctypes.c_int
        L.orc_tick.argtypes = [ctypes.POINTER(OrcParams), ctypes.c_int, _fp, _fp, _fp, ctypes.c_int, _dp, _dp, _dp, _dp, _dp,
                               _fp, _fp, _fp, _dp, _ip, _ip, _fp, _dp, ctypes.c_int]
        L.orc_expand_box.restype = ctypes.c_int
        L.orc_expand_box.argtypes = [ctypes.POINTER(OrcParams), ctypes.POINTER(OrcEdt), ctypes.c_double, _fp, _fp, ctypes.c_double, _dp]
        L.orc_bt_read.restype = ctypes.c_int
        L.orc_bt_read.argtypes = [ctypes.c_char_p, _dp, ctypes.POINTER(_ip), _ip]
        L.orc_edt_build.restype = ctypes.c_int
        L.orc_edt_build.argtypes = [_ip, ctypes.c_int, ctypes.c_double, _fp, _fp, ctypes.c_double, ctypes.POINTER(OrcEdt)]
        L.orc_edt_brushfire.restype = ctypes.c_int
        L.orc_edt_brushfire.argtypes = [_ip, ctypes.c_int, ctypes.c_double, _fp, _fp, ctypes.c_double, ctypes.POINTER(OrcEdt), _ip]
        L.orc_goal_prior_based.restype = None
        L.orc_goal_prior_based.argtypes = [ctypes.c_int, ctypes.c_int, _fp, _fp, _fp, ctypes.c_int, ctypes.c_double, ctypes.c_double,
                                           ctypes.c_double, ctypes.c_double, _fp]
        L.orc_tick_set_map.restype = None
        L.orc_tick_set_map.argtypes = [ctypes.c_void_p, ctypes.c_double, _ip]
        _ub = ctypes.POINTER(ctypes.c_ubyte)
        L.orc_grid_dims.restype = None
        L.orc_grid_dims.argtypes = [ctypes.POINTER(OrcParams), ctypes.c_double, _ip, _dp]
        L.orc_astar.argtypes = [_ub, _ip, _ip, _ip, _ip, ctypes.c_int]
        L.orc_goal_prior_based_map.restype = None
        L.orc_goal_prior_based_map.argtypes = [ctypes.POINTER(OrcParams), ctypes.POINTER(OrcEdt), ctypes.c_double, ctypes.c_double,
                                               ctypes.c_double, ctypes.c_int, ctypes.c_int, _fp, _fp, _fp, ctypes.c_int,
                                               ctypes.c_double, ctypes.c_double, ctypes.c_double, _dp, _dp, ctypes.POINTER(ctypes.c_ubyte),
                                               ctypes.c_int, _fp, _ip, ctypes.c_int, _ip, _ip]
        _ubp = ctypes.POINTER(ctypes.c_ubyte)
        pm, pp = ctypes.POINTER(OrcModes), ctypes.POINTER(OrcParams)
        L.orc_qp_solve_n.restype = ctypes.c_int
        L.orc_qp_solve_n.argtypes = [ctypes.c_int, _dp, _dp, ctypes.c_double, _dp, _dp, ctypes.POINTER(OrcRow), ctypes.c_int, _dp, _dp, _ip, _dp]
        L.orc_slack_count.restype = ctypes.c_int
        L.orc_slack_count.argtypes = [pm, ctypes.c_int, _ubp]
        L.orc_qp_assemble_ex.restype = ctypes.c_int
        L.orc_qp_assemble_ex.argtypes = [pp, pm, _fp, _fp, ctypes.c_double, _dp, _dp, ctypes.c_int, _fp, _fp, _dp, _fp, _ubp, _ip,
                                         _dp, _dp, _dp, _dp, _dp, ctypes.POINTER(OrcRow)]
        L.orc_bvc_pair.restype = None
        L.orc_bvc_pair.argtypes = [_fp, _fp, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, _fp, _dp]
        L.orc_disturbance_update.restype = None
        L.orc_disturbance_update.argtypes = [pp, pm, ctypes.c_int, _fp, _fp, ctypes.c_int, _ubp, _ip, _ubp]
        L.orc_goal_prior_based_ex.restype = None
        L.orc_goal_prior_based_ex.argtypes = [ctypes.c_int, ctypes.c_int, _fp, _fp, _fp, ctypes.c_int, ctypes.c_double, ctypes.c_double,
                                              ctypes.c_double, ctypes.c_double, _ubp, ctypes.c_int, _fp]
        L.orc_tick_ex.restype = ctypes.c_int
        L.orc_tick_ex.argtypes = [pp, pm, ctypes.c_int, _fp, _fp, _fp, ctypes.c_int, _dp, _dp, _dp, _dp, _dp, _fp, _ubp,
                                  ctypes.POINTER(OrcEdt), ctypes.c_double, _fp, _ip, _fp, _dp, _ip, _ip, _fp, _dp, ctypes.c_int]
        _libs[M] = L
    return _libs[M]


def _f(a):
    return a.ctypes.data_as(_fp)


def _d(a):
    return a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


def make_params(dt=0.2, w_control=0.01, w_terminal=1.0, world_min=(-10, -10, 0), world_max=(10, 10, 2.5),
                reset_threshold=0.15, use_sfc=False, obs_f32=False, world_dimension=3, world_z_2d=1.0):
    p = OrcParams()
    p.dt, p.w_control, p.w_terminal = dt, w_control, w_terminal
    for k in range(3):
        p.world_min[k] = np.float32(world_min[k])
        p.world_max[k] = np.float32(world_max[k])
    p.reset_threshold = reset_threshold
    p.use_sfc = int(use_sfc)
    p.obs_f32 = int(obs_f32)
    p.world_dimension, p.world_z_2d = int(world_dimension), float(world_z_2d)
    return p


def gjk_origin(pts):
    pts = np.ascontiguousarray(pts, dtype=np.float64)
    v = np.zeros(3)
    nv, it = ctypes.c_int(), ctypes.c_int()
    d = lib().orc_gjk_origin(_d(pts), len(pts), _d(v), ctypes.byref(nv), ctypes.byref(it))
    return d, v, nv.value, it.value


def qbase(dt=0.2):
    q = np.zeros((NC, NC))
    lib().orc_qbase(dt, _d(q))
    return q


def aeq_base(dt=0.2):
    a = np.zeros((3 * M, SEGV))
    lib().orc_aeq_base(dt, _d(a))
    return a


def lsc_pair(init_traj, obs_traj, r_a, r_o, dw_a, dw_o):
    it = np.ascontiguousarray(init_traj, dtype=np.float32)
    ot = np.ascontiguousarray(obs_traj, dtype=np.float32)
    nrm = np.zeros((M, 3), np.float32)
    d = np.zeros((M, NC))
    lib().orc_lsc_pair(_f(it), _f(ot), r_a, r_o, dw_a, dw_o, _f(nrm), _d(d))
    return nrm, d


def next_state(traj, dt=0.2):
    t = np.ascontiguousarray(traj, dtype=np.float32)
    s = np.zeros(9, np.float32)
    lib().orc_next_state(_f(t), dt, _f(s))
    return s


def shift_traj(prev):
    p = np.ascontiguousarray(prev, dtype=np.float32)
    o = np.zeros_like(p)
    lib().orc_shift_traj(_f(p), _f(o))
    return o


def const_vel_traj(pos, vel, dt=0.2):
    pos = np.ascontiguousarray(pos, np.float32)
    vel = np.ascontiguousarray(vel, np.float32)
    o = np.zeros(NV, np.float32)
    lib().orc_const_vel_traj(_f(pos), _f(vel), dt, _f(o))
    return o.reshape(3, SEGV)


class QP:
    """Assembled QP of one agent in the reference's row order."""

    def __init__(self, P, c, cst, lo, hi, rows, nrows):
        self.P, self.c, self.cst, self.lo, self.hi = P, c, cst, lo, hi
        self._rows, self.nrows = rows, nrows

    def row(self, r):
        R = self._rows[r]
        return [R.idx[j] for j in range(R.nnz)], [R.val[j] for j in range(R.nnz)], R.rhs, R.sense

    @property
    def nv(self):
        return len(self.c)

    def dense(self):
        """(Aeq, beq, G, h) with inequalities as G x <= h (bounds appended: +x<=hi then -x<=-lo per variable)."""
        eq, be, G, h = [], [], [], []
        NV = self.nv
        for r in range(self.nrows):
            idx, val, rhs, sense = self.row(r)
            a = np.zeros(NV)
            a[idx] = val
            if sense == 0:
                eq.append(a); be.append(rhs)
            elif sense == 1:
                G.append(-a); h.append(-rhs)
            else:
                G.append(a); h.append(rhs)
        for j in range(NV):
            if np.isfinite(self.hi[j]):
                a = np.zeros(NV); a[j] = 1; G.append(a); h.append(self.hi[j])
            if np.isfinite(self.lo[j]):
                a = np.zeros(NV); a[j] = -1; G.append(a); h.append(-self.lo[j])
        return np.array(eq), np.array(be), np.array(G), np.array(h)

    def solve(self):
        x = np.zeros(self.nv)
        cost = ctypes.c_double()
        it = ctypes.c_int()
        kkt = np.zeros(4)
        st = lib().orc_qp_solve_n(self.nv, _d(self.P), _d(self.c), self.cst, _d(self.lo), _d(self.hi), self._rows, self.nrows,
                                  _d(x), ctypes.byref(cost), ctypes.byref(it), _d(kkt))
        return st, x, cost.value, it.value, kkt


def qp_assemble(prm, state, goal, v_nom, vmax, amax, obs_traj, normal, d, sfc=None):
    state = np.ascontiguousarray(state, np.float32)
    goal = np.ascontiguousarray(goal, np.float32)
    vmax = np.ascontiguousarray(vmax, np.float64)
    amax = np.ascontiguousarray(amax, np.float64)
    obs_traj = np.ascontiguousarray(obs_traj, np.float32)
    normal = np.ascontiguousarray(normal, np.float32)
    d = np.ascontiguousarray(d, np.float64)
    n_obs = obs_traj.shape[0] if obs_traj.size else 0
    rows = (OrcRow * (51 + 27 * n_obs + 252 + 162))()
    nv = lib().orc_qp_nvars(ctypes.byref(prm))          # 90, or 60 in a planar world (src/traj_optimizer.cpp:8, 264-266)
    P = np.zeros((nv, nv)); c = np.zeros(nv); lo = np.zeros(nv); hi = np.zeros(nv)
    cst = ctypes.c_double()
    sfc_p = _f(np.ascontiguousarray(sfc, np.float32)) if sfc is not None else None
    nr = lib().orc_qp_assemble(ctypes.byref(prm), _f(state), _f(goal), v_nom, _d(vmax), _d(amax), n_obs,
                               _f(obs_traj), _f(normal), _d(d), sfc_p, _d(P), _d(c), ctypes.byref(cst), _d(lo), _d(hi), rows)
    return QP(P, c, cst.value, lo, hi, rows, nr)


def qp_assemble_ex(prm, modes, state, goal, v_nom, vmax, amax, obs_traj, normal, d, slack_flags=None, sfc=None):
    """populatebyrow with the alternate-mode switches: variables = 90 control-point coordinates + the slack variables."""
    state = np.ascontiguousarray(state, np.float32)
    goal = np.ascontiguousarray(goal, np.float32)
    vmax = np.ascontiguousarray(vmax, np.float64)
    amax = np.ascontiguousarray(amax, np.float64)
    obs_traj = np.ascontiguousarray(obs_traj, np.float32)
    normal = np.ascontiguousarray(normal, np.float32)
    d = np.ascontiguousarray(d, np.float64)
    n_obs = obs_traj.shape[0] if obs_traj.size else 0
    fl = np.ascontiguousarray(slack_flags if slack_flags is not None else np.zeros(max(n_obs, 1)), np.uint8)
    ubp = ctypes.POINTER(ctypes.c_ubyte)
    nv = lib().orc_qp_nvars(ctypes.byref(prm)) + lib().orc_slack_count(ctypes.byref(modes), n_obs, fl.ctypes.data_as(ubp))
    rows = (OrcRow * (51 + 27 * n_obs + 252 + 162))()
    P = np.zeros((nv, nv)); c = np.zeros(nv); lo = np.zeros(nv); hi = np.zeros(nv)
    cst = ctypes.c_double()
    nvo = ctypes.c_int()
    sfc_p = _f(np.ascontiguousarray(sfc, np.float32)) if sfc is not None else None
    nr = lib().orc_qp_assemble_ex(ctypes.byref(prm), ctypes.byref(modes), _f(state), _f(goal), v_nom, _d(vmax), _d(amax), n_obs,
                                  _f(obs_traj), _f(normal), _d(d), sfc_p, fl.ctypes.data_as(ubp), ctypes.byref(nvo), _d(P), _d(c),
                                  ctypes.byref(cst), _d(lo), _d(hi), rows)
    assert nvo.value == nv
    return QP(P, c, cst.value, lo, hi, rows, nr)


def bvc_pair(init_traj, obs_traj, r_a, r_o, dw_a, dw_o):
    it = np.ascontiguousarray(init_traj, dtype=np.float32)
    ot = np.ascontiguousarray(obs_traj, dtype=np.float32)
    nrm = np.zeros((M, 3), np.float32)
    d = np.zeros((M, NC))
    lib().orc_bvc_pair(_f(it), _f(ot), r_a, r_o, dw_a, dw_o, _f(nrm), _d(d))
    return nrm, d


class Swarm:
    """Persistent per-swarm oracle state: mirrors what MultiSyncSimulator + N TrajPlanners hold between ticks."""

    def __init__(self, prm, radius, downwash, vmax, amax, vnom):
        self.prm = prm
        self.N = len(radius)
        self.radius = np.ascontiguousarray(radius, np.float64)
        self.downwash = np.ascontiguousarray(downwash, np.float64)
        self.vmax = np.ascontiguousarray(vmax, np.float64).reshape(self.N, 3)
        self.amax = np.ascontiguousarray(amax, np.float64).reshape(self.N, 3)
        self.vnom = np.ascontiguousarray(vnom, np.float64)
        self.stale = np.zeros((self.N, 3, SEGV), np.float32)
        self.cost = np.zeros(self.N)
        self.sfc = np.zeros((self.N, M, 6), np.float32)
        self.sfc_init = np.ones(self.N, np.int32)       # flag_initialize_sfc (src/traj_planner.cpp:48)
        self.distmap = None
        self.world_res = 0.1

    def set_distmap(self, distmap, world_res=0.1):
        self.distmap, self.world_res = distmap, world_res

    def tick(self, state, goal, prev_traj, planner_seq, want_lsc=False, nthreads=1):
        N = self.N
        state = np.ascontiguousarray(state, np.float32).reshape(N, 9)
        goal = np.ascontiguousarray(goal, np.float32).reshape(N, 3)
        prev = np.ascontiguousarray(prev_traj, np.float32).reshape(N, 3, SEGV)
        out = np.zeros((N, 3, SEGV), np.float32)
        status = np.zeros(N, np.int32)
        iters = np.zeros(N, np.int32)
        nrm = np.zeros((N, max(N - 1, 1), M, 3), np.float32) if want_lsc else None
        dd = np.zeros((N, max(N - 1, 1), M, NC)) if want_lsc else None
        if self.prm.use_sfc:
            lib().orc_tick_set_map(ctypes.addressof(self.distmap.edt), self.world_res, _i(self.sfc_init))
        else:
            lib().orc_tick_set_map(None, 0.1, None)
        lib().orc_tick(ctypes.byref(self.prm), N, _f(state), _f(goal), _f(prev), planner_seq, _d(self.radius),
                       _d(self.downwash), _d(self.vmax), _d(self.amax), _d(self.vnom), _f(self.stale),
                       _f(self.sfc) if self.prm.use_sfc else None, _f(out), _d(self.cost), _i(status), _i(iters),
                       _f(nrm) if want_lsc else None, _d(dd) if want_lsc else None, nthreads)
        res = {"traj": out, "cost": self.cost.copy(), "status": status, "iters": iters, "sfc": self.sfc.copy()}
        if want_lsc:
            res["normal"], res["d"] = nrm, dd
        return res


class SwarmEx(Swarm):
    """Swarm with the alternate-mode switches: BVC planner mode, slack modes, disturbance reset with its persistent slack
    set (slack_set[qi][qj] = 1 once agent qi put agent qj into obs_slack_indices; never cleared, like the reference)."""

    def __init__(self, prm, modes, radius, downwash, vmax, amax, vnom):
        super().__init__(prm, radius, downwash, vmax, amax, vnom)
        self.modes = modes
        self.slack_set = np.zeros((self.N, self.N), np.uint8)

    def disturbance_update(self, state, prev_traj, planner_seq):
        """The two checks the reference runs BEFORE goal planning; returns own_reset [N]."""
        ubp = ctypes.POINTER(ctypes.c_ubyte)
        state = np.ascontiguousarray(state, np.float32).reshape(self.N, 9)
        prev = np.ascontiguousarray(prev_traj, np.float32).reshape(self.N, NV)
        own = np.zeros(self.N, np.uint8)
        lib().orc_disturbance_update(ctypes.byref(self.prm), ctypes.byref(self.modes), self.N, _f(state), _f(prev), planner_seq,
                                     self.slack_set.ctypes.data_as(ubp), _i(self.sfc_init) if self.prm.use_sfc else None,
                                     own.ctypes.data_as(ubp))
        return own

    def goal_prior_based(self, state, desired_goal, prev_traj, planner_seq, own_reset=None, dt=0.2, goal_threshold=0.1,
                         priority_dist_threshold=0.4, goal_radius=2.0):
        ubp = ctypes.POINTER(ctypes.c_ubyte)
        N = self.N
        state0 = np.ascontiguousarray(state, np.float32).reshape(N, 9)
        dg = np.ascontiguousarray(desired_goal, np.float32).reshape(N, 3)
        pt = np.ascontiguousarray(prev_traj, np.float32).reshape(N, NV)
        out = np.zeros((N, 3), np.float32)
        for qi in range(N):
            state = _own_view(self.prm, state0, qi)
            lib().orc_goal_prior_based_ex(N, qi, _f(state), _f(dg), _f(pt), planner_seq, dt, goal_threshold, priority_dist_threshold,
                                          goal_radius, self.slack_set[qi].ctypes.data_as(ubp),
                                          int(own_reset[qi]) if own_reset is not None else 0, _f(out[qi]))
        return out

    def tick(self, state, goal, prev_traj, planner_seq, want_lsc=False, nthreads=1):
        ubp = ctypes.POINTER(ctypes.c_ubyte)
        N = self.N
        state = np.ascontiguousarray(state, np.float32).reshape(N, 9)
        goal = np.ascontiguousarray(goal, np.float32).reshape(N, 3)
        prev = np.ascontiguousarray(prev_traj, np.float32).reshape(N, 3, SEGV)
        out = np.zeros((N, 3, SEGV), np.float32)
        status = np.zeros(N, np.int32)
        iters = np.zeros(N, np.int32)
        nrm = np.zeros((N, max(N - 1, 1), M, 3), np.float32) if want_lsc else None
        dd = np.zeros((N, max(N - 1, 1), M, NC)) if want_lsc else None
        use_map = bool(self.prm.use_sfc)
        lib().orc_tick_ex(ctypes.byref(self.prm), ctypes.byref(self.modes), N, _f(state), _f(goal), _f(prev), planner_seq,
                          _d(self.radius), _d(self.downwash), _d(self.vmax), _d(self.amax), _d(self.vnom), _f(self.stale),
                          self.slack_set.ctypes.data_as(ubp), ctypes.byref(self.distmap.edt) if use_map else None, self.world_res,
                          _f(self.sfc) if use_map else None, _i(self.sfc_init) if use_map else None, _f(out), _d(self.cost),
                          _i(status), _i(iters), _f(nrm) if want_lsc else None, _d(dd) if want_lsc else None, nthreads)
        res = {"traj": out, "cost": self.cost.copy(), "status": status, "iters": iters, "sfc": self.sfc.copy()}
        if want_lsc:
            res["normal"], res["d"] = nrm, dd
        return res


def _own_view(prm, state, qi):
    """The states as agent qi sees them: in a planar world its OWN position is read at z = world/z_2d
    (TrajPlanner::currentStateCallback, src/traj_planner.cpp:304-314); everybody else's message is used as it comes."""
    if prm is None or prm.world_dimension != 2:
        return state
    st = state.copy()
    st[qi, 2] = np.float32(prm.world_z_2d)
    return st


def goal_prior_based(state, desired_goal, prev_traj, planner_seq, dt=0.2, goal_threshold=0.1, priority_dist_threshold=0.4,
                     goal_radius=2.0, prm=None):
    """current_goal_position of every agent, mode/goal = prior_based on an empty map (prm: only read for world/dimension)."""
    state0 = np.ascontiguousarray(state, np.float32)
    N = len(state0)
    dg = np.ascontiguousarray(desired_goal, np.float32).reshape(N, 3)
    pt = np.ascontiguousarray(prev_traj, np.float32).reshape(N, NV)
    out = np.zeros((N, 3), np.float32)
    for qi in range(N):
        state = _own_view(prm, state0, qi)
        lib().orc_goal_prior_based(N, qi, _f(state), _f(dg), _f(pt), planner_seq, dt, goal_threshold, priority_dist_threshold,
                                   goal_radius, _f(out[qi]))
    return out


def grid_dims(prm, grid_res=0.3):
    """GridBasedPlanner::updateGridInfo: (dims int[3], grid_min double[3])."""
    dims = np.zeros(3, np.int32)
    gmin = np.zeros(3)
    lib().orc_grid_dims(ctypes.byref(prm), grid_res, _i(dims), _d(gmin))
    return dims, gmin


def astar(occ, start, goal, max_len=100000):
    """Astar-3D on an occupancy grid [ni][nj][nk] (0 free): int[n][3] path (empty when unreachable)."""
    occ = np.ascontiguousarray(occ, np.uint8)
    dims = np.asarray(occ.shape, np.int32)
    s = np.ascontiguousarray(start, np.int32)
    g = np.ascontiguousarray(goal, np.int32)
    out = np.zeros((max_len, 3), np.int32)
    n = lib().orc_astar(occ.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte)), _i(dims), _i(s), _i(g), _i(out), max_len)
    return out[:n].copy()


def goal_prior_based_map(prm, dm, state, desired_goal, prev_traj, planner_seq, radius, downwash, world_res=0.1, grid_res=0.3,
                         grid_margin=0.2, goal_threshold=0.1, priority_dist_threshold=0.4, goal_radius=2.0, want_paths=False,
                         slack_set=None, own_reset=None):
    """current_goal_position of every agent, mode/goal = prior_based WITH a distance field (grid A* + LOS goal).
    Returns goals [N][3] float32 (and, with want_paths, the list of grid paths and the flag words)."""
    state0 = np.ascontiguousarray(state, np.float32)
    N = len(state0)
    dg = np.ascontiguousarray(desired_goal, np.float32).reshape(N, 3)
    pt = np.ascontiguousarray(prev_traj, np.float32).reshape(N, NV)
    r = np.ascontiguousarray(radius, np.float64)
    dw = np.ascontiguousarray(downwash, np.float64)
    out = np.zeros((N, 3), np.float32)
    paths, flags = [], np.zeros(N, np.int32)
    buf = np.zeros((8192, 3), np.int32)
    ubp = ctypes.POINTER(ctypes.c_ubyte)
    for qi in range(N):
        n, fl = ctypes.c_int(), ctypes.c_int()
        row = np.ascontiguousarray(slack_set[qi], np.uint8) if slack_set is not None else None
        state = _own_view(prm, state0, qi)
        lib().orc_goal_prior_based_map(ctypes.byref(prm), ctypes.byref(dm.edt), world_res, grid_res, grid_margin, N, qi, _f(state),
                                       _f(dg), _f(pt), planner_seq, goal_threshold, priority_dist_threshold, goal_radius, _d(r),
                                       _d(dw), row.ctypes.data_as(ubp) if row is not None else None,
                                       int(own_reset[qi]) if (own_reset is not None and row is not None) else 0,
                                       _f(out[qi]), _i(buf), len(buf), ctypes.byref(n), ctypes.byref(fl))
        flags[qi] = fl.value
        if want_paths:
            paths.append(buf[:n.value].copy())
    return (out, paths, flags) if want_paths else out


def bt_read(path):
    """Occupied leaves of an octomap .bt file: (res, int[n][4] = min key x,y,z + cube edge in cells)."""
    res = ctypes.c_double()
    keys = _ip()
    n = ctypes.c_int()
    rc = lib().orc_bt_read(path.encode(), ctypes.byref(res), ctypes.byref(keys), ctypes.byref(n))
    if rc:
        raise IOError(f"orc_bt_read({path}) -> {rc}")
    arr = np.ctypeslib.as_array(keys, shape=(n.value, 4)).copy()
    ctypes.CDLL(None).free(keys)
    return res.value, arr


class DistMap:
    """Dense distance field as DynamicEDTOctomap holds it (restated): dist [nx][ny][nz] float32 metres."""

    def __init__(self, leaves, res, world_min, world_max, maxdist=1.0):
        self.edt = OrcEdt()
        wmin = np.ascontiguousarray(world_min, np.float32)
        wmax = np.ascontiguousarray(world_max, np.float32)
        leaves = np.ascontiguousarray(leaves, np.int32)
        rc = lib().orc_edt_build(_i(leaves), len(leaves), res, _f(wmin), _f(wmax), maxdist, ctypes.byref(self.edt))
        if rc:
            raise ValueError("orc_edt_build failed")
        e = self.edt
        self.dist = np.ctypeslib.as_array(e.dist, shape=(e.nx, e.ny, e.nz)).copy()
        ctypes.CDLL(None).free(e.dist)
        e.dist = _f(self.dist)
        self.key_min = np.array([e.key_min[0], e.key_min[1], e.key_min[2]], np.int32)
        self.res = res

    @classmethod
    def brushfire(cls, leaves, res, world_min, world_max, maxdist=1.0):
        """The field by dynamicEDT3D's published 26-neighbour propagation instead of the exact transform (map assumption tests)."""
        self = cls.__new__(cls)
        self.edt = OrcEdt()
        wmin = np.ascontiguousarray(world_min, np.float32)
        wmax = np.ascontiguousarray(world_max, np.float32)
        leaves = np.ascontiguousarray(leaves, np.int32)
        if lib().orc_edt_brushfire(_i(leaves), len(leaves), res, _f(wmin), _f(wmax), maxdist, ctypes.byref(self.edt), None):
            raise ValueError("orc_edt_brushfire failed")
        e = self.edt
        self.dist = np.ctypeslib.as_array(e.dist, shape=(e.nx, e.ny, e.nz)).copy()
        ctypes.CDLL(None).free(e.dist)
        e.dist = _f(self.dist)
        self.key_min = np.array([e.key_min[0], e.key_min[1], e.key_min[2]], np.int32)
        self.res = res
        return self

    @classmethod
    def from_array(cls, dist, key_min, res):
        """Wraps an existing dense field (tests with synthetic maps)."""
        self = cls.__new__(cls)
        self.edt = OrcEdt()
        self.dist = np.ascontiguousarray(dist, np.float32)
        self.edt.dist = _f(self.dist)
        self.edt.nx, self.edt.ny, self.edt.nz = self.dist.shape
        for k in range(3):
            self.edt.key_min[k] = int(key_min[k])
        self.edt.res = res
        self.key_min = np.asarray(key_min, np.int32)
        self.res = res
        return self

    def expand_box(self, prm, point, goal, radius, world_res=0.1):
        box = np.zeros(6)
        rc = lib().orc_expand_box(ctypes.byref(prm), ctypes.byref(self.edt), world_res, _f(np.ascontiguousarray(point, np.float32)),
                                  _f(np.ascontiguousarray(goal, np.float32)), radius, _d(box))
        return rc, box


def ref_gjk_lib():
    """The REFERENCE's openGJK (oracle/_ref), or None when it has not been built / shipped."""
    p = os.path.join(_HERE, "_ref", "libref_opengjk.so")
    if not os.path.exists(p):
        return None
    L = ctypes.CDLL(p)
    L.ref_gjk.restype = ctypes.c_double
    L.ref_gjk.argtypes = [_dp, ctypes.c_int, _dp, ctypes.c_int, _dp, _ip]
    return L
