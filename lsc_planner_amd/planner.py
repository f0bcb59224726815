"""Host-side mirror of the reference's per-agent planner surface, batched over the swarm.

The reference drives N TrajPlanner objects sequentially (src/multi_sync_simulator.cpp:320-328); each owns a
TrajOptimizer (include/traj_optimizer.hpp:20-28).  SwarmPlanner keeps the same vocabulary --
set_current_state / set_obs_prev_trajs / plan / get_traj / get_qp_cost / get_planning_report -- but one
`plan()` is one C-ABI call for all agents (include/lsc_planner_amd.h).
"""
import ctypes
import os
from dataclasses import dataclass, field

import numpy as np

from . import _lib
from ._lib import LscConfig, LscError, M, NC, NV, SEGV


@dataclass
class PlannerConfig:
    dt: float = 0.2                  # traj/dt
    control_input_weight: float = 0.01
    terminal_weight: float = 1.0
    use_octomap: bool = False
    world_resolution: float = 0.1
    device: int = 0
    max_rows_per_cp: int = 0
    max_iters: int = 50
    prune: bool = True
    warm_start_mu: float = 0.03    # 0 = cold (Mehrotra) start only
    goal_mode: str = "static"       # mode/goal: "static" or "prior_based" (the latter: goal input = desired goal)
    goal_threshold: float = 0.1
    priority_dist_threshold: float = 0.4
    goal_radius: float = 2.0
    grid_resolution: float = 0.3    # grid/resolution (goal planner's search grid; goal_mode prior_based + use_octomap)
    grid_margin: float = 0.2        # grid/margin
    horizon: float = 1.0            # traj/horizon; M = horizon / dt: 5 (liblsc_hip.so) or 4 (liblsc_hip_m4.so)
    goal_row_cap: int = 0           # > 0: smaller OPEN-row capacity of the goal search (tests of the overflow path)
    planner_mode: str = "lsc"       # mode/planner: "lsc" or "bvc"
    slack_mode: str = "none"        # SlackMode: "none", "dynamical_limit", "collision_constraint"
    slack_collision_weight: float = 100000.0
    n_constraint_segments: int = -1
    reset_threshold: float = 0.0    # multisim/reset_threshold; > 0 switches the disturbance checks on (launch files: 0.15)
    gap_tolerance: float = 1e-9     # interior point: relative duality gap at the optimum
    world_dimension: int = 3        # world/dimension: 2 = planar goal grid at z = world_z_2d
    world_z_2d: float = 1.0         # world/z_2d
    goal_search: str = "auto"      # goal planner's grid search: "auto" (register-resident, 32-bit keys when their table fits), "general", "key64" (register-resident, the double as key)
    # QP solver of the LSC fast path: "active_set" (dual active set first, interior point as fallback: the library's default) or
    # "interior_point" (the interior point alone, rounds 1-4).  LSC_SOLVER in the environment changes the default of this harness, so that
    # whole test files can be run through the other solver (tests/test_gpu_round5.py).
    solver: str = field(default_factory=lambda: os.environ.get("LSC_SOLVER", "active_set"))
    comm: tuple = None              # (world_size, rank, id bytes from comm_unique_id()): agent-sharded multi-GPU over RCCL


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _ip(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))


class SwarmPlanner:
    """All agents' TrajPlanner + TrajOptimizer state behind one context of liblsc_hip.so."""

    def __init__(self, mission, config=None):
        self.cfg = config or PlannerConfig()
        # M = horizon / dt picks the library (the kernels are unrolled for their segment count): 5 -> liblsc_hip.so, 4 -> liblsc_hip_m4.so
        self.M = _lib.segments_of(self.cfg.horizon, self.cfg.dt)
        self.SEGV, self.NV = NC * self.M, 3 * NC * self.M
        if self.M not in _lib.BUILT_SEGMENTS:
            raise LscError(f"horizon / dt = {self.cfg.horizon} / {self.cfg.dt} gives M = {self.M} segments; libraries are built for M in "
                           f"{_lib.BUILT_SEGMENTS} (the segment count is a build parameter: make LSC_SEGMENTS=k, src/traj_optimizer.cpp:9 "
                           "computes it at run time)")
        self.L = _lib.load_library(self.M)
        self.mission = mission
        c = LscConfig()
        self.L.lsc_default_config(ctypes.byref(c))
        c.dt, c.control_weight, c.terminal_weight = self.cfg.dt, self.cfg.control_input_weight, self.cfg.terminal_weight
        for k in range(3):
            c.world_min[k] = float(mission.world_min[k])
            c.world_max[k] = float(mission.world_max[k])
        c.use_octomap = int(self.cfg.use_octomap)
        c.world_resolution = self.cfg.world_resolution
        c.device = self.cfg.device
        c.max_rows_per_cp = self.cfg.max_rows_per_cp
        c.max_iters = self.cfg.max_iters
        c.prune = int(self.cfg.prune)           # bool or the ABI's 0..3
        c.warm_start_mu = float(self.cfg.warm_start_mu)
        c.goal_mode = {"static": 0, "prior_based": 1}[self.cfg.goal_mode]
        c.goal_threshold, c.priority_dist_threshold, c.goal_radius = (self.cfg.goal_threshold, self.cfg.priority_dist_threshold,
                                                                       self.cfg.goal_radius)
        c.grid_resolution, c.grid_margin = self.cfg.grid_resolution, self.cfg.grid_margin
        c.horizon, c.goal_row_cap = self.cfg.horizon, self.cfg.goal_row_cap
        c.planner_mode = {"lsc": 0, "bvc": 1}[self.cfg.planner_mode]
        c.slack_mode = {"none": 0, "dynamical_limit": 1, "collision_constraint": 2}[self.cfg.slack_mode]
        c.slack_collision_weight, c.n_constraint_segments = self.cfg.slack_collision_weight, self.cfg.n_constraint_segments
        c.reset_threshold = self.cfg.reset_threshold
        c.gap_tolerance = self.cfg.gap_tolerance
        c.world_dimension, c.world_z_2d = int(self.cfg.world_dimension), float(self.cfg.world_z_2d)
        c.goal_search = {"auto": 0, "general": 1, "key64": 2}[self.cfg.goal_search]
        c.solver = {"active_set": 1, "interior_point": 0, "hand_over": 2}[self.cfg.solver]      # (hand_over: test mode, see lsc_config.solver)
        self._c = c
        self.ctx = self.L.lsc_create(ctypes.byref(c))
        if not self.ctx:
            raise LscError("lsc_create failed: no usable gfx950 device (there is no CPU fallback)")
        self.N = mission.qn
        if self.cfg.comm is not None:
            world, rank, token = self.cfg.comm
            buf = (ctypes.c_ubyte * _lib.COMM_ID_BYTES).from_buffer_copy(bytes(token))
            self._check(self.L.lsc_comm_init(self.ctx, int(world), int(rank), buf))
        self._check(self.L.lsc_set_agents(self.ctx, self.N, _dp(np.ascontiguousarray(mission.radius, np.float64)),
                                          _dp(np.ascontiguousarray(mission.downwash, np.float64)),
                                          _dp(np.ascontiguousarray(mission.max_vel, np.float64)),
                                          _dp(np.ascontiguousarray(mission.max_acc, np.float64)),
                                          _dp(np.ascontiguousarray(mission.nominal_velocity, np.float64))))
        w, r, sr, tr = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        self._check(self.L.lsc_comm_info(self.ctx, ctypes.byref(w), ctypes.byref(r), ctypes.byref(sr), ctypes.byref(tr)))
        self.world, self.rank, self.shard_rows, self.table_rows = w.value, r.value, sr.value, tr.value
        self.first = min(self.rank * self.shard_rows, self.N)
        self.count = min(self.shard_rows, self.N - self.first)
        # TrajPlanner state (src/traj_planner.cpp:41-48)
        self.planner_seq = 0
        self.traj_curr = np.zeros((self.N, 3, self.SEGV), np.float32)
        self.qp_cost = np.zeros(self.N)
        self.planning_report = np.zeros(self.N, np.int32)
        self.iters = np.zeros(self.N, np.int32)

    def _check(self, rc):
        if rc != 0:
            raise LscError(f"lsc error {rc}: {self.L.lsc_last_error(self.ctx).decode()}")

    def close(self):
        if getattr(self, "ctx", None):
            self.L.lsc_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- TrajPlanner::setDistMap --------------------------------------------------------------------------
    def set_distmap(self, dist, key_min, res):
        """dist: float32 [nx][ny][nz] metres (DynamicEDTOctomap semantics), key_min: octomap key of cell (0,0,0)."""
        dist = np.ascontiguousarray(dist, np.float32)
        km = np.ascontiguousarray(key_min, np.int32)
        self._check(self.L.lsc_set_distmap(self.ctx, _fp(dist), dist.shape[0], dist.shape[1], dist.shape[2], _ip(km), float(res)))

    def load_octomap(self, bt_path, maxdist=1.0):
        """MultiSyncSimulator::setOctomap (src/multi_sync_simulator.cpp:153-167): .bt -> distance field -> every agent."""
        dist, key_min, res = edt_from_bt(bt_path, self.mission.world_min, self.mission.world_max, maxdist)
        self.set_distmap(dist, key_min, res)
        return dist, key_min, res

    def set_shard(self, first, count):
        self._check(self.L.lsc_set_shard(self.ctx, first, count))
        self.first, self.count = first, count

    # ---- host-buffer tick: TrajPlanner::plan for every agent of the shard -----------------------------
    def plan(self, state, current_goal, obs_prev_trajs, want_constraints=False):
        """state [N][9], current_goal [N][3], obs_prev_trajs [N][3][30] (every agent's previous getTraj()).
        Returns dict(traj [count][3][30], cost, status, iters[, normal, d])."""
        N, cnt = self.N, self.count
        state = np.ascontiguousarray(state, np.float32).reshape(N, 9)
        goal = np.ascontiguousarray(current_goal, np.float32).reshape(N, 3)
        prev = np.ascontiguousarray(obs_prev_trajs, np.float32).reshape(N, 3, self.SEGV)
        self.planner_seq += 1                                   # src/traj_planner.cpp:127
        out = np.zeros((cnt, 3, self.SEGV), np.float32)
        cost = self.qp_cost[self.first:self.first + cnt].copy()
        status = np.zeros(cnt, np.int32)
        iters = np.zeros(cnt, np.int32)
        nrm = np.zeros((cnt, max(N - 1, 1), self.M, 3), np.float32) if want_constraints else None
        dd = np.zeros((cnt, max(N - 1, 1), self.M, NC), np.float64) if want_constraints else None
        sfc = np.zeros((cnt, self.M, 6), np.float32) if self.cfg.use_octomap else None
        self._check(self.L.lsc_replan_tick(self.ctx, _fp(state), _fp(goal), _fp(prev), self.planner_seq, _fp(out), _dp(cost),
                                           _ip(status), _ip(iters), _fp(nrm) if want_constraints else None,
                                           _dp(dd) if want_constraints else None, _fp(sfc) if sfc is not None else None))
        sl = slice(self.first, self.first + cnt)
        self.traj_curr[sl] = out
        self.qp_cost[sl] = cost
        self.planning_report[sl] = status
        self.iters[sl] = iters
        res = {"traj": out, "cost": cost, "status": status, "iters": iters}
        if sfc is not None:
            res["sfc"] = sfc
        if want_constraints:
            res["normal"], res["d"] = nrm, dd
        return res

    def plan_all(self, state, current_goal, obs_prev_trajs):
        """Multi-GPU form of plan(): every rank passes all N agents' inputs, plans its shard and receives ALL N outputs
        (one RCCL all-gather group inside lsc_replan_tick_all)."""
        N = self.N
        state = np.ascontiguousarray(state, np.float32).reshape(N, 9)
        goal = np.ascontiguousarray(current_goal, np.float32).reshape(N, 3)
        prev = np.ascontiguousarray(obs_prev_trajs, np.float32).reshape(N, 3, self.SEGV)
        self.planner_seq += 1
        out = np.zeros((N, 3, self.SEGV), np.float32)
        cost = self.qp_cost.copy()
        status = np.zeros(N, np.int32)
        iters = np.zeros(N, np.int32)
        goals = np.zeros((N, 3), np.float32)
        self._check(self.L.lsc_replan_tick_all(self.ctx, _fp(state), _fp(goal), _fp(prev), self.planner_seq, _fp(out), _dp(cost),
                                               _ip(status), _ip(iters), _fp(goals)))
        self.traj_curr[:] = out
        self.qp_cost[:] = cost
        self.planning_report[:] = status
        self.iters[:] = iters
        return {"traj": out, "cost": cost, "status": status, "iters": iters, "goal": goals}

    # ---- getters with the reference's names ------------------------------------------------------------
    def get_traj(self):
        return self.traj_curr

    def get_qp_cost(self):
        return self.qp_cost

    def get_planning_report(self):
        return self.planning_report

    def get_planner_seq(self):
        return self.planner_seq

    def last_goals(self):
        """current_goal_position used by the last tick (getCurrentGoalPosition of every agent)."""
        g = np.zeros((self.N, 3), np.float32)
        self._check(self.L.lsc_last_goals(self.ctx, _fp(g)))
        return g

    def set_goal_trace(self, path_cap=512):
        """Keep every agent's grid path of the following ticks (goal_mode prior_based + use_octomap)."""
        self._goal_path_cap = path_cap
        self._check(self.L.lsc_set_goal_trace(self.ctx, path_cap))

    def goal_trace(self):
        """dict(paths = list of int[n][3] grid cells, flags, expansions, grid_dims, grid_min) of the last tick."""
        cap = getattr(self, "_goal_path_cap", 0)
        cnt = self.count
        cells = np.zeros((cnt, max(cap, 1), 3), np.int32)
        plen = np.zeros(cnt, np.int32)
        flags = np.zeros(cnt, np.int32)
        exp = np.zeros(cnt, np.int32)
        dims = np.zeros(3, np.int32)
        gmin = np.zeros(3)
        self._check(self.L.lsc_get_goal_trace(self.ctx, _ip(cells) if cap else None, _ip(plen) if cap else None, _ip(flags), _ip(exp),
                                              _ip(dims), _dp(gmin)))
        return {"paths": [cells[q, :min(plen[q], cap)].copy() for q in range(cnt)], "path_len": plen, "flags": flags,
                "expansions": exp, "grid_dims": dims, "grid_min": gmin}

    def row_counts(self):
        rows = np.zeros(self.N, np.int32)
        self._check(self.L.lsc_last_row_counts(self.ctx, _ip(rows)))
        return rows

    def neighbour_counts(self, priority=False):
        """Units (obstacle, segment) the LSC build of each agent was handed in the last tick (-1: no list) -- with priority=True also the
        number of candidates of the priority rule -- or None when the context builds no neighbour lists (swarms below 512 agents)."""
        units, prio = np.zeros(self.N, np.int32), np.zeros(self.N, np.int32)
        if self.L.lsc_neighbour_counts(self.ctx, _ip(units), _ip(prio) if priority else None) != 0:
            return None
        return (units, prio) if priority else units

    def row_capacity(self):
        """(rows of the LDS pass in the latency build, rows in the throughput build or 0)."""
        a, b = ctypes.c_int(), ctypes.c_int()
        self._check(self.L.lsc_row_capacity(self.ctx, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def bucket_max(self):
        """Rows of every agent's fullest control-point bucket in the last tick (first-pass LDS capacity needed)."""
        rows = np.zeros(self.N, np.int32)
        self._check(self.L.lsc_last_bucket_max(self.ctx, _ip(rows)))
        return rows

    def gjk_batch(self, pts):
        pts = np.ascontiguousarray(pts, np.float64).reshape(-1, 6, 3)
        v = np.zeros((len(pts), 3))
        d = np.zeros(len(pts))
        self._check(self.L.lsc_gjk_batch(self.ctx, _dp(pts), len(pts), _dp(v), _dp(d)))
        return v, d

    # ---- device-resident stepping (torch tensors own the memory) ---------------------------------------
    def tick_device(self, state, goal, traj_prev, traj_next, cost, status, iters, planner_seq, stream=0):
        self._check(self.L.lsc_tick_device(self.ctx, state.data_ptr(), goal.data_ptr(), traj_prev.data_ptr(), planner_seq,
                                           traj_next.data_ptr(), cost.data_ptr(), status.data_ptr(), iters.data_ptr(), stream))

    def tick_device_fused(self, state, goal, traj_prev, traj_next, state_next, cost, status, iters, planner_seq, stream=0):
        """One launch per tick: goal planning + LSC + QP + next ideal state of every agent of the shard."""
        self._check(self.L.lsc_tick_device_fused(self.ctx, state.data_ptr(), goal.data_ptr(), traj_prev.data_ptr(), planner_seq,
                                                 traj_next.data_ptr(), state_next.data_ptr(), cost.data_ptr(), status.data_ptr(),
                                                 iters.data_ptr(), stream))

    def tick_device_sharded(self, state, goal, traj_prev, traj_next, cost, status, iters, planner_seq, stream=0):
        """Sharded tick, everything enqueued on `stream`: plan the rank's agents, in-place RCCL all-gather of traj_next
        ([table_rows][90], padded), next ideal state of all N agents into `state`."""
        self._check(self.L.lsc_tick_device_sharded(self.ctx, state.data_ptr(), goal.data_ptr(), traj_prev.data_ptr(), planner_seq,
                                                   traj_next.data_ptr(), cost.data_ptr(), status.data_ptr(), iters.data_ptr(), stream))

    def propagate_device(self, traj, state, stream=0):
        self._check(self.L.lsc_propagate_device(self.ctx, traj.data_ptr(), state.data_ptr(), stream))

    def sweep_device(self, state, traj_prev, planner_seq, normal, d, stream=0):
        """Dense LSC sweep; d float64 (what the QP reads) or float32 (the compact dump: 180 B per ordered pair)."""
        import torch
        fn = self.L.lsc_sweep_device_f32 if d.dtype == torch.float32 else self.L.lsc_sweep_device
        self._check(fn(self.ctx, state.data_ptr(), traj_prev.data_ptr(), planner_seq, normal.data_ptr(), d.data_ptr(), stream))

    PHASES = ("setup", "lsc_build", "ip_init", "residual_pass", "row_reduce", "assemble", "cholesky", "tri_solves",
              "affine_pass", "corrector_pass", "step_update", "output", "(row_reduce: buckets, wave 0)", "(row_reduce: axis gather, last lane)",
              "(tri_solves: the substitutions of wave 0 alone)", "(spare)")

    def phase_profile(self, enable=-1):
        out = np.zeros((self.N, 16), np.int64)
        self._check(self.L.lsc_phase_profile(self.ctx, enable, out.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong))))
        return out

    GOAL_SECTIONS = ("prologue", "grid_setup", "search", "path_los", "find_min", "delete_min", "screening", "insertions",
                     "general_pop_cycles", "general_pops", "general_insert_cycles", "general_inserts")

    def goal_profile(self, enable=-1):
        out = np.zeros((self.N, 16), np.int64)
        self._check(self.L.lsc_goal_profile(self.ctx, enable, out.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong))))
        return out

    def safety_ratio(self, times):
        """savePlanningResult's agent-agent accounting for the plans of the last plan() call: (ratio [T][count], partner [T][count],
        minimum over everything -- over all ranks with a communicator)."""
        t = np.ascontiguousarray(times, np.float64)
        ratio = np.zeros((len(t), self.count), np.float64)
        partner = np.zeros((len(t), self.count), np.int32)
        mn = ctypes.c_double()
        self._check(self.L.lsc_safety_ratio(self.ctx, t.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), len(t),
                                            ratio.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                            partner.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), ctypes.byref(mn)))
        return ratio, partner, mn.value

    GENERAL_SECTIONS = ("setup", "start", "residual_pass", "row_reduce", "assemble", "factor", "solves", "affine_pass",
                        "corrector_rhs", "row_reduce_2", "assemble_2", "step", "iterations", "solves_of_agent")

    def general_profile(self):
        """Sections of lsc_general_kernel, collected while phase_profile is enabled: [N][16] shader cycles / counts."""
        out = np.zeros((self.N, 16), np.int64)
        self._check(self.L.lsc_general_profile(self.ctx, out.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong))))
        return out

    def dump_qp(self, agent, path):
        """TrajOptimizer::solve's failure export (log/QPmodel.lp): the agent's QP of the last plan() call as a CPLEX LP file."""
        self._check(self.L.lsc_dump_qp(self.ctx, int(agent), str(path).encode()))

    def solver_residuals(self):
        out = np.zeros((self.N, 4))
        self._check(self.L.lsc_solver_residuals(self.ctx, _dp(out)))
        return out

    def solver_trace(self, agent, read=False):
        ny = 3 * (3 * (self.M - 1) + 1)
        kld = ny + 2 - (ny % 2 == 0)
        wsz = 3 * self.NV + 6 * self.SEGV
        out = np.zeros(64 * 8 + ny * kld + wsz + 512)
        self._check(self.L.lsc_solver_trace(self.ctx, agent, _dp(out) if read else None))
        self.trace_K = out[512:512 + ny * kld].reshape(ny, kld)[:, :ny]
        self.trace_W = out[512 + ny * kld:512 + ny * kld + wsz]
        self.trace_kconst = out[512 + ny * kld + wsz:]
        return out[:512].reshape(64, 8)

    def iterations_total(self, reset=False):
        t = ctypes.c_longlong()
        self._check(self.L.lsc_iterations_total(self.ctx, ctypes.byref(t), int(reset)))
        return t.value

    def row_iterations_total(self):
        """Sum over agents of iterations x LSC rows carried since the last iterations_total(reset=True)."""
        t = ctypes.c_longlong()
        self._check(self.L.lsc_row_iterations_total(self.ctx, ctypes.byref(t)))
        return t.value

    def solver_stats(self):
        """Counters of the active-set solve since the last iterations_total(reset=True): dict(solved, handed_over, changes, ip_iterations)."""
        out = (ctypes.c_longlong * 4)()
        self._check(self.L.lsc_solver_stats(self.ctx, out))
        return dict(solved=out[0], handed_over=out[1], changes=out[2], ip_iterations=out[3])

    def set_timing(self, on):
        self._check(self.L.lsc_set_timing(self.ctx, int(on)))

    def kernel_time_ms(self, which=0):
        ms = ctypes.c_double()
        n = ctypes.c_long()
        self._check(self.L.lsc_kernel_time_ms(self.ctx, which, ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    def kernel_times_ms(self, which=0):
        """Per-launch device times (ms, HIP events on the launch stream) since set_timing(True)."""
        n = ctypes.c_long()
        self._check(self.L.lsc_kernel_times_ms(self.ctx, which, None, 0, ctypes.byref(n)))
        out = np.zeros(max(n.value, 1), np.float64)
        self._check(self.L.lsc_kernel_times_ms(self.ctx, which, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), n.value,
                                               ctypes.byref(n)))
        return out[:n.value]


def tick_device_fused_batch(planners, states, goals, trajs_prev, trajs_next, states_next, costs, statuses, iters, planner_seqs, stream=0):
    """One tick of several independent swarms -- one SwarmPlanner (context) each, all on one device -- in ONE launch
    (lsc_tick_device_fused_batch): every argument is a list with one entry per swarm; results are those of tick_device_fused."""
    n = len(planners)
    L = planners[0].L
    # one library (liblsc_hip.so and liblsc_hip_m4.so lay the same lsc_ctx symbol out with different array sizes: a context of the other
    # build would be read with the wrong sizes) and every context once (its stale-plan / hand-over buffers belong to ONE swarm of the launch)
    if not all(p.L is L for p in planners):
        raise LscError("tick_device_fused_batch: planners of different libraries (segment counts) in one batch")
    if len({int(getattr(p.ctx, "value", p.ctx) or 0) for p in planners}) != n:
        raise LscError("tick_device_fused_batch: the same planner (context) more than once in one batch")
    vp = ctypes.c_void_p

    def ptrs(ts):
        return (vp * n)(*[t.data_ptr() for t in ts])
    ctxs = (vp * n)(*[p.ctx for p in planners])
    seqs = (ctypes.c_int * n)(*[int(q) for q in planner_seqs])
    rc = L.lsc_tick_device_fused_batch(ctxs, n, ptrs(states), ptrs(goals), ptrs(trajs_prev), seqs, ptrs(trajs_next), ptrs(states_next),
                                       ptrs(costs), ptrs(statuses), ptrs(iters), stream)
    if rc != 0:
        raise LscError(f"lsc error {rc}: {L.lsc_last_error(planners[0].ctx).decode()}")


def comm_unique_id():
    """Rendezvous token of the RCCL communicator (ncclGetUniqueId): rank 0 makes it, every rank passes it to PlannerConfig.comm."""
    L = _lib.load_library()
    buf = (ctypes.c_ubyte * _lib.COMM_ID_BYTES)()
    rc = L.lsc_comm_unique_id(buf)
    if rc != 0:
        raise LscError(f"lsc_comm_unique_id failed: {rc} (librccl.so not loadable?)")
    return bytes(buf)


def edt_from_bt(bt_path, world_min, world_max, maxdist=1.0):
    """Host-only: octomap .bt -> (dist float32 [nx][ny][nz], key_min int32[3], res).  No GPU needed."""
    L = _lib.load_library()
    fpp = ctypes.POINTER(ctypes.c_float)
    wm = np.ascontiguousarray(world_min, np.float32)
    wM = np.ascontiguousarray(world_max, np.float32)
    edt = fpp()
    dims = (ctypes.c_int * 3)()
    kmin = (ctypes.c_int * 3)()
    res = ctypes.c_double()
    rc = L.lsc_edt_from_bt(str(bt_path).encode(), _fp(wm), _fp(wM), float(maxdist), ctypes.byref(edt), dims, kmin, ctypes.byref(res))
    if rc != 0:
        raise LscError(f"lsc_edt_from_bt({bt_path}) failed: {rc}")
    arr = np.ctypeslib.as_array(edt, shape=tuple(dims)).copy()
    L.lsc_free_host(edt)
    return arr, np.array(list(kmin), np.int32), res.value


def next_state_host(traj, dt=0.2):
    """getStateFromControlPoints at t = dt in float32 (include/polynomial.hpp:63-97), numpy, all agents."""
    t = np.asarray(traj, np.float32)
    t = t if t.ndim == 3 else t.reshape(len(t), 3, -1)          # [N][3][M (n + 1)] for any M (or the same rows flattened)
    c1 = t[:, :, NC:NC + 3]
    fn, fn1, finv = np.float32(5), np.float32(4), np.float32(dt ** -1)
    v0 = ((c1[:, :, 1] - c1[:, :, 0]) * fn) * finv
    v1 = ((c1[:, :, 2] - c1[:, :, 1]) * fn) * finv
    a0 = ((v1 - v0) * fn1) * finv
    return np.concatenate([c1[:, :, 0], v0, a0], axis=1).astype(np.float32)
