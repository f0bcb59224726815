"""ctypes binding of the C ABI declared in include/lsc_planner_amd.h.

There is no CPU fallback: if liblsc_hip.so is missing this raises, and every compute entry point fails
when no gfx950 device is usable.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}

# M = 5 (every shipped launch file) is the default build; load_library(4) loads the M = 4 build of the same sources
M, DEG, NC, SEGV, NV = 5, 5, 6, 30, 90


def segments_of(horizon, dt):
    """M = static_cast<int>((horizon + SP_EPSILON) / dt), src/traj_optimizer.cpp:9."""
    return int((horizon + 1e-9) / dt)

STATUS_OK, STATUS_INFEASIBLE, STATUS_SFC_BLOCKED, STATUS_GOAL_CAPACITY = 0, 1, 4, 5
COMM_ID_BYTES = 128


class LscError(RuntimeError):
    pass


class LscConfig(ctypes.Structure):
    _fields_ = [
        ("dt", ctypes.c_double),
        ("control_weight", ctypes.c_double),
        ("terminal_weight", ctypes.c_double),
        ("world_min", ctypes.c_float * 3),
        ("world_max", ctypes.c_float * 3),
        ("use_octomap", ctypes.c_int),
        ("world_resolution", ctypes.c_double),
        ("device", ctypes.c_int),
        ("max_rows_per_cp", ctypes.c_int),
        ("max_iters", ctypes.c_int),
        ("prune", ctypes.c_int),
        ("goal_mode", ctypes.c_int),
        ("goal_threshold", ctypes.c_double),
        ("priority_dist_threshold", ctypes.c_double),
        ("goal_radius", ctypes.c_double),
        ("warm_start_mu", ctypes.c_double),
        ("grid_resolution", ctypes.c_double),
        ("grid_margin", ctypes.c_double),
        ("horizon", ctypes.c_double),
        ("goal_row_cap", ctypes.c_int),
        ("planner_mode", ctypes.c_int),
        ("slack_mode", ctypes.c_int),
        ("slack_collision_weight", ctypes.c_double),
        ("n_constraint_segments", ctypes.c_int),
        ("reset_threshold", ctypes.c_double),
        ("gap_tolerance", ctypes.c_double),
        ("world_dimension", ctypes.c_int),
        ("world_z_2d", ctypes.c_double),
        ("goal_search", ctypes.c_int),
        ("solver", ctypes.c_int),
    ]


# every symbol include/lsc_planner_amd.h declares
EXPORTS = [
    "lsc_default_config", "lsc_create", "lsc_destroy", "lsc_last_error", "lsc_last_note", "lsc_segments", "lsc_set_agents", "lsc_set_shard",
    "lsc_set_distmap", "lsc_replan_tick", "lsc_tick_device", "lsc_tick_device_fused", "lsc_propagate_device", "lsc_safety_ratio", "lsc_sweep_device", "lsc_sweep_device_f32",
    "lsc_gjk_batch", "lsc_kernel_time_ms", "lsc_kernel_times_ms", "lsc_set_timing", "lsc_last_row_counts", "lsc_iterations_total", "lsc_phase_profile", "lsc_goal_profile", "lsc_goal_key_table", "lsc_general_profile", "lsc_dump_qp", "lsc_solver_residuals", "lsc_solver_trace", "lsc_edt_from_bt", "lsc_free_host", "lsc_last_goals", "lsc_set_goal_trace", "lsc_get_goal_trace",
    "lsc_last_bucket_max", "lsc_row_capacity", "lsc_comm_unique_id", "lsc_comm_init", "lsc_comm_info", "lsc_tick_device_sharded", "lsc_replan_tick_all",
    "lsc_row_iterations_total", "lsc_tick_device_fused_batch", "lsc_solver_stats", "lsc_neighbour_counts",
]


BUILT_SEGMENTS = (5, 4)      # csrc/Makefile builds these two; other M = recompile with -DLSC_SEGMENTS=k (3 <= k <= 5)


def lib_path(segments=5):
    """liblsc_hip.so (M = 5) / liblsc_hip_m4.so (M = 4) next to this file.  LSC_HIP_LIB overrides the M = 5 path and
    LSC_HIP_LIB_M4 the M = 4 one (A/B runs of two builds on the same GPU box; the LDS-poison builds of `make poison poison_m4`)."""
    if segments == 5:
        return os.environ.get("LSC_HIP_LIB") or os.path.join(_HERE, "liblsc_hip.so")
    if segments == 4:
        return os.environ.get("LSC_HIP_LIB_M4") or os.path.join(_HERE, "liblsc_hip_m4.so")
    return os.path.join(_HERE, f"liblsc_hip_m{segments}.so")


def load_library(segments=5):
    """Loads the library built for `segments` = horizon / dt (__graft_entry__.build() / csrc/Makefile).  Raises if absent."""
    if segments in _LIBS:
        return _LIBS[segments]
    p = lib_path(segments)
    if not os.path.exists(p):
        raise LscError(f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # PyTorch wheels bundle their own HIP/HSA runtime.  Two runtimes in one process do not coexist (the second one
    # finds "no HIP GPUs"), so when this harness runs next to torch, torch's runtime must be the one that is loaded
    # first; liblsc_hip.so then binds to it.  Stand-alone C++ users (lsc_sim) link /opt/rocm's runtime directly.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = ctypes.CDLL(p)
    vp, ip, dp, fp = ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_float)
    L.lsc_default_config.argtypes = [ctypes.POINTER(LscConfig)]
    L.lsc_default_config.restype = None
    L.lsc_create.argtypes = [ctypes.POINTER(LscConfig)]
    L.lsc_create.restype = vp
    L.lsc_destroy.argtypes = [vp]
    L.lsc_destroy.restype = None
    L.lsc_last_error.argtypes = [vp]
    L.lsc_last_error.restype = ctypes.c_char_p
    L.lsc_last_note.argtypes = [vp]
    L.lsc_last_note.restype = ctypes.c_char_p
    L.lsc_set_agents.argtypes = [vp, ctypes.c_int, dp, dp, dp, dp, dp]
    L.lsc_set_shard.argtypes = [vp, ctypes.c_int, ctypes.c_int]
    L.lsc_set_distmap.argtypes = [vp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ip, ctypes.c_double]
    L.lsc_replan_tick.argtypes = [vp, fp, fp, fp, ctypes.c_int, fp, dp, ip, ip, fp, dp, fp]
    L.lsc_tick_device.argtypes = [vp, vp, vp, vp, ctypes.c_int, vp, vp, vp, vp, vp]
    L.lsc_tick_device_fused.argtypes = [vp, vp, vp, vp, ctypes.c_int, vp, vp, vp, vp, vp, vp]
    vpp = ctypes.POINTER(vp)
    L.lsc_tick_device_fused_batch.argtypes = [vpp, ctypes.c_int, vpp, vpp, vpp, ip, vpp, vpp, vpp, vpp, vpp, vp]
    L.lsc_propagate_device.argtypes = [vp, vp, vp, vp]
    L.lsc_safety_ratio.argtypes = [vp, ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_double)]
    ubp = ctypes.POINTER(ctypes.c_ubyte)
    L.lsc_comm_unique_id.argtypes = [ubp]
    L.lsc_comm_init.argtypes = [vp, ctypes.c_int, ctypes.c_int, ubp]
    L.lsc_comm_info.argtypes = [vp, ip, ip, ip, ip]
    L.lsc_tick_device_sharded.argtypes = [vp, vp, vp, vp, ctypes.c_int, vp, vp, vp, vp, vp]
    L.lsc_replan_tick_all.argtypes = [vp, fp, fp, fp, ctypes.c_int, fp, dp, ip, ip, fp]
    L.lsc_sweep_device.argtypes = [vp, vp, vp, ctypes.c_int, vp, vp, vp]
    L.lsc_sweep_device_f32.argtypes = [vp, vp, vp, ctypes.c_int, vp, vp, vp]
    L.lsc_gjk_batch.argtypes = [vp, dp, ctypes.c_int, dp, dp]
    L.lsc_kernel_time_ms.argtypes = [vp, ctypes.c_int, dp, ctypes.POINTER(ctypes.c_long)]
    L.lsc_kernel_times_ms.argtypes = [vp, ctypes.c_int, dp, ctypes.c_long, ctypes.POINTER(ctypes.c_long)]
    L.lsc_set_timing.argtypes = [vp, ctypes.c_int]
    L.lsc_last_row_counts.argtypes = [vp, ip]
    L.lsc_neighbour_counts.argtypes = [vp, ip, ip]
    L.lsc_last_bucket_max.argtypes = [vp, ip]
    L.lsc_row_capacity.argtypes = [vp, ip, ip]
    L.lsc_phase_profile.argtypes = [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_longlong)]
    L.lsc_goal_profile.argtypes = [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_longlong)]
    L.lsc_general_profile.argtypes = [vp, ctypes.POINTER(ctypes.c_longlong)]
    L.lsc_goal_key_table.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_uint), ctypes.POINTER(ctypes.c_int)]
    L.lsc_dump_qp.argtypes = [vp, ctypes.c_int, ctypes.c_char_p]
    L.lsc_solver_residuals.argtypes = [vp, dp]
    L.lsc_solver_trace.argtypes = [vp, ctypes.c_int, dp]
    L.lsc_edt_from_bt.argtypes = [ctypes.c_char_p, fp, fp, ctypes.c_double, ctypes.POINTER(fp), ip, ip, dp]
    L.lsc_last_goals.argtypes = [vp, fp]
    L.lsc_set_goal_trace.argtypes = [vp, ctypes.c_int]
    L.lsc_get_goal_trace.argtypes = [vp, ip, ip, ip, ip, ip, dp]
    L.lsc_free_host.argtypes = [vp]
    L.lsc_free_host.restype = None
    L.lsc_iterations_total.argtypes = [vp, ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
    L.lsc_row_iterations_total.argtypes = [vp, ctypes.POINTER(ctypes.c_longlong)]
    L.lsc_solver_stats.argtypes = [vp, ctypes.POINTER(ctypes.c_longlong)]
    for name in EXPORTS:
        fn = getattr(L, name)
        if fn.restype is ctypes.c_int or name not in ("lsc_default_config", "lsc_create", "lsc_destroy", "lsc_last_error", "lsc_last_note", "lsc_free_host"):
            fn.restype = ctypes.c_int
    L.lsc_segments.argtypes = []
    if L.lsc_segments() != segments:
        raise LscError(f"{p} plans M = {L.lsc_segments()} segments, not {segments}")
    _LIBS[segments] = L
    return L
