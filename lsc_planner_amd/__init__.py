"""lsc_planner_amd -- MI355X-native per-agent replanning loop for qwerty35/lsc_planner.

Only what the hot path needs lives here:
  csrc/        gfx950 kernels + the C ABI (include/lsc_planner_amd.h)  -> liblsc_hip.so
  _lib.py      ctypes binding of that ABI (fails loudly when the library or a GPU is missing)
  planner.py   host-side mirror of the reference's TrajPlanner / TrajOptimizer surface, batched per tick
  mission.py   mission JSON loader + the circle-swap / random-swarm generators of the BASELINE configs
  maps.py      octomap .bt writer + the committed occupancy of the reference's simple_forest map (data/)
  csrc/host/   C++ host side above the ABI: Mission / Param / TrajPlanner-shaped facade, headless MultiSyncSimulator, result-CSV
               writer and reader -> lsc_sim
  sharded.py   partition arithmetic of the agent-sharded multi-GPU path + the torch.distributed fallback exchange (the native
               exchange is the library's own RCCL all-gather, lsc_comm_init / lsc_tick_device_sharded)
"""
from ._lib import LscError, load_library, lib_path  # noqa: F401
from .mission import Mission, load_mission, circle_swap, random_swarm  # noqa: F401
from .planner import SwarmPlanner, PlannerConfig, edt_from_bt, comm_unique_id, tick_device_fused_batch  # noqa: F401

__all__ = ["LscError", "load_library", "lib_path", "Mission", "load_mission", "circle_swap", "random_swarm",
           "SwarmPlanner", "PlannerConfig", "edt_from_bt", "comm_unique_id", "tick_device_fused_batch"]
