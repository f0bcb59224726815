"""Per-tick golden snapshots (inputs + oracle outputs) for the reference's own missions.

Run in the build container:  python tests/golden/make_tick_golden.py
Missions are read from /root/reference/missions (data files); each snapshot stores the frozen inputs of a tick
(states, goals, previous trajectories, planner_seq) and the oracle's LSC normals / margins, trajectories, costs.
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle import oracle as O  # noqa: E402
from lsc_planner_amd.mission import load_mission  # noqa: E402
from lsc_planner_amd.planner import next_state_host  # noqa: E402

REF = "/root/reference/missions"


def snapshots(path, keep):
    ms = load_mission(path)
    N = ms.qn
    prm = O.make_params(world_min=ms.world_min, world_max=ms.world_max, obs_f32=True)
    sw = O.Swarm(prm, ms.radius, ms.downwash, ms.max_vel, ms.max_acc, ms.nominal_velocity)
    state = np.zeros((N, 9), np.float32); state[:, :3] = ms.start
    traj = np.zeros((N, 3, 30), np.float32)
    out = {}
    for tick in range(1, max(keep) + 1):
        r = sw.tick(state, ms.goal, traj, tick, want_lsc=True, nthreads=8)
        if tick in keep:
            out[tick] = dict(state=state.copy(), prev=traj.copy(), stale=traj.copy(), normal=r["normal"], d=r["d"], traj=r["traj"],
                             cost=r["cost"], status=r["status"])
        traj = r["traj"]
        state = next_state_host(traj)
    return ms, out


def main():
    blob = {}
    for name, keep in (("multi_simple4", (1, 2, 3, 20)), ("multi_circle20", (1, 15))):
        ms, snaps = snapshots(f"{REF}/{name}.json", keep)
        for k in ("start", "goal", "world_min", "world_max", "radius", "downwash", "max_vel", "max_acc", "nominal_velocity"):
            blob[f"{name}/{k}"] = getattr(ms, k)
        for tick, s in snaps.items():
            for k, v in s.items():
                blob[f"{name}/tick{tick}/{k}"] = v
    np.savez_compressed(os.path.join(os.path.dirname(__file__), "ticks.npz"), **blob)
    print("wrote", len(blob), "arrays")


if __name__ == "__main__":
    main()
