"""Mission inputs of the hot path.

load_mission  : the reference's mission JSON (src/mission.cpp:20-132, missions/readme.txt); values are cast
                to float32 where the reference stores octomap::point3d.
circle_swap   : matlab/mission_generator.m:7-14 / Mission::generateCircleSwap (src/mission.cpp:321-335).
random_swarm  : seeded uniform starts/goals of the 256 / 1024-agent BASELINE configs.
"""
import json
from dataclasses import dataclass, field

import numpy as np


@dataclass
class Mission:
    start: np.ndarray            # [N][3] float32
    goal: np.ndarray             # [N][3] float32 (desired goal)
    world_min: np.ndarray        # [3] float32
    world_max: np.ndarray        # [3] float32
    radius: np.ndarray           # [N] float64
    downwash: np.ndarray         # [N] float64
    max_vel: np.ndarray          # [N][3] float64
    max_acc: np.ndarray          # [N][3] float64
    nominal_velocity: np.ndarray  # [N] float64
    name: str = ""
    world_file: str = ""
    meta: dict = field(default_factory=dict)

    @property
    def qn(self):
        return len(self.start)


def load_mission(path, world_dimension=3, world_z_2d=1.0):
    """Mission::initialize (src/mission.cpp:20-132).  world_dimension == 2 puts every start and goal at z = world_z_2d (:88-112)."""
    doc = json.load(open(path))
    world = doc["world"]
    if len(world) != 1:
        raise ValueError("[Mission] World must have one element")
    dim = world[0]["dimension"]
    quads = doc["quadrotors"]
    agents = doc["agents"]
    n = len(agents)
    start = np.zeros((n, 3), np.float32)
    goal = np.zeros((n, 3), np.float32)
    radius, downwash, vnom = np.zeros(n), np.zeros(n), np.zeros(n)
    vmax, amax = np.zeros((n, 3)), np.zeros((n, 3))
    for i, a in enumerate(agents):
        if "type" not in a or "start" not in a or "goal" not in a:
            raise ValueError("[Mission] Agent must have type, start and goal elements")
        q = quads[a["type"]]
        start[i] = np.asarray(a["start"], np.float64).astype(np.float32)
        goal[i] = np.asarray(a["goal"], np.float64).astype(np.float32)
        if world_dimension == 2:
            start[i, 2] = goal[i, 2] = np.float32(world_z_2d)
        radius[i] = q["radius"]
        downwash[i] = a.get("downwash", q["downwash"])
        vnom[i] = a.get("nominal_velocity", q["nominal_velocity"])
        vmax[i] = q["max_vel"]
        amax[i] = q["max_acc"]
    return Mission(start, goal, np.asarray(dim[:3], np.float32), np.asarray(dim[3:], np.float32), radius, downwash,
                   vmax, amax, vnom, name=str(path))


def _uniform_agents(n, radius=0.15, downwash=2.0, max_vel=(1.0, 1.0, 1.0), max_acc=(2.0, 2.0, 2.0), vnom=1.0):
    return (np.full(n, radius), np.full(n, downwash), np.tile(np.asarray(max_vel, float), (n, 1)),
            np.tile(np.asarray(max_acc, float), (n, 1)), np.full(n, vnom))


def circle_swap(n, circle_radius=8.0, z=1.0, world=(-10, -10, 0, 10, 10, 2.5)):
    th = 2.0 * np.pi * np.arange(n) / n
    start = np.stack([circle_radius * np.cos(th), circle_radius * np.sin(th), np.full(n, z)], 1)
    goal = start.copy()
    goal[:, :2] *= -1.0
    r, dw, vm, am, vn = _uniform_agents(n)
    return Mission(start.astype(np.float32), goal.astype(np.float32), np.asarray(world[:3], np.float32),
                   np.asarray(world[3:], np.float32), r, dw, vm, am, vn, name=f"circle_swap{n}_R{circle_radius}")


def random_swarm(n, world=(-20, -20, 0, 20, 20, 5), seed=20260929, min_sep=0.6, shrink=0.5, downwash=2.0, edt=None,
                 edt_key_min=None, edt_res=0.1, min_clearance=0.45):
    """Uniform starts / goals in the world box shrunk by `shrink`, pairwise (downwash-scaled) distance >= min_sep;
    with a distance field, samples closer than `min_clearance` to an obstacle are rejected."""
    rng = np.random.default_rng(seed)
    lo = np.asarray(world[:3], float) + shrink
    hi = np.asarray(world[3:], float) - shrink

    def sample():
        # (the same draws and the same accept / reject decisions as a scan over all points placed so far, found through a cell hash:
        #  a point closer than min_sep lies in one of the 27 cells around p -- swarms of 8192 agents in seconds instead of minutes)
        pts = []
        cells = {}
        cs = np.array([min_sep, min_sep, min_sep * downwash])
        draws = 0
        while len(pts) < n:
            draws += 1
            if draws > 20000 + 2000 * n:
                raise ValueError(f"random_swarm: cannot place {n} agents in this world (separation {min_sep}, clearance {min_clearance})")
            p = rng.uniform(lo, hi)
            ok = True
            if edt is not None:
                c = np.floor(p / edt_res).astype(int) + 32768 - np.asarray(edt_key_min)
                if (c < 0).any() or (c >= np.asarray(edt.shape)).any() or edt[c[0], c[1], c[2]] < min_clearance:
                    continue
            key = tuple(np.floor(p / cs).astype(int))
            for dx in (-1, 0, 1):
                for dy in (-1, 0, 1):
                    for dz in (-1, 0, 1):
                        for q in cells.get((key[0] + dx, key[1] + dy, key[2] + dz), ()):
                            d = p - q
                            d[2] /= downwash
                            if np.dot(d, d) < min_sep * min_sep:
                                ok = False
                                break
                        if not ok:
                            break
                    if not ok:
                        break
                if not ok:
                    break
            if ok:
                pts.append(p)
                cells.setdefault(key, []).append(p)
        return np.asarray(pts)

    start = sample()
    goal = sample()[rng.permutation(n)]
    r, dw, vm, am, vn = _uniform_agents(n, downwash=downwash)
    return Mission(start.astype(np.float32), goal.astype(np.float32), np.asarray(world[:3], np.float32),
                   np.asarray(world[3:], np.float32), r, dw, vm, am, vn, name=f"random_swarm{n}_seed{seed}")
