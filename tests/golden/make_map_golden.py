"""Occupied leaves of the reference's DATA file world/simple_forest.bt (octomap binary tree) as a small fixture.

Run in the build container:  python tests/golden/make_map_golden.py   (writes lsc_planner_amd/data/simple_forest_leaves.npz)
Stores int32 [n][4] = min-corner key (x, y, z) and cube edge in max-depth cells, plus the resolution.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle import oracle as O  # noqa: E402

res, leaves = O.bt_read("/root/reference/world/simple_forest.bt")
np.savez_compressed(os.path.join(os.path.dirname(__file__), "..", "..", "lsc_planner_amd", "data", "simple_forest_leaves.npz"), leaves=leaves.astype(np.int32), res=res)
print(len(leaves), "leaves,", int((leaves[:, 3] ** 3).sum()), "occupied cells, res", res)
