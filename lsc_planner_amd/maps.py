"""Map-side helpers of the harness: an octomap .bt writer (the counterpart of the library's reader, lsc_edt_from_bt) and the
occupancy of the reference's data file world/simple_forest.bt as a committed leaf list (data/simple_forest_leaves.npz, made by
tests/golden/make_map_golden.py from the reference's map; data, not source), so that bench.py, tools/ and the tests can rebuild
BASELINE configs[3]'s map on a box that has no /root/reference."""
import os

import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def forest_leaves():
    """Occupied leaves int32 [n][4] (min-corner key x, y, z and cube edge in max-depth cells) and the resolution."""
    z = np.load(os.path.join(_DATA, "simple_forest_leaves.npz"))
    return z["leaves"], float(z["res"])


def write_bt(path, leaves, res):
    """Serialises occupied leaves [n][4] (min key x,y,z, edge) as an octomap binary tree: depth-first, two bytes per
    inner node, child bits 00 unknown / 01 free / 10 occupied / 11 inner (LSB first)."""

    def build(mx, my, mz, size, items):
        # items: leaves fully inside this node
        half = size >> 1
        bits = 0
        kids = []
        for ch in range(8):
            cx, cy, cz = mx + (half if ch & 1 else 0), my + (half if ch & 2 else 0), mz + (half if ch & 4 else 0)
            sub = [l for l in items if cx <= l[0] < cx + half and cy <= l[1] < cy + half and cz <= l[2] < cz + half]
            if not sub:
                continue
            if len(sub) == 1 and sub[0][3] == half and (sub[0][0], sub[0][1], sub[0][2]) == (cx, cy, cz):
                bits |= 2 << (2 * ch)
            else:
                bits |= 3 << (2 * ch)
                kids.append((cx, cy, cz, half, sub))
        out = bytes([bits & 0xff, bits >> 8])
        for k in kids:
            out += build(*k)
        return out

    items = [tuple(int(v) for v in l) for l in leaves]
    data = build(0, 0, 0, 65536, items)
    with open(path, "wb") as f:
        f.write(b"# Octomap OcTree binary file\n# written by lsc_planner_amd.maps.write_bt\nid OcTree\n")
        f.write(f"size {len(data) // 2}\nres {res}\ndata\n".encode())
        f.write(data)
