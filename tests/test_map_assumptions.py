"""The map-side assumptions as executable statements.  octomap / dynamicEDT3D are external to the reference and absent here, so
what the product and the oracle assume about them (csrc/lsc_octomap.cpp header) cannot be pinned against the
libraries; it can be checked against what octomap PUBLISHES about its own geometry, on the reference's own data files:

  * a binary tree file is a depth-first stream of 2-byte nodes, 2 bits per child (01 free leaf, 10 occupied leaf, 11 inner),
    child index bit0 = x, bit1 = y, bit2 = z; the root cube is centred on the origin with edge 2^16 res; a child's centre is
    its parent's centre -+ a quarter of the parent's edge per axis (OcTreeBaseImpl / readBinaryNode);
  * a coordinate belongs to the max-depth cell  key = (int)floor(coord / res) + 32768  (OcTreeBaseImpl::coordToKey), and a
    cell's centre is  (key - 32768 + 0.5) res  (keyToCoord).

The walker below derives every occupied leaf GEOMETRICALLY (cube centres by halving, in metres, no keys anywhere) and only then
applies the published key formula to the cells' centre coordinates; the product's reader (lsc_edt_from_bt, key arithmetic
throughout) must put its zeros -- distance 0 = occupied cell -- on exactly those cells, for three files of the reference.
Needs /root/reference (the data files do not travel); runs without a GPU.
"""
import os

import numpy as np
import pytest

REF = "/root/reference/world"
FILES = ["simple_forest.bt", "forest/forest1.bt", "office.bt"]

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference's data files (build container only)")


def _occupied_cell_centres(path):
    """Centres (metres) of all max-depth cells inside occupied leaves, by cube halving from the root; plus res."""
    raw = open(path, "rb").read()
    head, data = raw.split(b"data\n", 1)
    res = float([l for l in head.decode("latin-1").splitlines() if l.startswith("res ")][0].split()[1])
    pos = 0
    cubes = []                                           # (centre xyz, edge) of occupied leaves
    stack = [((0.0, 0.0, 0.0), res * 65536.0)]
    while stack:                                         # depth first, children in index order
        centre, edge = stack.pop()
        bits = data[pos] | (data[pos + 1] << 8)
        pos += 2
        kids = []
        for ch in range(8):
            code = (bits >> (2 * ch)) & 3
            if code == 0:
                continue
            c = tuple(centre[a] + (edge / 4.0 if (ch >> a) & 1 else -edge / 4.0) for a in range(3))
            if code == 2:
                cubes.append((c, edge / 2.0))
            elif code == 3:
                kids.append((c, edge / 2.0))
        stack.extend(reversed(kids))
    assert pos == len(data) or pos == 2 * int([l for l in head.decode("latin-1").splitlines() if l.startswith("size ")][0].split()[1])
    pts = []
    for c, edge in cubes:
        n = int(round(edge / res))
        off = (np.arange(n) + 0.5) * res - edge / 2.0
        g = np.stack(np.meshgrid(c[0] + off, c[1] + off, c[2] + off, indexing="ij"), -1).reshape(-1, 3)
        pts.append(g)
    return np.concatenate(pts), res


@pytest.mark.parametrize("name", FILES)
def test_product_reader_puts_the_occupied_cells_where_octomaps_key_formula_puts_them(name):
    import lsc_planner_amd as L
    path = os.path.join(REF, name)
    centres, res = _occupied_cell_centres(path)
    lo, hi = centres.min(0) - 0.35, centres.max(0) + 0.35
    wmin, wmax = np.asarray(lo, np.float32), np.asarray(hi, np.float32)
    dist, kmin, r = L.edt_from_bt(path, wmin, wmax)
    assert r == res
    key = np.floor(centres / res).astype(np.int64) + 32768                       # OcTreeBaseImpl::coordToKey on the cell centres
    assert np.allclose((key - 32768 + 0.5) * res, centres, atol=1e-9)             # keyToCoord gives the centres back
    idx = key - np.asarray(kmin, np.int64)
    assert (idx >= 0).all() and (idx < np.asarray(dist.shape)).all()              # the queried box holds every occupied cell
    occ = np.zeros(dist.shape, bool)
    occ[idx[:, 0], idx[:, 1], idx[:, 2]] = True
    assert occ.sum() == len(np.unique(key, axis=0))
    assert np.array_equal(dist == 0, occ)                                         # zeros of the distance field == occupied cells
    # the field's own grid follows the same formula: cell (0,0,0) is the cell of world_min
    assert np.array_equal(np.asarray(kmin), np.floor(np.asarray(wmin, np.float64) / res).astype(np.int64) + 32768)
    # distances are lattice distances between cell indices times res, truncated at (int)(maxdist / res + 1) cells
    cap = int(1.0 / res + 1)
    assert abs(float(dist.max()) - cap * res) < 1e-6
    rng = np.random.default_rng(0)
    cells = np.unique(idx, axis=0)
    for _ in range(60):
        c = rng.integers(0, dist.shape)
        d2 = ((cells - c) ** 2).sum(1).min()
        assert dist[tuple(c)] == np.float32(np.float64(np.float32(np.sqrt(min(d2, cap * cap)))) * res), c


def test_simple_forest_is_the_map_survey_describes():
    centres, res = _occupied_cell_centres(os.path.join(REF, "simple_forest.bt"))
    assert res == 0.1 and len(centres) == 4384
    lo, hi = centres.min(0) - res / 2, centres.max(0) + res / 2
    assert np.allclose(lo, [-2.9, -2.5, 0.0], atol=1e-9) and np.allclose(hi, [2.6, 3.2, 2.5], atol=1e-9)


def test_reference_maps_exact_transform_equals_the_published_propagation():
    """On every map the reference ships (world/*.bt and world/forest/*.bt) the exact lattice transform of the oracle / product and
    dynamicEDT3D's published 26-neighbour propagation (orc_edt_brushfire) give the same field, cell for cell, within the 1 m
    truncation -- so every threshold the path reads (corridor 0.2 m, grid 0.35 m, castRay up to 1 m) sees the same numbers."""
    import glob
    from oracle import oracle as O
    maps = sorted(glob.glob(os.path.join(REF, "*.bt")) + glob.glob(os.path.join(REF, "forest", "*.bt")))
    assert len(maps) >= 3
    for path in maps:
        res, leaves = O.bt_read(path)
        world = (-6, -6, 0, 6, 6, 2.5) if "forest" in path else (-10, -10, 0, 10, 10, 2.5)
        a = O.DistMap(leaves, res, world[:3], world[3:])
        b = O.DistMap.brushfire(leaves, res, world[:3], world[3:])
        assert np.array_equal(a.dist, b.dist), (path, int((a.dist != b.dist).sum()))
