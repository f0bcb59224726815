"""Python model of the product's A*: per-row open lists kept as arrays in libstdc++ unordered_map iteration order."""
import math, sys
import numpy as np

# libstdc++ _Prime_rehash_policy (max load 1.0, growth factor 2): a fresh container has one bucket, the first insertion
# takes it to 13, afterwards the count becomes the next entry of the library's prime table >= twice the current count
# whenever an insertion would exceed one element per bucket.  (The product reads this sequence from the container.)
PRIMES = [13, 29, 59, 127, 257, 541, 1109, 2357, 5087, 10273, 20753, 42043]

class Row:
    __slots__ = ("ent", "nb", "minr")
    def __init__(self):
        self.ent = []      # list of [key, g] in iteration order
        self.nb = 1
        self.minr = -1
    def _place(self, lst, nb, e):
        b = e[0] % nb
        for p, x in enumerate(lst):
            if x[0] % nb == b:
                lst.insert(p, e); return
        lst.insert(0, e)
    def insert(self, key, g):
        if len(self.ent) + 1 > self.nb or self.nb == 1:
            # rehash
            if self.nb == 1:
                nb = 13
            else:
                nb = next(p for p in PRIMES if p >= 2 * self.nb)
            old = self.ent; self.ent = []
            for e in old:
                self._place(self.ent, nb, e)
            self.nb = nb
        self._place(self.ent, self.nb, [key, g])
    def find(self, key):
        for e in self.ent:
            if e[0] == key: return e
        return None
    def erase(self, key):
        for p, e in enumerate(self.ent):
            if e[0] == key:
                del self.ent[p]; return

def astar(occ, start, goal):
    H, W, A = occ.shape
    keyf = lambda i, j, z: H * W * z + W * i + j
    def coords(key):
        z = key // (H * W); rem = key % (H * W); return rem // W, rem % W, z
    def hf(i, j, z): return 10.0 * math.sqrt((goal[0]-i)**2 + (goal[1]-j)**2 + (goal[2]-z)**2)
    def F(key, g):
        i, j, z = coords(key); return g + hf(i, j, z)
    rows = [Row() for _ in range(H)]
    closed = {}
    parent = {}
    def add_open(i, key, g, par):
        r = rows[i]; inserted = False
        e = r.find(key)
        if e is not None:
            if F(key, g) < F(key, e[1]):
                e[1] = g; parent[key] = par; inserted = True
        else:
            r.insert(key, g); parent[key] = par; inserted = True
        if len(r.ent) == 1:
            r.minr = key
        else:
            m = r.find(r.minr)
            fn, fm = F(key, g), F(m[0], m[1])
            if inserted and fn <= fm:
                if fn == fm:
                    if g >= m[1]: r.minr = key
                else:
                    r.minr = key
    sk = keyf(*start)
    add_open(start[0], sk, 0.0, -1)
    nopen = 1
    found = None
    nexp = 0
    while nopen:
        nexp += 1
        best = None
        for i in range(H):
            r = rows[i]
            if not r.ent: continue
            m = r.find(r.minr); f = F(m[0], m[1])
            if best is None or f < best[0] or (f == best[0] and m[1] >= best[1]):
                best = (f, m[1], m[0], i)
        f, g, key, i = best
        closed[key] = g
        r = rows[i]
        r.erase(key)
        mF, mg = float("inf"), 0.0
        for e in r.ent:
            fe = F(e[0], e[1])
            if fe < mF or (fe == mF and e[1] >= mg):
                r.minr = e[0]; mF = fe; mg = e[1]
        nopen -= 1
        ci, cj, cz = coords(key)
        if ci == goal[0] and cj == goal[1]:
            found = key; break
        for d in ((-1,0,0),(0,-1,0),(0,0,-1),(0,0,1),(0,1,0),(1,0,0)):
            ni, nj, nz = ci+d[0], cj+d[1], cz+d[2]
            if ni < 0 or ni >= H or nj < 0 or nj >= W or nz < 0 or nz >= A or occ[ni, nj, nz]: continue
            nk = keyf(ni, nj, nz)
            if nk in closed: continue
            before = rows[ni].find(nk) is None
            add_open(ni, nk, g + 10.0, key)
            if before: nopen += 1
    if found is None: return np.zeros((0,3), int), nexp
    path = []; k = found
    while k != -1:
        path.append(coords(k)); k = parent[k]
    return np.array(path[::-1]), nexp

if __name__ == "__main__":  # ad-hoc run; the pytest entry is tests/test_goal_planning.py
    sys.path.insert(0, "/root/repo")
    from oracle import oracle as O
    rng = np.random.default_rng(1)
    bad = 0
    for trial in range(300):
        dims = (rng.integers(3, 40), rng.integers(3, 40), rng.integers(1, 10))
        occ = (rng.random(dims) < rng.choice([0.0, 0.1, 0.25, 0.35])).astype(np.uint8)
        s = [rng.integers(0, d) for d in dims]; g = [rng.integers(0, d) for d in dims]
        occ[tuple(s)] = 0
        po = O.astar(occ, s, g)
        pm, nexp = astar(occ, s, g)
        ok = po.shape == pm.shape and np.array_equal(po, pm)
        if not ok:
            bad += 1
            print("MISMATCH trial", trial, dims, s, g, len(po), len(pm))
    print("bad", bad)
