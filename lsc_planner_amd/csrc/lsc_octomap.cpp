// lsc_octomap.cpp -- host-side map inputs of the SFC stage: octomap ".bt" reader and the dense Euclidean distance
// field that TrajPlanner::setDistMap hands to every agent (src/multi_sync_simulator.cpp:153-167 builds it with
// octomap::OcTree::readBinary + DynamicEDTOctomap(1.0, tree, world_min, world_max, false).update()).
//
// octomap / dynamicEDT3D are external libraries that are not part of the reference tree; what is implemented here is
// their documented file format and semantics (assumptions are listed in DESIGN.md, section 2):
//   * .bt : depth-first stream, 2 bytes per inner node (8 children x 2 bits: 01 free, 10 occupied, 11 inner), 16 levels
//   * key = floor(coord / res) + 32768
//   * distance in cells = sqrt(exact squared lattice distance to the nearest occupied max-depth cell), truncated at
//     (int)(maxdist/res + 1) cells; value returned in metres as float.
// The transform is the separable lower-envelope algorithm (three 1-D passes over integer squared distances), i.e. a
// different algorithm than the test suite's brute-force checker -- both are exact on the lattice, so they must agree.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/lsc_planner_amd.h"

namespace {

struct Leaf { int x, y, z, size; };

bool read_bt(const char *path, double &res, std::vector<Leaf> &occ)
{
    FILE *f = std::fopen(path, "rb");
    if (!f) return false;
    std::vector<unsigned char> buf;
    unsigned char tmp[65536];
    size_t n;
    while ((n = std::fread(tmp, 1, sizeof(tmp), f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
    std::fclose(f);
    // header: text lines until "data"
    size_t pos = 0, data_at = std::string::npos;
    res = 0.0;
    while (pos < buf.size()) {
        size_t e = pos;
        while (e < buf.size() && buf[e] != '\n') e++;
        std::string line(buf.begin() + pos, buf.begin() + e);
        if (line.rfind("res ", 0) == 0) res = std::atof(line.c_str() + 4);
        if (line == "data") { data_at = e + 1; break; }
        pos = e + 1;
    }
    if (data_at == std::string::npos || !(res > 0.0)) return false;
    // iterative depth-first walk (explicit stack, children pushed in reverse so that child 0 is read first)
    struct Node { int x, y, z, size; };
    std::vector<Node> stack;
    stack.push_back({0, 0, 0, 65536});
    size_t p = data_at;
    while (!stack.empty()) {
        Node nd = stack.back();
        stack.pop_back();
        if (p + 2 > buf.size() || nd.size < 2) return false;
        const unsigned bits = buf[p] | (unsigned(buf[p + 1]) << 8);
        p += 2;
        const int half = nd.size >> 1;
        Node kids[8];
        int nk = 0;
        for (int ch = 0; ch < 8; ch++) {
            const unsigned v = (bits >> (2 * ch)) & 3u;
            const Node c{nd.x + ((ch & 1) ? half : 0), nd.y + ((ch & 2) ? half : 0), nd.z + ((ch & 4) ? half : 0), half};
            if (v == 2u) occ.push_back({c.x, c.y, c.z, c.size});
            else if (v == 3u) kids[nk++] = c;
        }
        for (int j = nk - 1; j >= 0; j--) stack.push_back(kids[j]);
    }
    return true;
}

inline int coord_key(double c, double res) { return (int)std::floor((1.0 / res) * c) + 32768; }

// 1-D squared distance transform (lower envelope of parabolas, Felzenszwalb & Huttenlocher); values are exact
// integers held in doubles, BIG marks "no obstacle on this line yet"
constexpr double BIG = 1e12;
void dt1d(const double *f, int n, double *out, int *v, double *z)
{
    int k = 0;
    v[0] = 0;
    z[0] = -1e300; z[1] = 1e300;
    for (int q = 1; q < n; q++) {
        double s;
        for (;;) {
            s = ((f[q] + (double)q * q) - (f[v[k]] + (double)v[k] * v[k])) / (2.0 * q - 2.0 * v[k]);
            if (s <= z[k]) k--;
            else break;
        }
        k++;
        v[k] = q; z[k] = s; z[k + 1] = 1e300;
    }
    k = 0;
    for (int q = 0; q < n; q++) {
        while (z[k + 1] < q) k++;
        out[q] = (double)(q - v[k]) * (q - v[k]) + f[v[k]];
    }
}

}  // namespace

extern "C" {

int lsc_edt_from_bt(const char *path, const float world_min[3], const float world_max[3], double maxdist, float **edt_out,
                    int dims_out[3], int key_min_out[3], double *res_out)
{
    if (!path || !world_min || !world_max || !edt_out || !dims_out || !key_min_out || !res_out) return LSC_EINVAL;
    double res;
    std::vector<Leaf> occ;
    if (!read_bt(path, res, occ)) return LSC_EINVAL;
    int kmin[3], dims[3];
    for (int a = 0; a < 3; a++) {
        kmin[a] = coord_key((double)world_min[a], res);
        dims[a] = coord_key((double)world_max[a], res) - kmin[a] + 1;
        if (dims[a] < 1) return LSC_EINVAL;
    }
    const int nx = dims[0], ny = dims[1], nz = dims[2];
    const int md = (int)(maxdist / res + 1);
    const double trunc2 = (double)md * md;           // DynamicEDT3D(maxDist_squared)
    std::vector<double> g((size_t)nx * ny * nz, BIG);
    for (const Leaf &l : occ) {
        // a pruned leaf covers size^3 cells (up to 32768^3): intersect it with the grid before walking it
        const long x0 = std::max<long>(l.x - kmin[0], 0), x1 = std::min<long>((long)l.x + l.size - kmin[0], nx);
        const long y0 = std::max<long>(l.y - kmin[1], 0), y1 = std::min<long>((long)l.y + l.size - kmin[1], ny);
        const long z0 = std::max<long>(l.z - kmin[2], 0), z1 = std::min<long>((long)l.z + l.size - kmin[2], nz);
        for (long x = x0; x < x1; x++)
            for (long y = y0; y < y1; y++)
                for (long z = z0; z < z1; z++) g[((size_t)x * ny + y) * nz + z] = 0.0;
    }
    const int nmax = std::max(nx, std::max(ny, nz));
    std::vector<double> f(nmax), o(nmax), zz(nmax + 2);
    std::vector<int> v(nmax + 1);
    // along z (contiguous), then y, then x
    for (int x = 0; x < nx; x++)
        for (int y = 0; y < ny; y++) {
            double *line = &g[((size_t)x * ny + y) * nz];
            dt1d(line, nz, o.data(), v.data(), zz.data());
            std::memcpy(line, o.data(), sizeof(double) * nz);
        }
    for (int x = 0; x < nx; x++)
        for (int z = 0; z < nz; z++) {
            for (int y = 0; y < ny; y++) f[y] = g[((size_t)x * ny + y) * nz + z];
            dt1d(f.data(), ny, o.data(), v.data(), zz.data());
            for (int y = 0; y < ny; y++) g[((size_t)x * ny + y) * nz + z] = o[y];
        }
    for (int y = 0; y < ny; y++)
        for (int z = 0; z < nz; z++) {
            for (int x = 0; x < nx; x++) f[x] = g[((size_t)x * ny + y) * nz + z];
            dt1d(f.data(), nx, o.data(), v.data(), zz.data());
            for (int x = 0; x < nx; x++) g[((size_t)x * ny + y) * nz + z] = o[x];
        }
    float *edt = (float *)std::malloc(sizeof(float) * (size_t)nx * ny * nz);
    if (!edt) return LSC_ENOMEM;
    for (size_t i = 0; i < (size_t)nx * ny * nz; i++) {
        const float cells = (float)std::sqrt(g[i] < trunc2 ? g[i] : trunc2);
        edt[i] = (float)((double)cells * res);
    }
    *edt_out = edt;
    for (int a = 0; a < 3; a++) { dims_out[a] = dims[a]; key_min_out[a] = kmin[a]; }
    *res_out = res;
    return LSC_OK;
}

void lsc_free_host(void *p) { std::free(p); }

}  // extern "C"
