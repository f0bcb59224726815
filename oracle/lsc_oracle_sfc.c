/*
 * lsc_oracle_sfc.c -- oracle part 2: octomap .bt reader, Euclidean distance field, Safe Flight Corridor growth.
 * TEST INFRASTRUCTURE ONLY (see lsc_oracle.h).
 *
 * Follows include/corridor_constructor.hpp:18-245 (box growth) and src/traj_planner.cpp:1451-1491
 * (generateFeasibleSFC).  octomap and dynamicEDT3D are NOT in /root/reference (apt packages, version unpinned,
 * CMakeLists.txt:17-25): PARITY UNPINNED for the map side.  Their published behaviour is restated:
 *   - .bt stream: depth-first, two bytes per inner node = 8 children x 2 bits (LSB first; 01 free leaf,
 *     10 occupied leaf, 11 inner node), 16 levels, key 32768 = origin (octomap AbstractOccupancyOcTree::readBinary)
 *   - OcTreeBaseImpl::coordToKey: key = (int)floor(coord / res) + 32768
 *   - DynamicEDTOctomap(maxdist, tree, bbxMin, bbxMax, unknownOccupied=false): grid over the keys of the bounding
 *     box, occupied leaves (pruned cubes expanded) are obstacles, distance in cells = sqrt of the exact squared
 *     lattice distance, truncated at maxDist = (int)(maxdist/res + 1) cells; getDistance(p) = cell distance * res,
 *     -1 outside the grid.  The SFC test only compares against r + 0.05 - 1e-5 (< 0.2 m = 2 cells), where the
 *     brushfire propagation of the library is exact, so the restatement does not depend on its far-field errors.
 */
#include "lsc_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- .bt reader ---------------------------- */
typedef struct {
    const unsigned char *p, *end;
    int *keys;      /* [n][4] : min-corner key x,y,z and cube size (in max-depth cells) */
    int n, cap;
    int bad;
} bt_ctx;

static void bt_node(bt_ctx *c, int mx, int my, int mz, int size)
{
    if (c->p + 2 > c->end || size < 2) { c->bad = 1; return; }
    unsigned bits = c->p[0] | (c->p[1] << 8);
    c->p += 2;
    int half = size >> 1;
    int inner[8], ni = 0;
    for (int ch = 0; ch < 8; ch++) {
        unsigned v = (bits >> (2 * ch)) & 3u;
        if (v == 2u) {
            if (c->n == c->cap) { c->cap = c->cap ? 2 * c->cap : 1024; c->keys = (int *)realloc(c->keys, sizeof(int) * 4 * c->cap); }
            int *k = c->keys + 4 * c->n++;
            k[0] = mx + ((ch & 1) ? half : 0); k[1] = my + ((ch & 2) ? half : 0); k[2] = mz + ((ch & 4) ? half : 0); k[3] = half;
        } else if (v == 3u) inner[ni++] = ch;
    }
    for (int j = 0; j < ni && !c->bad; j++) {
        int ch = inner[j];
        bt_node(c, mx + ((ch & 1) ? half : 0), my + ((ch & 2) ? half : 0), mz + ((ch & 4) ? half : 0), half);
    }
}

/* returns the occupied leaves as [n][4] (min key, cube edge in cells); caller frees *keys */
int orc_bt_read(const char *path, double *res, int **keys, int *n)
{
    FILE *f = fopen(path, "rb");
    if (!f) return -1;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    unsigned char *buf = (unsigned char *)malloc((size_t)sz + 1);
    if (fread(buf, 1, (size_t)sz, f) != (size_t)sz) { fclose(f); free(buf); return -1; }
    fclose(f);
    buf[sz] = 0;
    *res = 0;
    long pos = 0, data_at = -1;
    while (pos < sz) {
        long e = pos;
        while (e < sz && buf[e] != '\n') e++;
        if (e - pos >= 4 && !memcmp(buf + pos, "res ", 4)) *res = atof((const char *)buf + pos + 4);
        if (e - pos == 4 && !memcmp(buf + pos, "data", 4)) { data_at = e + 1; break; }
        pos = e + 1;
    }
    if (data_at < 0 || !(*res > 0)) { free(buf); return -2; }
    bt_ctx c = {buf + data_at, buf + sz, NULL, 0, 0, 0};
    bt_node(&c, 0, 0, 0, 65536);
    free(buf);
    if (c.bad) { free(c.keys); return -3; }
    *keys = c.keys;
    *n = c.n;
    return 0;
}

static int coord_to_key(double coord, double res) { return (int)floor((1.0 / res) * coord) + 32768; }

/* Dense distance field over the key box of [world_min, world_max]; caller frees edt->dist. */
int orc_edt_build(const int *leaves, int n, double res, const float world_min[3], const float world_max[3],
                  double maxdist, orc_edt *edt)
{
    int kmin[3], dims[3];
    for (int a = 0; a < 3; a++) {
        kmin[a] = coord_to_key((double)world_min[a], res);
        dims[a] = coord_to_key((double)world_max[a], res) - kmin[a] + 1;
        if (dims[a] < 1) return -1;
    }
    const int nx = dims[0], ny = dims[1], nz = dims[2];
    const int md = (int)(maxdist / res + 1);               /* DynamicEDT3D(maxdist_squared = md*md) */
    const int md2 = md * md;
    int *sq = (int *)malloc(sizeof(int) * (size_t)nx * ny * nz);
    for (size_t i = 0; i < (size_t)nx * ny * nz; i++) sq[i] = md2;
    for (int l = 0; l < n; l++) {
        const int *k = leaves + 4 * l;
        for (int dx = 0; dx < k[3]; dx++)
            for (int dy = 0; dy < k[3]; dy++)
                for (int dz = 0; dz < k[3]; dz++) {
                    int x = k[0] + dx - kmin[0], y = k[1] + dy - kmin[1], z = k[2] + dz - kmin[2];
                    if (x < 0 || y < 0 || z < 0 || x >= nx || y >= ny || z >= nz) continue;   /* outside the bounding box */
                    /* brute-force stamp of the exact squared lattice distance within the truncation radius */
                    for (int ox = -md; ox <= md; ox++) {
                        int xx = x + ox;
                        if (xx < 0 || xx >= nx) continue;
                        for (int oy = -md; oy <= md; oy++) {
                            int yy = y + oy;
                            if (yy < 0 || yy >= ny) continue;
                            int base = ox * ox + oy * oy;
                            if (base >= md2) continue;
                            for (int oz = -md; oz <= md; oz++) {
                                int zz = z + oz;
                                if (zz < 0 || zz >= nz) continue;
                                int d2 = base + oz * oz;
                                int *c = &sq[((size_t)xx * ny + yy) * nz + zz];
                                if (d2 < *c) *c = d2;
                            }
                        }
                    }
                }
    }
    float *dist = (float *)malloc(sizeof(float) * (size_t)nx * ny * nz);
    for (size_t i = 0; i < (size_t)nx * ny * nz; i++) {
        float cells = (float)sqrt((double)sq[i]);          /* dataCell::dist (float), in cells */
        dist[i] = (float)((double)cells * res);             /* getDistance: dist * treeResolution -> float */
    }
    free(sq);
    edt->dist = dist;
    edt->nx = nx; edt->ny = ny; edt->nz = nz;
    edt->key_min[0] = kmin[0]; edt->key_min[1] = kmin[1]; edt->key_min[2] = kmin[2];
    edt->res = res;
    return 0;
}

/* The same field by the propagation dynamicEDT3D itself runs (Lau, Sprunk, Burgard, "Efficient grid-based spatial representations
 * for robot navigation in dynamic environments", RAS 2013, algorithm "lower"; dynamicEDT3D is an apt package absent from this image,
 * so this is a restatement of the PUBLISHED algorithm, not of its source): every occupied cell enters a queue ordered by squared
 * distance with itself as closest obstacle; a popped cell offers its closest obstacle to its 26 neighbours, a neighbour takes it when
 * the squared distance to THAT obstacle is smaller than what it holds (and within the truncation radius) and is queued in turn.
 * This is a vector propagation, not an exact Euclidean transform: a cell whose true nearest obstacle is not the nearest obstacle of any
 * of its 26 neighbours keeps a slightly larger value.  tests/test_map_assumptions.py runs it next to orc_edt_build on the reference's
 * three maps to see whether -- and from which distance on -- the two differ.  sq_out (optional): the squared distances in cells. */
int orc_edt_brushfire(const int *leaves, int n, double res, const float world_min[3], const float world_max[3],
                      double maxdist, orc_edt *edt, int *sq_out)
{
    int kmin[3], dims[3];
    for (int a = 0; a < 3; a++) {
        kmin[a] = coord_to_key((double)world_min[a], res);
        dims[a] = coord_to_key((double)world_max[a], res) - kmin[a] + 1;
        if (dims[a] < 1) return -1;
    }
    const int nx = dims[0], ny = dims[1], nz = dims[2];
    const size_t C = (size_t)nx * ny * nz;
    const int md = (int)(maxdist / res + 1), md2 = md * md;
    int *sq = (int *)malloc(sizeof(int) * C);
    int *ob = (int *)malloc(sizeof(int) * C);              /* closest obstacle: its cell index, -1 none */
    for (size_t i = 0; i < C; i++) { sq[i] = md2; ob[i] = -1; }
    /* bucket queue by squared distance: bucket d holds cell indices, appended as they are (re)inserted */
    int **bk = (int **)calloc((size_t)md2 + 1, sizeof(int *));
    int *bn = (int *)calloc((size_t)md2 + 1, sizeof(int)), *bc = (int *)calloc((size_t)md2 + 1, sizeof(int));
#define ORC_PUSH(d, idx) do { if (bn[d] == bc[d]) { bc[d] = bc[d] ? 2 * bc[d] : 256; bk[d] = (int *)realloc(bk[d], sizeof(int) * (size_t)bc[d]); } bk[d][bn[d]++] = (idx); } while (0)
    for (int l = 0; l < n; l++) {
        const int *k = leaves + 4 * l;
        for (int dx = 0; dx < k[3]; dx++)
            for (int dy = 0; dy < k[3]; dy++)
                for (int dz = 0; dz < k[3]; dz++) {
                    int x = k[0] + dx - kmin[0], y = k[1] + dy - kmin[1], z = k[2] + dz - kmin[2];
                    if (x < 0 || y < 0 || z < 0 || x >= nx || y >= ny || z >= nz) continue;
                    const int idx = (x * ny + y) * nz + z;
                    if (sq[idx] == 0) continue;
                    sq[idx] = 0; ob[idx] = idx;
                    ORC_PUSH(0, idx);
                }
    }
    for (int d = 0; d < md2; d++)
        for (int q = 0; q < bn[d]; q++) {                   /* (a bucket may grow while it is walked: entries of the same distance) */
            const int s = bk[d][q];
            if (sq[s] != d) continue;                       /* superseded by a smaller distance since it was queued */
            const int sx = s / (ny * nz), sy = (s / nz) % ny, sz = s % nz;
            const int o = ob[s], ox = o / (ny * nz), oy = (o / nz) % ny, oz = o % nz;
            for (int ax = -1; ax <= 1; ax++)
                for (int ay = -1; ay <= 1; ay++)
                    for (int az = -1; az <= 1; az++) {
                        if (!ax && !ay && !az) continue;
                        const int x = sx + ax, y = sy + ay, z = sz + az;
                        if (x < 0 || y < 0 || z < 0 || x >= nx || y >= ny || z >= nz) continue;
                        const int nd = (x - ox) * (x - ox) + (y - oy) * (y - oy) + (z - oz) * (z - oz);
                        const int ni = (x * ny + y) * nz + z;
                        if (nd < sq[ni] && nd < md2) { sq[ni] = nd; ob[ni] = o; ORC_PUSH(nd, ni); }
                    }
        }
#undef ORC_PUSH
    float *dist = (float *)malloc(sizeof(float) * C);
    for (size_t i = 0; i < C; i++) {
        float cells = (float)sqrt((double)sq[i]);
        dist[i] = (float)((double)cells * res);
    }
    if (sq_out) memcpy(sq_out, sq, sizeof(int) * C);
    for (int d = 0; d <= md2; d++) free(bk[d]);
    free(bk); free(bn); free(bc); free(sq); free(ob);
    edt->dist = dist;
    edt->nx = nx; edt->ny = ny; edt->nz = nz;
    edt->key_min[0] = kmin[0]; edt->key_min[1] = kmin[1]; edt->key_min[2] = kmin[2];
    edt->res = res;
    return 0;
}

/* DynamicEDTOctomap::getDistance(point3d): -1 outside the grid */
static float edt_lookup(const orc_edt *e, const float p[3])
{
    int c[3];
    const int dims[3] = {e->nx, e->ny, e->nz};
    for (int a = 0; a < 3; a++) {
        c[a] = coord_to_key((double)p[a], e->res) - e->key_min[a];
        if (c[a] < 0 || c[a] >= dims[a]) return -1.0f;
    }
    return e->dist[((size_t)c[0] * e->ny + c[1]) * e->nz + c[2]];
}

/* ---------------------------------------------------------------- corridor_constructor.hpp ---------------- */
/* :81-122 */
static int obstacle_in_box(const orc_params *prm, const orc_edt *e, double wres, const double box[6], double margin)
{
    int size[3];
    for (int i = 0; i < 3; i++) size[i] = (int)round((box[i + 3] - box[i]) / wres) + 1;
    const int n0 = size[0] > 2 ? size[0] : 2, n1 = size[1] > 2 ? size[1] : 2, n2 = size[2] > 2 ? size[2] : 2;
    for (int i0 = 0; i0 < n0; i0++)
        for (int i1 = 0; i1 < n1; i1++)
            for (int i2 = 0; i2 < n2; i2++) {
                const int it[3] = {i0, i1, i2};
                float sp[3], delta[3];
                for (int i = 0; i < 3; i++) {
                    if (size[i] == 1 && it[i] > 0) sp[i] = (float)box[i];
                    else sp[i] = (float)(box[i] + it[i] * wres);
                    if (it[i] == 0 && box[i] > (double)prm->world_min[i] + 1e-5) delta[i] = (float)-1e-5;
                    else delta[i] = (float)1e-5;
                }
                for (int i = 0; i < 3; i++) sp[i] = sp[i] + delta[i];
                float dist = edt_lookup(e, sp);
                if ((double)dist < margin + 0.5 * wres - 1e-5) return 1;
            }
    return 0;
}

/* :124-131 with margin 0 */
static int box_in_boundary(const orc_params *prm, const double box[6])
{
    return box[0] > (double)prm->world_min[0] - 1e-9 && box[1] > (double)prm->world_min[1] - 1e-9 &&
           box[2] > (double)prm->world_min[2] - 1e-9 && box[3] < (double)prm->world_max[0] + 1e-9 &&
           box[4] < (double)prm->world_max[1] + 1e-9 && box[5] < (double)prm->world_max[2] + 1e-9;
}

/* :142-182 */
static void axis_candidates(const double box[6], const float goal[3], int cand[6])
{
    float mid[3], delta[3];
    for (int k = 0; k < 3; k++) { mid[k] = (float)(0.5 * (box[k] + box[k + 3])); delta[k] = goal[k] - mid[k]; }
    int offs[3];
    double val[3];
    for (int k = 0; k < 3; k++) { offs[k] = delta[k] > 0 ? 3 : 0; val[k] = fabs((double)delta[k]); }
    int order[3], n = 0;
    double maxv = -1, minv = 1e9;
    for (int i = 0; i < 3; i++) {
        if (val[i] > maxv) {                 /* insert at the front */
            for (int j = n; j > 0; j--) order[j] = order[j - 1];
            order[0] = i; n++;
            maxv = val[i];
        } else if (val[i] < minv) {          /* append */
            order[n++] = i;
            minv = val[i];
        } else {                             /* insert at position 1 */
            for (int j = n; j > 1; j--) order[j] = order[j - 1];
            order[1] = i; n++;
        }
    }
    for (int i = 0; i < 3; i++) {
        cand[i] = order[i] + offs[order[i]];
        cand[5 - i] = order[i] + (3 - offs[order[i]]);
    }
}

/* expandBoxFromPoint :18-44 + expandSFCFromBox :234-245 + expand_box :184-232.
 * returns 0 ok, 1 when the seed box already touches an obstacle (the reference throws std::invalid_argument) */
int orc_expand_box(const orc_params *prm, const orc_edt *e, double wres, const float point[3], const float goal[3],
                   double radius, double out[6])
{
    double box[6];
    for (int i = 0; i < 3; i++) {
        double p = (double)point[i];
        double rp = round(p / wres) * wres;
        if (fabs(p - rp) < 0.01) { box[i] = rp; box[i + 3] = rp; }
        else { box[i] = floor(p / wres) * wres; box[i + 3] = ceil(p / wres) * wres; }
    }
    if (obstacle_in_box(prm, e, wres, box, radius)) return 1;

    int cand[6], ncand = 6;
    axis_candidates(box, goal, cand);
    int i = -1;
    double cur[6], cnd[6], upd[6];
    memcpy(cur, box, sizeof(cur));
    while (ncand > 0) {
        memcpy(cnd, cur, sizeof(cur));
        memcpy(upd, cur, sizeof(cur));
        while (!obstacle_in_box(prm, e, wres, upd, radius) && box_in_boundary(prm, upd)) {
            i++;
            if (i >= ncand) i = 0;
            int axis = cand[i];
            memcpy(cur, cnd, sizeof(cur));
            memcpy(upd, cnd, sizeof(cur));
            if (axis < 3) {
                upd[axis + 3] = cnd[axis];
                cnd[axis] = cnd[axis] - wres;
                upd[axis] = cnd[axis];
            } else {
                upd[axis - 3] = cnd[axis];
                cnd[axis] = cnd[axis] + wres;
                upd[axis] = cnd[axis];
            }
        }
        if (i < 0) return 2;   /* seed box outside the world: undefined in the reference */
        for (int j = i; j < ncand - 1; j++) cand[j] = cand[j + 1];
        ncand--;
        if (i > 0) i--;
        else i = ncand - 1;
    }
    memcpy(out, cur, sizeof(cur));
    return 0;
}

/* TrajPlanner::generateFeasibleSFC (src/traj_planner.cpp:1451-1491) for one agent.
 * sfc [M][6] float (Box = two point3d), init_flag in/out (flag_initialize_sfc). */
int orc_update_sfc(const orc_params *prm, const orc_edt *e, double wres, const float pos[3], const float goal[3],
                   const float *prev_traj /*[3][30]*/, double radius, float *sfc, int *init_flag)
{
    double box[6];
    if (*init_flag) {
        int rc = orc_expand_box(prm, e, wres, pos, goal, radius, box);
        if (rc) return rc;
        for (int m = 0; m < ORC_M; m++)
            for (int j = 0; j < 6; j++) sfc[m * 6 + j] = (float)box[j];
        *init_flag = 0;
    } else {
        float last[3];
        for (int k = 0; k < 3; k++) last[k] = prev_traj[k * ORC_SEGV + (ORC_M - 1) * ORC_NC + ORC_N];
        int rc = orc_expand_box(prm, e, wres, last, goal, radius, box);
        if (rc) return rc;
        for (int m = 1; m < ORC_M; m++)
            for (int j = 0; j < 6; j++) sfc[(m - 1) * 6 + j] = sfc[m * 6 + j];
        for (int j = 0; j < 6; j++) sfc[(ORC_M - 1) * 6 + j] = (float)box[j];
    }
    return 0;
}
