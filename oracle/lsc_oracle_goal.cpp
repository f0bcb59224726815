/*
 * lsc_oracle_goal.cpp -- CPU restatement of goal planning with a distance field (mode/goal = prior_based on an
 * octomap world).  TEST INFRASTRUCTURE ONLY (see lsc_oracle.h).
 *
 *   TrajPlanner::goalPlanningWithPriority          src/traj_planner.cpp:540-608
 *   GridBasedPlanner::plan / updateGridInfo / updateGridMap / updateGridMission / gridVectorToPoint3D /
 *       point3DToGridVector / findLOSFreeGoal / castRay        src/grid_based_planner.cpp:53-433
 *   Astar-3D (ISearch::startSearch, findSuccessors, findMin, deleteMin, addOpen, Astar::computeHFromCellToCell)
 *                                                              src/Astar-3D/isearch.cpp:46-283, astar.cpp:18-52
 *
 * PARITY UNPINNED: Astar-3D's map.h includes tinyxml2.h, which this image lacks, so the reference's A* cannot be
 * compiled here as a pin.  What makes an exact restatement possible at all: the search keeps its OPEN list as one
 * std::unordered_map per grid row and, after every pop, rescans that row in the container's iteration order, keeping
 * the LAST entry among equal (F, g) -- so equal-cost ties are broken by libstdc++'s hash-table order.  This file is
 * C++ precisely so that it can use the same std::unordered_map<uint_least32_t, ...> with the same sequence of
 * insertions and erasures and inherit that order from the library instead of guessing it (the product emulates the
 * order explicitly; the tests compare the two).
 */
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <unordered_map>
#include <vector>

#include "lsc_oracle.h"

namespace {

constexpr double kEps = 1e-9;        /* SP_EPSILON        include/sp_const.hpp:3 */
constexpr double kEpsF = 1e-5;       /* SP_EPSILON_FLOAT  include/sp_const.hpp:4 */
constexpr double kLine = 10.0;       /* CN_MC_LINE        include/Astar-3D/gl_const.h:175 */

inline double f32_norm(const float v[3])
{
    float n2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    return std::sqrt((double)n2);
}
inline double f32_dist(const float a[3], const float b[3])
{
    const float d[3] = {a[0] - b[0], a[1] - b[1], a[2] - b[2]};
    return f32_norm(d);
}
inline void f32_normalize(float v[3])
{
    const double len = f32_norm(v);
    if (len > 0) { const float l = (float)len; v[0] /= l; v[1] /= l; v[2] /= l; }
}

/* DynamicEDTOctomap::getDistance(point3d): key = floor(coord / res) + 32768, -1 outside the box */
float edt_at(const orc_edt *e, const float p[3])
{
    const int dims[3] = {e->nx, e->ny, e->nz};
    int c[3];
    for (int a = 0; a < 3; a++) {
        c[a] = (int)std::floor((1.0 / e->res) * (double)p[a]) + 32768 - e->key_min[a];
        if (c[a] < 0 || c[a] >= dims[a]) return -1.0f;
    }
    return e->dist[((size_t)c[0] * e->ny + c[1]) * e->nz + c[2]];
}

struct Grid {
    double gmin[3], gmax[3], res;
    int dim[3];
    std::vector<unsigned char> occ;                     /* [i][j][k] */
    unsigned char &at(int i, int j, int k) { return occ[((size_t)i * dim[1] + j) * dim[2] + k]; }
    unsigned char get(int i, int j, int k) const { return occ[((size_t)i * dim[1] + j) * dim[2] + k]; }
    void point(int i, int j, int k, float p[3]) const   /* gridVectorToPoint3D :300-305 */
    {
        p[0] = (float)(gmin[0] + i * res); p[1] = (float)(gmin[1] + j * res); p[2] = (float)(gmin[2] + k * res);
    }
    void cell(const float p[3], int c[3]) const         /* point3DToGridVector :325-330 */
    {
        for (int a = 0; a < 3; a++) c[a] = (int)std::round(((double)p[a] - gmin[a]) / res);
    }
};

/* updateGridInfo :72-93 */
void grid_info(const orc_params *prm, double res, Grid &g)
{
    g.res = res;
    for (int a = 0; a < 3; a++) {
        g.gmin[a] = -std::floor((-(double)prm->world_min[a] + kEps) / res) * res;
        g.gmax[a] = std::floor(((double)prm->world_max[a] + kEps) / res) * res;
    }
    if (prm->world_dimension == 2) { g.gmin[2] = prm->world_z_2d; g.gmax[2] = prm->world_z_2d; }   /* :82-85 */
    for (int a = 0; a < 3; a++) g.dim[a] = (int)std::round((g.gmax[a] - g.gmin[a]) / res) + 1;
}

struct Node { int i, j, z; double F, g, H; int64_t parent; };
thread_local long g_last_expansions = 0;     /* nodes popped by the last search (diagnostics for tests / bench notes) */

/* ISearch::startSearch with Astar(1.0, CN_SP_BT_GMAX), EnvironmentOptions(): euclidean metric, no diagonals */
bool astar(const Grid &G, const int start[3], const int goal[3], std::vector<int> &path /* flat i,j,k */)
{
    const int height = G.dim[0], width = G.dim[1], alt = G.dim[2];
    auto key_of = [&](int i, int j, int z) { return (uint_least32_t)((uint_least32_t)height * width * z + width * i + j); };
    auto hfun = [&](int i, int j, int z) {
        return kLine * std::sqrt((double)((goal[0] - i) * (goal[0] - i) + (goal[1] - j) * (goal[1] - j) + (goal[2] - z) * (goal[2] - z)));
    };
    std::vector<std::unordered_map<uint_least32_t, Node>> open(height);
    std::vector<int_least64_t> open_min(height, -1);
    std::unordered_map<uint_least32_t, Node> close;
    int open_size = 0;

    auto add_open = [&](const Node &nn, uint_least32_t key) {         /* isearch.cpp:243-283 */
        bool inserted = false;
        const size_t idx = (size_t)nn.i;
        if (open[idx].find(key) != open[idx].end()) {
            if (nn.F < open[idx][key].F) { open[idx][key] = nn; inserted = true; }
        } else {
            open[idx][key] = nn;
            inserted = true;
            ++open_size;
        }
        if (open[idx].size() == 1) {
            open_min[idx] = key;
        } else {
            const Node mn = open[idx][(uint_least32_t)open_min[idx]];
            if (inserted && nn.F <= mn.F) {
                if (nn.F == mn.F) { if (nn.g >= mn.g) open_min[idx] = key; }
                else open_min[idx] = key;
            }
        }
    };
    auto find_min = [&]() {                                            /* isearch.cpp:181-209 */
        Node mn{};
        mn.F = std::numeric_limits<double>::infinity();
        mn.g = 0;
        for (int i = 0; i < height; i++) {
            if (open[i].empty()) continue;
            const Node cur = open[i][(uint_least32_t)open_min[i]];
            if (cur.F <= mn.F) {
                if (cur.F == mn.F) { if (cur.g >= mn.g) mn = cur; }
                else mn = cur;
            }
        }
        return mn;
    };
    auto delete_min = [&](const Node &m, uint_least32_t key) {          /* isearch.cpp:211-241 */
        const size_t idx = (size_t)m.i;
        open[idx].erase(key);
        Node mn{};
        mn.F = (double)std::numeric_limits<float>::infinity();
        mn.g = 0;
        for (auto it = open[idx].begin(); it != open[idx].end(); ++it) {
            if (it->second.F <= mn.F) {
                if (it->second.F == mn.F) {
                    if (it->second.g >= mn.g) { open_min[idx] = it->first; mn = it->second; }
                } else { open_min[idx] = it->first; mn = it->second; }
            }
        }
    };

    Node cur{start[0], start[1], start[2], 0, 0, 0, -1};
    cur.H = hfun(cur.i, cur.j, cur.z);
    cur.F = 1.0f * cur.H;
    add_open(cur, key_of(cur.i, cur.j, cur.z));
    open_size = 1;
    bool found = false;
    g_last_expansions = 0;
    while (open_size != 0) {
        g_last_expansions++;
        cur = find_min();
        const uint_least32_t ck = key_of(cur.i, cur.j, cur.z);
        close.insert({ck, cur});
        delete_min(cur, ck);
        --open_size;
        if (cur.i == goal[0] && cur.j == goal[1]) { found = true; break; }   /* the altitude is not part of the goal test */
        for (int di = -1; di <= 1; ++di)
            for (int dj = -1; dj <= 1; ++dj)
                for (int dh = -1; dh <= 1; ++dh) {
                    if (std::abs(di) + std::abs(dj) + std::abs(dh) != 1) continue;   /* allowdiagonal = false */
                    const int ni = cur.i + di, nj = cur.j + dj, nz = cur.z + dh;
                    if (ni < 0 || ni >= height || nj < 0 || nj >= width || nz < 0 || nz > alt - 1) continue;
                    if (G.get(ni, nj, nz) != 0) continue;
                    /* allowcutcorners = false: the three cells tested are the target or the current cell for axis moves */
                    if (G.get(cur.i + di, cur.j + dj, cur.z) != 0 || G.get(cur.i + di, cur.j, cur.z + dh) != 0 ||
                        G.get(cur.i, cur.j + dj, cur.z + dh) != 0)
                        continue;
                    const uint_least32_t nk = key_of(ni, nj, nz);
                    if (close.find(nk) != close.end()) continue;
                    Node nn{ni, nj, nz, 0, cur.g + kLine, hfun(ni, nj, nz), (int64_t)ck};
                    nn.F = nn.g + 1.0f * nn.H;
                    add_open(nn, nk);
                }
    }
    path.clear();
    if (!found) return false;
    std::vector<Node> rev;
    Node c = cur;
    for (;;) {
        rev.push_back(c);
        if (c.parent < 0) break;
        c = close.find((uint_least32_t)c.parent)->second;
    }
    for (size_t t = rev.size(); t-- > 0;) { path.push_back(rev[t].i); path.push_back(rev[t].j); path.push_back(rev[t].z); }
    return true;
}

/* castRay :409-433 */
bool cast_ray(const orc_edt *e, double wres, const float a[3], const float b[3], double radius)
{
    const double d = f32_dist(a, b);
    const double thr = std::sqrt(0.25 * d * d + radius * radius);
    const double sa = (double)edt_at(e, a), sb = (double)edt_at(e, b);
    if (sa < radius + 0.5 * wres - kEpsF) return false;
    if (sb < radius + 0.5 * wres - kEpsF) return false;
    if (thr < 1.0 && sa > thr && sb > thr) return true;
    float mid[3];
    for (int k = 0; k < 3; k++) { float s = a[k] + b[k]; mid[k] = s * 0.5f; }
    return cast_ray(e, wres, a, mid, radius) && cast_ray(e, wres, mid, b, radius);
}

}  // namespace

extern "C" {

long orc_astar_last_expansions(void) { return g_last_expansions; }

void orc_grid_dims(const orc_params *prm, double grid_res, int dims[3], double gmin[3])
{
    Grid g;
    grid_info(prm, grid_res, g);
    for (int a = 0; a < 3; a++) { dims[a] = g.dim[a]; gmin[a] = g.gmin[a]; }
}

/* bare search on a caller-supplied occupancy grid [ni][nj][nk] (0 free); path_out [max_len][3]; returns the number
 * of cells of the path, 0 when the goal cannot be reached */
int orc_astar(const unsigned char *occ, const int dims[3], const int start[3], const int goal[3], int *path_out, int max_len)
{
    Grid g;
    for (int a = 0; a < 3; a++) { g.dim[a] = dims[a]; g.gmin[a] = 0; g.gmax[a] = 0; }
    g.res = 1;
    g.occ.assign(occ, occ + (size_t)dims[0] * dims[1] * dims[2]);
    std::vector<int> p;
    if (!astar(g, start, goal, p)) return 0;
    const int n = (int)p.size() / 3;
    for (int t = 0; t < n && t < max_len; t++) for (int a = 0; a < 3; a++) path_out[3 * t + a] = p[3 * t + a];
    return n;
}

/* TrajPlanner::goalPlanningWithPriority with a distance field.  radius/downwash are the mission values; when
 * prm->obs_f32 the obstacle copies pass through float32 (dynamic_msgs::Obstacle).  path_out (optional, [max][3]
 * grid cells) receives the grid path that findLOSFreeGoal walks; *path_len its length; flags bit 0: retreat rule
 * fired, bit 1: the prioritised search failed and the search without priorities was used. */
/* disturbance reset (src/traj_planner.cpp:547-551, 866-878, 1047-1061): the slack obstacles are stamped as higher priority */
/* slack_row [N] (may be NULL): the agent's slack set, indexed by agent; own_reset: its own initial trajectory was reset.  Explicit
 * arguments, no process-wide state: a threaded caller (the QP stage already is) must not pick up another agent's slack set. */
void orc_goal_prior_based_map(const orc_params *prm, const orc_edt *edt, double world_res, double grid_res, double grid_margin,
                              int N, int qi, const float *state, const float *desired_goal, const float *prev_traj,
                              int planner_seq, double goal_threshold, double priority_dist_threshold, double goal_radius,
                              const double *radius, const double *downwash, const unsigned char *slack_row, int own_reset,
                              float out_goal[3], int *path_out, int max_path, int *path_len, int *flags)
{
    const float *pos = state + 9 * qi;
    const float *goal_i = desired_goal + 3 * qi;
    const double dist_to_goal = f32_dist(pos, goal_i);
    double min_dist_to_obs = 1e9;                 /* SP_INFINITY */
    int closest = -1;
    std::vector<char> high(N, 0);
    if (path_len) *path_len = 0;
    if (flags) *flags = 0;
    for (int qj = 0; qj < N; qj++) {              /* :547-577 */
        if (qj == qi) continue;
        if (slack_row && slack_row[qj]) { high[qj] = 1; continue; }   /* :548-551 */
        const float *opos = state + 9 * qj, *ogoal = desired_goal + 3 * qj;
        const double obs_dist_to_goal = f32_dist(opos, ogoal);
        const double dist_to_obs = f32_dist(opos, pos);
        if (obs_dist_to_goal < goal_threshold) continue;
        const float *pt = prev_traj + (size_t)qj * ORC_NV;
        float a[3], b[3];
        for (int k = 0; k < 3; k++) {
            const float last = pt[k * ORC_SEGV + (ORC_M - 1) * ORC_NC + ORC_N], first = pt[k * ORC_SEGV + ORC_N];
            a[k] = last - first;
            b[k] = first - pos[k];
        }
        const float dot = a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
        if (dist_to_goal > goal_threshold && (double)dot > 0) continue;
        if (dist_to_goal < goal_threshold || obs_dist_to_goal < dist_to_goal) {
            if (dist_to_obs < min_dist_to_obs) { min_dist_to_obs = dist_to_obs; closest = qj; }
            high[qj] = 1;
        }
    }
    if (min_dist_to_obs < priority_dist_threshold) {      /* :580-587 */
        const double dist_keep = priority_dist_threshold + 0.1;
        const float *opos = state + 9 * closest;
        float dir[3] = {opos[0] - pos[0], opos[1] - pos[1], opos[2] - pos[2]};
        f32_normalize(dir);
        for (int k = 0; k < 3; k++) { const float s = dir[k] * (float)dist_keep; out_goal[k] = pos[k] - s; }
        if (flags) *flags |= 1;
        return;
    }

    /* ---- GridBasedPlanner::plan (:53-70), first with the higher-priority agents as obstacles, then without */
    const double r_a = radius[qi], dw_a = downwash[qi];
    Grid G;
    grid_info(prm, grid_res, G);
    std::vector<unsigned char> static_occ((size_t)G.dim[0] * G.dim[1] * G.dim[2], 0);
    {
        const float margin = (float)grid_margin;          /* `float grid_margin = param.grid_margin` :111 */
        G.occ = static_occ;
        for (int i = 0; i < G.dim[0]; i++)
            for (int j = 0; j < G.dim[1]; j++)
                for (int k = 0; k < G.dim[2]; k++) {
                    float p[3];
                    G.point(i, j, k, p);
                    const float dist = edt_at(edt, p);
                    if ((double)dist < r_a + (double)margin) static_occ[((size_t)i * G.dim[1] + j) * G.dim[2] + k] = 1;
                }
    }
    std::vector<int> path;
    for (int attempt = 0; attempt < 2; attempt++) {
        G.occ = static_occ;
        if (attempt == 0) {
            for (int qj = 0; qj < N; qj++) {              /* :163-189 */
                if (qj == qi || !high[qj]) continue;
                const double r_o = prm->obs_f32 ? (double)(float)radius[qj] : radius[qj];
                const double dw_o = prm->obs_f32 ? (double)(float)downwash[qj] : downwash[qj];
                const double px = (double)state[9 * qj], py = (double)state[9 * qj + 1], pz = (double)state[9 * qj + 2];
                const int oi_ = (int)std::round((px - G.gmin[0] + kEps) / grid_res);
                const int oj_ = (int)std::round((py - G.gmin[1] + kEps) / grid_res);
                /* `obs_k = 0` stays when world/dimension is 2 (:127-133) */
                const int ok_ = prm->world_dimension == 2 ? 0 : (int)std::round((pz - G.gmin[2] + kEps) / grid_res);
                const int sxy = (int)std::ceil((r_a + r_o) / grid_res);
                const int sz = (int)std::ceil((r_a * dw_a + r_o * dw_o) / grid_res);
                const double dwt = (r_a * dw_a + r_o * dw_o) / (r_a + r_o);
                for (int i = std::max(oi_ - sxy, 0); i <= std::min(oi_ + sxy, G.dim[0] - 1); i++)
                    for (int j = std::max(oj_ - sxy, 0); j <= std::min(oj_ + sxy, G.dim[1] - 1); j++)
                        for (int k = std::max(ok_ - sz, 0); k <= std::min(ok_ + sz, G.dim[2] - 1); k++) {
                            float p[3];
                            G.point(i, j, k, p);
                            const double dist = std::sqrt(std::pow((double)p[0] - px, 2) + std::pow((double)p[1] - py, 2) +
                                                          std::pow(((double)p[2] - pz) / dwt, 2));
                            if (dist < r_a + r_o) G.at(i, j, k) = 1;
                        }
            }
        } else if (flags) {
            *flags |= 2;
        }
        /* updateGridMission :193-239 */
        int s[3], g[3];
        G.cell(pos, s);
        G.cell(goal_i, g);
        if (prm->world_dimension == 2) { s[2] = 0; g[2] = 0; }     /* :199-202 */
        /* the reference indexes the grid with the start cell unchecked (a position within half a cell of world_max
         * rounds to dim): clamped here and in the product instead of reading out of bounds */
        for (int a = 0; a < 3; a++) s[a] = s[a] < 0 ? 0 : (s[a] > G.dim[a] - 1 ? G.dim[a] - 1 : s[a]);
        if (G.get(s[0], s[1], s[2]) == 1) {
            const int wdim = prm->world_dimension == 2 ? 2 : 3;
            int best = 1000000000, c[3] = {s[0], s[1], s[2]};
            for (int i = -2; i < 3; i++)
                for (int j = -2; j < 3; j++)
                    for (int k = 2 - wdim; k < wdim - 1; k++) {     /* :204-206: -1..1 in 3-D, 0 in a planar world */
                        const int x = s[0] + i, y = s[1] + j, z = s[2] + k;
                        const bool occupied = x < 0 || x > G.dim[0] - 1 || y < 0 || y > G.dim[1] - 1 || z < 0 || z > G.dim[2] - 1 ||
                                              G.get(x, y, z) == 1;
                        if (!occupied) {
                            const int dist = std::abs(i) + std::abs(j) + std::abs(k);
                            if (dist < best) { best = dist; c[0] = x; c[1] = y; c[2] = z; }
                        }
                    }
            s[0] = c[0]; s[1] = c[1]; s[2] = c[2];
            if (G.get(s[0], s[1], s[2]) == 1) G.at(s[0], s[1], s[2]) = 0;
        }
        if (astar(G, s, g, path)) break;
        path.clear();
    }
    const int n_path = (int)path.size() / 3;
    if (path_len) *path_len = n_path;
    if (path_out) for (int t = 0; t < n_path && t < max_path; t++) for (int a = 0; a < 3; a++) path_out[3 * t + a] = path[3 * t + a];

    /* ---- findLOSFreeGoal(initial_traj[M-1][n], desired goal, ...) :350-407 */
    float cur[3];
    if (own_reset) {
        for (int k = 0; k < 3; k++) cur[k] = pos[k];               /* initial trajectory reset to the current position */
    } else if (planner_seq < 2) {
        float tmp[ORC_NV];
        orc_const_vel_traj(pos, pos + 3, prm->dt, tmp);
        for (int k = 0; k < 3; k++) cur[k] = tmp[k * ORC_SEGV + (ORC_M - 1) * ORC_NC + ORC_N];
    } else {
        const float *pt = prev_traj + (size_t)qi * ORC_NV;
        for (int k = 0; k < 3; k++) cur[k] = pt[k * ORC_SEGV + (ORC_M - 1) * ORC_NC + ORC_N];
    }
    float los[3] = {cur[0], cur[1], cur[2]};
    for (int it = 0; it < 6; it++) {
        const double margin_ratio = 1.5 - 0.1 * it;
        for (int t = 0; t <= n_path; t++) {
            float p[3];
            if (t < n_path) G.point(path[3 * t], path[3 * t + 1], path[3 * t + 2], p);
            else { p[0] = goal_i[0]; p[1] = goal_i[1]; p[2] = goal_i[2]; }
            if (cast_ray(edt, world_res, cur, p, r_a * margin_ratio)) { los[0] = p[0]; los[1] = p[1]; los[2] = p[2]; }
            else break;
        }
        if (f32_dist(los, cur) > 0.3) break;
    }
    float delta[3] = {los[0] - cur[0], los[1] - cur[1], los[2] - cur[2]};
    if (f32_norm(delta) > goal_radius) {
        f32_normalize(delta);
        for (int k = 0; k < 3; k++) { const float s = delta[k] * (float)goal_radius; los[k] = cur[k] + s; }
    }
    out_goal[0] = los[0]; out_goal[1] = los[1]; out_goal[2] = los[2];
}

}  // extern "C"
