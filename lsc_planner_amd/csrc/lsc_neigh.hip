// lsc_neigh.hip -- neighbour lists of a large swarm: which (obstacle, segment) units of phase B can carry a row that survives its pruning,
// found through a uniform grid in O(neighbours) per agent instead of a walk over all N - 1 obstacles.
//
// Reference behaviour being replaced: the obstacle loop of TrajPlanner::generateLSC, `for (int oi = 0; oi < N_obs; oi++)`
// (src/traj_planner.cpp:1335-1407) -- there every agent builds the 27 rows against every other agent, O(N^2) per tick; phase B of the plan
// kernel (lsc_kernels.hip) drops the rows that are provably redundant inside the box a control point can reach, and since round 2 decided
// WHICH units to look at by walking all obstacles' bounding spheres (O(N) per agent: 19 of an agent's 43 us at N = 1024).
//
// The test that may drop a unit (obstacle o, segment m) before the GJK -- phase B's own, lsc_kernels.hip "spatial pre-cull" -- is
//      |w_c| >= 2 s B_m + r_a + r_o + 2e-4 + R_w,     w_j = S (p_j - q_j),  S = diag(1, 1, 1 / downwash),  s = max(1, 1 / downwash),
// with p_j / q_j the six predicted control points of the agent / the obstacle in segment m, w_c their centroid, R_w = max_j |w_j - w_c| and
// B_m = max_i (|c_{0,2} - p_{m,i}| + radius of the box c_{m,i} can reach).  With a bounding sphere (C, rho) of each side's six points,
// D = S (C_a - C_o):  |w_j - D| <= s (rho_a + rho_o), hence |w_c - D| <= s (rho_a + rho_o) and R_w <= 2 s (rho_a + rho_o), and
//      |D| >= s (2 B_m + 3 (rho_a + rho_o)) + r_a + r_o + 2e-4 (+ 1e-5 for the rounding of this file's own arithmetic)
// implies the test above: a unit that fails it is listed, everything else is dropped.  The list is a superset of the units the in-kernel
// test keeps, in ascending (obstacle, segment) order; phase B runs its exact per-row test on every listed unit, so rows, their order
// and the plan are bit-identical to `prune = 3` (no cull at all).
//
// Two launches in front of the tick (launch_neigh):
//   lsc_neigh_build_kernel : 32 lanes per agent of the WHOLE swarm: bounding sphere of all predicted control points (the in-kernel cull's bound,
//                            kept for agents whose list overflows), bounding sphere per segment, B_m, and ONE insertion into the grid at the
//                            cell of the agent's centre.  Nothing is ever cleared: a bucket's counter carries the tick's tag in its upper word
//                            (atomic max with tag << 32 resets a stale bucket), and so do the swarm-wide maxima.
//   lsc_neigh_query_kernel : one 128-lane workgroup per agent of the SHARD: the cells its five query boxes overlap (clamped to the
//                            bounding box of the occupied cells), one lane per cell -> candidate obstacles in LDS -> one lane per candidate,
//                            M sphere tests -> a bit per unit in LDS -> sorted list in HBM.  Hash collisions and buckets met twice only
//                            add candidates; setting a bit twice changes nothing.
//
// The same grid serves the two other walks over all agents a workgroup of the plan kernel used to make in phase A:
//   * goalPlanningWithPriority (src/traj_planner.cpp:540-608) looks for the closest higher-priority agent and acts only when that agent is
//     closer than priority_dist_threshold: the agents within that distance of the own position are all it needs.  The current position of
//     every agent is one of the points of the sphere it is filed under, the query covers a box of threshold + G around the own position,
//     and the candidates come as a second short list (the rule itself -- priorities, directions, the tie order -- stays where it was);
//   * the disturbance checks (obstaclePredictionCheck / initialTrajPlanningCheck, :866-878, 1047-1061: "is ANY agent off its plan?")
//     are made ONCE per agent by the build kernel (same float32 test, the persistent flags set there) and reach phase A as one bit.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "lsc_kernels.h"
#include "lsc_predict.hpp"

namespace lsc {

namespace {

constexpr int NQ = 128;                 // lanes of a query workgroup
constexpr int QUEUE_CAP = 4096;         // candidate obstacles per agent (LDS)
constexpr int BITMAP_WORDS = 2048;      // one bit per unit: n_units <= 0xffff (the lists hold 16-bit units, like phase B's own)
constexpr int MAX_CELLS = 8192;         // cells a query may visit
constexpr int PRIO_CAP = NEIGH_PRIO_CAP; // candidates of the priority rule per agent (LDS)
constexpr int CELL_LIM = 1 << 20;       // cell coordinates are clamped to +- this (monotone: a clamped range still contains a clamped point)
constexpr unsigned BIAS = 0x80000000u;

__device__ __forceinline__ int cell_of(double x, double inv)
{
    const double c = floor(x * inv);
    return (int)fmin(fmax(c, -(double)CELL_LIM), (double)CELL_LIM);      // (NaN -> -CELL_LIM: fmax returns the other operand)
}
__device__ __forceinline__ unsigned cell_hash(int ix, int iy, int iz)
{
    return ((unsigned)ix * 73856093u) ^ ((unsigned)iy * 19349663u) ^ ((unsigned)iz * 83492791u);
}
__device__ __forceinline__ unsigned long long tagged(unsigned tag, unsigned v) { return ((unsigned long long)tag << 32) | v; }
__device__ __forceinline__ unsigned untag(unsigned long long w, unsigned tag, unsigned otherwise) { return (unsigned)(w >> 32) == tag ? (unsigned)w : otherwise; }

// slots of NeighArgs::glob
enum { G_RADIUS = 0, G_OVF = 1, G_MAXX = 2, G_MAXY = 3, G_MAXZ = 4, G_MINX = 5, G_MINY = 6, G_MINZ = 7, G_SLACK = 8, G_COUNT = 9 };

}  // namespace

constexpr int NB_THREADS = 256, NB_AGENTS = NB_THREADS / 32;      // lanes / agents of a build workgroup
__global__ __launch_bounds__(NB_THREADS) void lsc_neigh_build_kernel(NeighArgs a)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    const int qa = q >> 5, j = q & 31;
    const bool live = qa < a.N;
    const int ql = live ? qa : 0;
    const int pi = j < SEGV ? j : SEGV - 1, m = pi / NC;
    // ---- everything this kernel reads, in one batch (what follows is arithmetic; the atomics come last, because whatever is issued
    // behind an atomic that returns a value waits for it)
    F3 po[6];
    load_segment(a.state, a.traj_prev, ql, m, a.planner_seq, a.dtf, po);
    const float *s = a.state + 9 * ql;
    float sv[9];
#pragma unroll
    for (int i = 0; i < 9; i++) sv[i] = s[i];
    double vm[3], am[3];
#pragma unroll
    for (int k = 0; k < 3; k++) { vm[k] = a.vmax[3 * ql + k]; am[k] = a.amax[3 * ql + k]; }
    const double r_obs = a.radius_obs[ql];
    float t1[3] = {0.f, 0.f, 0.f};
    int ev = 0;
    if (a.checks) {
        const float *t = a.traj_prev + (size_t)ql * NV + NC;
        t1[0] = t[0]; t1[1] = t[SEGV]; t1[2] = t[2 * SEGV];
        ev = (int)a.ever[ql];
    }
    F3 me = po[0];
#pragma unroll
    for (int i = 1; i < 6; i++) if (pi % NC == i) me = po[i];
    if (j == SEGV) me = F3{sv[0], sv[1], sv[2]};      // the current position is one of the points of the sphere (priority candidates are found by position)
    // ---- sphere around all predicted control points and the current position (lanes beyond SEGV + 1 repeat the last point): centre =
    // float32 of the mean, radius taken around the centre that is stored and rounded up
    double cx = (double)me.x, cy = (double)me.y, cz = (double)me.z;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) { cx += __shfl_xor(cx, o, 32); cy += __shfl_xor(cy, o, 32); cz += __shfl_xor(cz, o, 32); }
    const float fx = (float)(cx * (1.0 / 32.0)), fy = (float)(cy * (1.0 / 32.0)), fz = (float)(cz * (1.0 / 32.0));
    double r2;
    {
        const double ex = (double)me.x - (double)fx, ey = (double)me.y - (double)fy, ez = (double)me.z - (double)fz;
        r2 = ex * ex + ey * ey + ez * ez;
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) r2 = fmax(r2, __shfl_xor(r2, o, 32));
    const float rad = (float)(sqrt(r2) * (1.0 + 1e-6) + 1e-6);
    // ---- sphere around the six points of this lane's segment
    float sx[6], sy[6], sz[6];
    double mx = 0.0, my = 0.0, mz = 0.0;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        sx[i] = __shfl(me.x, m * NC + i, 32); sy[i] = __shfl(me.y, m * NC + i, 32); sz[i] = __shfl(me.z, m * NC + i, 32);
        mx += (double)sx[i]; my += (double)sy[i]; mz += (double)sz[i];
    }
    const float gx = (float)(mx * (1.0 / 6.0)), gy = (float)(my * (1.0 / 6.0)), gz = (float)(mz * (1.0 / 6.0));
    double s2 = 0.0;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const double ex = (double)sx[i] - (double)gx, ey = (double)sy[i] - (double)gy, ez = (double)sz[i] - (double)gz;
        s2 = fmax(s2, ex * ex + ey * ey + ez * ez);
    }
    const float srad = (float)(sqrt(s2) * (1.0 + 1e-6) + 1e-6);
    float smax = j < SEGV ? srad : 0.f;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) smax = fmaxf(smax, __shfl_xor(smax, o, 32));
    // ---- B_m: the arithmetic of phase A / B of plan_agent (state constants, reach of every control point by prefix sums over the steps)
    double c2[3], lo[3], hi[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        double c0 = (double)sv[k];
        double c1 = c0 + (double)sv[3 + k] * a.hv_scale;
        double cc = (double)sv[6 + k] * a.ha_scale + 2.0 * c1 - c0;
        if (a.dim2 && k == 2) c0 = c1 = cc = a.z2d;
        c2[k] = cc;
        const double V = vm[k] * a.hv_scale, A = am[k] * a.ha_scale, d0 = cc - c1;
        const bool in = j >= 1 && j < 28;
        lo[k] = in ? fmax(-V, d0 - (double)j * A) - 1e-9 : 0.0;
        hi[k] = in ? fmin(V, d0 + (double)j * A) + 1e-9 : 0.0;
#pragma unroll
        for (int d = 1; d < 32; d <<= 1) {
            const double ul = __shfl_up(lo[k], d, 32), uh = __shfl_up(hi[k], d, 32);
            if (j >= d) { lo[k] += ul; hi[k] += uh; }
        }
    }
    const int K = DEG * (pi / NC) + (pi % NC) - 2, Kc = K >= 1 ? K : 1;
    double d2 = 0.0, e2 = 0.0;
    {
        const double mec[3] = {(double)me.x, (double)me.y, (double)me.z};
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const double el = __shfl(lo[k], Kc, 32), eh = __shfl(hi[k], Kc, 32);
            const double e = fmax(fabs(el), fabs(eh)), dd = c2[k] - mec[k];
            d2 += dd * dd; e2 += e * e;
        }
    }
    const double bc = K >= 1 ? sqrt(d2) + sqrt(e2) : 0.0;
    double bm = 0.0;
#pragma unroll
    for (int i = 0; i < 6; i++) bm = fmax(bm, __shfl(bc, m * NC + i, 32));
    // ---- disturbance check of this agent (off_plan of plan_agent: float32, no contraction): the persistent flag is set HERE, once
    bool slack = false;
    if (a.checks) {
#pragma clang fp contract(off)
        const float dx = t1[0] - sv[0], dy = t1[1] - sv[1], dz = t1[2] - sv[2];
        const float n2 = dx * dx + dy * dy + dz * dz;
        const bool nw = sqrt((double)n2) > a.reset_thr;
        if (nw && live && j == 0) a.ever[qa] = 1;
        slack = live && (nw || ev != 0);
    }
    if (live && j < SEGV && j % NC == 0) {
        reinterpret_cast<float4 *>(a.seg_bound)[(size_t)qa * M + m] = make_float4(gx, gy, gz, srad);
        a.reach[(size_t)qa * M + m] = (float)(bm * (1.0 + 1e-6) + 1e-6);
    }
    // ---- what this agent adds, as an OBSTACLE, to the query radius of everybody else: its segment centres and its position lie within rad of
    // the centre it is filed under, and a segment's sphere test reaches 3 s rho_o + r_o further
    const float gr = (float)((3.0 * a.sc_max * (double)smax + r_obs + (double)rad) * (1.0 + 1e-6) + 1e-6);
    const int ix = cell_of((double)fx, a.inv_cell), iy = cell_of((double)fy, a.inv_cell), iz = cell_of((double)fz, a.inv_cell_z);
    // ---- one insertion into the grid, at the cell of the centre.  atomic max with (tag, 0) first: a bucket last touched in an older tick
    // counts as empty.  The counter's round trip (0.3 us on its own address, tools/microbench/launch_atomics.hip) is issued HERE and its slot
    // used at the very end, so that the reduction below runs under it.
    unsigned long long *const bucket = a.cells + 4 * (size_t)(cell_hash(ix, iy, iz) & a.hmask);
    unsigned slot = 0;
    if (live && j == 0) {
        reinterpret_cast<float4 *>(a.obs_bound)[qa] = make_float4(fx, fy, fz, rad);
        atomicMax(bucket, tagged(a.tag, 0u));
        slot = (unsigned)atomicAdd(bucket, 1ull);
    }
    // ---- swarm-wide maxima (query radius, bounding box of the occupied cells, "somebody is off its plan"): reduced over the workgroup's
    // agents first, and an atomic only when it would change what stands there (one atomic per agent on seven addresses was 40 us of a
    // 1024-agent tick)
    __shared__ unsigned red[8][NB_AGENTS];
    if (j == 0) {
        const int s_ = threadIdx.x >> 5;
        red[0][s_] = live ? __float_as_uint(gr) : 0u;
        red[1][s_] = live ? BIAS + (unsigned)ix : 0u; red[2][s_] = live ? BIAS + (unsigned)iy : 0u; red[3][s_] = live ? BIAS + (unsigned)iz : 0u;
        red[4][s_] = live ? BIAS - (unsigned)ix : 0u; red[5][s_] = live ? BIAS - (unsigned)iy : 0u; red[6][s_] = live ? BIAS - (unsigned)iz : 0u;
        red[7][s_] = slack ? 1u : 0u;
    }
    __syncthreads();
    {
        const int which = threadIdx.x >> 5;
        unsigned v = j < NB_AGENTS ? red[which][j] : 0u;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, o, 32));
        if (j == 0 && v != 0u) {
            unsigned long long *g = &a.glob[which == 0 ? G_RADIUS : (which == 7 ? G_SLACK : G_MAXX + (which - 1))];
            const unsigned long long mine = tagged(a.tag, v);
            if (__atomic_load_n(g, __ATOMIC_RELAXED) < mine) atomicMax(g, mine);
        }
    }
    if (live && j == 0) {
        if (slot < (unsigned)NEIGH_SLOTS) reinterpret_cast<unsigned short *>(bucket + 1)[slot] = (unsigned short)qa;
        else {
            atomicMax(&a.glob[G_OVF], tagged(a.tag, 0u));
            const unsigned ov = (unsigned)atomicAdd(&a.glob[G_OVF], 1ull);
            if (ov < (unsigned)a.ovf_cap) a.ovf[ov] = (unsigned short)qa;
        }
    }
}

__global__ __launch_bounds__(NQ) void lsc_neigh_query_kernel(NeighArgs a)
{
    __shared__ unsigned bitmap[BITMAP_WORDS];
    __shared__ unsigned short queue[QUEUE_CAP];
    __shared__ unsigned short pcand[PRIO_CAP];
    __shared__ int qn, pn, wtot[NQ / 64], rnk;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int al = blockIdx.x, qa = a.first + al;
    const unsigned tag = a.tag;
    const int n_units = (a.N - 1) * M, words = (n_units + 31) >> 5;
    // ---- this agent's side of the tests, the swarm-wide numbers: one batch of loads
    const double r_a = a.radius[qa], dw_a = a.downwash[qa];
    float4 sb[M];
    float rc[M];
#pragma unroll
    for (int m = 0; m < M; m++) { sb[m] = reinterpret_cast<const float4 *>(a.seg_bound)[(size_t)qa * M + m]; rc[m] = a.reach[(size_t)qa * M + m]; }
    unsigned long long gl[G_COUNT];
#pragma unroll
    for (int i = 0; i < G_COUNT; i++) gl[i] = a.glob[i];
    const bool prio = a.goal_mode == 1;
    float px = 0.f, py = 0.f, pz = 0.f;
    if (prio) { const float *s = a.state + 9 * qa; px = s[0]; py = s[1]; pz = s[2]; }
    for (int w = tid; w < words; w += NQ) bitmap[w] = 0u;
    if (tid == 0) { qn = 0; pn = 0; rnk = 0; }
    double ca[M][3], ra[M], bm[M], qrad[M];
    const double G = (double)__uint_as_float(untag(gl[G_RADIUS], tag, 0u));
#pragma unroll
    for (int m = 0; m < M; m++) {
        ca[m][0] = (double)sb[m].x; ca[m][1] = (double)sb[m].y; ca[m][2] = (double)sb[m].z; ra[m] = (double)sb[m].w;
        bm[m] = (double)rc[m];
        qrad[m] = a.sc_max * (2.0 * bm[m] + 3.0 * ra[m]) + r_a + 2e-4 + 1e-5 + G;      // reach of the query for segment m along x and y
    }
    const double pthr = a.prio_thr * (1.0 + 1e-6) + 1e-9;      // (a candidate list may hold more than the rule needs, never less)
    // ---- cells the query boxes overlap, clamped to the cells that hold somebody
    int c0[3], c1[3];
    bool fail = false;
    {
        const int bmax[3] = {(int)(untag(gl[G_MAXX], tag, BIAS) - BIAS), (int)(untag(gl[G_MAXY], tag, BIAS) - BIAS), (int)(untag(gl[G_MAXZ], tag, BIAS) - BIAS)};
        const int bmin[3] = {(int)(BIAS - untag(gl[G_MINX], tag, BIAS)), (int)(BIAS - untag(gl[G_MINY], tag, BIAS)), (int)(BIAS - untag(gl[G_MINZ], tag, BIAS))};
        const double pp[3] = {(double)px, (double)py, (double)pz};
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const double sck = k == 2 ? a.zscale : 1.0, inv = k == 2 ? a.inv_cell_z : a.inv_cell;
            double lo = 1e300, hi = -1e300;
#pragma unroll
            for (int m = 0; m < M; m++) { lo = fmin(lo, ca[m][k] - qrad[m] * sck); hi = fmax(hi, ca[m][k] + qrad[m] * sck); }
            if (prio) { lo = fmin(lo, pp[k] - (pthr + G)); hi = fmax(hi, pp[k] + (pthr + G)); }
            c0[k] = max(cell_of(lo, inv), bmin[k]); c1[k] = min(cell_of(hi, inv), bmax[k]);
            if (!(lo <= hi)) fail = true;                  // (NaN inputs: no list, the agent's own phase B decides)
        }
    }
    const int nx = c1[0] - c0[0] + 1, ny = c1[1] - c0[1] + 1, nz = c1[2] - c0[2] + 1;
    long long ncell = (nx > 0 && ny > 0 && nz > 0) ? (long long)nx * ny * nz : 0;
    if (ncell > MAX_CELLS) { fail = true; ncell = 0; }
    const unsigned ovn = untag(gl[G_OVF], tag, 0u);
    if (ovn > (unsigned)a.ovf_cap) fail = true;            // somebody is in no bucket and in no overflow slot
    __syncthreads();
    // ---- stage 1: one lane per cell, the bucket's agents -> candidate queue
    for (int ci = tid; ci < (int)ncell; ci += NQ) {
        const int ix = c0[0] + ci % nx, iy = c0[1] + (ci / nx) % ny, iz = c0[2] + ci / (nx * ny);
        const unsigned long long *b = a.cells + 4 * (size_t)(cell_hash(ix, iy, iz) & a.hmask);
        const unsigned long long w0 = b[0], w1 = b[1], w2 = b[2], w3 = b[3];
        unsigned n = untag(w0, tag, 0u);
        n = n < (unsigned)NEIGH_SLOTS ? n : (unsigned)NEIGH_SLOTS;
        if (n == 0) continue;
        const int at = atomicAdd(&qn, (int)n);
#pragma unroll
        for (int e = 0; e < NEIGH_SLOTS; e++) {
            const unsigned long long w = e < 4 ? w1 : (e < 8 ? w2 : w3);
            if (e < (int)n && at + e < QUEUE_CAP) queue[at + e] = (unsigned short)(w >> (16 * (e & 3)));
        }
    }
    for (unsigned e = tid; e < ovn && e < (unsigned)a.ovf_cap; e += NQ) {
        const int at = atomicAdd(&qn, 1);
        if (at < QUEUE_CAP) queue[at] = a.ovf[e];
    }
    __syncthreads();
    const int nq = qn;
    if (nq > QUEUE_CAP) fail = true;
    // ---- stage 2: one lane per candidate obstacle, M sphere tests (+ the distance of the priority rule)
    for (int ci = tid; ci < nq && ci < QUEUE_CAP; ci += NQ) {
        const int o = (int)queue[ci];
        if (o == qa || o >= a.N) continue;
        const double r_o = a.radius_obs[o], dw_o = a.downwash_obs[o];
        float4 so[M];
#pragma unroll
        for (int m = 0; m < M; m++) so[m] = reinterpret_cast<const float4 *>(a.seg_bound)[(size_t)o * M + m];
        float ox = 0.f, oy = 0.f, oz = 0.f;
        if (prio) { const float *s = a.state + 9 * o; ox = s[0]; oy = s[1]; oz = s[2]; }
        const double downwash = (dw_a * r_a + dw_o * r_o) / (r_a + r_o);
        const double idw = 1.0 / downwash, sc = fmax(1.0, idw);
        const int oi = o < qa ? o : o - 1;
#pragma unroll
        for (int m = 0; m < M; m++) {
            const double dx = ca[m][0] - (double)so[m].x, dy = ca[m][1] - (double)so[m].y, dz = (ca[m][2] - (double)so[m].z) * idw;
            const double need = sc * (2.0 * bm[m] + 3.0 * (ra[m] + (double)so[m].w)) + (r_a + r_o) + 2e-4 + 1e-5;
            if (!(dx * dx + dy * dy + dz * dz >= need * need)) {
                const int u = oi * M + m;
                atomicOr(&bitmap[u >> 5], 1u << (u & 31));
            }
        }
        if (prio) {
#pragma clang fp contract(off)      // distf of goalPlanningWithPriority: float32 differences and squares, square root in double
            const float dx = ox - px, dy = oy - py, dz = oz - pz;
            const float n2 = dx * dx + dy * dy + dz * dz;
            if (!(sqrt((double)n2) >= pthr)) {
                const int at = atomicAdd(&pn, 1);
                if (at < PRIO_CAP) pcand[at] = (unsigned short)o;
            }
        }
    }
    __syncthreads();
    // ---- stage 3: the set bits in ascending order -> the agent's list (each lane a contiguous run of words)
    const int per = (words + NQ - 1) / NQ;
    int mine = 0;
    for (int w = tid * per; w < (tid + 1) * per && w < words; w++) mine += __popc(bitmap[w]);
    int incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int up = __shfl_up(incl, d, 64);
        if (lane >= d) incl += up;
    }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    int off = incl - mine, total = 0;
#pragma unroll
    for (int w = 0; w < NQ / 64; w++) { off += w < wave ? wtot[w] : 0; total += wtot[w]; }
    const bool lfail = fail || total > a.list_cap;
    if (!lfail) {
        unsigned short *out = a.list + (size_t)qa * a.list_cap;
        for (int w = tid * per; w < (tid + 1) * per && w < words; w++) {
            unsigned bits = bitmap[w];
            while (bits) {
                const int bpos = __ffs(bits) - 1;
                bits &= bits - 1;
                out[off++] = (unsigned short)(w * 32 + bpos);
            }
        }
    }
    // the candidates of the priority rule (any order, an agent may appear twice: the rule takes a minimum) and the disturbance bit
    const int np = pn;
    const bool pfail = fail || np > PRIO_CAP || np > a.plist_cap;
    if (prio && !pfail)
        for (int i = tid; i < np; i += NQ) a.plist[(size_t)qa * a.plist_cap + i] = pcand[i];
    if (tid == 0) {
        a.cnt[qa] = lfail ? -1 : total;
        const int any = untag(gl[G_SLACK], tag, 0u) ? (1 << 30) : 0;
        a.pcnt[qa] = (prio && pfail) ? -1 : ((prio ? np : 0) | any);
    }
    // ---- launch order of the throughput build (more than one round of workgroups): rank of this agent among the shard's agents by the cost
    // of its previous tick, descending, ties by index (what lsc_prep_kernel does without the lists)
    if (a.order) {
        auto cost = [&](int p) { return (unsigned)a.iters[a.first + p] * (unsigned)(a.nrows[a.first + p] + 600); };
        const unsigned cq = cost(al);
        int r = 0;
        for (int p = tid; p < a.count; p += NQ) {
            const unsigned cp = cost(p);
            r += (cp > cq) || (cp == cq && p < al);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) r += __shfl_xor(r, o, 64);
        if (lane == 0) atomicAdd(&rnk, r);
        __syncthreads();
        if (tid == 0) a.order[rnk] = al;
    }
}

hipError_t launch_neigh(const NeighArgs &a, hipStream_t st)
{
    if (a.N < 2 || (a.N - 1) * M > 0xffff) return hipErrorInvalidValue;
    hipLaunchKernelGGL(lsc_neigh_build_kernel, dim3((a.N + NB_AGENTS - 1) / NB_AGENTS), dim3(NB_THREADS), 0, st, a);
    if (a.count > 0) hipLaunchKernelGGL(lsc_neigh_query_kernel, dim3(a.count), dim3(NQ), 0, st, a);
    return hipGetLastError();
}

}  // namespace lsc
