// lsc_neigh.hip -- neighbour lists of a large swarm: which (obstacle, segment) units of phase B can carry a row that survives its pruning,
// found through a uniform grid in O(neighbours) per agent instead of a walk over all N - 1 obstacles.
//
// Reference behaviour being replaced: the obstacle loop of TrajPlanner::generateLSC, `for (int oi = 0; oi < N_obs; oi++)`
// (src/traj_planner.cpp:1335-1407) -- there every agent builds the 27 rows against every other agent, O(N^2) per tick; phase B of the plan
// kernel (lsc_kernels.hip) drops the rows that are provably redundant inside the box a control point can reach, and since round 2 decided
// WHICH units to look at by walking all obstacles' bounding spheres (O(N) per agent: 6.7 us of an agent's 46 at N = 1024, ~270 of ~315 at
// N = 8192 together with phase A's two walks below).
//
// The test that may drop a unit (obstacle o, segment m) before the GJK -- phase B's own, lsc_kernels.hip "spatial pre-cull" -- is
//      |w_c| >= 2 s B_m + r_a + r_o + 2e-4 + R_w,     w_j = S (p_j - q_j),  S = diag(1, 1, 1 / downwash),  s = max(1, 1 / downwash),
// with p_j / q_j the six predicted control points of the agent / the obstacle in segment m, w_c their centroid, R_w = max_j |w_j - w_c| and
// B_m = max_i (|c_{0,2} - p_{m,i}| + radius of the box c_{m,i} can reach).  With a bounding sphere (C, rho) of each side's six points,
// D = S (C_a - C_o):  |w_j - D| <= s (rho_a + rho_o), hence |w_c - D| <= s (rho_a + rho_o) and R_w <= 2 s (rho_a + rho_o), and
//      |D| >= s (2 B_m + 3 (rho_a + rho_o)) + r_a + r_o + 2e-4   (evaluated in float32 and rounded up: x (1 + 2e-5) + 5e-4 instead of + 2e-4)
// implies the test above: a unit that fails it is listed, everything else is dropped.  The list is a superset of the units the in-kernel
// test keeps, in ascending (obstacle, segment) order; phase B runs its exact per-row test on every listed unit, so rows, their order
// and the plan are bit-identical to `prune = 3` (no cull at all).
//
// Two launches in front of the tick (launch_neigh):
//   lsc_neigh_build_kernel : 8 lanes per agent of the WHOLE swarm: bounding sphere of all predicted control points (the in-kernel cull's bound,
//                            kept for agents whose list overflows), bounding sphere per segment, B_m, and ONE insertion into the grid at the
//                            cell of the agent's centre.  Nothing is ever cleared: a bucket's counter carries the tick's tag in its upper word
//                            (atomic max with tag << 32 resets a stale bucket), and so do the swarm-wide maxima.
//   lsc_neigh_query_kernel : one 256-lane workgroup per agent of the SHARD: the cells its five query boxes overlap (clamped to the
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

constexpr int NQ = 256;                 // lanes of a query workgroup
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
    unsigned h = ((unsigned)ix * 0x9E3779B1u) ^ ((unsigned)iy * 0x85EBCA77u) ^ ((unsigned)iz * 0xC2B2AE3Du);
    return h ^ (h >> 15);                  // (the mask keeps the low bits: fold the high ones in)
}
__device__ __forceinline__ unsigned long long tagged(unsigned tag, unsigned v) { return ((unsigned long long)tag << 32) | v; }
__device__ __forceinline__ unsigned untag(unsigned long long w, unsigned tag, unsigned otherwise) { return (unsigned)(w >> 32) == tag ? (unsigned)w : otherwise; }

// slots of NeighArgs::glob
enum { G_RADIUS = 0, G_OVF = 1, G_MAXX = 2, G_MAXY = 3, G_MAXZ = 4, G_MINX = 5, G_MINY = 6, G_MINZ = 7, G_SLACK = 8, G_COUNT = 9 };

}  // namespace

// ---- build: eight lanes per agent, lane m < M owns segment m.  float32 throughout -- these are bounds, not results: every quantity that
// widens a test is rounded up by far more than float32 arithmetic can lose (B_m: + 1e-5 B_m + 1e-3 m; radii: + 1e-5 r + 1e-5 m) --, no LDS
// round trips between the lanes of an agent (three DPP steps reduce over its eight lanes), one lane's work ~1 200 instructions: a lone wave
// pays 6-8 cycles per instruction and ~100 per LDS round trip, and the first version (32 lanes per agent, float64, ~150 ds_bpermute) was
// 14 us of a 1024-agent tick.
constexpr int NB_THREADS = 256, NB_LANES = 8, NB_AGENTS = NB_THREADS / NB_LANES;      // lanes / lanes per agent / agents of a build workgroup
static_assert(M <= NB_LANES, "one lane per segment");

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true)); }
// sum / maximum over the eight lanes of an agent, result in all of them: quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror
__device__ __forceinline__ float sum8(float v) { v += dpp_f<0xB1>(v); v += dpp_f<0x4E>(v); v += dpp_f<0x141>(v); return v; }
__device__ __forceinline__ float max8(float v) { v = fmaxf(v, dpp_f<0xB1>(v)); v = fmaxf(v, dpp_f<0x4E>(v)); v = fmaxf(v, dpp_f<0x141>(v)); return v; }

// sum_{j=1..K} min(V, d0 + j A) and sum_{j=1..K} max(-V, d0 - j A) in closed form (A >= 0: the terms are monotone in j, the first n of them
// unclamped; nh / nl = how many that would be without the limit K, formed once per axis).  A term next to the clamp is within rounding of
// either branch, so a floor() off by one moves the sum by rounding only.
__device__ __forceinline__ float reach_hi(float d0, float V, float A, float nh, int K)
{
    const float n1 = fminf(nh, (float)K);
    return n1 * d0 + 0.5f * A * n1 * (n1 + 1.f) + ((float)K - n1) * V;
}
__device__ __forceinline__ float reach_lo(float d0, float V, float A, float nl, int K)
{
    const float n1 = fminf(nl, (float)K);
    return n1 * d0 - 0.5f * A * n1 * (n1 + 1.f) - ((float)K - n1) * V;
}

__global__ __launch_bounds__(NB_THREADS) void lsc_neigh_build_kernel(NeighArgs a)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    const int qa = q >> 3, l = q & 7;
    const bool live = qa < a.N, seg = l < M;
    const int ql = live ? qa : 0, m = seg ? l : M - 1;
    // ---- everything this kernel reads, in one batch (what follows is arithmetic; the atomics come last)
    F3 p[6];
    load_segment(a.state, a.traj_prev, ql, m, a.planner_seq, a.dtf, p);
    const float *s = a.state + 9 * ql;
    float sv[9];
#pragma unroll
    for (int i = 0; i < 9; i++) sv[i] = s[i];
    float V[3], A[3];
#pragma unroll
    for (int k = 0; k < 3; k++) { V[k] = (float)(a.vmax[3 * ql + k] * a.hv_scale); A[k] = (float)(a.amax[3 * ql + k] * a.ha_scale); }
    const float r_obs = (float)a.radius_obs[ql];
    float t1[3] = {0.f, 0.f, 0.f};
    int ev = 0;
    if (a.checks) {
        const float *t = a.traj_prev + (size_t)ql * NV + NC;
        t1[0] = t[0]; t1[1] = t[SEGV]; t1[2] = t[2 * SEGV];
        ev = (int)a.ever[ql];
    }
    // ---- sphere around all predicted control points and the current position (priority candidates are found by position): any centre will
    // do as long as the radius is taken around the one that is stored
    float sx = 0.f, sy = 0.f, sz = 0.f;
    if (seg) {
#pragma unroll
        for (int i = 0; i < 6; i++) { sx += p[i].x; sy += p[i].y; sz += p[i].z; }
    }
    const float gx = sx * (1.f / 6.f), gy = sy * (1.f / 6.f), gz = sz * (1.f / 6.f);      // this lane's segment: centre of its own sphere
    if (l == 0) { sx += sv[0]; sy += sv[1]; sz += sv[2]; }
    const float inv_n = 1.f / (float)(6 * M + 1);
    const float fx = sum8(sx) * inv_n, fy = sum8(sy) * inv_n, fz = sum8(sz) * inv_n;
    float r2 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const float ex = p[i].x - fx, ey = p[i].y - fy, ez = p[i].z - fz;
        r2 = fmaxf(r2, ex * ex + ey * ey + ez * ez);
        const float hx = p[i].x - gx, hy = p[i].y - gy, hz = p[i].z - gz;
        s2 = fmaxf(s2, hx * hx + hy * hy + hz * hz);
    }
    if (!seg) { r2 = 0.f; s2 = 0.f; }
    if (l == 0) { const float ex = sv[0] - fx, ey = sv[1] - fy, ez = sv[2] - fz; r2 = fmaxf(r2, ex * ex + ey * ey + ez * ez); }
    const float rad = sqrtf(max8(r2)) * (1.f + 1e-5f) + 1e-5f;
    const float srad = sqrtf(s2) * (1.f + 1e-5f) + 1e-5f;
    const float smax = max8(seg ? srad : 0.f);
    // ---- B_m = max_i (|c_{0,2} - p_{m,i}| + reach radius of c_{m,i}) over the control points K = 5 m + i - 2 >= 1 steps from c_{0,2}, bounded by
    // (largest distance) + (largest reach radius); the state constants and the step limits as phase A of plan_agent forms them
    float c2[3], d0[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float hv = (float)a.hv_scale, ha = (float)a.ha_scale;
        d0[k] = sv[3 + k] * hv + sv[6 + k] * ha;                   // c_{0,2} - c_{0,1}
        c2[k] = sv[k] + 2.f * sv[3 + k] * hv + sv[6 + k] * ha;
        if (a.dim2 && k == 2) { c2[k] = (float)a.z2d; d0[k] = 0.f; }
    }
    float nh[3], nl[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float ia = A[k] > 0.f ? __frcp_rn(A[k]) : 0.f;
        nh[k] = A[k] > 0.f ? fmaxf(floorf((V[k] - d0[k]) * ia), 0.f) : (d0[k] <= V[k] ? 1e9f : 0.f);
        nl[k] = A[k] > 0.f ? fmaxf(floorf((d0[k] + V[k]) * ia), 0.f) : (d0[k] >= -V[k] ? 1e9f : 0.f);
    }
    float dm2 = 0.f, em2 = 0.f;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const int K = DEG * m + i - 2;                             // (K < 1: c_{0,0..2} are fixed by the state and carry no row)
        const float dx = c2[0] - p[i].x, dy = c2[1] - p[i].y, dz = c2[2] - p[i].z;
        float e2 = 0.f;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float e = fmaxf(fabsf(reach_lo(d0[k], V[k], A[k], nl[k], K)), fabsf(reach_hi(d0[k], V[k], A[k], nh[k], K)));
            e2 += e * e;
        }
        dm2 = fmaxf(dm2, K >= 1 ? dx * dx + dy * dy + dz * dz : 0.f);
        em2 = fmaxf(em2, K >= 1 ? e2 : 0.f);
    }
    // (+ 1e-5 of the coordinates' size: c_{0,2} is formed in float32 here and in float64 in phase A -- half an ulp of a coordinate apart)
    const float bm = (sqrtf(dm2) + sqrtf(em2)) * (1.f + 1e-5f) + 1e-3f + 1e-5f * (fabsf(c2[0]) + fabsf(c2[1]) + fabsf(c2[2]));
    // ---- disturbance check of this agent (off_plan of plan_agent: float32, no contraction): the persistent flag is set HERE, once
    bool slack = false;
    if (a.checks) {
#pragma clang fp contract(off)
        const float dx = t1[0] - sv[0], dy = t1[1] - sv[1], dz = t1[2] - sv[2];
        const float n2 = dx * dx + dy * dy + dz * dz;
        const bool nw = sqrt((double)n2) > a.reset_thr;
        if (nw && live && l == 0) a.ever[qa] = 1;
        slack = live && (nw || ev != 0);
    }
    if (live && seg) {
        reinterpret_cast<float4 *>(a.seg_bound)[(size_t)qa * M + m] = make_float4(gx, gy, gz, srad);
        a.reach[(size_t)qa * M + m] = bm;
    }
    // ---- what this agent adds, as an OBSTACLE, to the query radius of everybody else: its segment centres and its position lie within rad of
    // the centre it is filed under, and a segment's sphere test reaches 3 s rho_o + r_o further
    const float gr = (3.f * (float)a.sc_max * smax + r_obs + rad) * (1.f + 1e-5f) + 1e-5f;
    const int ix = cell_of((double)fx, a.inv_cell), iy = cell_of((double)fy, a.inv_cell), iz = cell_of((double)fz, a.inv_cell_z);
    // ---- one insertion into the grid, at the cell of the centre.  atomic max with (tag, 0) first: a bucket last touched in an older tick
    // counts as empty.  The counter's round trip (0.3 us on its own address, tools/microbench/launch_atomics.hip) is issued HERE and its slot
    // used at the very end, so that the reduction below runs under it.
    unsigned long long *const bucket = a.cells + 4 * (size_t)(cell_hash(ix, iy, iz) & a.hmask);
    unsigned slot = 0;
    if (live && l == 0) {
        reinterpret_cast<float4 *>(a.obs_bound)[qa] = make_float4(fx, fy, fz, rad);
        atomicMax(bucket, tagged(a.tag, 0u));
        slot = (unsigned)atomicAdd(bucket, 1ull);
    }
    // ---- swarm-wide maxima (query radius, bounding box of the occupied cells, "somebody is off its plan"): reduced over the workgroup's
    // agents first, and an atomic only when it would change what stands there (one atomic per agent on seven addresses was 40 us of a
    // 1024-agent tick)
    __shared__ unsigned red[8][NB_AGENTS];
    if (l == 0) {
        const int s_ = threadIdx.x >> 3;
        red[0][s_] = live ? __float_as_uint(gr) : 0u;
        red[1][s_] = live ? BIAS + (unsigned)ix : 0u; red[2][s_] = live ? BIAS + (unsigned)iy : 0u; red[3][s_] = live ? BIAS + (unsigned)iz : 0u;
        red[4][s_] = live ? BIAS - (unsigned)ix : 0u; red[5][s_] = live ? BIAS - (unsigned)iy : 0u; red[6][s_] = live ? BIAS - (unsigned)iz : 0u;
        red[7][s_] = slack ? 1u : 0u;
    }
    __syncthreads();
    {
        static_assert(NB_AGENTS == 32 && NB_THREADS == 8 * 32, "eight values, 32 agents: one 32-lane group per value");
        const int which = threadIdx.x >> 5, j = threadIdx.x & 31;
        unsigned v = red[which][j];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, o, 32));
        if (j == 0 && v != 0u) {
            unsigned long long *g = &a.glob[which == 0 ? G_RADIUS : (which == 7 ? G_SLACK : G_MAXX + (which - 1))];
            const unsigned long long mine = tagged(a.tag, v);
            if (__atomic_load_n(g, __ATOMIC_RELAXED) < mine) atomicMax(g, mine);
        }
    }
    if (live && l == 0) {
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
    if (a.prof && tid == 0) a.prof[8 * (size_t)al + 0] = (long long)__builtin_amdgcn_s_memrealtime();
    const int n_units = (a.N - 1) * M, words = (n_units + 31) >> 5;
    // ---- this agent's side of the tests, the swarm-wide numbers: one batch of loads.  float32 like the build kernel: every `need` below is
    // rounded up by 2e-5 of itself + 1e-4 m, far beyond what float32 loses on distances of metres
    const float r_a = (float)a.radius[qa], dw_a = (float)a.downwash[qa];
    float4 sb[M];
    float bm[M];
#pragma unroll
    for (int m = 0; m < M; m++) { sb[m] = reinterpret_cast<const float4 *>(a.seg_bound)[(size_t)qa * M + m]; bm[m] = a.reach[(size_t)qa * M + m]; }
    unsigned long long gl[G_COUNT];
#pragma unroll
    for (int i = 0; i < G_COUNT; i++) gl[i] = a.glob[i];
    const bool prio = a.goal_mode == 1;
    float px = 0.f, py = 0.f, pz = 0.f;
    if (prio) { const float *s = a.state + 9 * qa; px = s[0]; py = s[1]; pz = s[2]; }
    for (int w = tid; w < words; w += NQ) bitmap[w] = 0u;
    if (tid == 0) { qn = 0; pn = 0; rnk = 0; }
    float qrad[M];
    const float G = __uint_as_float(untag(gl[G_RADIUS], tag, 0u));
    const float scm = (float)a.sc_max;
#pragma unroll
    for (int m = 0; m < M; m++)
        qrad[m] = (scm * (2.f * bm[m] + 3.f * sb[m].w) + r_a + G) * (1.f + 4e-5f) + 1e-3f;      // reach of the query for segment m along x and y (>= the test's `need` + the obstacle's own radius)
    const double pthr = a.prio_thr * (1.0 + 1e-6) + 1e-9;      // (a candidate list may hold more than the rule needs, never less)
    const float pq = ((float)pthr + G) * (1.f + 4e-5f) + 1e-3f;
    // ---- cells the query boxes overlap, clamped to the cells that hold somebody
    int c0[3], c1[3];
    bool fail = false;
    {
        const int bmax[3] = {(int)(untag(gl[G_MAXX], tag, BIAS) - BIAS), (int)(untag(gl[G_MAXY], tag, BIAS) - BIAS), (int)(untag(gl[G_MAXZ], tag, BIAS) - BIAS)};
        const int bmin[3] = {(int)(BIAS - untag(gl[G_MINX], tag, BIAS)), (int)(BIAS - untag(gl[G_MINY], tag, BIAS)), (int)(BIAS - untag(gl[G_MINZ], tag, BIAS))};
        const float pp[3] = {px, py, pz};
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float sck = k == 2 ? (float)a.zscale : 1.f;
            const double inv = k == 2 ? a.inv_cell_z : a.inv_cell;
            float lo = 3e38f, hi = -3e38f;
#pragma unroll
            for (int m = 0; m < M; m++) {
                const float c = k == 0 ? sb[m].x : (k == 1 ? sb[m].y : sb[m].z);
                lo = fminf(lo, c - qrad[m] * sck); hi = fmaxf(hi, c + qrad[m] * sck);
            }
            if (prio) { lo = fminf(lo, pp[k] - pq); hi = fmaxf(hi, pp[k] + pq); }
            // (the box is widened by 1e-5 of the coordinate: the float32 sums above against the float64 product that filed the obstacle)
            c0[k] = max(cell_of((double)(lo - 1e-5f * fabsf(lo)), inv), bmin[k]); c1[k] = min(cell_of((double)(hi + 1e-5f * fabsf(hi)), inv), bmax[k]);
            if (!(lo <= hi)) fail = true;                  // (NaN inputs: no list, the agent's own phase B decides)
        }
    }
    const int nx = c1[0] - c0[0] + 1, ny = c1[1] - c0[1] + 1, nz = c1[2] - c0[2] + 1;
    long long ncell = (nx > 0 && ny > 0 && nz > 0) ? (long long)nx * ny * nz : 0;
    if (ncell > MAX_CELLS) { fail = true; ncell = 0; }
    const unsigned ovn = untag(gl[G_OVF], tag, 0u);
    if (ovn > (unsigned)a.ovf_cap) fail = true;            // somebody is in no bucket and in no overflow slot
    __syncthreads();
    if (a.prof && tid == 0) a.prof[8 * (size_t)al + 1] = (long long)__builtin_amdgcn_s_memrealtime();
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
    if (a.prof && tid == 0) a.prof[8 * (size_t)al + 2] = (long long)__builtin_amdgcn_s_memrealtime();
    const int nq = qn;
    if (nq > QUEUE_CAP) fail = true;
    // ---- stage 2: one lane per candidate obstacle, M sphere tests (+ the distance of the priority rule)
    for (int ci = tid; ci < nq && ci < QUEUE_CAP; ci += NQ) {
        const int o = (int)queue[ci];
        if (o == qa || o >= a.N) continue;
        const float r_o = (float)a.radius_obs[o], dw_o = (float)a.downwash_obs[o];
        float4 so[M];
#pragma unroll
        for (int m = 0; m < M; m++) so[m] = reinterpret_cast<const float4 *>(a.seg_bound)[(size_t)o * M + m];
        float ox = 0.f, oy = 0.f, oz = 0.f;
        if (prio) { const float *s = a.state + 9 * o; ox = s[0]; oy = s[1]; oz = s[2]; }
        const float downwash = (dw_a * r_a + dw_o * r_o) / (r_a + r_o);
        const float idw = 1.f / downwash, sc = fmaxf(1.f, idw);
        const int oi = o < qa ? o : o - 1;
#pragma unroll
        for (int m = 0; m < M; m++) {
            const float dx = sb[m].x - so[m].x, dy = sb[m].y - so[m].y, dz = (sb[m].z - so[m].z) * idw;
            const float need = (sc * (2.f * bm[m] + 3.f * (sb[m].w + so[m].w)) + (r_a + r_o)) * (1.f + 2e-5f) + 5e-4f;      // (2e-4 of the test itself + rounding)
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
    if (a.prof && tid == 0) a.prof[8 * (size_t)al + 3] = (long long)__builtin_amdgcn_s_memrealtime();
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
    if (a.prof && tid == 0) { a.prof[8 * (size_t)al + 4] = (long long)__builtin_amdgcn_s_memrealtime(); a.prof[8 * (size_t)al + 5] = (long long)nq | ((long long)ovn << 32); a.prof[8 * (size_t)al + 6] = ncell; a.prof[8 * (size_t)al + 7] = total; }
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
