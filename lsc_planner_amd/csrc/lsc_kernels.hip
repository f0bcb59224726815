// lsc_kernels.hip -- gfx950 kernels of the replanning tick (one workgroup of 256 lanes per agent).
//
//   lsc_plan_kernel      : prediction shift -> LSC build (GJK, pruning, bucketing) -> reduced-space
//                          primal-dual interior point QP -> float32 trajectory      (the hot path)
//   lsc_sweep_kernel     : dense LSC sweep for every ordered (agent, obstacle, segment) -> HBM
//   lsc_propagate_kernel : next ideal state from the planned trajectory at t = dt
//   lsc_gjk_kernel       : batched GJK test hook
//
// Reference behaviour being replaced (file:line in /root/reference):
//   TrajPlanner::obstaclePredictionWithPrevSol / initialTrajPlanningPrevSol  src/traj_planner.cpp:829-864, 997-1016
//   TrajPlanner::generateLSC                                                 src/traj_planner.cpp:1310-1407
//   TrajOptimizer::populatebyrow + IloCplex::solve                           src/traj_optimizer.cpp:261-539, 76-96
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "lsc_gjk.hpp"
#include "lsc_model.hpp"
#include "lsc_kernels.h"

namespace lsc {

constexpr int NT = 256;
constexpr int NWAVE = NT / 64;

// ---------------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_min(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
    return v;
}

// Predicted / initial control points of agent q for segment m.
//   planner_seq < 2 : pos + vel * m_intp * dt   (float32, src/traj_planner.cpp:699-712, 1030-1037)
//   else            : previous plan shifted by one segment, last segment = 6 x previous end point
__device__ __forceinline__ void load_segment(const float *__restrict__ state, const float *__restrict__ traj_prev, int q,
                                             int m, int planner_seq, float dtf, F3 out[6])
{
    if (planner_seq < 2) {
        const float *s = state + 9 * q;
#pragma unroll
        for (int i = 0; i < 6; i++) {
            float mi = (float)((double)m + (double)i / (double)DEG);
            float ax = (s[3] * mi) * dtf, ay = (s[4] * mi) * dtf, az = (s[5] * mi) * dtf;
            out[i] = F3{s[0] + ax, s[1] + ay, s[2] + az};
        }
    } else {
        const float *t = traj_prev + (size_t)q * NV;
        if (m < M - 1) {
#pragma unroll
            for (int i = 0; i < 6; i++) {
                int c = (m + 1) * NC + i;
                out[i] = F3{t[c], t[SEGV + c], t[2 * SEGV + c]};
            }
        } else {
            int c = (M - 1) * NC + DEG;
            F3 e = F3{t[c], t[SEGV + c], t[2 * SEGV + c]};
#pragma unroll
            for (int i = 0; i < 6; i++) out[i] = e;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Dense LSC sweep: one lane per (agent, obstacle, segment).  Output is what CollisionConstraints holds
// after generateLSC (normal shared by the 6 rows of a segment, six margins).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lsc_sweep_kernel(SweepArgs a)
{
    const int n_obs = a.N - 1;
    const long total = (long)a.count * n_obs * M;
    for (long u = (long)blockIdx.x * blockDim.x + threadIdx.x; u < total; u += (long)gridDim.x * blockDim.x) {
        const int m = (int)(u % M);
        const int oi = (int)((u / M) % n_obs);
        const int al = (int)(u / ((long)M * n_obs));
        const int qi = a.first + al;
        const int qj = oi < qi ? oi : oi + 1;
        F3 pa[6], po[6];
        load_segment(a.state, a.traj_prev, qi, m, a.planner_seq, a.dtf, pa);
        load_segment(a.state, a.traj_prev, qj, m, a.planner_seq, a.dtf, po);
        const double r_a = a.radius[qi], r_o = a.radius_obs[qj];
        const double downwash = (a.downwash[qi] * r_a + a.downwash_obs[qj] * r_o) / (r_a + r_o);
        F3 n;
        double d[6];
        lsc_segment(pa, po, downwash, r_o + r_a, n, d);
        float *on = a.out_normal + u * 3;
        on[0] = n.x; on[1] = n.y; on[2] = n.z;
        double *od = a.out_d + u * 6;
#pragma unroll
        for (int i = 0; i < 6; i++) od[i] = d[i];
    }
}

__global__ __launch_bounds__(256) void lsc_gjk_kernel(const double *__restrict__ pts, int count, double *__restrict__ v,
                                                      double *__restrict__ dist)
{
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= count) return;
    const double *p = pts + (size_t)c * 18;
    D3 q[6];
#pragma unroll
    for (int i = 0; i < 6; i++) q[i] = D3{p[3 * i], p[3 * i + 1], p[3 * i + 2]};
    D3 w;
    int nv;
    dist[c] = gjk_origin_hull6(q[0], q[1], q[2], q[3], q[4], q[5], w, nv);
    v[3 * c] = w.x; v[3 * c + 1] = w.y; v[3 * c + 2] = w.z;
}

// getStateFromControlPoints at t = dt (include/polynomial.hpp:63-97): segment 1, local time 0.
__global__ void lsc_propagate_kernel(const float *__restrict__ traj, float *__restrict__ state, int N, float finv)
{
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * 3) return;
    int q = idx / 3, k = idx % 3;
    const float *c1 = traj + (size_t)q * NV + k * SEGV + NC;
    const float fn = (float)DEG, fn1 = (float)(DEG - 1);
    float v0 = ((c1[1] - c1[0]) * fn) * finv;
    float v1 = ((c1[2] - c1[1]) * fn) * finv;
    float a0 = ((v1 - v0) * fn1) * finv;
    state[9 * q + k] = c1[0];
    state[9 * q + 3 + k] = v0;
    state[9 * q + 6 + k] = a0;
}

// ---------------------------------------------------------------------------------------------------
// The per-agent planning kernel
// ---------------------------------------------------------------------------------------------------
constexpr int NB = 27;  // control points that carry LSC rows: all but (m=0, i<3)

struct Smem {
    // PDIP vectors
    double x[96], dx[96];       // control points (axis-major, 90 used) and their step
    double gx[96], gz[96];      // x-space gradients: cost + sum a_r v_r ; cost + sum a_r z_r
    double y[40], dy[40], rhs[40];
    double W[W_SIZE];           // x-space Hessian weights (see lsc_model.hpp)
    double Tv[NCP * 3], Tz[NCP * 3];
    double K[NY * KLD];         // reduced Hessian (lower band), overwritten by its Cholesky factor
    double red[3][NWAVE];
    double sc[8];               // broadcast scalars
    // axis rows: slot = type*90 + k*30 + t ; type 0 x<=hi, 1 -x<=-lo, 2/3 +-velocity, 4/5 +-acceleration
    double as_[AXROWS], az[AXROWS], at1[AXROWS], at2[AXROWS], ah[AXROWS];
    unsigned char avalid[AXROWS];
    // agent constants
    double s0[3][3];            // c_{0,0..2} per axis
    double lo[3][M], hi[3][M];  // bounds per axis and segment (world box, intersected with the SFC)
    double goal[3];
    float pinit[NV];            // own initial trajectory (float32)
    int cnt[32];                // rows per control-point bucket
    int wcnt[NWAVE][32];
    int tseg;                   // terminal segments
    int flag;                   // capacity overflow
    alignas(8) uint32_t dyn[2]; // dynamic part starts here: terms, entry table, kconst, LSC rows
};

__device__ __forceinline__ double ax_row(const double *x, int type, int k, int t)
{
    const double *xk = x + k * SEGV;
    switch (type) {
    case 0: return xk[t];
    case 1: return -xk[t];
    case 2: return xk[t + 1] - xk[t];
    case 3: return -(xk[t + 1] - xk[t]);
    case 4: return xk[t + 2] - 2.0 * xk[t + 1] + xk[t];
    default: return -(xk[t + 2] - 2.0 * xk[t + 1] + xk[t]);
    }
}

// value of lane L (compile-time) broadcast to the wave: two v_readlane_b32, no LDS round trip
template <int L>
__device__ __forceinline__ double bcast_lane(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, L);
    hi = __builtin_amdgcn_readlane(hi, L);
    return __hiloint2double(hi, lo);
}

// Banded Cholesky of the 39x39 matrix in LDS (lower band, leading dimension KLD), executed by wave 0:
// lane i owns row i in registers; finished columns are published through LDS and re-read as broadcasts.
template <int J>
__device__ __forceinline__ void chol_column(double *K, double (&row)[NY], double (&invd)[NY], int lane, bool act, bool &ok)
{
    constexpr int k0 = (J - BAND) > 0 ? (J - BAND) : 0;
    double acc = row[J];
    double dj = K[J * KLD + J];
#pragma unroll
    for (int k = k0; k < J; k++) {
        double ljk = K[J * KLD + k];
        acc -= row[k] * ljk;
        dj -= ljk * ljk;
    }
    if (!(dj > 0.0)) ok = false;
    double inv = 1.0 / sqrt(dj);
    invd[J] = inv;
    double l = acc * inv;
    if (lane == J) l = dj * inv;
    if (lane < J) l = 0.0;
    row[J] = l;
    if (act && lane >= J) K[lane * KLD + J] = l;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if constexpr (J + 1 < NY) chol_column<J + 1>(K, row, invd, lane, act, ok);
}

template <int J>
__device__ __forceinline__ void fwd_step(const double (&row)[NY], const double (&invd)[NY], double &b, int lane)
{
    double bj = bcast_lane<J>(b) * invd[J];
    if (lane == J) b = bj;
    else if (lane > J) b -= row[J] * bj;
    if constexpr (J + 1 < NY) fwd_step<J + 1>(row, invd, b, lane);
}
template <int I>
__device__ __forceinline__ void bwd_step(const double (&col)[NY], const double (&invd)[NY], double &b, int lane)
{
    double xi = bcast_lane<I>(b) * invd[I];
    if (lane == I) b = xi;
    else if (lane < I) b -= col[I] * xi;
    if constexpr (I > 0) bwd_step<I - 1>(col, invd, b, lane);
}

// phase stamps (PROF variant only): cycles of lane 0 spent per phase, accumulated per agent
enum { PH_SETUP = 0, PH_LSC, PH_INIT, PH_P1, PH_REDUCE, PH_ASSEMBLE, PH_FACTOR, PH_SOLVE, PH_P2, PH_P3, PH_P45, PH_OUT, PH_COUNT };

template <bool PROF>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(1, 1))) void lsc_plan_kernel(PlanArgs a)
{
    long long t_last = 0;
    long long t_acc[PH_COUNT];
    if constexpr (PROF) {
#pragma unroll
        for (int i = 0; i < PH_COUNT; i++) t_acc[i] = 0;
        t_last = wall_clock64();
    }
    auto stamp = [&](int ph) {
        if constexpr (PROF) {
            long long now = wall_clock64();
            t_acc[ph] += now - t_last;
            t_last = now;
        }
    };
    extern __shared__ __align__(16) unsigned char smem_raw[];
    Smem &S = *reinterpret_cast<Smem *>(smem_raw);
    const Model &md = *a.model;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qi = a.first + blockIdx.x;
    const int N = a.N, n_obs = N - 1, CAP = a.cap;
    const int CS = (CAP & 1) ? CAP : CAP + 1;   // odd bucket stride: spreads the buckets over LDS banks
    const int R = NB * CS;
    const int n_terms = md.n_terms, n_entries = md.n_entries;

    // dynamic LDS carve-up
    uint32_t *terms = S.dyn;                                        // [n_terms]
    uint32_t *ent = terms + ((n_terms + 1) & ~1);                   // [n_entries+1][2] : (gi<<16|gj), first term
    double *kconst = reinterpret_cast<double *>(ent + 2 * n_entries + 2);
    double *rrhs = kconst + n_entries;
    double *rs = rrhs + R, *rz = rs + R, *rt1 = rz + R, *rt2 = rt1 + R;
    float *rn = reinterpret_cast<float *>(rt2 + R);                 // [3][R]
    unsigned char *rcp = reinterpret_cast<unsigned char *>(rn + 3 * R);  // control point of the row, 255 = empty

    // ------------------------------------------------------------------ phase A: agent constants
    for (int i = tid; i < n_terms; i += NT) terms[i] = a.terms[i];
    for (int i = tid; i < 2 * n_entries + 2; i += NT) ent[i] = a.entries[i];
    for (int i = tid; i < R; i += NT) rcp[i] = 255;
    if (tid < 32) S.cnt[tid] = 0;
    if (tid == 0) S.flag = 0;
    const float dtf = (float)md.dt;
    if (tid < NV) {
        const int k = tid / SEGV, c = tid % SEGV, m = c / NC, i = c % NC;
        float val;
        if (a.planner_seq < 2) {
            const float *s = a.state + 9 * qi;
            float mi = (float)((double)m + (double)i / (double)DEG);
            val = s[k] + (s[3 + k] * mi) * dtf;
        } else {
            const float *t = a.traj_prev + (size_t)qi * NV + k * SEGV;
            val = (m < M - 1) ? t[(m + 1) * NC + i] : t[(M - 1) * NC + DEG];
        }
        S.pinit[tid] = val;
    }
    if (tid < 3) {
        const int k = tid;
        const float *s = a.state + 9 * qi;
        double c0 = (double)s[k];
        double c1 = c0 + (double)s[3 + k] * md.hv_scale;
        double c2 = (double)s[6 + k] * md.ha_scale + 2.0 * c1 - c0;
        S.s0[k][0] = c0; S.s0[k][1] = c1; S.s0[k][2] = c2;
        S.goal[k] = (double)a.goal[3 * qi + k];
        for (int m = 0; m < M; m++) {
            double lo = (double)md.world_min[k], hi = (double)md.world_max[k];
            if (md.use_sfc && a.sfc) {
                const float *b = a.sfc + ((size_t)qi * M + m) * 6;
                lo = fmax(lo, (double)b[k]);
                hi = fmin(hi, (double)b[3 + k]);
            }
            S.lo[k][m] = lo; S.hi[k][m] = hi;
        }
    }
    if (tid == 0) {
        // getTerminalSegments (src/traj_optimizer.cpp:541-548), float32 norm like octomath
        const float *s = a.state + 9 * qi;
        const float *g = a.goal + 3 * qi;
        float dxg = g[0] - s[0], dyg = g[1] - s[1], dzg = g[2] - s[2];
        float n2 = dxg * dxg + dyg * dyg + dzg * dzg;
        double flight = sqrt((double)n2) / a.vnom[qi];
        int T = (int)((M * md.dt - flight + 1e-9) / md.dt);
        S.tseg = T > 1 ? T : 1;
    }
    // per-lane constants kept in registers for the whole solve
    int xk = 0, xt = 0, xn = 0, xg0 = 0, xg1 = 0, xg2 = 0;     // lanes < 90: x_t = sum xc*y[xg]
    double xc0 = 0, xc1 = 0, xc2 = 0, qr[NC] = {0, 0, 0, 0, 0, 0};
    if (tid < NV) {
        xk = tid / SEGV; xt = tid % SEGV;
        xn = md.x_n[xt];
        xg0 = yglob(xk, md.x_i[xt][0]); xg1 = yglob(xk, md.x_i[xt][1]); xg2 = yglob(xk, md.x_i[xt][2]);
        xc0 = md.x_c[xt][0]; xc1 = md.x_c[xt][1]; xc2 = md.x_c[xt][2];
        if (xn < 1) xc0 = 0; if (xn < 2) xc1 = 0; if (xn < 3) xc2 = 0;
#pragma unroll
        for (int j = 0; j < NC; j++) qr[j] = md.Qh[(xt % NC) * NC + j];
    }
    int yk = 0, yo0 = 0, yo1 = 0, yo2 = 0, yo3 = 0;            // lanes < 39: gy = sum yc * gx[yo]
    int yp0 = 0, yp1 = 0, yp2 = 0, yp3 = 0;
    double yc0 = 0, yc1 = 0, yc2 = 0, yc3 = 0;
    if (tid < NY) {
        const int g = tid;
        yk = g < 36 ? (g % 9) / 3 : g - 36;
        const int va = g < 36 ? (g / 9) * 3 + (g % 3) : 12;
        const int n = md.t_n[va];
        yc0 = n > 0 ? md.t_c[va][0] : 0.0; yc1 = n > 1 ? md.t_c[va][1] : 0.0;
        yc2 = n > 2 ? md.t_c[va][2] : 0.0; yc3 = n > 3 ? md.t_c[va][3] : 0.0;
        const int t0 = n > 0 ? md.t_t[va][0] : 0, t1 = n > 1 ? md.t_t[va][1] : 0;
        const int t2 = n > 2 ? md.t_t[va][2] : 0, t3 = n > 3 ? md.t_t[va][3] : 0;
        yo0 = yk * SEGV + t0; yo1 = yk * SEGV + t1; yo2 = yk * SEGV + t2; yo3 = yk * SEGV + t3;
        yp0 = t0 * 3 + yk; yp1 = t1 * 3 + yk; yp2 = t2 * 3 + yk; yp3 = t3 * 3 + yk;
    }
    __syncthreads();
    const bool xterm = (tid < NV) && (xt % NC == DEG) && (xt / NC >= M - S.tseg);
    // constant part of every Hessian entry: cost Hessian (same axis) + terminal weight on c_{m,5}
    for (int e = tid; e < n_entries; e += NT) {
        const uint32_t id = ent[2 * e];
        const int gi = id >> 16, gj = id & 0xffff;
        const int ki = gi < 36 ? (gi % 9) / 3 : gi - 36, kj = gj < 36 ? (gj % 9) / 3 : gj - 36;
        const int va = gi < 36 ? (gi / 9) * 3 + (gi % 3) : 12, vb = gj < 36 ? (gj / 9) * 3 + (gj % 3) : 12;
        double v = 0.0;
        if (ki == kj) {
            v = md.Hc[va * NYA + vb];
            if (va == vb) {
                const int mterm = va == 12 ? 4 : ((va % 3) == 2 ? va / 3 : -1);
                if (mterm >= M - S.tseg) v += 2.0 * md.w_t;
            }
        }
        kconst[e] = v;
    }

    stamp(PH_SETUP);
    // ------------------------------------------------------------------ phase B: LSC rows
    // unit = (obstacle oi, segment m); rows that cannot be active inside the reachable box are dropped
    // (redundant constraints: removing them does not change the feasible set, hence not the optimum).
    {
        const double r_a = a.radius[qi], dw_a = a.downwash[qi];
        const double dlx = a.vmax[3 * qi] * md.hv_scale, dly = a.vmax[3 * qi + 1] * md.hv_scale,
                     dlz = a.vmax[3 * qi + 2] * md.hv_scale;   // largest step between consecutive control points
        const int n_units = n_obs * M;
        const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        for (int base = 0; base < n_units; base += NT) {
            const int u = base + tid;
            const bool live = u < n_units;
            const int oi = live ? u / M : 0, m = live ? u % M : 0;
            const int qj = oi < qi ? oi : oi + 1;
            F3 nrm = F3{0.f, 0.f, 0.f};
            double rhs[6];
            bool actv[6];
#pragma unroll
            for (int i = 0; i < 6; i++) { rhs[i] = 0.0; actv[i] = false; }
            if (live) {
                F3 pa[6], po[6];
#pragma unroll
                for (int i = 0; i < 6; i++) {
                    int c = m * NC + i;
                    pa[i] = F3{S.pinit[c], S.pinit[SEGV + c], S.pinit[2 * SEGV + c]};
                }
                load_segment(a.state, a.traj_prev, qj, m, a.planner_seq, dtf, po);
                const double r_o = a.radius_obs[qj];
                const double downwash = (dw_a * r_a + a.downwash_obs[qj] * r_o) / (r_a + r_o);
                double d[6];
                lsc_segment(pa, po, downwash, r_o + r_a, nrm, d);
                if (a.out_normal) {
                    size_t o = ((size_t)blockIdx.x * n_obs + oi) * M + m;
                    a.out_normal[o * 3] = nrm.x; a.out_normal[o * 3 + 1] = nrm.y; a.out_normal[o * 3 + 2] = nrm.z;
#pragma unroll
                    for (int i = 0; i < 6; i++) a.out_d[o * 6 + i] = d[i];
                }
                const double nx = (double)nrm.x, ny = (double)nrm.y, nz = (double)nrm.z;
                const double reach1 = fabs(nx) * dlx + fabs(ny) * dly + fabs(nz) * dlz;
                const double centre = nx * S.s0[0][2] + ny * S.s0[1][2] + nz * S.s0[2][2];
#pragma unroll
                for (int i = 0; i < 6; i++) {
                    double r = d[i];
                    r += nx * (double)po[i].x;
                    r += ny * (double)po[i].y;
                    r += nz * (double)po[i].z;
                    rhs[i] = r;
                    bool on = !(m == 0 && i < 3);
                    if (on && md.prune) {
                        // every feasible c_{m,i} lies within (5m+i-2) velocity-limited steps of c_{0,2}
                        double worst = centre - (double)(5 * m + i - 2) * reach1;
                        if (worst >= r + 1e-6) on = false;
                    }
                    actv[i] = on;
                }
            }
            // deterministic bucketing by control point: rows of a bucket stay in increasing-obstacle order
            int rank[6];
#pragma unroll
            for (int i = 0; i < 6; i++) rank[i] = 0;
#pragma unroll
            for (int mm = 0; mm < M; mm++) {
#pragma unroll
                for (int i = 0; i < 6; i++) {
                    unsigned long long mask = __ballot(live && m == mm && actv[i]);
                    if (m == mm) rank[i] = __popcll(mask & lt_mask);
                    if (lane == 0) S.wcnt[wave][mm * NC + i] = __popcll(mask);
                }
            }
            __syncthreads();
            if (live) {
#pragma unroll
                for (int i = 0; i < 6; i++) {
                    if (!actv[i]) continue;
                    const int cp = m * NC + i;
                    int pos = S.cnt[cp] + rank[i];
                    for (int w = 0; w < wave; w++) pos += S.wcnt[w][cp];
                    if (pos < CAP) {
                        const int r = (cp - 3) * CS + pos;
                        rn[r] = nrm.x; rn[R + r] = nrm.y; rn[2 * R + r] = nrm.z;
                        rrhs[r] = rhs[i];
                        rcp[r] = (unsigned char)cp;
                    } else {
                        S.flag = 1;
                    }
                }
            }
            __syncthreads();
            if (tid < NCP) {
                int c = S.cnt[tid];
                for (int w = 0; w < NWAVE; w++) c += S.wcnt[w][tid];
                S.cnt[tid] = c;
            }
            __syncthreads();
        }
    }
    if (tid < NCP && S.cnt[tid] > CAP) S.cnt[tid] = CAP;
    stamp(PH_LSC);

    // ------------------------------------------------------------------ phase C: interior point
    for (int sl = tid; sl < AXROWS; sl += NT) {
        const int type = sl / NV, kt = sl % NV, k = kt / SEGV, t = kt % SEGV, m = t / NC, i = t % NC;
        bool valid;
        double h;
        if (type < 2) { valid = !(m == 0 && i < 3); h = type == 0 ? S.hi[k][m] : -S.lo[k][m]; }
        else if (type < 4) { valid = i <= 4 && !(m == 0 && i < 2); h = a.vmax[3 * qi + k] * md.hv_scale; }
        else { valid = i <= 3 && !(m == 0 && i == 0); h = a.amax[3 * qi + k] * md.ha_scale; }
        S.avalid[sl] = valid ? 1 : 0;
        S.ah[sl] = h;
        S.as_[sl] = 1.0; S.az[sl] = 0.0; S.at1[sl] = 0.0; S.at2[sl] = 0.0;
    }
    if (tid < 40) { S.y[tid] = 0.0; S.dy[tid] = 0.0; }
    for (int i = tid; i < NY * KLD; i += NT) S.K[i] = 0.0;
    for (int i = tid; i < NCP * 3; i += NT) { S.Tv[i] = 0.0; S.Tz[i] = 0.0; }
    for (int i = tid; i < W_SIZE; i += NT) S.W[i] = 0.0;
    __syncthreads();
    const bool overflow = S.flag != 0;

    // x from y : x_t = sum coef * y_glob  (+ state constants for t < 3)
    auto compute_x = [&](const double *yv, double *xv, bool with_const) {
        if (tid < NV) {
            double v;
            if (xt < 3) v = with_const ? S.s0[xk][xt] : 0.0;
            else v = xc0 * yv[xg0] + xc1 * yv[xg1] + xc2 * yv[xg2];
            xv[tid] = v;
        }
    };
    // cost gradient in x-space for this lane's variable: 2 w_c Q x within the segment + terminal term
    auto cost_grad = [&]() -> double {
        const double *xs = S.x + xk * SEGV + (xt / NC) * NC;
        double g = 0.0;
#pragma unroll
        for (int j = 0; j < NC; j++) g += qr[j] * xs[j];
        return g;
    };
    // block reductions: op 0 sum, 1 max, 2 min ; results in S.sc[0..2]
    auto block_reduce3 = [&](double v0, double v1, double v2, int op0, int op1, int op2) {
        double r0 = op0 == 0 ? wave_sum(v0) : (op0 == 1 ? wave_max(v0) : wave_min(v0));
        double r1 = op1 == 0 ? wave_sum(v1) : (op1 == 1 ? wave_max(v1) : wave_min(v1));
        double r2 = op2 == 0 ? wave_sum(v2) : (op2 == 1 ? wave_max(v2) : wave_min(v2));
        if (lane == 0) { S.red[0][wave] = r0; S.red[1][wave] = r1; S.red[2][wave] = r2; }
        __syncthreads();
        if (tid == 0) {
            double t0 = S.red[0][0], t1 = S.red[1][0], t2 = S.red[2][0];
            for (int w = 1; w < NWAVE; w++) {
                t0 = op0 == 0 ? t0 + S.red[0][w] : (op0 == 1 ? fmax(t0, S.red[0][w]) : fmin(t0, S.red[0][w]));
                t1 = op1 == 0 ? t1 + S.red[1][w] : (op1 == 1 ? fmax(t1, S.red[1][w]) : fmin(t1, S.red[1][w]));
                t2 = op2 == 0 ? t2 + S.red[2][w] : (op2 == 1 ? fmax(t2, S.red[2][w]) : fmin(t2, S.red[2][w]));
            }
            S.sc[0] = t0; S.sc[1] = t1; S.sc[2] = t2;
        }
        __syncthreads();
    };

    // Reduction of the per-row values (w = z*t1 or 1, v = t2) into x-space weights and gradients.
    auto reduce_rows = [&](bool with_w, bool unit_w) {
        if (tid < NV) {
            const int k = xk, t = xt, i = t % NC;
            const int b = k * SEGV + t;
            auto wof = [&](int type, int tt) -> double {
                int sl = type * NV + k * SEGV + tt;
                if (!S.avalid[sl]) return 0.0;
                return unit_w ? 1.0 : S.az[sl] * S.at1[sl];
            };
            auto vof = [&](int type, int tt) -> double { int sl = type * NV + k * SEGV + tt; return S.avalid[sl] ? S.at2[sl] : 0.0; };
            auto zof = [&](int type, int tt) -> double { int sl = type * NV + k * SEGV + tt; return S.avalid[sl] ? S.az[sl] : 0.0; };
            double g = 0.0, gzv = 0.0;
            g += vof(0, t) - vof(1, t);
            gzv += zof(0, t) - zof(1, t);
            g += -vof(2, t) + vof(3, t);
            gzv += -zof(2, t) + zof(3, t);
            g += vof(4, t) - vof(5, t);
            gzv += zof(4, t) - zof(5, t);
            if (i >= 1) {
                g += vof(2, t - 1) - vof(3, t - 1) - 2.0 * (vof(4, t - 1) - vof(5, t - 1));
                gzv += zof(2, t - 1) - zof(3, t - 1) - 2.0 * (zof(4, t - 1) - zof(5, t - 1));
            }
            if (i >= 2) {
                g += vof(4, t - 2) - vof(5, t - 2);
                gzv += zof(4, t - 2) - zof(5, t - 2);
            }
            double cg = cost_grad();
            if (xterm) cg += 2.0 * md.w_t * (S.x[b] - S.goal[k]);
            S.gx[b] = cg + g;
            S.gz[b] = cg + gzv;
            if (with_w) {
                double wB = wof(0, t) + wof(1, t);
                double wV0 = wof(2, t) + wof(3, t);
                double wA0 = wof(4, t) + wof(5, t);
                double wV1 = i >= 1 ? wof(2, t - 1) + wof(3, t - 1) : 0.0;
                double wA1 = i >= 1 ? wof(4, t - 1) + wof(5, t - 1) : 0.0;
                double wA2 = i >= 2 ? wof(4, t - 2) + wof(5, t - 2) : 0.0;
                S.W[W_D + b] = wB + wV0 + wV1 + wA0 + 4.0 * wA1 + wA2;
                S.W[W_1 + b] = -wV0 - 2.0 * wA0 - 2.0 * wA1;
                S.W[W_2 + b] = wA0;
            }
        }
        // LSC buckets: unit (bucket, c): c 0..5 -> sum w n n^T, 6..8 -> -sum v n, 9..11 -> -sum z n  (a_r = -n)
        for (int u = tid; u < NB * 12; u += NT) {
            const int bkt = u / 12, c = u % 12, cp = bkt + 3;
            const int cnt = S.cnt[cp];
            if (c < 6 && !with_w) continue;
            double acc = 0.0;
            const int r0 = bkt * CS;
            int ia = 0, ib = 0;
            if (c < 6) { ia = c < 3 ? 0 : (c < 5 ? 1 : 2); ib = c < 3 ? c : (c < 5 ? c - 2 : 2); }
            const float *na = rn + ia * R + r0, *nb = rn + ib * R + r0, *nc = rn + ((c >= 9 ? c - 9 : (c >= 6 ? c - 6 : 0))) * R + r0;
            if (c < 6) {
                for (int j = 0; j < cnt; j++) {
                    double w = unit_w ? 1.0 : rz[r0 + j] * rt1[r0 + j];
                    acc += w * (double)na[j] * (double)nb[j];
                }
                S.W[W_S + cp * 6 + c] = acc;
            } else if (c < 9) {
                for (int j = 0; j < cnt; j++) acc += rt2[r0 + j] * (double)nc[j];
                S.Tv[cp * 3 + (c - 6)] = -acc;
            } else {
                for (int j = 0; j < cnt; j++) acc += rz[r0 + j] * (double)nc[j];
                S.Tz[cp * 3 + (c - 9)] = -acc;
            }
        }
        __syncthreads();
    };

    // K (lower band) from W ;  rhs = -Z^T (gx + Tv) ;  stationarity residual Z^T (gz + Tz) -> S.dy (scratch)
    auto assemble = [&](bool with_k) {
        if (with_k) {
            for (int e = tid; e < n_entries; e += NT) {
                const uint32_t id = ent[2 * e], t0 = ent[2 * e + 1], t1 = ent[2 * e + 3];
                double v = kconst[e];
                for (uint32_t q = t0; q < t1; q++) {
                    const uint32_t tm = terms[q];
                    v += (double)((int)(tm & 0xff) - 128) * S.W[(tm >> 8) & 0x3ff];
                }
                S.K[(id >> 16) * KLD + (id & 0xffff)] = v;
            }
        }
        if (tid < NY) {
            double r = yc0 * (S.gx[yo0] + S.Tv[yp0]) + yc1 * (S.gx[yo1] + S.Tv[yp1]) + yc2 * (S.gx[yo2] + S.Tv[yp2]) +
                       yc3 * (S.gx[yo3] + S.Tv[yp3]);
            double rd = yc0 * (S.gz[yo0] + S.Tz[yp0]) + yc1 * (S.gz[yo1] + S.Tz[yp1]) + yc2 * (S.gz[yo2] + S.Tz[yp2]) +
                        yc3 * (S.gz[yo3] + S.Tz[yp3]);
            S.rhs[tid] = -r;
            S.dy[tid] = rd;
        }
        __syncthreads();
    };

    double lrow[NY], lcol[NY], linvd[NY];  // wave 0: Cholesky factor rows / columns / inverse pivots
#pragma unroll
    for (int j = 0; j < NY; j++) { lrow[j] = 0.0; lcol[j] = 0.0; linvd[j] = 0.0; }

    auto factor = [&]() -> bool {
        if (wave == 0) {
            const bool act = lane < NY;
#pragma unroll
            for (int j = 0; j < NY; j++) lrow[j] = (act && j <= lane && lane - j <= BAND) ? S.K[lane * KLD + j] : 0.0;
            bool ok = true;
            chol_column<0>(S.K, lrow, linvd, lane, act, ok);
            // column of L owned by this lane (L[i][lane], i >= lane) for the backward sweep
#pragma unroll
            for (int i = 0; i < NY; i++) lcol[i] = (act && i >= lane && i - lane <= BAND) ? S.K[i * KLD + lane] : 0.0;
            if (lane == 0) S.sc[7] = ok ? 1.0 : 0.0;
        }
        __syncthreads();
        return S.sc[7] != 0.0;
    };
    auto solve = [&]() {
        if (wave == 0) {
            double b = lane < NY ? S.rhs[lane] : 0.0;
            fwd_step<0>(lrow, linvd, b, lane);
            bwd_step<NY - 1>(lcol, linvd, b, lane);
            if (lane < NY) S.dy[lane] = b;
        }
        __syncthreads();
        compute_x(S.dy, S.dx, false);
        __syncthreads();
    };
    // LSC row helpers (row r: -n.x <= -rhs)
    auto lsc_ax = [&](const double *xv, int r, int cp) -> double {
        return -((double)rn[r] * xv[cp] + (double)rn[R + r] * xv[SEGV + cp] + (double)rn[2 * R + r] * xv[2 * SEGV + cp]);
    };

    int status = LSC_STATUS_INFEASIBLE_K;
    int iters = 0;
    double obj = 0.0;

    if (overflow) {
        status = LSC_STATUS_CAPACITY_K;
    } else {
        // ---------------- initial point: (H + A^T A) y = -grad(x0) + A^T (h - A x0)
        compute_x(S.y, S.x, true);
        __syncthreads();
        for (int sl = tid; sl < AXROWS; sl += NT) {
            if (!S.avalid[sl]) continue;
            const int type = sl / NV, kt = sl % NV;
            S.at2[sl] = ax_row(S.x, type, kt / SEGV, kt % SEGV) - S.ah[sl];
        }
        for (int r = tid; r < R; r += NT) {
            const int cp = rcp[r];
            if (cp == 255) continue;
            rt2[r] = lsc_ax(S.x, r, cp) + rrhs[r];
        }
        __syncthreads();
        reduce_rows(true, true);
        assemble(true);
        if (factor()) {
            solve();
            if (tid < NY) S.y[tid] = S.dy[tid];
            __syncthreads();
            compute_x(S.y, S.x, true);
            __syncthreads();
            double mins = 1e300, minz = 1e300;
            for (int sl = tid; sl < AXROWS; sl += NT) {
                if (!S.avalid[sl]) continue;
                const int type = sl / NV, kt = sl % NV;
                double sv = S.ah[sl] - ax_row(S.x, type, kt / SEGV, kt % SEGV);
                S.as_[sl] = sv; S.az[sl] = -sv;
                mins = fmin(mins, sv); minz = fmin(minz, -sv);
            }
            for (int r = tid; r < R; r += NT) {
                const int cp = rcp[r];
                if (cp == 255) continue;
                double sv = -rrhs[r] - lsc_ax(S.x, r, cp);
                rs[r] = sv; rz[r] = -sv;
                mins = fmin(mins, sv); minz = fmin(minz, -sv);
            }
            block_reduce3(mins, minz, 0.0, 2, 2, 0);
            const double shs = S.sc[0] <= 0.0 ? 1.0 - S.sc[0] : 0.0;
            const double shz = S.sc[1] <= 0.0 ? 1.0 - S.sc[1] : 0.0;
            for (int sl = tid; sl < AXROWS; sl += NT) {
                if (!S.avalid[sl]) continue;
                S.as_[sl] += shs; S.az[sl] += shz;
            }
            for (int r = tid; r < R; r += NT) {
                if (rcp[r] == 255) continue;
                rs[r] += shs; rz[r] += shz;
            }
            __syncthreads();

            stamp(PH_INIT);
            // ---------------- Mehrotra predictor-corrector iterations
            const int max_iters = md.max_iters;
            const double hmax = fmax(1.0, fmax(fabs((double)md.world_max[0]), fabs((double)md.world_min[0])));
            for (iters = 0; iters < max_iters; iters++) {
                // P1: residuals, 1/s, v = w rp
                double gap = 0.0, rpmax = 0.0, nrow = 0.0;
                for (int sl = tid; sl < AXROWS; sl += NT) {
                    if (!S.avalid[sl]) continue;
                    const int type = sl / NV, kt = sl % NV;
                    double sv = S.as_[sl], zv = S.az[sl];
                    double rp = ax_row(S.x, type, kt / SEGV, kt % SEGV) + sv - S.ah[sl];
                    double is = 1.0 / sv;
                    S.at1[sl] = is;
                    S.at2[sl] = zv * is * rp;
                    gap += sv * zv; rpmax = fmax(rpmax, fabs(rp)); nrow += 1.0;
                }
                for (int r = tid; r < R; r += NT) {
                    const int cp = rcp[r];
                    if (cp == 255) continue;
                    double sv = rs[r], zv = rz[r];
                    double rp = lsc_ax(S.x, r, cp) + sv + rrhs[r];
                    double is = 1.0 / sv;
                    rt1[r] = is;
                    rt2[r] = zv * is * rp;
                    gap += sv * zv; rpmax = fmax(rpmax, fabs(rp)); nrow += 1.0;
                }
                // objective: sum x'(w_c Q)x + w_t sum |c - g|^2  (src/traj_optimizer.cpp:329-372)
                double objp = 0.0;
                if (tid < NV) {
                    objp = 0.5 * cost_grad() * S.x[tid];
                    if (xterm) { double e = S.x[tid] - S.goal[xk]; objp += md.w_t * e * e; }
                }
                block_reduce3(gap, rpmax, nrow, 0, 1, 0);
                gap = S.sc[0]; rpmax = S.sc[1]; nrow = S.sc[2];
                block_reduce3(objp, 0.0, 0.0, 0, 0, 0);
                obj = S.sc[0];
                const double mu = gap / nrow;
                stamp(PH_P1);

                reduce_rows(true, false);
                stamp(PH_REDUCE);
                assemble(true);
                stamp(PH_ASSEMBLE);
                double rdn = (tid < NY) ? fabs(S.dy[tid]) : 0.0;
                block_reduce3(rdn, 0.0, 0.0, 1, 0, 0);
                rdn = S.sc[0];
                if (tid == 0) { S.sc[3] = gap; S.sc[4] = rpmax; S.sc[5] = rdn; S.sc[6] = obj; }
                const bool gap_ok = gap <= 1e-9 * (1.0 + fabs(obj));
                if (rpmax <= 1e-9 * hmax && rdn <= 1e-5 * (1.0 + fabs(obj)) && gap_ok) { status = LSC_STATUS_OK_K; break; }
                if (!(gap == gap) || !(rpmax == rpmax)) break;

                const bool fok = factor();
                stamp(PH_FACTOR);
                if (!fok) {
                    // K lost definiteness to round-off: accept only if already within 1e-7 relative gap
                    if (rpmax <= 1e-8 * hmax && gap <= 1e-7 * (1.0 + fabs(obj))) status = LSC_STATUS_OK_K;
                    break;
                }
                solve();  // affine direction in dy / dx
                stamp(PH_SOLVE);
                {
                    // Newton-step test: with gap and primal residual at tolerance, the affine (pure Newton) step
                    // measures the distance to the optimum; the stationarity residual can stall at the round-off
                    // level of the ill-conditioned normal equations when z/s is huge.
                    double dxa = (tid < NV) ? fabs(S.dx[tid]) : 0.0, xa = (tid < NV) ? fabs(S.x[tid]) : 0.0;
                    block_reduce3(dxa, xa, 0.0, 1, 1, 0);
                    if (rpmax <= 1e-9 * hmax && gap_ok && S.sc[0] <= 1e-9 * fmax(1.0, S.sc[1])) { status = LSC_STATUS_OK_K; break; }
                }

                // P2: affine step length and centring statistics
                double amin = 1.0, s1 = 0.0, s2 = 0.0;
                for (int sl = tid; sl < AXROWS; sl += NT) {
                    if (!S.avalid[sl]) continue;
                    const int type = sl / NV, kt = sl % NV, k = kt / SEGV, t = kt % SEGV;
                    double sv = S.as_[sl], zv = S.az[sl], w = zv * S.at1[sl];
                    double rp = ax_row(S.x, type, k, t) + sv - S.ah[sl];
                    double adx = ax_row(S.dx, type, k, t);
                    double ds = -rp - adx, dz = -zv - w * ds;
                    if (ds < 0.0) amin = fmin(amin, -sv / ds);
                    if (dz < 0.0) amin = fmin(amin, -zv / dz);
                    s1 += sv * dz + zv * ds; s2 += ds * dz;
                    S.at2[sl] = ds * dz;
                }
                for (int r = tid; r < R; r += NT) {
                    const int cp = rcp[r];
                    if (cp == 255) continue;
                    double sv = rs[r], zv = rz[r], w = zv * rt1[r];
                    double rp = lsc_ax(S.x, r, cp) + sv + rrhs[r];
                    double adx = lsc_ax(S.dx, r, cp);
                    double ds = -rp - adx, dz = -zv - w * ds;
                    if (ds < 0.0) amin = fmin(amin, -sv / ds);
                    if (dz < 0.0) amin = fmin(amin, -zv / dz);
                    s1 += sv * dz + zv * ds; s2 += ds * dz;
                    rt2[r] = ds * dz;
                }
                block_reduce3(amin, s1, s2, 2, 0, 0);
                const double aaff = S.sc[0];
                const double mu_aff = (gap + aaff * S.sc[1] + aaff * aaff * S.sc[2]) / nrow;
                double sigma = mu > 0.0 ? mu_aff / mu : 0.0;
                sigma = sigma * sigma * sigma;
                const double smu = sigma * mu;
                stamp(PH_P2);

                // P3: corrector right-hand side  v = w rp - (ds dz - sigma mu)/s
                for (int sl = tid; sl < AXROWS; sl += NT) {
                    if (!S.avalid[sl]) continue;
                    const int type = sl / NV, kt = sl % NV;
                    double sv = S.as_[sl], zv = S.az[sl], is = S.at1[sl];
                    double rp = ax_row(S.x, type, kt / SEGV, kt % SEGV) + sv - S.ah[sl];
                    S.at2[sl] = zv * is * rp - (S.at2[sl] - smu) * is;
                }
                for (int r = tid; r < R; r += NT) {
                    const int cp = rcp[r];
                    if (cp == 255) continue;
                    double sv = rs[r], zv = rz[r], is = rt1[r];
                    double rp = lsc_ax(S.x, r, cp) + sv + rrhs[r];
                    rt2[r] = zv * is * rp - (rt2[r] - smu) * is;
                }
                __syncthreads();
                stamp(PH_P3);
                reduce_rows(false, false);
                stamp(PH_REDUCE);
                assemble(false);
                stamp(PH_ASSEMBLE);
                solve();  // combined direction
                stamp(PH_SOLVE);

                // P4: step length (ds = -rp - a.dx ; dz = -z + v + w a.dx)
                amin = 1e300;
                for (int sl = tid; sl < AXROWS; sl += NT) {
                    if (!S.avalid[sl]) continue;
                    const int type = sl / NV, kt = sl % NV, k = kt / SEGV, t = kt % SEGV;
                    double sv = S.as_[sl], zv = S.az[sl], w = zv * S.at1[sl];
                    double rp = ax_row(S.x, type, k, t) + sv - S.ah[sl];
                    double adx = ax_row(S.dx, type, k, t);
                    double ds = -rp - adx, dz = -zv + S.at2[sl] + w * adx;
                    if (ds < 0.0) amin = fmin(amin, -sv / ds);
                    if (dz < 0.0) amin = fmin(amin, -zv / dz);
                }
                for (int r = tid; r < R; r += NT) {
                    const int cp = rcp[r];
                    if (cp == 255) continue;
                    double sv = rs[r], zv = rz[r], w = zv * rt1[r];
                    double rp = lsc_ax(S.x, r, cp) + sv + rrhs[r];
                    double adx = lsc_ax(S.dx, r, cp);
                    double ds = -rp - adx, dz = -zv + rt2[r] + w * adx;
                    if (ds < 0.0) amin = fmin(amin, -sv / ds);
                    if (dz < 0.0) amin = fmin(amin, -zv / dz);
                }
                block_reduce3(amin, 0.0, 0.0, 2, 0, 0);
                const double alpha = fmin(1.0, 0.99 * S.sc[0]);

                // P5: update (recomputes ds, dz; nothing else is stored per row)
                for (int sl = tid; sl < AXROWS; sl += NT) {
                    if (!S.avalid[sl]) continue;
                    const int type = sl / NV, kt = sl % NV, k = kt / SEGV, t = kt % SEGV;
                    double sv = S.as_[sl], zv = S.az[sl], w = zv * S.at1[sl];
                    double rp = ax_row(S.x, type, k, t) + sv - S.ah[sl];
                    double adx = ax_row(S.dx, type, k, t);
                    double ds = -rp - adx, dz = -zv + S.at2[sl] + w * adx;
                    S.as_[sl] = sv + alpha * ds; S.az[sl] = zv + alpha * dz;
                }
                for (int r = tid; r < R; r += NT) {
                    const int cp = rcp[r];
                    if (cp == 255) continue;
                    double sv = rs[r], zv = rz[r], w = zv * rt1[r];
                    double rp = lsc_ax(S.x, r, cp) + sv + rrhs[r];
                    double adx = lsc_ax(S.dx, r, cp);
                    double ds = -rp - adx, dz = -zv + rt2[r] + w * adx;
                    rs[r] = sv + alpha * ds; rz[r] = zv + alpha * dz;
                }
                __syncthreads();
                if (tid < NY) S.y[tid] += alpha * S.dy[tid];
                __syncthreads();
                compute_x(S.y, S.x, true);
                __syncthreads();
                stamp(PH_P45);
            }
        }
    }

    // ------------------------------------------------------------------ output
    // success: float32 rounding of the optimum (src/traj_optimizer.cpp:79-96); failure: the optimiser's
    // previous trajectory is reused (src/traj_planner.cpp:1553-1584)
    float *out = a.traj_next + (size_t)qi * NV;
    float *stale = a.stale + (size_t)qi * NV;
    if (tid < NV) {
        if (status == LSC_STATUS_OK_K) {
            float v = (float)S.x[tid];
            out[tid] = v; stale[tid] = v;
        } else {
            out[tid] = stale[tid];
        }
    }
    if (tid == 0) {
        if (status == LSC_STATUS_OK_K) a.cost[qi] = obj;
        a.status[qi] = status;
        a.iters[qi] = iters;
        if (a.iters_acc) a.iters_acc[qi] += iters;
        if (a.nrows) {
            int tot = 0;
            for (int c = 0; c < NCP; c++) tot += S.cnt[c];
            a.nrows[qi] = tot;
        }
        if (a.dbg) { a.dbg[4 * qi] = S.sc[3]; a.dbg[4 * qi + 1] = S.sc[4]; a.dbg[4 * qi + 2] = S.sc[5]; a.dbg[4 * qi + 3] = S.sc[6]; }
        if constexpr (PROF) {
            stamp(PH_OUT);
            if (a.prof)
                for (int i = 0; i < PH_COUNT; i++) a.prof[(size_t)qi * PH_COUNT + i] += t_acc[i];
        }
    }
}

}  // namespace lsc

// ---------------------------------------------------------------------------------------------------
// launch wrappers (called from lsc_abi.cpp)
// ---------------------------------------------------------------------------------------------------
namespace lsc {

size_t plan_smem_bytes(int n_terms, int n_entries, int cap)
{
    size_t b = sizeof(Smem);
    b += sizeof(uint32_t) * (size_t)((n_terms + 1) & ~1);
    b += sizeof(uint32_t) * (size_t)(2 * n_entries + 2);
    b += sizeof(double) * (size_t)n_entries;
    const int cs = (cap & 1) ? cap : cap + 1;
    size_t R = (size_t)NB * cs;
    b += R * (5 * sizeof(double) + 3 * sizeof(float) + 1);
    return (b + 15) & ~(size_t)15;
}

hipError_t launch_plan(const PlanArgs &a, size_t smem, hipStream_t st)
{
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&lsc_plan_kernel<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&lsc_plan_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    if (a.prof) hipLaunchKernelGGL(lsc_plan_kernel<true>, dim3(a.count), dim3(NT), smem, st, a);
    else hipLaunchKernelGGL(lsc_plan_kernel<false>, dim3(a.count), dim3(NT), smem, st, a);
    return hipGetLastError();
}

hipError_t launch_sweep(const SweepArgs &a, hipStream_t st)
{
    long total = (long)a.count * (a.N - 1) * M;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(lsc_sweep_kernel, dim3(blocks), dim3(256), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_propagate(const float *traj, float *state, int N, double dt, hipStream_t st)
{
    int n = N * 3;
    hipLaunchKernelGGL(lsc_propagate_kernel, dim3((n + 127) / 128), dim3(128), 0, st, traj, state, N, (float)pow(dt, -1));
    return hipGetLastError();
}

hipError_t launch_gjk(const double *pts, int count, double *v, double *dist, hipStream_t st)
{
    hipLaunchKernelGGL(lsc_gjk_kernel, dim3((count + 255) / 256), dim3(256), 0, st, pts, count, v, dist);
    return hipGetLastError();
}

}  // namespace lsc
