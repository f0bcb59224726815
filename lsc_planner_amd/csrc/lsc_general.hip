// lsc_general.hip -- the reference's ALTERNATE planner modes on gfx950, one 256-lane workgroup per agent:
//
//   BVC planner mode        TrajPlanner::generateBVC                                   src/traj_planner.cpp:1409-1440
//                           prediction / initial trajectory = current position          :796-807, :1039-1045 (param.cpp:40-45)
//                           no stop-at-horizon equalities (LSC only)                    src/traj_optimizer.cpp:527-536
//                           opt/N_constraint_segments                                   src/traj_optimizer.cpp:410, 438
//   slack variables         SlackMode::DYNAMICALLIMIT / COLLISIONCONSTRAINT             src/traj_optimizer.cpp:306-326, 375-390,
//                                                                                       455-457, 476-510
//   disturbance reset       obstaclePredictionCheck / initialTrajPlanningCheck and the slack rows they leave behind for the
//                           rest of the mission (obs_slack_indices is never cleared)    src/traj_planner.cpp:866-878, 1047-1061
//
// These modes change the SHAPE of the QP (45 instead of 39 free coordinates without the stop rows; slack variables that
// couple all control points of a segment), which the banded, register-resident solver of lsc_plan_kernel is built around.
// They are off the reference's default path (every shipped launch file runs mode/planner = lsc, slack none, and the
// disturbance checks only fire on a real disturbance), so this kernel trades speed for generality: dense reduced-space
// Mehrotra interior point over all rows (no pruning; row arrays in LDS as far as it reaches, the rest in an HBM workspace),
// warm start with the cold start as fallback -- the same algorithm as the fast path, little of its structure.  Agents reach it through status LSC_STATUS_GENERAL_K set by lsc_plan_kernel's phase A.
//
// Unknowns: y (3 x nya free control-point coordinates, nya = 13 with / 15 without the stop rows), the 2M slack variables
// of DYNAMICALLIMIT as explicit unknowns, and one slack variable per (slack obstacle, segment) that is eliminated from
// every Newton system analytically (its Hessian block is diagonal): K = Kyy - sum_g m_g m_g^T / D_g.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "lsc_gjk.hpp"
#include "lsc_model.hpp"
#include "lsc_kernels.h"
#include "lsc_wave.hpp"
#include <type_traits>

namespace lsc {

namespace {

// (KArgs -- the kernel's argument block, read where it lies in the kernarg segment -- is declared in lsc_kernels.h)

constexpr int GT = 512;            // lanes per agent
constexpr int GW = GT / 64;
constexpr int PMAX = 3 * GNYA + 2 * M;      // 45 + 10 for M = 5
constexpr int P_STOP = 3 * NYA, P_FREE = 3 * GNYA;   // unknowns with / without the stop-at-horizon rows (39 / 45 for M = 5)
constexpr int KL = PMAX + 2;       // leading dimension of the dense matrices in LDS (57 doubles: rows of one column fall in different banks)
constexpr int NBK = NCP - 3;       // control points that carry collision rows (27 for M = 5)
constexpr int SEG_E = 171;
// sections of lsc_general_profile
enum { GP_SETUP = 0, GP_START, GP_RESID, GP_REDUCE, GP_ASSEMBLE, GP_FACTOR, GP_SOLVE, GP_AFFINE, GP_CORR_RHS, GP_REDUCE2, GP_ASSEMBLE2, GP_STEP, GP_ITERS, GP_AGENTS };         // symmetric 18 x 18 block of one segment (6 control points x 3 axes)

struct GS {
    double x[96], dx[96];
    double y[PMAX + 1], dy[PMAX + 1], rhs[PMAX + 1];
    double K[PMAX * KL];
    double invd[PMAX + 1];         // 1 / D of K = L D L^T
    double gv[96], gz[96];         // x-space: cost gradient + sum vv_r a_r  /  + sum z_r a_r
    double Wd[NV], W1[NV], W2[NV]; // x-space Hessian pieces of the bound / velocity / acceleration rows
    double Ws[NCP * 6];            // per control point: sum w n n^T (xx xy xz yy yz zz)
    double Tv[NCP * 3], Tz[NCP * 3];   // per control point: -sum vv n, -sum z n
    double part[3][NBK * 3][2];    // partial sums of the control-point reductions
    double Wu[2 * M][NV];          // DYNAMICALLIMIT: cross terms x-space <-> slack variable j
    double Huu[2 * M], qu[2 * M], gu[2 * M];
    double Cm[M][SEG_E];           // per segment: sum_g m_g m_g^T / D_g in x-space
    double cq[NCP * 3];            // x-space: sum_g m_g q_g / D_g
    double Z[SEGV][GNYA];
    double tc[GNYA][4];            // column a of Z as a short list: the (at most four) control points y_a moves ...
    int tt[GNYA][4], tn[GNYA];     // ... their indices t and the list length
    double Hc[GNYA * GNYA];
    double Qh[NC * NC];
    double s0[3][3], lo[3][M], hi[3][M], goal[3];
    double ah[AXROWS];
    double red[8][GW];
    double sc[8];
    float pinit[NV];
    float goalf[3];
    unsigned char avalid[AXROWS];
    double reachL[3][28], reachU[3][28];   // per axis: bounds of c_{m,i} - c_{0,2} after K = 5m+i-2 steps (row pruning, as in phase B of lsc_plan_kernel)
    int wkept[GW];
    int tseg, ok, any_slack, nk;
};

// predicted control points of agent q for segment m in the general modes: current position (BVC, or after a
// disturbance reset), else like the fast path
__device__ __forceinline__ void g_segment(KArgs &a, int q, int m, bool at_rest, float dtf, F3 out[6])
{
#pragma clang fp contract(off)
    const float *s = a.state + 9 * q;
    if (at_rest) {
#pragma unroll
        for (int i = 0; i < 6; i++) out[i] = F3{s[0], s[1], s[2]};
        return;
    }
    if (a.planner_seq < 2) {
#pragma unroll
        for (int i = 0; i < 6; i++) {
            float mi = (float)((double)m + (double)i / (double)DEG);
            float ax = (s[3] * mi) * dtf, ay = (s[4] * mi) * dtf, az = (s[5] * mi) * dtf;
            out[i] = F3{s[0] + ax, s[1] + ay, s[2] + az};
        }
    } else {
        const float *t = a.traj_prev + (size_t)q * NV;
        if (m < M - 1) {
#pragma unroll
            for (int i = 0; i < 6; i++) { int c = (m + 1) * NC + i; out[i] = F3{t[c], t[SEGV + c], t[2 * SEGV + c]}; }
        } else {
            int c = (M - 1) * NC + DEG;
            F3 e = F3{t[c], t[SEGV + c], t[2 * SEGV + c]};
#pragma unroll
            for (int i = 0; i < 6; i++) out[i] = e;
        }
    }
}

// obstaclePredictionCheck / initialTrajPlanningCheck for agent q: its plan says it should be at traj_prev[q](t = dt) now
__device__ __forceinline__ bool disturbed_now(KArgs &a, int q)
{
#pragma clang fp contract(off)
    if (!(a.reset_thr > 0.0) || a.planner_seq < 2 || a.planner_mode != 0) return false;
    const float *t = a.traj_prev + (size_t)q * NV + NC;          // shifted plan, segment 0, point 0
    const float *s = a.state + 9 * q;
    const float dx = t[0] - s[0], dy = t[SEGV] - s[1], dz = t[2 * SEGV] - s[2];
    const float n2 = dx * dx + dy * dy + dz * dz;
    return sqrt((double)n2) > a.reset_thr;
}

__device__ __forceinline__ double ax_x(const double *x, int type, int k, int t)
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

// ---- dense K = L D L^T and the two substitutions, on wave 0 alone and OUT OF LINE: one body each, whatever the number of
// call sites, with a register allocation of its own (inlined, the unrolled row of the factor pushed the whole kernel into
// scratch and the scalar registers of the caller into vector lanes).  lane = row.  The factor runs right-looking in registers
// (av[c] = K[lane][c]; step j: pivot through v_readlane, every later column k takes -l_j l_k d_j with l_k again a v_readlane:
// ~PU^2/2 readlane + fma pairs, no square roots, no LDS traffic); L (strictly lower, zeros above) and 1/D go back to LDS once.
// The substitutions hold the row / the column of L in registers -- loaded ahead of the dependent chain readlane -> fma, which
// is all that remains on it.  PU: compile-time bound of the unrolled loops (45 without, 55 with the explicit slack variables);
// rows and columns P..PU-1 are the identity (written once at set-up).  K is stored as a full symmetric matrix.
extern __shared__ __align__(16) unsigned char gsm_general[];
__device__ __forceinline__ int uni_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ unsigned long long uni_u64(unsigned long long v)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
template <typename T>
__device__ __forceinline__ T *uni_p(T *p) { return (T *)uni_u64((unsigned long long)p); }
// column J of the right-looking factor: the entries l_k d_J of the later columns reach every lane as broadcasts (v_readlane -> scalar pair), in
// batches of eight -- eight broadcasts, then their eight updates: a broadcast directly in front of its update costs a wait state each, and
// left to itself the scheduler either does exactly that or hoists whole columns of broadcasts and keeps ~100 more registers alive
template <int PU, int J, int K0>
__device__ __forceinline__ void dense_factor_batch(double (&av)[PU], const double l)
{
    constexpr int NB8 = (PU - K0) < 8 ? (PU - K0) : 8;
    double sk[NB8];
#pragma unroll
    for (int q = 0; q < NB8; q++) sk[q] = lane_value(av[J], K0 + q);
#pragma unroll
    for (int q = 0; q < NB8; q++) av[K0 + q] = fma(-l, sk[q], av[K0 + q]);
    if constexpr (K0 + 8 < PU) dense_factor_batch<PU, J, K0 + 8>(av, l);
}
template <int PU, int J>
__device__ __forceinline__ void dense_factor_step(double (&av)[PU], double *invd, int &npos)
{
    const double d = lane_value(av[J], J);
    npos += d > 0.0 ? 1 : 0;                               // (a count, not a flag: the chain of ANDs was kept in 2 PU scalar registers to the end)
    const double dd = d > 0.0 ? d : 1.0;
    double inv = __builtin_amdgcn_rcp(dd);                 // 1 / d: hardware estimate + two Newton steps (full division is twice the chain)
    inv = fma(fma(-dd, inv, 1.0), inv, inv);
    inv = fma(fma(-dd, inv, 1.0), inv, inv);
    const double l = av[J] * inv;                          // column J of L
    invd[J] = inv;                                         // uniform over the wave: every lane stores the same word (selecting lane J's copy at the
                                                           // end kept all PU reciprocals alive: 2 PU registers, the callee-saved ones among them)
    if constexpr (J + 1 < PU) dense_factor_batch<PU, J, J + 1>(av, l);
    av[J] = l;
    if constexpr (J + 1 < PU) dense_factor_step<PU, J + 1>(av, invd, npos);
}
template <int PU>
__device__ __attribute__((noinline)) void dense_factor_w0()
{
    GS &S = *reinterpret_cast<GS *>(gsm_general);
    const int lane = (int)threadIdx.x & 63;
    const int lr = lane < PU ? lane : PU - 1;
    double av[PU];
#pragma unroll
    for (int c = 0; c < PU; c++) av[c] = S.K[lr * KL + c];
    int npos = 0;
    dense_factor_step<PU, 0>(av, S.invd, npos);
    if (lane < PU) {
        int lw = lr;                                         // (an opaque copy: the ~PU / 2 addresses of the loads above are not kept for these stores)
        asm volatile("" : "+v"(lw));
#pragma unroll
        for (int c = 0; c < PU; c++) S.K[lw * KL + c] = c < lane ? av[c] : 0.0;
    }
    if (lane == 0) S.ok = npos == PU ? 1 : 0;
}
template <int PU>
__device__ __attribute__((noinline)) void dense_solve_w0(int P_in)
{
    GS &S = *reinterpret_cast<GS *>(gsm_general);
    const int lane = (int)threadIdx.x & 63;
    const int P = uni_i(P_in);
    const int lr = lane < PU ? lane : PU - 1;
    double b = lane < PU ? S.rhs[lr] : 0.0;
    const double myinv = S.invd[lr];
    {
        double lrow[PU];                                   // L[lane][j], zero for j >= lane
#pragma unroll
        for (int j = 0; j < PU; j++) lrow[j] = S.K[lr * KL + j];
#pragma unroll
        for (int j = 0; j < PU; j++) b = fma(-lrow[j], lane_value(b, j), b);
    }
    b *= myinv;
    {
        double lcol[PU];                                   // L[j][lane], zero for j <= lane
#pragma unroll
        for (int j = 0; j < PU; j++) lcol[j] = S.K[j * KL + lr];
#pragma unroll
        for (int j = PU - 1; j >= 0; j--) b = fma(-lcol[j], lane_value(b, j), b);
    }
    if (lane < P) S.dy[lane] = b;
}

// Per-workgroup row workspace (per-row state of the interior point, collision rows of all obstacles).  As much of it as
// fits behind the solver state lives in LDS (all but one array at N = 64), the rest in HBM: the row passes are chains of
// dependent loads, an order of magnitude shorter out of LDS than out of L2.  The code is the same either way (flat
// addressing).
__host__ __device__ inline size_t ws_main_bytes(int N)
{
    const size_t nob = N - 1 > 1 ? N - 1 : 1;
    const size_t RT = AXROWS + 2 * M + NBK * nob + M * nob;
    size_t b = sizeof(double) * (4 * RT + NBK * nob + 5 * M * nob) + sizeof(float) * 3 * M * nob + nob * (1 + NBK + M) + 16 * 14;
    return (b + 255) & ~(size_t)255;
}
// + the staging area of the row build (rows of all obstacles before the compaction)
__host__ __device__ inline size_t ws_bytes_of(int N)
{
    const size_t nob = N - 1 > 1 ? N - 1 : 1;
    size_t b = ws_main_bytes(N) + ((12 * M * nob + 15) & ~(size_t)15) + sizeof(double) * NBK * nob + ((NBK * nob + 15) & ~(size_t)15) + 2 * nob + 16;
    return (b + 255) & ~(size_t)255;
}
__host__ __device__ inline size_t gs_bytes() { return (sizeof(GS) + 255) & ~(size_t)255; }
__host__ __device__ inline size_t ws_lds_bytes(int N)
{
    const size_t room = 160 * 1024 - gs_bytes(), all = ws_main_bytes(N);
    return all < room ? all : room;
}

}  // namespace

// (noinline: the kernel below must be able to leave before this function's frame -- it keeps part of its state in
// scratch -- is set up; the common launch is the one that finds nobody flagged)
// LDSP: every row array in LDS, addressed as such (ds_read / ds_write instead of flat accesses through generic pointers: the row
// passes are chains of dependent loads).  Returns false, before touching anything but its own set-up, when the kept rows do not
// fit -- the caller then runs the generic-pointer build, which spills the last arrays to the HBM workspace.
#if defined(__HIP_DEVICE_COMPILE__)
#define LSC_LDS_PTR(T) __attribute__((address_space(3))) T *
#else
#define LSC_LDS_PTR(T) T *
#endif
// (inlined into the kernel: as a function of its own it saved the ~110 callee-saved vector registers of the calling convention at
// entry -- a third of the scratch writes; with the spills gone there is no frame left whose set-up the early exit would have to dodge)
template <bool LDSP>
static __device__ __forceinline__ bool general_agent(KArgs &a_in, const int al_in, unsigned char *smem_raw_in, unsigned char *wsb_in,
                                                         unsigned char *lds_ws_in, size_t lds_ws_bytes_in)
{
    // The arguments of an out-of-line device function arrive in VECTOR registers, and everything derived from them -- every row-array
    // pointer, every offset -- stays there: ~230 loop-invariant values were spilled once per agent (the 40 MB of scratch writes per
    // launch in round 3's PMC pass) and reloaded ~600 times per iteration.  They are uniform by construction: back to scalars.
    KArgs &a = *(KArgs *)uni_u64((unsigned long long)&a_in);
    const int al = uni_i(al_in);
    unsigned char *smem_raw = uni_p(smem_raw_in), *wsb = uni_p(wsb_in), *lds_ws = uni_p(lds_ws_in);
    size_t lds_ws_bytes = (size_t)uni_u64((unsigned long long)lds_ws_bytes_in);
    using FP = typename std::conditional<LDSP, LSC_LDS_PTR(float), float *>::type;
    using BP = typename std::conditional<LDSP, LSC_LDS_PTR(unsigned char), unsigned char *>::type;
    using DP = typename std::conditional<LDSP, LSC_LDS_PTR(double), double *>::type;
    GS &S = *reinterpret_cast<GS *>(smem_raw);
    const GModel &gm = *a.gmodel;
    const Model &md = *a.model;
    // `tid` is re-read through an opaque copy at the start of every phase (fresh()): the compiler otherwise hoists the per-lane address
    // arithmetic of ALL phases to the top of the function -- ~120 values per lane that do not fit the register file and went to
    // scratch once per agent, to be reloaded ~600 times per iteration (the rest of round 3's 40 MB of scratch writes per launch).
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));       // (opaque from the first use on: what the set-up derives from it is not hoisted out of the kernel's loop over agents)
    const int lane = tid & 63, wave = tid >> 6;
    auto fresh = [&]() { int t = threadIdx.x; asm volatile("" : "+v"(t)); tid = t; };
    const int qi = a.first + al;
    const int N = a.N, n_all = N - 1, nob_all = n_all > 0 ? n_all : 1;
    const int nya = uni_i(gm.nya), P0 = 3 * nya;
    const int nu = a.slack_mode == 1 ? 2 * M : 0;
    const int P = P0 + nu;
    const int ncs = a.ncs < 0 ? M : (a.ncs > M ? M : a.ncs);
    const bool bvc = a.planner_mode == 1;
    const float dtf = (float)md.dt;
    const double hv = md.hv_scale, ha = md.ha_scale;
    // planar world (world/dimension == 2, src/traj_optimizer.cpp:8): no z variables -- here: the z unknowns stay without any row,
    // decoupled (every n_z zeroed), resting at z_2d, and are overwritten on output (see lsc_model.hpp)
    const bool dim2 = md.dim2 != 0;

    // block reduction of up to five values: op 0 sum, 1 max, 2 min, < 0 slot unused; results in S.sc[0..4].  DPP wave reductions
    // (lsc_wave.hpp), one barrier pair, the per-wave partials combined by five lanes.
    auto block_reduce = [&](double v0, double v1, double v2, double v3, double v4, int op0, int op1, int op2, int op3, int op4) {
        auto wr = [&](double v, int op) { return op < 0 ? 0.0 : (op == 0 ? wave_sum(v) : (op == 1 ? wave_max(v) : wave_min(v))); };
        const double r0 = wr(v0, op0), r1 = wr(v1, op1), r2 = wr(v2, op2), r3 = wr(v3, op3), r4 = wr(v4, op4);
        if (lane == 0) {
            S.red[0][wave] = r0;
            if (op1 >= 0) S.red[1][wave] = r1;
            if (op2 >= 0) S.red[2][wave] = r2;
            if (op3 >= 0) S.red[3][wave] = r3;
            if (op4 >= 0) S.red[4][wave] = r4;
        }
        __syncthreads();
        const int nused = op4 >= 0 ? 5 : (op3 >= 0 ? 4 : (op2 >= 0 ? 3 : (op1 >= 0 ? 2 : 1)));
        if (tid < nused) {
            const int op = tid == 0 ? op0 : (tid == 1 ? op1 : (tid == 2 ? op2 : (tid == 3 ? op3 : op4)));
            double t = S.red[tid][0];
#pragma unroll
            for (int w = 1; w < GW; w++) t = op == 0 ? t + S.red[tid][w] : (op == 1 ? fmax(t, S.red[tid][w]) : fmin(t, S.red[tid][w]));
            S.sc[tid] = t;
        }
        __syncthreads();
    };

    // optional section profile (lsc_general_profile): shader cycles seen by lane 0 between the stamps
    long long *const gp = a.prof ? a.prof + ((size_t)N + qi) * PROF_PHASES : nullptr;
    long long tk = gp ? (long long)__builtin_readcyclecounter() : 0;
    auto gstamp = [&](int slot) {
        if (gp && tid == 0) { const long long t = (long long)__builtin_readcyclecounter(); gp[slot] += t - tk; tk = t; }
    };
    // ------------------------------------------------------------------ setup
    const bool own_now = disturbed_now(a, qi);
    const bool ever_i = a.ever ? (a.ever[qi] != 0) : false;
    const bool own_rest = bvc || own_now;
    if (tid < NV) {
#pragma clang fp contract(off)
        const int k = tid / SEGV, c = tid % SEGV, m = c / NC, i = c % NC;
        const float *s = a.state + 9 * qi;
        float val;
        if (own_rest) val = s[k];
        else if (a.planner_seq < 2) {
            float mi = (float)((double)m + (double)i / (double)DEG);
            val = s[k] + (s[3 + k] * mi) * dtf;
        } else {
            const float *t = a.traj_prev + (size_t)qi * NV + k * SEGV;
            val = (m < M - 1) ? t[(m + 1) * NC + i] : t[(M - 1) * NC + DEG];
        }
        S.pinit[tid] = val;
    }
    for (int i = tid; i < SEGV * GNYA; i += GT) S.Z[i / GNYA][i % GNYA] = gm.Z[i / GNYA][i % GNYA];
    for (int i = tid; i < GNYA * GNYA; i += GT) S.Hc[i] = gm.Hc[i];
    if (tid < NC * NC) S.Qh[tid] = md.Qh[tid];
    if (tid >= 64 && tid < 64 + GNYA) {
        const int aa = tid - 64;
        int n = 0;
        for (int t = 3; t < SEGV && aa < nya; t++)
            if (gm.Z[t][aa] != 0.0 && n < 4) { S.tt[aa][n] = t; S.tc[aa][n] = gm.Z[t][aa]; n++; }
        S.tn[aa] = n;
    }
    if (tid < 3) {
        const int k = tid;
        const float *s = a.state + 9 * qi;
        double c0 = (double)s[k], c1 = c0 + (double)s[3 + k] * hv, c2 = (double)s[6 + k] * ha + 2.0 * c1 - c0;
        if (dim2 && k == 2) c0 = c1 = c2 = md.z2d;
        S.s0[k][0] = c0; S.s0[k][1] = c1; S.s0[k][2] = c2;
        S.goalf[k] = a.goal_out[3 * qi + k];                  // current_goal_position, planned by phase A of lsc_plan_kernel
        S.goal[k] = (dim2 && k == 2) ? md.z2d : (double)S.goalf[k];
        for (int m = 0; m < M; m++) {
            double lo = (double)md.world_min[k], hi = (double)md.world_max[k];
            if (md.use_sfc && a.sfc && m < ncs) {
                const float *b = a.sfc + ((size_t)qi * M + m) * 6;
                lo = fmax(lo, (double)b[k]);
                hi = fmin(hi, (double)b[3 + k]);
            }
            S.lo[k][m] = lo; S.hi[k][m] = hi;
        }
    }
    __syncthreads();
    if (tid == 0) {
#pragma clang fp contract(off)
        const float *s = a.state + 9 * qi;
        const float *g = S.goalf;
        float dxg = g[0] - s[0], dyg = g[1] - s[1], dzg = g[2] - s[2];
        float n2 = dxg * dxg + dyg * dyg + dzg * dzg;
        double flight = sqrt((double)n2) / a.vnom[qi];
        int T = (int)((M * md.dt - flight + 1e-9) / md.dt);
        S.tseg = T > 1 ? T : 1;
    }
    for (int sl = tid; sl < AXROWS; sl += GT) {
        const int type = sl / NV, kt = sl % NV, k = kt / SEGV, t = kt % SEGV, m = t / NC, i = t % NC;
        bool valid;
        double h;
        if (type < 2) { valid = !(m == 0 && i < 3); h = type == 0 ? S.hi[k][m] : -S.lo[k][m]; }
        else if (type < 4) { valid = i <= 4 && !(m == 0 && i < 2); h = a.vmax[3 * qi + k] * hv; }
        else { valid = i <= 3 && !(m == 0 && i == 0); h = a.amax[3 * qi + k] * ha; }
        if (dim2 && k == 2) valid = false;                    // `for (k < dim)`: src/traj_optimizer.cpp:274, 469
        S.avalid[sl] = valid ? 1 : 0;
        S.ah[sl] = h;
    }
    fresh();
    // ---- collision rows of every obstacle: LSC via GJK, or the BVC half-space.  Rows that cannot be active inside the
    // reachable box of their control point are redundant (the test of lsc_plan_kernel's phase B; it rests on the velocity and
    // acceleration rows being hard, so not with DYNAMICALLIMIT's slack on them; a slack variable on the row only relaxes it
    // further).  Obstacles without an active row are left out altogether: the arrays below are per KEPT obstacle, in
    // increasing order of the obstacle index.
    const bool prune = md.prune != 0 && a.slack_mode != 1 && !a.out_normal;
    // staging area (HBM, behind the fallback workspace): rows of all obstacles before the compaction
    float *t_nrm = reinterpret_cast<float *>(wsb + ws_main_bytes(N));                                  // [n_all * M][3]
    double *t_crhs = reinterpret_cast<double *>(wsb + ws_main_bytes(N) + (((size_t)12 * M * nob_all + 15) & ~(size_t)15));   // [n_all][NBK]
    unsigned char *t_act = reinterpret_cast<unsigned char *>(t_crhs + (size_t)NBK * nob_all);         // [n_all][NBK]
    unsigned short *omap = reinterpret_cast<unsigned short *>(t_act + (((size_t)NBK * nob_all + 15) & ~(size_t)15));   // [kept] -> obstacle
    if (tid < 3) {
        const int k = tid;
        const double V = a.vmax[3 * qi + k] * hv, A = a.amax[3 * qi + k] * ha;
        const double d0 = S.s0[k][2] - S.s0[k][1];
        double lo = 0.0, hi = 0.0;
        S.reachL[k][0] = 0.0; S.reachU[k][0] = 0.0;
        for (int j = 1; j < 28; j++) {
            lo += fmax(-V, d0 - (double)j * A) - 1e-9;
            hi += fmin(V, d0 + (double)j * A) + 1e-9;
            S.reachL[k][j] = lo; S.reachU[k][j] = hi;
        }
    }
    __syncthreads();
    auto in_set_of = [&](int qj) {
        return a.slack_mode == 2 || (a.slack_mode == 0 && (ever_i || own_now || (a.ever && a.ever[qj]) || disturbed_now(a, qj)));
    };
    {
        const double r_a = a.radius[qi], dw_a = a.downwash[qi];
        for (int u = tid; u < n_all * M; u += GT) {
            const int oi = u / M, m = u % M;
            const int qj = oi < qi ? oi : oi + 1;
            F3 pa[6], po[6];
#pragma unroll
            for (int i = 0; i < 6; i++) { int c = m * NC + i; pa[i] = F3{S.pinit[c], S.pinit[SEGV + c], S.pinit[2 * SEGV + c]}; }
            g_segment(a, qj, m, bvc || disturbed_now(a, qj), dtf, po);
            const double r_o = a.radius_obs[qj];
            const double downwash = (dw_a * r_a + a.downwash_obs[qj] * r_o) / (r_a + r_o);
            F3 n;
            double d[6];
            if (bvc) {
#pragma clang fp contract(off)
                // generateBVC: normal from the two current positions, one margin for all rows of the obstacle
                const float pz = (float)((double)S.pinit[2 * SEGV] / downwash), qz = (float)((double)po[0].z / downwash);
                const F3 rel = F3{S.pinit[0] - po[0].x, S.pinit[SEGV] - po[0].y, pz - qz};
                n = normalized_f32(rel);
                const float dp = rel.x * n.x + rel.y * n.y + rel.z * n.z;
                const double dd = 0.5 * ((r_o + r_a) + (double)dp);
                n.z = (float)((double)n.z / downwash);
#pragma unroll
                for (int i = 0; i < 6; i++) d[i] = dd;
            } else {
                lsc_segment(pa, po, downwash, r_o + r_a, n, d);
            }
            if (a.out_normal) {
                size_t o = ((size_t)al * n_all + oi) * M + m;
                a.out_normal[o * 3] = n.x; a.out_normal[o * 3 + 1] = n.y; a.out_normal[o * 3 + 2] = n.z;
#pragma unroll
                for (int i = 0; i < 6; i++) a.out_d[o * 6 + i] = d[i];
            }
            if (dim2) n.z = 0.0f;                              // the row's z term exists only `if (dim == 3)` (:446-453)
            t_nrm[3 * u] = n.x; t_nrm[3 * u + 1] = n.y; t_nrm[3 * u + 2] = n.z;
            const double nx = (double)n.x, ny = (double)n.y, nz = (double)n.z;
            const double centre = nx * S.s0[0][2] + ny * S.s0[1][2] + nz * S.s0[2][2];
            const double (*rx)[28] = nx >= 0.0 ? S.reachL : S.reachU, (*ry)[28] = ny >= 0.0 ? S.reachL : S.reachU,
                         (*rzb)[28] = nz >= 0.0 ? S.reachL : S.reachU;
#pragma unroll
            for (int i = 0; i < 6; i++) {
                const int cp = m * NC + i;
                if (cp < 3) continue;
                double r = d[i];
                r += nx * (double)po[i].x;
                r += ny * (double)po[i].y;
                r += nz * (double)po[i].z;
                t_crhs[oi * NBK + cp - 3] = r;
                bool on = m < ncs;
                if (on && prune) {
                    const int K = 5 * m + i - 2;             // smallest n.c over the reachable box of c_{m,i}
                    const double worst = centre + nx * rx[0][K] + ny * ry[1][K] + nz * rzb[2][K];
                    if (worst >= r + 1e-6) on = false;
                }
                t_act[oi * NBK + cp - 3] = on ? 1 : 0;
            }
        }
    }
    __threadfence_block();
    __syncthreads();
    fresh();
    // kept obstacles, in order: ballot ranks per wave, wave offsets through LDS
    {
        int base = 0;
        for (int o0 = 0; o0 < n_all; o0 += GT) {
            const int oi = o0 + tid;
            bool keep = false;
            if (oi < n_all) {
                if (!prune) keep = true;
                else
                    for (int c = 0; c < NBK; c++) keep |= t_act[oi * NBK + c] != 0;
            }
            const unsigned long long mk = __ballot(keep);
            if (lane == 0) S.wkept[wave] = __popcll(mk);
            __syncthreads();
            int off = base;
            for (int w = 0; w < wave; w++) off += S.wkept[w];
            if (keep) omap[off + __popcll(mk & ((1ull << lane) - 1ull))] = (unsigned short)oi;
            int tot = 0;
            for (int w = 0; w < GW; w++) tot += S.wkept[w];
            base += tot;
            __syncthreads();
        }
        if (tid == 0) S.nk = base;
    }
    __threadfence_block();
    __syncthreads();
    const int n_obs = uni_i(S.nk), nob = n_obs > 0 ? n_obs : 1;
    // ---- workspace carve-up: per-row state of the interior point, collision rows of the kept obstacles
    const int NCL = NBK * nob, NGR = M * nob;
    const int US0 = AXROWS, CL0 = AXROWS + 2 * M, GS0 = CL0 + NCL, RT = GS0 + NGR;
    // Each array goes to LDS while there is room (most latency-critical first: the ones the per-control-point reductions
    // walk obstacle by obstacle), else to the workgroup's HBM workspace; the code below only sees flat pointers.
    bool fits = true;
    auto take = [&](size_t bytes) -> unsigned char * {
        bytes = (bytes + 15) & ~(size_t)15;
        unsigned char *p;
        if (bytes <= lds_ws_bytes) { p = lds_ws; lds_ws += bytes; lds_ws_bytes -= bytes; }
        else { p = wsb; wsb += bytes; fits = false; }
        return p;
    };
    FP nrm = (FP)take(sizeof(float) * 3 * NGR);                                          // [NGR][3]
    BP slk = (BP)take(nob);                                                              // [nob]
    BP cact = (BP)take((size_t)NBK * nob);                                               // [NCL] row is active
    BP gact = (BP)take((size_t)M * nob);                                                 // [NGR] group has an active row
    DP rt1 = (DP)take(sizeof(double) * RT);
    DP rt2 = (DP)take(sizeof(double) * RT);
    DP rz = (DP)take(sizeof(double) * RT);
    DP crhs = (DP)take(sizeof(double) * NCL);                                            // [NCL]   d + n.q of a collision row
    DP ev = (DP)take(sizeof(double) * NGR);                                              // [NGR]   group slack variables
    DP dev = (DP)take(sizeof(double) * NGR);
    DP Dg = (DP)take(sizeof(double) * NGR);
    DP iDg = (DP)take(sizeof(double) * NGR);                                             // 1 / D_g
    DP qg = (DP)take(sizeof(double) * NGR);
    DP rs = (DP)take(sizeof(double) * RT);
    if (LDSP && !fits) return false;                                                     // (uniform: sizes only)

    fresh();
    for (int oe = tid; oe < n_obs; oe += GT) {
        const int oi = omap[oe];
        slk[oe] = in_set_of(oi < qi ? oi : oi + 1) ? 1 : 0;
    }
    if (tid == 0 && n_obs == 0) slk[0] = 0;
    for (int u = tid; u < n_obs * M; u += GT) {
        const int oe = u / M, m = u % M, oi = omap[oe];
        const float *tn = t_nrm + 3 * (oi * M + m);
        nrm[3 * u] = tn[0]; nrm[3 * u + 1] = tn[1]; nrm[3 * u + 2] = tn[2];
        unsigned char any = 0;
        for (int i = 0; i < NC; i++) {
            const int cp = m * NC + i;
            if (cp >= 3) any |= t_act[oi * NBK + cp - 3];
        }
        gact[u] = any;
    }
    for (int c = tid; c < NCL; c += GT) {
        const int oe = c / NBK, cpi = c % NBK;
        const bool there = oe < n_obs;                                                   // (NCL is NBK even with nobody kept)
        const int src = there ? omap[oe] * NBK + cpi : 0;
        crhs[c] = there ? t_crhs[src] : 0.0;
        cact[c] = there ? t_act[src] : 0;
        rt1[CL0 + c] = 0.0; rt2[CL0 + c] = 0.0; rz[CL0 + c] = 0.0; rs[CL0 + c] = 1.0;     // rows left out stay zero in every sum
    }
    __syncthreads();
    if (tid == 0) {
        int any = 0;
        for (int oe = 0; oe < n_obs; oe++) any |= slk[oe];
        S.any_slack = any;
    }
    if (tid <= PMAX) { S.y[tid] = 0.0; S.dy[tid] = 0.0; S.rhs[tid] = 0.0; }
    for (int e = tid; e < PMAX * KL; e += GT) S.K[e] = (e / KL == e % KL) ? 1.0 : 0.0;      // rows / columns P.. of the factor's bound: identity
    for (int g = tid; g < NGR; g += GT) { ev[g] = 0.0; dev[g] = 0.0; Dg[g] = 1.0; iDg[g] = 0.0; qg[g] = 0.0; if (g >= n_obs * M) gact[g] = 0; }   // (groups without an active row stay like this)
    __syncthreads();

    const int tseg = uni_i(S.tseg);
    auto compute_x = [&](const double *yv, double *xv, bool with_const) {
        fresh();
        const int xk = tid < NV ? tid / SEGV : 0, xt = tid < NV ? tid % SEGV : 0;
        if (tid < NV) {
            double v = (xt < 3 && with_const) ? S.s0[xk][xt] : 0.0;
            if (xt >= 3)
                for (int j = 0; j < nya; j++) v += S.Z[xt][j] * yv[xk * nya + j];
            xv[tid] = v;
        }
    };
    // row bookkeeping ---------------------------------------------------------------------------------
    // kinds: axis slot sl in [0, AXROWS) (valid mask), slack sign rows US0 + j (DYNAMICALLIMIT), collision rows
    // CL0 + oi*27 + (cp-3) (segment < ncs), group sign rows GS0 + oi*M + m (slack obstacles, segment < ncs)
    auto coll_valid = [&](int c) { return cact[c] != 0; };                         // (segment < ncs and not pruned)
    auto grp_valid = [&](int g) { return slk[g / M] != 0 && gact[g] != 0; };
    // a_r . v for the three variable blocks (xv: control points, uv: explicit slack, gv: group slack)
    auto val_axis = [&](int sl, const double *xv, const double *uv) {
        const int type = sl / NV, kt = sl % NV, k = kt / SEGV, t = kt % SEGV;
        double v = ax_x(xv, type, k, t);
        if (nu && type >= 2) v += type < 4 ? hv * uv[t / NC] : ha * uv[M + t / NC];
        return v;
    };
    auto val_coll = [&](int c, const double *xv, const double *gv) {
        const int oi = c / NBK, cp = c % NBK + 3, m = cp / NC;
        const float *n = nrm + 3 * (oi * M + m);
        double v = -((double)n[0] * xv[cp] + (double)n[1] * xv[SEGV + cp] + (double)n[2] * xv[2 * SEGV + cp]);
        if (slk[oi]) v += gv[oi * M + m];
        return v;
    };
    int nrow_i = 0;
    for (int sl = tid; sl < AXROWS; sl += GT) nrow_i += S.avalid[sl];
    if (tid < nu) nrow_i++;
    for (int c = tid; c < NCL; c += GT) nrow_i += (c / NBK < n_obs && coll_valid(c)) ? 1 : 0;
    for (int g = tid; g < NGR; g += GT) nrow_i += (g / M < n_obs && grp_valid(g)) ? 1 : 0;
    block_reduce((double)nrow_i, 0, 0, 0, 0, 0, -1, -1, -1, -1);
    const double nrow = S.sc[0];
    double hmax = 1.0;
    {
        double hm = 1.0;
        for (int sl = tid; sl < AXROWS; sl += GT) if (S.avalid[sl]) hm = fmax(hm, fabs(S.ah[sl]));
        for (int c = tid; c < NCL; c += GT) if (c / NBK < n_obs && coll_valid(c)) hm = fmax(hm, fabs(crhs[c]));
        block_reduce(hm, 0, 0, 0, 0, 1, -1, -1, -1, -1);
        hmax = S.sc[0];
    }
    const double wg_base = 2.0 * a.slack_w / (double)M;       // Hessian of slack_w (M - m)/M eps^2 is 2 slack_w (M - m)/M
    auto cost_grad = [&](int xk, int xt) -> double {
        const double *xs = S.x + xk * SEGV + (xt / NC) * NC;
        double g = 0.0;
        for (int j = 0; j < NC; j++) g += S.Qh[(xt % NC) * NC + j] * xs[j];
        return g;
    };

    // x-space sums of a per-row coefficient (rt2 = vv, rz = z) and, with_w, of the weights rt1 = w: the only place where
    // the rows meet the unknowns.  Fixed summation orders: results do not depend on scheduling.
    auto reduce_rows = [&](bool with_w, bool unit_w) {
        fresh();
        const int xk = tid < NV ? tid / SEGV : 0, xt = tid < NV ? tid % SEGV : 0;
        if (tid < NV) {
            const int k = xk, t = xt, i = t % NC, b = tid;
            const int t1i = t >= 1 ? t - 1 : 0, t2i = t >= 2 ? t - 2 : 0;
            const int o0 = k * SEGV + t, o1 = k * SEGV + t1i, o2 = k * SEGV + t2i;
            auto V = [&](const double *arr, int type, int o) { return S.avalid[type * NV + o] ? arr[type * NV + o] : 0.0; };
            const double m1 = (t >= 1 && (t1i / NC == t / NC)) ? 1.0 : 0.0, m2 = (t >= 2 && (t2i / NC == t / NC)) ? 1.0 : 0.0;
            auto gather = [&](const double *arr) {
                return (V(arr, 0, o0) - V(arr, 1, o0)) + (V(arr, 3, o0) - V(arr, 2, o0)) + (V(arr, 4, o0) - V(arr, 5, o0)) +
                       m1 * ((V(arr, 2, o1) - V(arr, 3, o1)) - 2.0 * (V(arr, 4, o1) - V(arr, 5, o1))) + m2 * (V(arr, 4, o2) - V(arr, 5, o2));
            };
            double cg = cost_grad(xk, xt);
            if (i == DEG && t / NC >= M - tseg) cg += 2.0 * md.w_t * (S.x[b] - S.goal[k]);
            S.gv[b] = cg + gather(rt2);
            S.gz[b] = cg + gather(rz);
            if (with_w) {
                auto Wt = [&](int type, int o) { return S.avalid[type * NV + o] ? (unit_w ? 1.0 : rt1[type * NV + o]) : 0.0; };
                const double wB = Wt(0, o0) + Wt(1, o0), wV0 = Wt(2, o0) + Wt(3, o0), wA0 = Wt(4, o0) + Wt(5, o0);
                const double wV1 = m1 * (Wt(2, o1) + Wt(3, o1)), wA1 = m1 * (Wt(4, o1) + Wt(5, o1)), wA2 = m2 * (Wt(4, o2) + Wt(5, o2));
                S.Wd[b] = wB + wV0 + wV1 + wA0 + 4.0 * wA1 + wA2;
                S.W1[b] = -wV0 - 2.0 * wA0 - 2.0 * wA1;
                S.W2[b] = wA0;
            }
        }
        // collision rows per control point: unit (cpi, component) x 3 obstacle stripes, combined in a fixed order
        if (tid < 3 * NBK * 3) {
            const int part = tid / (NBK * 3), u = tid % (NBK * 3), cpi = u / 3, k = u % 3, m = (cpi + 3) / NC;
            double sv = 0.0, sz = 0.0;
            if (m < ncs)
                for (int oi = part; oi < n_obs; oi += 3) {
                    const double nk = (double)nrm[3 * (oi * M + m) + k];
                    const int r = CL0 + oi * NBK + cpi;
                    sv += rt2[r] * nk; sz += rz[r] * nk;
                }
            S.part[part][u][0] = sv; S.part[part][u][1] = sz;
        }
        if (with_w && tid < NBK * 6) {
            const int cpi = tid / 6, c = tid % 6, m = (cpi + 3) / NC;
            const int ia = c < 3 ? 0 : (c < 5 ? 1 : 2), ib = c < 3 ? c : (c < 5 ? c - 2 : 2);
            double acc = 0.0;
            if (m < ncs)
                for (int oi = 0; oi < n_obs; oi++) {
                    const float *n = nrm + 3 * (oi * M + m);
                    const double w = unit_w ? (double)cact[oi * NBK + cpi] : rt1[CL0 + oi * NBK + cpi];
                    acc += w * (double)n[ia] * (double)n[ib];
                }
            S.Ws[(cpi + 3) * 6 + c] = acc;
        }
        __syncthreads();
        if (tid < NBK * 3) {
            const int cp = tid / 3 + 3, k = tid % 3;
            S.Tv[cp * 3 + k] = -((S.part[0][tid][0] + S.part[1][tid][0]) + S.part[2][tid][0]);
            S.Tz[cp * 3 + k] = -((S.part[0][tid][1] + S.part[1][tid][1]) + S.part[2][tid][1]);
        }
        // explicit slack variables (DYNAMICALLIMIT): gradient entries and, with_w, their Hessian row
        if (nu && tid < nu) {
            const int j = tid, m = j % M, isacc = j >= M;
            const double cu = isacc ? ha : hv;
            double gvv = 0.0, gzz = 0.0, huu = 0.0;
            for (int k = 0; k < 3; k++)
                for (int i = 0; i < NC; i++)
                    for (int sg = 0; sg < 2; sg++) {
                        const int sl = ((isacc ? 4 : 2) + sg) * NV + k * SEGV + m * NC + i;
                        if (!S.avalid[sl]) continue;
                        gvv += rt2[sl] * cu; gzz += rz[sl] * cu;
                        if (with_w) huu += (unit_w ? 1.0 : rt1[sl]) * cu * cu;
                    }
            const double hq = wg_base * (double)(M - m);
            S.gu[j] = hq * S.y[P0 + j] + gzz + rz[US0 + j];                 // stationarity residual of u_j
            S.qu[j] = -(hq * S.y[P0 + j] + gvv + rt2[US0 + j]);
            if (with_w) S.Huu[j] = hq + huu + (unit_w ? 1.0 : rt1[US0 + j]);
        }
        if (nu && with_w) {
            for (int e = tid; e < nu * NV; e += GT) {
                const int j = e / NV, b = e % NV, k = b / SEGV, t = b % SEGV, m = j % M, isacc = j >= M;
                double acc = 0.0;
                if (t / NC == m) {
                    const int ty = isacc ? 4 : 2, i = t % NC;
                    auto Wt = [&](int type, int tt) {
                        const int sl = type * NV + k * SEGV + tt;
                        return (tt / NC == m && tt >= 0 && S.avalid[sl]) ? (unit_w ? 1.0 : rt1[sl]) : 0.0;
                    };
                    if (!isacc) {
                        // rows +-(x[t+1] - x[t]) + hv u: coefficient on x[t] is -+1 (row starting at t), +-1 (row starting at t-1)
                        acc = -(Wt(ty, t) - Wt(ty + 1, t)) + (i >= 1 ? (Wt(ty, t - 1) - Wt(ty + 1, t - 1)) : 0.0);
                    } else {
                        acc = (Wt(ty, t) - Wt(ty + 1, t)) - (i >= 1 ? 2.0 * (Wt(ty, t - 1) - Wt(ty + 1, t - 1)) : 0.0) +
                              (i >= 2 ? (Wt(ty, t - 2) - Wt(ty + 1, t - 2)) : 0.0);
                    }
                    acc *= isacc ? ha : hv;
                }
                S.Wu[j][b] = acc;
            }
        }
        // group slack variables: diagonal D_g, right-hand side q_g, stationarity residual (kept in dev for the test)
        for (int g = tid; g < NGR; g += GT) {
            if (!(g / M < n_obs) || !grp_valid(g)) continue;
            const int oi = g / M, m = g % M;
            double sw = 0.0, svv = 0.0, szz = 0.0;
            for (int i = 0; i < NC; i++) {
                const int cp = m * NC + i;
                if (cp < 3) continue;
                const int r = CL0 + oi * NBK + cp - 3;
                sw += unit_w ? (double)cact[r - CL0] : rt1[r]; svv += rt2[r]; szz += rz[r];
            }
            const double hq = wg_base * (double)(M - m);
            if (with_w) { const double dg = hq + sw + (unit_w ? 1.0 : rt1[GS0 + g]); Dg[g] = dg; iDg[g] = 1.0 / dg; }
            qg[g] = -(hq * ev[g] + svv + rt2[GS0 + g]);
            dev[g] = hq * ev[g] + szz + rz[GS0 + g];
        }
        __syncthreads();
        // elimination of the group slack variables: C_m = sum_g m_g m_g^T / D_g and c_q = sum_g m_g q_g / D_g in x-space,
        // m_g = -w_r n_g at the control points of segment m
        if (with_w && S.any_slack)
            for (int e = tid; e < M * SEG_E; e += GT) {
                const int m = e / SEG_E;
                int r = e % SEG_E, p = 0;
                while (r >= 18 - p) { r -= 18 - p; p++; }
                const int q = p + r;                                   // p <= q in 0..17 : (i, k) = (p / 3, p % 3)
                const int i1 = p / 3, k1 = p % 3, i2 = q / 3, k2 = q % 3;
                double acc = 0.0;
                if (m < ncs && m * NC + i1 >= 3 && m * NC + i2 >= 3)
#pragma unroll 4
                    for (int oi = 0; oi < n_obs; oi++) {                  // branch-free: the loads of several obstacles in flight
                        const float *n = nrm + 3 * (oi * M + m);
                        const double w1 = unit_w ? (double)cact[oi * NBK + m * NC + i1 - 3] : rt1[CL0 + oi * NBK + m * NC + i1 - 3];
                        const double w2 = unit_w ? (double)cact[oi * NBK + m * NC + i2 - 3] : rt1[CL0 + oi * NBK + m * NC + i2 - 3];
                        const double t = w1 * w2 * (double)n[k1] * (double)n[k2] * iDg[oi * M + m];
                        acc += slk[oi] ? t : 0.0;
                    }
                S.Cm[m][e % SEG_E] = acc;
            }
        if (tid < NBK * 3) {
            const int cp = tid / 3 + 3, k = tid % 3, m = cp / NC;
            double acc = 0.0;
            if (m < ncs && S.any_slack)
#pragma unroll 4
                for (int oi = 0; oi < n_obs; oi++) {
                    const int g = oi * M + m;
                    const double w = unit_w ? (double)cact[oi * NBK + cp - 3] : rt1[CL0 + oi * NBK + cp - 3];
                    const double t = -w * (double)nrm[3 * g + k] * qg[g] * iDg[g];
                    acc += slk[oi] ? t : 0.0;
                }
            S.cq[cp * 3 + k] = acc;
        }
        __syncthreads();
    };
    auto seg_c = [&](int m, int i1, int k1, int i2, int k2) {
        int p = i1 * 3 + k1, q = i2 * 3 + k2;
        if (p > q) { const int t = p; p = q; q = t; }
        return S.Cm[m][p * 18 - p * (p - 1) / 2 + (q - p)];
    };
    // dense reduced system: K (lower triangle) and rhs = q_y - sum_g m_g q_g / D_g ; stationarity residual in dy
    auto assemble = [&](bool with_k) {
        fresh();
        if (with_k && tid >= P && tid < PMAX) S.K[tid * KL + tid] = 1.0;     // identity beyond P (the factor left its L there: zero)
        if (with_k)
            for (int e = tid; e < P * (P + 1) / 2; e += GT) {       // lower triangle, row-major: e = r (r + 1) / 2 + c
                int r = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
                if (r * (r + 1) / 2 > e) r--;
                else if ((r + 1) * (r + 2) / 2 <= e) r++;
                const int c = e - r * (r + 1) / 2;
                double v = 0.0;
                if (r < P0 && c < P0) {
                    const int k = r / nya, aa = r % nya, kk = c / nya, bb = c % nya;
                    const int sc6 = k == kk ? (k == 0 ? 0 : (k == 1 ? 3 : 5)) : ((k < kk ? k : kk) == 0 ? (k + kk) : 4);   // xx xy xz yy yz zz
                    if (k == kk) v = S.Hc[aa * GNYA + bb];
                    for (int pz = 0; pz < S.tn[aa]; pz++) {
                        const int t = S.tt[aa][pz];
                        const double za = S.tc[aa][pz];
                        const int m = t / NC, i = t % NC;
                        double row = S.Ws[t * 6 + sc6] * S.Z[t][bb];
                        if (k == kk) {
                            double dg = S.Wd[k * SEGV + t];
                            if (i == DEG && m >= M - tseg) dg += 2.0 * md.w_t;
                            row += dg * S.Z[t][bb];
                            if (i + 1 < NC) row += S.W1[k * SEGV + t] * S.Z[t + 1][bb];
                            if (i >= 1) row += S.W1[k * SEGV + t - 1] * S.Z[t - 1][bb];
                            if (i + 2 < NC) row += S.W2[k * SEGV + t] * S.Z[t + 2][bb];
                            if (i >= 2) row += S.W2[k * SEGV + t - 2] * S.Z[t - 2][bb];
                        }
                        if (m < ncs && S.any_slack)
                            for (int i2 = 0; i2 < NC; i2++) {
                                const double zb = S.Z[m * NC + i2][bb];
                                if (zb != 0.0 && m * NC + i2 >= 3) row -= seg_c(m, i, k, i2, kk) * zb;
                            }
                        v += za * row;
                    }
                } else if (r >= P0 && c < P0) {
                    const int j = r - P0, kk = c / nya, bb = c % nya;
                    for (int pz = 0; pz < S.tn[bb]; pz++) v += S.tc[bb][pz] * S.Wu[j][kk * SEGV + S.tt[bb][pz]];
                } else if (r == c) {
                    v = S.Huu[r - P0];
                }
                S.K[r * KL + c] = v;
                S.K[c * KL + r] = v;
            }
        if (tid < P0) {
            const int k = tid / nya, aa = tid % nya;
            double r = 0.0, rdv = 0.0;
            for (int pz = 0; pz < S.tn[aa]; pz++) {
                const int t = S.tt[aa][pz];
                const double za = S.tc[aa][pz];
                r += za * (S.gv[k * SEGV + t] + S.Tv[t * 3 + k] + S.cq[t * 3 + k]);
                rdv += za * (S.gz[k * SEGV + t] + S.Tz[t * 3 + k]);
            }
            S.rhs[tid] = -r;
            S.dy[tid] = rdv;
        } else if (tid < P) {
            S.rhs[tid] = S.qu[tid - P0];
            S.dy[tid] = S.gu[tid - P0];
        }
        __syncthreads();
    };
    // dense K = L D L^T, then L D L^T dy = rhs: wave 0, out of line (dense_factor_w0 / dense_solve_w0 above)
    auto factor = [&]() -> bool {
        if (wave == 0) {
            if (P <= P_STOP) dense_factor_w0<P_STOP>();
            else if (P <= P_FREE) dense_factor_w0<P_FREE>();
            else dense_factor_w0<PMAX>();
        }
        __syncthreads();
        return S.ok != 0;
    };
    auto solve = [&]() {
        fresh();
        if (wave == 0) {
            long long t0 = 0;
            if (gp) t0 = (long long)__builtin_readcyclecounter();
            if (P <= P_STOP) dense_solve_w0<P_STOP>(P);
            else if (P <= P_FREE) dense_solve_w0<P_FREE>(P);
            else dense_solve_w0<PMAX>(P);
            if (gp && tid == 0) gp[14] += (long long)__builtin_readcyclecounter() - t0;
        }
        __syncthreads();
        compute_x(S.dy, S.dx, false);
        __syncthreads();
        // back-substitution of the eliminated group slack variables: de_g = (q_g - m_g . dx) / D_g
        for (int g = tid; g < NGR; g += GT) {
            if (!(g / M < n_obs) || !grp_valid(g)) { dev[g] = 0.0; continue; }
            const int oi = g / M, m = g % M;
            const float *n = nrm + 3 * g;
            double mdx = 0.0;
            for (int i = 0; i < NC; i++) {
                const int cp = m * NC + i;
                if (cp < 3) continue;
                const double w = S.sc[7] != 0.0 ? (double)cact[oi * NBK + cp - 3] : rt1[CL0 + oi * NBK + cp - 3];
                mdx += -w * ((double)n[0] * S.dx[cp] + (double)n[1] * S.dx[SEGV + cp] + (double)n[2] * S.dx[2 * SEGV + cp]);
            }
            dev[g] = (qg[g] - mdx) / Dg[g];
        }
        __syncthreads();
    };
    // generic sweep over all valid rows: f(row index r, value a_r.v at (xv, uv, gv), value at the step, rhs h)
    auto for_rows = [&](auto &&f) {
        fresh();
        for (int sl = tid; sl < AXROWS; sl += GT)
            if (S.avalid[sl]) f(sl, val_axis(sl, S.x, S.y + P0), val_axis(sl, S.dx, S.dy + P0), S.ah[sl]);
        if (tid < nu) f(US0 + tid, S.y[P0 + tid], S.dy[P0 + tid], 0.0);
        for (int c = tid; c < NCL; c += GT)
            if (c / NBK < n_obs && coll_valid(c)) f(CL0 + c, val_coll(c, S.x, ev), val_coll(c, S.dx, dev), -crhs[c]);
        for (int g = tid; g < NGR; g += GT)
            if (g / M < n_obs && grp_valid(g)) f(GS0 + g, ev[g], dev[g], 0.0);
    };
    auto objective = [&]() -> double {
        fresh();
        const int xk = tid < NV ? tid / SEGV : 0, xt = tid < NV ? tid % SEGV : 0;
        double o = 0.0;
        if (tid < NV && !(dim2 && xk == 2)) {
            o = 0.5 * cost_grad(xk, xt) * S.x[tid];
            if (xt % NC == DEG && xt / NC >= M - tseg) { const double e = S.x[tid] - S.goal[xk]; o += md.w_t * e * e; }
        }
        if (tid < nu) o += 0.5 * wg_base * (double)(M - tid % M) * S.y[P0 + tid] * S.y[P0 + tid];
        for (int g = tid; g < NGR; g += GT)
            if (g / M < n_obs && grp_valid(g)) o += 0.5 * wg_base * (double)(M - g % M) * ev[g] * ev[g];
        return o;
    };

    // ------------------------------------------------------------------ starts
    // Warm (from the third tick on): y = free control points of the shifted previous plan, every row centred on mu0 -- the
    // start of lsc_plan_kernel -- with the cold start (least-squares point, then shift) as fallback; cold only otherwise.
    gstamp(GP_SETUP);
    int status = LSC_STATUS_INFEASIBLE_K, iters = 0, spent = 0;
    double obj = 0.0;
    bool can = true;
    if (a.goal_err && a.goal_err[qi] != 0) { status = LSC_STATUS_GOAL_K; can = false; }
    else if (a.sfc_err && a.sfc_err[qi] != 0) { status = LSC_STATUS_SFC_K; can = false; }
    const bool try_warm = md.ws_mu0 > 0.0 && a.planner_seq >= 2;
    for (int attempt = try_warm ? 0 : 1; can && attempt < 2; attempt++) {
        fresh();
        bool run = true;
        iters = 0;
        if (attempt == 0) {
            const double mu0 = md.ws_mu0, smin = sqrt(mu0);
            if (tid < P0) {
                const int k = tid / nya, aa = tid % nya;
                const int t = aa < NYL ? (aa / 3) * NC + 3 + aa % 3 : (M - 1) * NC + 3 + (aa - NYL);
                const int m = t / NC, i = t % NC;
                const float *tp = a.traj_prev + (size_t)qi * NV + k * SEGV;
                S.y[tid] = (dim2 && k == 2) ? md.z2d : (double)((m < M - 1) ? tp[(m + 1) * NC + i] : tp[(M - 1) * NC + DEG]);
            } else if (tid < P) S.y[tid] = 0.0;
            for (int g = tid; g < NGR; g += GT) ev[g] = 0.0;
            __syncthreads();
            compute_x(S.y, S.x, true);
            if (tid == 0) S.sc[7] = 0.0;
            __syncthreads();
            for_rows([&](int r, double av, double, double h) {
                const int type = r < AXROWS ? r / NV : 0;
                const double floor_s = type < 2 ? smin : (type < 4 ? smin * hv : smin * ha);
                const double sv = fmax(h - av, floor_s);
                rs[r] = sv; rz[r] = mu0 / sv;
            });
            __syncthreads();
        } else {
            if (gp && tid == 0) gp[15]++;
            if (tid <= PMAX) { S.y[tid] = 0.0; S.dy[tid] = 0.0; }
            for (int g = tid; g < NGR; g += GT) { ev[g] = 0.0; dev[g] = 0.0; }
            __syncthreads();
            if (run) {
                compute_x(S.y, S.x, true);
                __syncthreads();
                for_rows([&](int r, double av, double, double h) { rt2[r] = av - h; rz[r] = 0.0; });
                if (tid == 0) S.sc[7] = 1.0;                                  // unit weights in solve()'s back-substitution
                __syncthreads();
                reduce_rows(true, true);
                assemble(true);
                if (!factor()) run = false;
            }
            if (run) {
                solve();
                if (tid < P) S.y[tid] = S.dy[tid];
                for (int g = tid; g < NGR; g += GT) ev[g] = dev[g];
                __syncthreads();
                compute_x(S.y, S.x, true);
                __syncthreads();
                double mins = 1e300, minz = 1e300;
                for_rows([&](int r, double av, double, double h) {
                    const double sl = h - av;
                    rs[r] = sl; rz[r] = -sl;
                    mins = fmin(mins, sl); minz = fmin(minz, -sl);
                });
                block_reduce(mins, minz, 0, 0, 0, 2, 2, -1, -1, -1);
                const double shs = S.sc[0] <= 0.0 ? 1.0 - S.sc[0] : 0.0, shz = S.sc[1] <= 0.0 ? 1.0 - S.sc[1] : 0.0;
                for_rows([&](int r, double, double, double) { rs[r] += shs; rz[r] += shz; });
                if (tid == 0) S.sc[7] = 0.0;
                __syncthreads();
            }

        }

        // -------------------------------------------------------------- Mehrotra predictor-corrector
        gstamp(GP_START);
        const int max_iters = attempt == 0 ? 30 : 80;
        while (run) {
            if (iters >= max_iters) break;
            // residuals, weights, predictor right-hand side
            double gpart = 0.0, rpm = 0.0;
            for_rows([&](int r, double av, double, double h) {
                const double sv = rs[r], zv = rz[r];
                const double rp = av + sv - h, w = zv / sv;
                rt1[r] = w; rt2[r] = w * rp;
                gpart += sv * zv; rpm = fmax(rpm, fabs(rp));
            });
            block_reduce(gpart, rpm, objective(), 0, 0, 0, 1, 0, -1, -1);
            gstamp(GP_RESID);
            const double gap = S.sc[0], rpmax = S.sc[1];
            obj = S.sc[2];
            const double mu = gap / nrow;
            const bool gap_ok = gap <= 1e-9 * (1.0 + fabs(obj));
            const bool tracing = a.trace && qi == a.trace_agent && tid == 0 && iters + spent < 64;       // lsc_solver_trace
            if (tracing) { double *tr = a.trace + (iters + spent) * 8; tr[0] = gap; tr[1] = rpmax; tr[2] = obj; tr[3] = -1; tr[5] = -1; tr[7] = mu; }
            if (!(gap == gap) || !(rpmax == rpmax)) break;
            reduce_rows(true, false);
            gstamp(GP_REDUCE);
            assemble(true);
            gstamp(GP_ASSEMBLE);
            {
                double rda = tid < P ? fabs(S.dy[tid]) : 0.0;
                for (int g = tid; g < NGR; g += GT)
                    if (g / M < n_obs && grp_valid(g)) rda = fmax(rda, fabs(dev[g]));
                block_reduce(rda, 0, 0, 0, 0, 1, -1, -1, -1, -1);
                if (rpmax <= 1e-9 * hmax && gap_ok && S.sc[0] <= 1e-5 * (1.0 + fabs(obj))) { status = LSC_STATUS_OK_K; break; }
            }
            const bool fok = factor();
            gstamp(GP_FACTOR);
            if (!fok) {
                if (rpmax <= 1e-8 * hmax && gap <= 1e-7 * (1.0 + fabs(obj))) status = LSC_STATUS_OK_K;
                break;
            }
            solve();
            gstamp(GP_SOLVE);
            // affine step length and centring statistics
            double amin = 1.0, s1 = 0.0, s2 = 0.0;
            for_rows([&](int r, double av, double adv, double h) {
                const double sv = rs[r], zv = rz[r], w = rt1[r];
                const double rp = av + sv - h;
                const double ds = -rp - adv, dz = -zv - w * ds;
                if (ds < 0.0) amin = fmin(amin, -sv / ds);
                if (dz < 0.0) amin = fmin(amin, -zv / dz);
                s1 += sv * dz + zv * ds; s2 += ds * dz;
                rt2[r] = ds * dz;
            });
            const double dxa = tid < NV ? fabs(S.dx[tid]) : 0.0, xa = tid < NV ? fabs(S.x[tid]) : 0.0;
            block_reduce(amin, s1, s2, dxa, xa, 2, 0, 0, 1, 1);
            const double aaff = S.sc[0], ss1 = S.sc[1], ss2 = S.sc[2], dxn = S.sc[3], xn = S.sc[4];
            gstamp(GP_AFFINE);
            if (rpmax <= 1e-9 * hmax && gap_ok && dxn <= 1e-9 * fmax(1.0, xn)) { status = LSC_STATUS_OK_K; break; }
            const double mu_aff = (gap + aaff * ss1 + aaff * aaff * ss2) / nrow;
            double sigma = mu > 0.0 ? mu_aff / mu : 0.0;
            sigma = sigma * sigma * sigma;
            const double smu = sigma * mu;
            if (tracing) { double *tr = a.trace + (iters + spent) * 8; tr[3] = aaff; tr[4] = sigma; tr[6] = dxn; }
            // corrector right-hand side, same factor
            for_rows([&](int r, double av, double, double h) {
                const double sv = rs[r];
                const double rp = av + sv - h;
                rt2[r] = rt1[r] * rp - (rt2[r] - smu) / sv;
            });
            __syncthreads();
            gstamp(GP_CORR_RHS);
            reduce_rows(false, false);
            gstamp(GP_REDUCE2);
            assemble(false);
            gstamp(GP_ASSEMBLE2);
            solve();
            gstamp(GP_SOLVE);
            double amax = 1e300;
            for_rows([&](int r, double av, double adv, double h) {
                const double sv = rs[r], zv = rz[r], w = rt1[r];
                const double rp = av + sv - h;
                const double ds = -rp - adv, dz = -zv + rt2[r] + w * adv;
                if (ds < 0.0) amax = fmin(amax, -sv / ds);
                if (dz < 0.0) amax = fmin(amax, -zv / dz);
                rt1[r] = ds; rt2[r] = dz;
            });
            block_reduce(amax, 0, 0, 0, 0, 2, -1, -1, -1, -1);
            // fraction to the boundary like the fast path's: the closer the affine step came to a full step, the closer the combined
            // step may go to the boundary (the last iterations then converge faster than the factor 100 a fixed 0.99 allows)
            const double tau = fmin(1.0 - 1e-5, fmax(0.99, aaff));
            const double alpha = fmin(1.0, tau * S.sc[0]);
            if (tracing) a.trace[(iters + spent) * 8 + 5] = alpha;
            for_rows([&](int r, double, double, double) { rs[r] += alpha * rt1[r]; rz[r] += alpha * rt2[r]; });
            if (tid < P) S.y[tid] += alpha * S.dy[tid];
            for (int g = tid; g < NGR; g += GT) ev[g] += alpha * dev[g];
            __syncthreads();
            compute_x(S.y, S.x, true);
            __syncthreads();
            gstamp(GP_STEP);
            if (gp && tid == 0) gp[GP_ITERS]++;
            iters++;
        }

        if (status == LSC_STATUS_OK_K) break;
        spent += iters;
        iters = 0;
    }
    iters += spent;

    // ------------------------------------------------------------------ output (same conventions as lsc_plan_kernel)
    fresh();
    float *out = a.traj_next + (size_t)qi * NV;
    float *stale = a.stale + (size_t)qi * NV;
    __syncthreads();
    if (tid < NV) {
        if (status == LSC_STATUS_OK_K) {
            float v = (float)S.x[tid];
            if (dim2 && tid >= 2 * SEGV) v = (float)md.z2d;   // src/traj_optimizer.cpp:87-90
            out[tid] = v; stale[tid] = v;
        }
        else out[tid] = stale[tid];
    }
    if (a.state_next && tid < 3) {
#pragma clang fp contract(off)
        const int k = tid;
        float c0, c1, c2;
        if (status == LSC_STATUS_OK_K) {
            c0 = (float)S.x[k * SEGV + NC]; c1 = (float)S.x[k * SEGV + NC + 1]; c2 = (float)S.x[k * SEGV + NC + 2];
            if (dim2 && k == 2) c0 = c1 = c2 = (float)md.z2d;
        } else { c0 = stale[k * SEGV + NC]; c1 = stale[k * SEGV + NC + 1]; c2 = stale[k * SEGV + NC + 2]; }
        const float fn = (float)DEG, fn1 = (float)(DEG - 1), finv = a.finv;
        const float v0 = ((c1 - c0) * fn) * finv, v1 = ((c2 - c1) * fn) * finv, a0 = ((v1 - v0) * fn1) * finv;
        a.state_next[9 * qi + k] = c0; a.state_next[9 * qi + 3 + k] = v0; a.state_next[9 * qi + 6 + k] = a0;
    }
    if (gp && tid == 0) gp[GP_AGENTS]++;
    if (tid == 0) {
        if (status == LSC_STATUS_OK_K) a.cost[qi] = obj;
        a.status[qi] = status;
        a.iters[qi] = iters;
        if (a.iters_acc) { a.iters_acc[qi] += iters; a.iters_acc[a.N + qi] += (long long)iters * ((long long)nrow - md.n_ax - nu); }
        if (a.nrows) a.nrows[qi] = (int)nrow - md.n_ax - nu;  // collision rows + group sign rows
    }
    __syncthreads();
    return true;
}

// The agents of one workgroup, out of line.  (Round 2 copied the argument block into private memory here: 2.2 KB of scratch per
// lane, 50 MB of writes per launch in the PMC counters.)
static __device__ __forceinline__ void general_entry(KArgs *ka, unsigned char *smem_raw)
{
    KArgs &a = *ka;
#ifdef LSC_POISON_LDS
    // debugging aid (not built into the product, see lsc_kernels.hip): the workgroup's LDS starts as 0xff bytes
    for (size_t i = threadIdx.x; i < (gs_bytes() + ws_lds_bytes(a.N)) / 4; i += GT) reinterpret_cast<uint32_t *>(smem_raw)[i] = 0xffffffffu;
    __syncthreads();
#endif
    unsigned char *ws = a.gen_ws + (size_t)blockIdx.x * a.gen_stride;
    for (int al = blockIdx.x; al < a.count; al += gridDim.x) {
        if (a.status[a.first + al] != LSC_STATUS_GENERAL_K) continue;
        __syncthreads();
        if (!general_agent<true>(a, al, smem_raw, ws, smem_raw + gs_bytes(), ws_lds_bytes(a.N))) {
            __syncthreads();
            general_agent<false>(a, al, smem_raw, ws, smem_raw + gs_bytes(), ws_lds_bytes(a.N));
        }
        __syncthreads();
    }
}

// Most launches of this kernel find nobody flagged (it follows the plan kernel whenever the disturbance checks are on):
// that case must cost a launch and nothing else.  The argument block is therefore read in place, through the kernarg
// segment pointer -- naming the by-value parameter would make the compiler copy all of it into scratch in the prologue,
// before the exit test.
__global__ __launch_bounds__(GT) void lsc_general_kernel(PlanArgs)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
#if defined(__HIP_DEVICE_COMPILE__)
    KArgs *ka = (KArgs *)__builtin_amdgcn_kernarg_segment_ptr();
#else
    KArgs *ka = nullptr;                                                             // (host pass of the single-source build)
#endif
    bool work = false;
    for (int al = blockIdx.x; al < ka->count; al += gridDim.x) work |= ka->status[ka->first + al] == LSC_STATUS_GENERAL_K;
    if (!work) return;
    general_entry(ka, smem_raw);
}

// The same for a batch of independent swarms (blockIdx.y = swarm; lsc_kernels.h: PlanBatch).
__global__ __launch_bounds__(GT) void lsc_general_batch_kernel(PlanBatch)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
#if defined(__HIP_DEVICE_COMPILE__)
    KArgs *ka = (KArgs *)__builtin_amdgcn_kernarg_segment_ptr() + blockIdx.y;
#else
    KArgs *ka = nullptr;
#endif
    bool work = false;
    for (int al = blockIdx.x; al < ka->count; al += gridDim.x) work |= ka->status[ka->first + al] == LSC_STATUS_GENERAL_K;
    if (!work) return;
    general_entry(ka, smem_raw);
}

size_t general_ws_bytes(int N) { return ws_bytes_of(N); }

size_t general_smem_bytes() { return gs_bytes(); }

hipError_t init_device_general_kernel()
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&lsc_general_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(&lsc_general_batch_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

hipError_t launch_general(const PlanArgs &a, int slots, hipStream_t st)
{
    if (a.count == 0 || slots < 1 || !a.gen_ws) return hipSuccess;
    const int grid = a.count < slots ? a.count : slots;
    const size_t smem = gs_bytes() + ws_lds_bytes(a.N);
    hipLaunchKernelGGL(lsc_general_kernel, dim3(grid), dim3(GT), smem, st, a);
    return hipGetLastError();
}

// n swarms of the same size class in one launch; every swarm's workgroups use ITS workspace (gen_ws of its own block), so `slots`
// is the smallest slot count among them
hipError_t launch_general_batch(const PlanArgs *a, int n, int slots, hipStream_t st)
{
    if (n < 1 || n > PLAN_BATCH_MAX || slots < 1) return hipErrorInvalidValue;
    PlanBatch b;
    int grid = 0, Nmax = 0;
    for (int i = 0; i < n; i++) {
        if (!a[i].gen_ws) return hipErrorInvalidValue;
        b.a[i] = a[i];
        const int g = a[i].count < slots ? a[i].count : slots;
        grid = g > grid ? g : grid;
        Nmax = a[i].N > Nmax ? a[i].N : Nmax;
    }
    for (int i = n; i < PLAN_BATCH_MAX; i++) { b.a[i] = a[0]; b.a[i].count = 0; }
    if (grid == 0) return hipSuccess;
    const size_t smem = gs_bytes() + ws_lds_bytes(Nmax);
    hipLaunchKernelGGL(lsc_general_batch_kernel, dim3(grid, n), dim3(GT), smem, st, b);
    return hipGetLastError();
}

}  // namespace lsc
