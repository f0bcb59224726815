// lsc_model.hpp -- problem constants shared by host and device.
//
// What TrajOptimizer builds once per agent (buildQBase / buildAeqBase, src/traj_optimizer.cpp:169-236)
// is the same for every agent and tick, so it is built ONCE per context on the host and kept in HBM.
// The 51 equality rows (initial state, C^2 continuity at 4 junctions, stop-at-horizon:
// src/traj_optimizer.cpp:394-405, 527-536) are eliminated analytically instead of being handed to a
// solver: per axis the 13 free variables are
//     y[3m + (i-3)] = c_{m,i}  for m = 0..3, i = 3..5,      y[12] = c_{4,3} = c_{4,4} = c_{4,5}
// and the other control points follow from
//     c_{0,0} = p,  c_{0,1} = p + v dt/n,  c_{0,2} = a dt^2/(n(n-1)) + 2 c_{0,1} - c_{0,0}
//     c_{m,0} = c_{m-1,5},  c_{m,1} = 2 c_{m-1,5} - c_{m-1,4},  c_{m,2} = 4 c_{m-1,5} - 4 c_{m-1,4} + c_{m-1,3}
// (same feasible set as the reference's equality rows; 39 unknowns per agent instead of 90 + 51 duals).
#pragma once
#include <stdint.h>

namespace lsc {

// Segments of a plan: M = horizon / dt is a run-time number in the reference (src/traj_optimizer.cpp:9, src/traj_planner.cpp:22);
// here it is a build parameter -- every array below is sized by it and the factorisation is unrolled for it.  Two instantiations
// are built: M = 5 (every shipped launch file: dt 0.2, horizon 1.0) -> liblsc_hip.so, and M = 4 (the C++ defaults of
// src/param.cpp:66-67: dt 0.5, horizon 2.0) -> liblsc_hip_m4.so, from the same sources with -DLSC_SEGMENTS=4.
#ifndef LSC_SEGMENTS
#define LSC_SEGMENTS 5
#endif
constexpr int M = LSC_SEGMENTS, DEG = 5, NC = 6, SEGV = M * NC, NV = 3 * SEGV;
static_assert(M >= 3 && M <= 5, "built and tested for 3 <= M <= 5 (the LDS tables and the twisted factorisation are sized for it)");
constexpr int NYL = 3 * (M - 1);   // free variables of an axis inside the clusters: c_{m,3..5} for m < M - 1
constexpr int NYA = NYL + 1;       // + c_{M-1,3} (= c_{M-1,4} = c_{M-1,5}: stop at the horizon)
constexpr int NYC = 9 * (M - 1);   // cluster part of the global order: (M - 1) clusters of 3 axes x 3 locals, then the 3 last unknowns
constexpr int NY = 3 * NYA;        // free variables per agent (39 for M = 5)
constexpr int NCP = SEGV;          // control points per agent (all but 3 carry constraints)
constexpr int KLD = NY + 2 - (NY % 2 == 0);   // leading dimension of the NY x NY matrices in LDS, odd (bank spread): 41 for NY = 39
constexpr int BAND = 11;           // half bandwidth of the reduced Hessian in cluster-major order (9 + 2, whatever M is)
constexpr int AXROWS = 6 * NV;     // bound(2) / velocity(2) / acceleration(2) row slots per variable
// rows that exist (src/traj_optimizer.cpp:274-303, 468-525): bounds on all but c_{0,0..2}; 5 M - 2 velocity and 4 M - 1 acceleration
// differences per axis, two signs each
constexpr int AXVALID_3D = 2 * 3 * (SEGV - 3) + 2 * 3 * (5 * M - 2) + 2 * 3 * (4 * M - 1);
constexpr int AXVALID_2D = AXVALID_3D / 3 * 2;
static_assert(M != 5 || (NY == 39 && KLD == 41 && AXVALID_3D == 414 && AXVALID_2D == 276), "the M = 5 layout of rounds 1-3");

// global (cluster-major) index of free variable a of axis k: clusters of 3 axes x 3 locals
__host__ __device__ inline int yglob(int k, int a) { return a < NYL ? (a / 3) * 9 + k * 3 + (a % 3) : NYC + k; }
// ... and back: axis and per-axis index of global unknown g
__host__ __device__ inline int yaxis(int g) { return g < NYC ? (g % 9) / 3 : g - NYC; }
__host__ __device__ inline int yvar(int g) { return g < NYC ? (g / 9) * 3 + (g % 3) : NYL; }

struct Model {
    double dt, w_c, w_t;
    double hv_scale;          // dt / n          : velocity rows are kept as  +-(c_{i+1}-c_i) <= vmax dt/n
    double ha_scale;          // dt^2 / (n(n-1)) : acceleration rows as +-(c_{i+2}-2c_{i+1}+c_i) <= amax dt^2/(n(n-1))
    double Qh[NC * NC];       // 2 w_c Q_base : Hessian block of one segment (objective has no 1/2)
    double Hc[NYA * NYA];     // Z^T blockdiag(Qh) Z for one axis
    // x_t = sum_j xc[t][j] * y[xi[t][j]]   (t >= 3);  x_0..2 come from the state
    int    x_n[SEGV];
    int    x_i[SEGV][3];
    double x_c[SEGV][3];
    // y_a enters x_t with coefficient t_c  (transpose of the map above)
    int    t_n[NYA];
    int    t_t[NYA][4];
    double t_c[NYA][4];
    // Hessian assembly terms: K[dest] += coef * W[src]; packed dest(11b) | src(10b) | coef+128 (8b), grouped by
    // destination; entry e owns terms [k_off[e], k_off[e+1])
    int    n_entries;         // lower-band entries of the 39x39 matrix
    int    n_terms;
    float  world_min[3], world_max[3];
    int    use_sfc, prune, max_iters, cap;
    double dx_tol;            // Newton-step convergence tolerance (relative to max(1, |x|_inf))
    double gap_tol;           // duality-gap tolerance, relative to 1 + |objective|
    double ws_mu0;            // warm start: initial complementarity target (0 = cold start only)
    int sigma_pow;            // Mehrotra centering exponent: sigma = (mu_aff / mu)^sigma_pow  (2, 3 or 4)
    unsigned short amap[AXVALID_3D + 2]; // compact list of the valid axis-row slots (414 for M = 5; 276 in a planar world)
    int    n_ax;              // entries of amap
    // world/dimension == 2 (src/traj_optimizer.cpp:8): the QP has the x and y variables only -- no z bounds, no z velocity /
    // acceleration rows, collision and corridor rows without their z term (:264-266, 330, 367, 394, 411, 423, 450, 469, 529) --
    // and every stored control point gets z = world/z_2d (:87-90).  The kernels keep their 39 unknowns: the z unknowns stay in
    // the system without any row, decoupled from x / y (every n_z is zeroed), pinned at z_2d by the state constants, and are
    // overwritten on output; the x / y iterates are those of the 26-unknown problem.
    int    dim2;
    double z2d;               // (double)(float)world/z_2d
    // Inverse of the reduced cost Hessian of ONE axis, Z'(blockdiag(Qh) + 2 w_t E_T) Z, for T = 1 .. M terminal segments
    // (src/traj_optimizer.cpp:329-372: the terminal weight sits on c_{m,5} of the last T segments).  The three axes share it and the
    // cost does not couple them, so the 39 x 39 reduced Hessian is three copies of this 13 x 13 block: what the active-set solve
    // (lsc_kernels.hip, gi_solve) needs of it is its inverse, formed once on the host in extended precision.
    double ginv[M][NYA * NYA];
    // ghz[T-1][t][a] = (ginv[T-1] Z' e_t)_a: H^-1 times the y-space image of the variable t of an axis.  H^-1 times a row's normal (three
    // x-variables) is three of these per lane.
    double ghz[M][SEGV * NYA];
    // gzt[t][a] = Z[t][a]: the y-space image of the variable t of an axis as a dense row (the same lookup for a row's normal itself)
    double gzt[SEGV * NYA];
    // gy0[T-1][a][0..2 | 3]: the unconstrained optimum of an axis is linear in its three state constants and its goal coordinate,
    //   y*_a = sum_j gy0[a][j] s0_j + gy0[a][3] goal  ( = -ginv Z' (Qh x0 + terminal gradient) ), formed on the host.
    double gy0[M][NYA * 4];
    // The kernels' LDS tables that do not depend on the agent, as the kernels hold them (built once here instead of by every workgroup in every
    // tick behind dependent loads of the fields above): amap32[i] = slot | type << 10 | axis << 13 | t << 15 of valid axis row i;
    // xgp32[k * SEGV + t] = the (at most three) unknowns of variable (k, t) as y-indices, one per byte; xtcm[t][j] = their coefficients, zero
    // beyond x_n[t].
    uint32_t amap32[AXVALID_3D + 2];
    uint32_t xgp32[NV];
    double   xtcm[SEGV][3];
};

// Dense variant of the same elimination for the alternate planner modes (lsc_general.hip): axis-major y, with
// (nya = 13) or without (nya = 15, BVC: c_{4,3..5} stay free) the stop-at-horizon rows.
constexpr int GNYA = NYL + 3;      // 15 for M = 5
struct GModel {
    int    nya;
    double Z[SEGV][GNYA];     // x_t = (state constants for t < 3) + sum_a Z[t][a] y[a]
    double Hc[GNYA * GNYA];   // Z^T blockdiag(Qh) Z for one axis
};

// offsets inside the x-space weight array W that the assembly terms read from
constexpr int W_D = 0;          // [3][SEGV] diagonal        (bounds + velocity + acceleration stencils)
constexpr int W_1 = NV;         // [3][SEGV] (t, t+1) coupling
constexpr int W_2 = 2 * NV;     // [3][SEGV] (t, t+2) coupling
constexpr int W_S = 3 * NV;     // [NCP][6]  LSC blocks  sum w n n^T : xx, xy, xz, yy, yz, zz
constexpr int W_SIZE = 3 * NV + 6 * NCP;
static_assert(W_SIZE < 1024, "an assembly term addresses W with 10 bits");

}  // namespace lsc
