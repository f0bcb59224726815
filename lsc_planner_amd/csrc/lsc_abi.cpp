// lsc_abi.cpp -- C ABI (include/lsc_planner_amd.h) over the gfx950 kernels.  Host side only: builds the
// per-context constants once (what TrajOptimizer's constructor does per agent in the reference,
// src/traj_optimizer.cpp:4-25), owns HBM buffers and persistent per-agent state, launches kernels.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types and prototypes only: the library itself is bound at run time (see rccl_api())
#include <dlfcn.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <algorithm>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/lsc_planner_amd.h"
#include "lsc_kernels.h"

using namespace lsc;

namespace {

int binom(int n, int k)
{
    if (k < 0 || k > n) return 0;
    long r = 1;
    for (int i = 1; i <= k; i++) r = r * (n - k + i) / i;
    return (int)r;
}

// Q_base = B Z B^T dt^(-2 phi + 1), phi = 3, phi_n = 1   (src/traj_optimizer.cpp:169-184; B from
// include/polynomial.hpp:415-426)
void build_qbase(double dt, double Q[NC * NC])
{
    const int n = DEG, phi = 3;
    double B[NC][NC], Z[NC][NC], T[NC][NC];
    for (int i = 0; i <= n; i++)
        for (int j = 0; j <= n; j++) B[i][j] = j >= i ? binom(n, i) * binom(n - i, n - j) * (((j - i) & 1) ? -1.0 : 1.0) : 0.0;
    auto fall = [](int a, int p) { if (a < p) return 0; int c = 1; for (int i = 0; i < p; i++) c *= a - i; return c; };
    for (int i = 0; i <= n; i++)
        for (int j = 0; j <= n; j++) {
            int den = i + j - 2 * phi + 1;
            Z[i][j] = den > 0 ? (double)fall(i, phi) * fall(j, phi) / den : 0.0;
        }
    for (int i = 0; i <= n; i++)
        for (int j = 0; j <= n; j++) { double s = 0; for (int l = 0; l <= n; l++) s += B[i][l] * Z[l][j]; T[i][j] = s; }
    const double sc = std::pow(dt, -2 * phi + 1);
    for (int i = 0; i <= n; i++)
        for (int j = 0; j <= n; j++) { double s = 0; for (int l = 0; l <= n; l++) s += T[i][l] * B[j][l]; Q[i * NC + j] = s * sc; }
}

struct HostModel {
    Model m;
    std::vector<uint32_t> terms;
    std::vector<uint32_t> entries;
};

void build_model(const lsc_config &cfg, HostModel &H)
{
    Model &m = H.m;
    std::memset(&m, 0, sizeof(m));
    m.dt = cfg.dt; m.w_c = cfg.control_weight; m.w_t = cfg.terminal_weight;
    m.hv_scale = cfg.dt / DEG;
    m.ha_scale = cfg.dt * cfg.dt / (DEG * (DEG - 1));
    double Q[NC * NC];
    build_qbase(cfg.dt, Q);
    for (int i = 0; i < NC * NC; i++) m.Qh[i] = 2.0 * cfg.control_weight * Q[i];

    // x_t as a combination of the 13 free variables of an axis (see lsc_model.hpp)
    double Zm[SEGV][NYA];
    std::memset(Zm, 0, sizeof(Zm));
    for (int mm = 0; mm < M; mm++)
        for (int i = 0; i < NC; i++) {
            const int t = mm * NC + i;
            if (mm == M - 1 && i >= 3) { Zm[t][NYL] = 1.0; continue; }       // stop at the horizon
            if (i >= 3) { Zm[t][3 * mm + (i - 3)] = 1.0; continue; }
            if (mm == 0) continue;                                         // fixed by the current state
            const int u3 = 3 * (mm - 1), u4 = u3 + 1, u5 = u3 + 2;          // c_{m-1,3..5}
            if (i == 0) Zm[t][u5] = 1.0;
            if (i == 1) { Zm[t][u5] = 2.0; Zm[t][u4] = -1.0; }
            if (i == 2) { Zm[t][u5] = 4.0; Zm[t][u4] = -4.0; Zm[t][u3] = 1.0; }
        }
    for (int t = 0; t < SEGV; t++) {
        int n = 0;
        for (int a = 0; a < NYA; a++)
            if (Zm[t][a] != 0.0) { m.x_i[t][n] = a; m.x_c[t][n] = Zm[t][a]; n++; }
        m.x_n[t] = n;
    }
    for (int a = 0; a < NYA; a++) {
        int n = 0;
        for (int t = 0; t < SEGV; t++)
            if (Zm[t][a] != 0.0) { m.t_t[a][n] = t; m.t_c[a][n] = Zm[t][a]; n++; }
        m.t_n[a] = n;
    }
    // Hc = Z^T blockdiag(Qh) Z
    for (int a = 0; a < NYA; a++)
        for (int b = 0; b < NYA; b++) {
            double s = 0;
            for (int mm = 0; mm < M; mm++)
                for (int i = 0; i < NC; i++)
                    for (int j = 0; j < NC; j++) s += Zm[mm * NC + i][a] * m.Qh[i * NC + j] * Zm[mm * NC + j][b];
            m.Hc[a * NYA + b] = s;
        }

    for (int t = 0; t < SEGV; t++)
        for (int a = 0; a < NYA; a++) m.gzt[t * NYA + a] = Zm[t][a];
    // inverse of the axis block of the reduced cost Hessian for T = 1 .. M terminal segments (Model::ginv): Gauss-Jordan with partial
    // pivoting in long double (the block is positive definite: a plan whose jerk cost vanishes is fixed by the initial state)
    for (int T = 1; T <= M; T++) {
        long double A[NYA][2 * NYA];
        for (int a = 0; a < NYA; a++) {
            for (int b = 0; b < NYA; b++) { A[a][b] = m.Hc[a * NYA + b]; A[a][NYA + b] = a == b ? 1.0L : 0.0L; }
            const int mterm = a == NYL ? M - 1 : ((a % 3) == 2 ? a / 3 : -1);       // y_a is c_{m,5} (or the last segment's end point)
            if (mterm >= M - T) A[a][a] += 2.0L * cfg.terminal_weight;
        }
        for (int c = 0; c < NYA; c++) {
            int piv = c;
            for (int r = c + 1; r < NYA; r++) if (fabsl(A[r][c]) > fabsl(A[piv][c])) piv = r;
            if (piv != c) for (int j = 0; j < 2 * NYA; j++) std::swap(A[c][j], A[piv][j]);
            const long double d = A[c][c];
            for (int j = 0; j < 2 * NYA; j++) A[c][j] /= d;
            for (int r = 0; r < NYA; r++) {
                if (r == c) continue;
                const long double f = A[r][c];
                if (f != 0.0L) for (int j = 0; j < 2 * NYA; j++) A[r][j] -= f * A[c][j];
            }
        }
        for (int a = 0; a < NYA; a++)
            for (int b = 0; b < NYA; b++) m.ginv[T - 1][a * NYA + b] = (double)(0.5L * (A[a][NYA + b] + A[b][NYA + a]));
        for (int t = 0; t < SEGV; t++)
            for (int a = 0; a < NYA; a++) {
                double acc = 0.0;
                for (int j = 0; j < m.x_n[t]; j++) acc += m.x_c[t][j] * m.ginv[T - 1][a * NYA + m.x_i[t][j]];
                m.ghz[T - 1][t * NYA + a] = acc;
            }
        // unconstrained optimum as a linear map of (s0, goal): gradient at y = 0 in x-space = Qh (segment 0) x0 - 2 w_t goal on the terminal points
        for (int j = 0; j < 4; j++) {
            long double gy[NYA];
            for (int b = 0; b < NYA; b++) {
                long double acc = 0.0L;
                if (j < 3) { for (int t = 0; t < NC; t++) acc += (long double)Zm[t][b] * (long double)m.Qh[t * NC + j]; }
                else { for (int mm = M - T; mm < M; mm++) acc += (long double)Zm[mm * NC + DEG][b] * (-2.0L * cfg.terminal_weight); }
                gy[b] = acc;
            }
            for (int a = 0; a < NYA; a++) {
                long double acc = 0.0L;
                for (int b = 0; b < NYA; b++) acc += (long double)m.ginv[T - 1][a * NYA + b] * gy[b];
                m.gy0[T - 1][a * 4 + j] = (double)(-acc);
            }
        }
    }

    // Hessian assembly terms: K[(k,a),(k',b)] += Z[t][a] Z[t'][b] * Wx[(k,t),(k',t')]
    auto scomp = [](int k, int kk) { if (k > kk) std::swap(k, kk); return k == 0 ? kk : (k == 1 ? 2 + kk : 5); };  // xx xy xz yy yz zz
    auto axis_of = [](int g) { return yaxis(g); };
    auto var_of = [](int g) { return yvar(g); };
    H.terms.clear(); H.entries.clear();
    int n_entries = 0;
    for (int gi = 0; gi < NY; gi++)
        for (int gj = 0; gj <= gi; gj++) {
            const int k = axis_of(gi), kk = axis_of(gj), a = var_of(gi), b = var_of(gj);
            std::map<int, int> acc;  // src -> integer coefficient
            for (int p = 0; p < m.t_n[a]; p++)
                for (int q = 0; q < m.t_n[b]; q++) {
                    const int t = m.t_t[a][p], tt = m.t_t[b][q];
                    const int c = (int)std::lround(m.t_c[a][p] * m.t_c[b][q]);
                    if (t == tt) {
                        acc[W_S + t * 6 + scomp(k, kk)] += c;
                        if (k == kk) acc[W_D + k * SEGV + t] += c;
                    } else if (k == kk && t / NC == tt / NC) {
                        const int lo = t < tt ? t : tt, df = t < tt ? tt - t : t - tt;
                        if (df == 1) acc[W_1 + k * SEGV + lo] += c;
                        if (df == 2) acc[W_2 + k * SEGV + lo] += c;
                    }
                }
            bool any = false;
            for (auto &kv : acc) if (kv.second != 0) any = true;
            const bool structural = any || (k == kk && m.Hc[a * NYA + b] != 0.0);
            if (gi - gj > BAND) {
                if (structural) { std::fprintf(stderr, "lsc: band violated (%d,%d)\n", gi, gj); std::abort(); }
                continue;
            }
            // every position of the band gets an entry, also the structurally zero ones: the Cholesky factor is
            // published in place and fills the band, so each assembly must rewrite all of it
            H.entries.push_back(((uint32_t)gi << 16) | (uint32_t)gj);
            H.entries.push_back((uint32_t)H.terms.size());
            for (auto &kv : acc) {
                if (kv.second == 0) continue;
                if (kv.second < -128 || kv.second > 127 || kv.first >= 1024) { std::fprintf(stderr, "lsc: term overflow\n"); std::abort(); }
                H.terms.push_back(((uint32_t)gi << 0) * 0u | ((uint32_t)kv.first << 8) | (uint32_t)(kv.second + 128));
            }
            n_entries++;
        }
    H.entries.push_back(0);
    H.entries.push_back((uint32_t)H.terms.size());
    m.n_entries = n_entries;
    m.n_terms = (int)H.terms.size();
    for (int k = 0; k < 3; k++) { m.world_min[k] = cfg.world_min[k]; m.world_max[k] = cfg.world_max[k]; }
    m.use_sfc = cfg.use_octomap;
    m.prune = cfg.prune;
    m.max_iters = cfg.max_iters > 0 ? cfg.max_iters : 50;
    m.dx_tol = 2e-8;
    m.gap_tol = cfg.gap_tolerance > 0.0 ? cfg.gap_tolerance : 1e-9;
    m.ws_mu0 = cfg.warm_start_mu >= 0.0 ? cfg.warm_start_mu : 0.03;
    m.dim2 = cfg.world_dimension == 2 ? 1 : 0;
    m.z2d = (double)(float)cfg.world_z_2d;
    int n = 0;
    for (int sl = 0; sl < AXROWS; sl++) {
        const int type = sl / NV, k = (sl % NV) / SEGV, t = (sl % NV) % SEGV, mm = t / NC, i = t % NC;
        const bool valid = type < 2 ? !(mm == 0 && i < 3) : (type < 4 ? (i <= 4 && !(mm == 0 && i < 2)) : (i <= 3 && !(mm == 0 && i == 0)));
        if (valid && !(m.dim2 && k == 2)) m.amap[n++] = (unsigned short)sl;      // planar world: `for (k < dim)`, src/traj_optimizer.cpp:274, 469
    }
    if (n != (m.dim2 ? AXVALID_2D : AXVALID_3D)) { std::fprintf(stderr, "lsc: axis row count %d\n", n); std::abort(); }
    m.n_ax = n;
    for (int i = 0; i < n; i++) {
        const uint32_t sl = m.amap[i], type = sl / NV, kt = sl % NV;
        m.amap32[i] = sl | (type << 10) | ((kt / SEGV) << 13) | ((kt % SEGV) << 15);
    }
    for (int v = 0; v < NV; v++) {
        const int k = v / SEGV, t = v % SEGV;
        m.xgp32[v] = (uint32_t)yglob(k, m.x_i[t][0]) | ((uint32_t)yglob(k, m.x_i[t][1]) << 8) | ((uint32_t)yglob(k, m.x_i[t][2]) << 16);
    }
    for (int t = 0; t < SEGV; t++)
        for (int j = 0; j < 3; j++) m.xtcm[t][j] = m.x_n[t] < j + 1 ? 0.0 : m.x_c[t][j];
    m.sigma_pow = 3;
    // (no environment overrides: everything that changes the solve is an lsc_config field)
}

// dense elimination tables of the alternate-mode kernel: axis-major y, with (LSC) or without (BVC) the stop rows
void build_gmodel(const lsc_config &cfg, const Model &m, GModel &g)
{
    std::memset(&g, 0, sizeof(g));
    const bool stop = cfg.planner_mode == 0;
    g.nya = stop ? NYA : GNYA;
    for (int mm = 0; mm < M; mm++)
        for (int i = 0; i < NC; i++) {
            const int t = mm * NC + i;
            if (mm == M - 1 && i >= 3) { g.Z[t][stop ? NYL : NYL + (i - 3)] = 1.0; continue; }
            if (i >= 3) { g.Z[t][3 * mm + (i - 3)] = 1.0; continue; }
            if (mm == 0) continue;
            const int u3 = 3 * (mm - 1), u4 = u3 + 1, u5 = u3 + 2;
            if (i == 0) g.Z[t][u5] = 1.0;
            if (i == 1) { g.Z[t][u5] = 2.0; g.Z[t][u4] = -1.0; }
            if (i == 2) { g.Z[t][u5] = 4.0; g.Z[t][u4] = -4.0; g.Z[t][u3] = 1.0; }
        }
    for (int a = 0; a < g.nya; a++)
        for (int b = 0; b < g.nya; b++) {
            double s = 0;
            for (int mm = 0; mm < M; mm++)
                for (int i = 0; i < NC; i++)
                    for (int j = 0; j < NC; j++) s += g.Z[mm * NC + i][a] * m.Qh[i * NC + j] * g.Z[mm * NC + j][b];
            g.Hc[a * GNYA + b] = s;
        }
}

}  // namespace

struct lsc_ctx {
    lsc_config cfg;
    HostModel hm;
    int N = 0, first = 0, count = 0, cap = 0, cap_tp = 0, n_cu = 256;
    size_t smem_tp = 0;
    std::string err;
    std::string note;                    // informational remarks of lsc_create (not errors): lsc_last_note
    bool timing = false;
    // device
    Model *d_model = nullptr;
    uint32_t *d_terms = nullptr, *d_entries = nullptr;
    double *d_radius = nullptr, *d_radius_obs = nullptr, *d_downwash = nullptr, *d_downwash_obs = nullptr;
    double *d_vmax = nullptr, *d_amax = nullptr, *d_vnom = nullptr;
    float *d_stale = nullptr, *d_sfc = nullptr;
    float *d_goal_cur = nullptr;
    // alternate planner modes (lsc_general.hip)
    GModel *d_gmodel = nullptr;
    unsigned char *d_ever = nullptr, *d_gen_ws = nullptr;
    size_t gen_stride = 0;
    int gen_slots = 0;
    int last_host_seq = 0;               // planner_seq of the last host-buffer tick (lsc_dump_qp reads its inputs back)
    bool next_has_all_rows = false;      // d_next holds the new plan of ALL N agents (whole-swarm shard, or lsc_replan_tick_all's gather)
    bool h_ever_stale = false;           // device-resident ticks ran since h_ever was last in step with d_ever
    std::vector<unsigned char> h_ever;   // host mirror for the host-buffer ticks (they decide on the host whether anybody is off plan)
    unsigned char *d_spill = nullptr;    // HBM row workspaces of the second pass (agents beyond the LDS row capacity)
    size_t spill_stride = 0;
    int spill_slots = 0;
    float *fused_state_next = nullptr;   // set by lsc_tick_device_fused for one call
    int *d_sfc_init = nullptr, *d_sfc_err = nullptr, *d_img_of_agent = nullptr, *d_integral = nullptr;
    std::vector<float> h_edt;   // host copy of the distance field (integral images are rebuilt when agents change)
    int edt_dims[3] = {0, 0, 0}, edt_kmin[3] = {0, 0, 0};
    double edt_res = 0.0;
    std::vector<double> h_radius;
    // goal planner with a distance field (lsc_goal.hip)
    float *d_edt = nullptr, *d_goal_planned = nullptr, *d_ray_stack = nullptr;
    unsigned char *d_occ_static = nullptr;
    int *d_goal_err = nullptr, *d_goal_flags = nullptr, *d_goal_exp = nullptr, *d_goal_path = nullptr, *d_goal_plen = nullptr;
    int goal_path_cap = 0;
    std::vector<uint32_t> h_fcode;       // Key32 table of the goal search (empty: Key64)
    uint32_t *d_fcode = nullptr;
    int fcode_rb = 0;
    unsigned char *d_safety = nullptr;           // lsc_safety_ratio's buffers
    size_t safety_bytes = 0;
    long long *d_goal_prof = nullptr;
    bool goal_profiling = false;
    int grid_dims[3] = {0, 0, 0}, grid_row_cap = 0;
    double grid_min[3] = {0, 0, 0};
    std::vector<int> nb_seq;
    int *d_nrows = nullptr, *d_bmax = nullptr, *d_order = nullptr;
    float *d_obs_bound = nullptr;
    // neighbour lists of large swarms (lsc_neigh.hip): one allocation (base neigh.seg_bound ... see lsc_set_agents); neigh.cnt == nullptr: not in use
    NeighArgs neigh = {};
    void *d_neigh = nullptr;
    bool neigh_always = false;           // LSC_NEIGH_ALWAYS (measurements, tests): lists whenever the context has them, not only where they pay
    long long *d_iters_acc = nullptr;
    long long *d_prof = nullptr;
    double *d_dbg = nullptr;
    double *d_trace = nullptr;
    int trace_agent = -1;
    bool profiling = false;
    // buffers of the host-pointer tick
    float *d_state = nullptr, *d_goal = nullptr, *d_prev = nullptr, *d_next = nullptr;
    double *d_cost = nullptr;
    int *d_status = nullptr, *d_iters = nullptr;
    float *d_onormal = nullptr;
    double *d_od = nullptr;
    // host-pointer tick: inputs and outputs travel as ONE copy each, through pinned staging buffers.
    //   d_state | d_goal | d_prev  are consecutive in one allocation (base d_state),
    //   d_cost | d_next | d_status | d_iters  in another (base d_cost)
    float *h_in = nullptr;
    unsigned char *h_out = nullptr;
    hipStream_t stream = nullptr;
    // kernel timing: one HIP event pair per launch, recorded on the launch stream, read back on query
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool[5];   // 0 plan kernel(s), 1 dense sweep, 2 trajectory exchange, 3 goal kernel, 4 corridor kernel
    std::vector<double> host_tick_ms;    // which = 5: wall clock of every lsc_replan_tick since lsc_set_timing(1), entry to return
    size_t ev_used[5] = {0, 0, 0, 0, 0};
    // agent-sharded multi-GPU: this context is rank `rank` of `world`; every rank owns a block of shard_rows agents of a
    // table padded to table_rows = shard_rows * world rows, so that the exchange is ONE in-place equal-sized all-gather
    int world = 1, rank = 0, shard_rows = 0, table_rows = 0;
    ncclComm_t comm = nullptr;
};

// RCCL is bound with dlopen instead of a link-time dependency: inside a PyTorch process the wheel's own librccl (and
// HIP runtime) is already mapped and must be the one that is used -- two RCCL copies in one process would each bring
// their own view of the devices; stand-alone C++ users (lsc_sim) get /opt/rocm's.
struct RcclApi {
    void *h = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

static bool rccl_bind(RcclApi &api);
static const RcclApi *rccl_api()
{
    static RcclApi api;
    static std::once_flag once;                               // contexts may be created from several threads
    std::call_once(once, [] { (void)rccl_bind(api); });
    return api.h ? &api : nullptr;
}
static bool rccl_bind(RcclApi &api)
{
    const char *names[] = {"librccl.so.1", "librccl.so"};
    void *h = nullptr;
    for (const char *n : names) if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);   // the copy the process already has
    for (const char *n : names) if (!h) h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return false;
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    api.AllGather = reinterpret_cast<decltype(api.AllGather)>(dlsym(h, "ncclAllGather"));
    api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(dlsym(h, "ncclAllReduce"));
    api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(dlsym(h, "ncclGroupStart"));
    api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather || !api.AllReduce || !api.GroupStart || !api.GroupEnd ||
        !api.GetErrorString)
        return false;
    api.h = h;
    return true;
}

#define NCCLCHK(ctx, api, call)                                                               \
    do {                                                                                      \
        ncclResult_t r_ = (call);                                                             \
        if (r_ != ncclSuccess) {                                                              \
            (ctx)->err = std::string(#call) + ": " + (api)->GetErrorString(r_);               \
            return LSC_ECOMM;                                                                 \
        }                                                                                     \
    } while (0)

static int timing_begin(lsc_ctx *c, int which, hipStream_t st, hipEvent_t *e1)
{
    if (c->ev_used[which] == c->ev_pool[which].size()) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return LSC_EHIP;
        c->ev_pool[which].push_back({a, b});
    }
    auto &p = c->ev_pool[which][c->ev_used[which]++];
    *e1 = p.second;
    return hipEventRecord(p.first, st) == hipSuccess ? LSC_OK : LSC_EHIP;
}

#define HIPCHK(ctx, call)                                                                     \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                   \
            return LSC_EHIP;                                                                  \
        }                                                                                     \
    } while (0)

static int build_integrals(lsc_ctx *c);
static int build_goal_grid(lsc_ctx *c, const std::vector<double> &radii);

extern "C" {

void lsc_default_config(lsc_config *cfg)
{
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->dt = 0.2; cfg->control_weight = 0.01; cfg->terminal_weight = 1.0;
    cfg->world_min[0] = -10; cfg->world_min[1] = -10; cfg->world_min[2] = 0;
    cfg->world_max[0] = 10; cfg->world_max[1] = 10; cfg->world_max[2] = 2.5f;
    cfg->use_octomap = 0; cfg->world_resolution = 0.1; cfg->device = 0;
    cfg->max_rows_per_cp = 0; cfg->max_iters = 50; cfg->prune = 1; cfg->warm_start_mu = 0.03;
    cfg->goal_mode = 0; cfg->goal_threshold = 0.1; cfg->priority_dist_threshold = 0.4; cfg->goal_radius = 2.0;
    cfg->grid_resolution = 0.3; cfg->grid_margin = 0.2;   // launch/testall_forest.launch:88-89
    cfg->horizon = 1.0; cfg->goal_row_cap = 0;
    cfg->planner_mode = 0; cfg->slack_mode = 0; cfg->slack_collision_weight = 100000.0; cfg->n_constraint_segments = -1;
    cfg->reset_threshold = 0.0;
    cfg->gap_tolerance = 1e-9;
    cfg->world_dimension = 3; cfg->world_z_2d = 1.0; cfg->goal_search = 0; cfg->solver = 1;
}

lsc_ctx *lsc_create(const lsc_config *cfg)
{
    if (!cfg || !(cfg->dt > 0)) return nullptr;
    // M = static_cast<int>((horizon + SP_EPSILON) / dt) (src/traj_optimizer.cpp:9, src/traj_planner.cpp:22).  The kernels are unrolled
    // for their segment count: this library plans M segments (liblsc_hip.so: 5, every shipped launch file; liblsc_hip_m4.so: 4, the
    // C++ defaults of src/param.cpp:66-67) and says so instead of silently planning another horizon.
    if ((int)((cfg->horizon + 1e-9) / cfg->dt) != M) {
        std::fprintf(stderr, "lsc_create: horizon %g / dt %g = %d segments, this library plans M = %d (lsc_segments(); the sibling build "
                             "liblsc_hip%s.so plans %d)\n", cfg->horizon, cfg->dt, (int)((cfg->horizon + 1e-9) / cfg->dt), M, M == 5 ? "_m4" : "", M == 5 ? 4 : 5);
        return nullptr;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device >= ndev) {
        std::fprintf(stderr, "lsc_create: no usable HIP device (there is no CPU fallback)\n");
        return nullptr;
    }
    if (hipSetDevice(cfg->device) != hipSuccess) return nullptr;
    if (init_device_kernels() != hipSuccess) {
        std::fprintf(stderr, "lsc_create: device %d does not accept the gfx950 kernels' LDS request\n", cfg->device);
        return nullptr;
    }
    if (cfg->planner_mode < 0 || cfg->planner_mode > 1 || cfg->slack_mode < 0 || cfg->slack_mode > 2 ||
        (cfg->planner_mode == 1 && cfg->use_octomap)) {
        // BVC + octomap: the reference's generateSFC throws for every planner mode but LSC (src/traj_planner.cpp:1442-1449)
        std::fprintf(stderr, "lsc_create: unsupported planner_mode / slack_mode combination\n");
        return nullptr;
    }
    lsc_ctx *c = new lsc_ctx();
    c->cfg = *cfg;
    if (c->cfg.planner_mode == 0 && c->cfg.slack_mode != 0) {
        // TrajPlanner::checkPlannerMode (src/traj_planner.cpp:445-448): "LSC does not need slack variables, fix to none".  The
        // slack rows of LSC mode are the ones a disturbance reset leaves behind (obs_slack_indices), nothing else.
        c->cfg.slack_mode = 0;
        c->note = "planner_mode lsc with a slack mode: slack_mode fixed to none (src/traj_planner.cpp:445-448)";
    }
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess && prop.multiProcessorCount > 0) c->n_cu = prop.multiProcessorCount;
    }
    build_model(*cfg, c->hm);
    bool ok = hipStreamCreate(&c->stream) == hipSuccess;
    ok = ok && hipMalloc(&c->d_terms, sizeof(uint32_t) * (c->hm.terms.size() + 2)) == hipSuccess;
    ok = ok && hipMalloc(&c->d_entries, sizeof(uint32_t) * c->hm.entries.size()) == hipSuccess;
    ok = ok && hipMalloc(&c->d_model, sizeof(Model)) == hipSuccess;
    {
        GModel g;
        build_gmodel(*cfg, c->hm.m, g);
        ok = ok && hipMalloc(&c->d_gmodel, sizeof(GModel)) == hipSuccess;
        ok = ok && hipMemcpy(c->d_gmodel, &g, sizeof(GModel), hipMemcpyHostToDevice) == hipSuccess;
    }
    ok = ok && hipMemcpy(c->d_terms, c->hm.terms.data(), sizeof(uint32_t) * c->hm.terms.size(), hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && hipMemcpy(c->d_entries, c->hm.entries.data(), sizeof(uint32_t) * c->hm.entries.size(), hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) { lsc_destroy(c); return nullptr; }
    return c;
}

static void free_agents(lsc_ctx *c)
{
    void *ptrs[] = {c->d_radius, c->d_radius_obs, c->d_downwash, c->d_downwash_obs, c->d_vmax, c->d_amax, c->d_vnom,
                    c->d_stale, c->d_sfc, c->d_goal_cur, c->d_sfc_init, c->d_sfc_err, c->d_img_of_agent, c->d_integral, c->d_nrows, c->d_iters_acc, c->d_prof, c->d_dbg, c->d_state, c->d_cost,
                    c->d_onormal, c->d_od, c->d_spill, c->d_ever, c->d_gen_ws, c->d_bmax, c->d_order, c->d_obs_bound};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    if (c->neigh.prof) {
        std::vector<long long> h(8 * (size_t)c->N);
        if (hipDeviceSynchronize() == hipSuccess && hipMemcpy(h.data(), c->neigh.prof, sizeof(long long) * h.size(), hipMemcpyDeviceToHost) == hipSuccess) {
            double st[4] = {0, 0, 0, 0}, nq = 0, nc = 0, nu = 0, novf = 0;
            int n = 0;
            for (int q = 0; q < c->N; q++) {
                const long long *p = &h[8 * (size_t)q];
                if (p[4] == 0) continue;
                for (int k = 0; k < 4; k++) st[k] += (double)(p[k + 1] - p[k]) / 100.0;
                nq += (double)(p[5] & 0xffffffff); nc += (double)p[6]; nu += (double)p[7]; novf = (double)(p[5] >> 32); n++;
            }
            if (n) fprintf(stderr, "[lsc] query kernel of the neighbour lists, last tick, mean over %d agents: set-up %.2f us, cells -> candidates %.2f us, sphere tests %.2f us, "
                                   "sorted list %.2f us; %.0f cells, %.0f candidates (%.0f of them from the overflow list), %.0f units per agent\n", n, st[0] / n, st[1] / n, st[2] / n, st[3] / n, nc / n, nq / n, novf, nu / n);
        }
        (void)hipFree(c->neigh.prof);
    }
    if (c->d_neigh) (void)hipFree(c->d_neigh);
    c->d_neigh = nullptr; c->neigh = NeighArgs{};
    c->d_spill = nullptr; c->spill_slots = 0; c->spill_stride = 0;
    c->d_bmax = nullptr; c->d_order = nullptr; c->d_obs_bound = nullptr; c->d_ever = nullptr; c->d_gen_ws = nullptr; c->gen_slots = 0; c->gen_stride = 0;
    if (c->h_in) { (void)hipHostFree(c->h_in); c->h_in = nullptr; }
    if (c->h_out) { (void)hipHostFree(c->h_out); c->h_out = nullptr; }
    void *gp[] = {c->d_edt, c->d_goal_planned, c->d_ray_stack, c->d_occ_static, c->d_goal_err, c->d_goal_flags, c->d_goal_exp,
                  c->d_goal_path, c->d_goal_plen, c->d_goal_prof, c->d_fcode, c->d_safety};
    for (void *p : gp) if (p) (void)hipFree(p);
    c->d_goal_prof = nullptr; c->d_fcode = nullptr; c->d_safety = nullptr; c->safety_bytes = 0;
    c->d_edt = c->d_goal_planned = c->d_ray_stack = nullptr; c->d_occ_static = nullptr;
    c->d_goal_err = c->d_goal_flags = c->d_goal_exp = c->d_goal_path = c->d_goal_plen = nullptr;
    c->d_radius = c->d_radius_obs = c->d_downwash = c->d_downwash_obs = c->d_vmax = c->d_amax = c->d_vnom = nullptr;
    c->d_stale = c->d_sfc = c->d_state = c->d_goal = c->d_prev = c->d_next = nullptr;
    c->d_cost = nullptr; c->d_status = c->d_iters = c->d_nrows = nullptr; c->d_iters_acc = nullptr; c->d_prof = nullptr; c->d_dbg = nullptr;
    c->d_sfc_init = c->d_sfc_err = c->d_img_of_agent = c->d_integral = nullptr;
    c->d_goal_cur = nullptr; c->d_onormal = nullptr; c->d_od = nullptr;
}

void lsc_destroy(lsc_ctx *c)
{
    if (!c) return;
    if (c->comm) { if (const RcclApi *api = rccl_api()) (void)api->CommDestroy(c->comm); c->comm = nullptr; }
    free_agents(c);
    if (c->d_trace) (void)hipFree(c->d_trace);
    if (c->d_model) (void)hipFree(c->d_model);
    if (c->d_gmodel) (void)hipFree(c->d_gmodel);
    if (c->d_terms) (void)hipFree(c->d_terms);
    if (c->d_entries) (void)hipFree(c->d_entries);
    for (int w = 0; w < 5; w++)
        for (auto &p : c->ev_pool[w]) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char *lsc_last_error(const lsc_ctx *c) { return c ? c->err.c_str() : "null context"; }
const char *lsc_last_note(const lsc_ctx *c) { return c ? c->note.c_str() : ""; }
int lsc_segments(void) { return M; }

int lsc_set_agents(lsc_ctx *c, int N, const double *radius, const double *downwash, const double *max_vel,
                   const double *max_acc, const double *nominal_vel)
{
    if (!c || N < 1 || !radius || !downwash || !max_vel || !max_acc || !nominal_vel) return LSC_EINVAL;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    free_agents(c);
    c->N = N;
    // rank's block of the (padded) table; world = 1 without lsc_comm_init: the whole swarm
    c->shard_rows = (N + c->world - 1) / c->world;
    c->table_rows = c->shard_rows * c->world;
    c->first = std::min(c->rank * c->shard_rows, N);
    c->count = std::min(c->shard_rows, N - c->first);
    const size_t Np = (size_t)c->table_rows;
    // Row capacity of the LDS pass: 27 x max_rows_per_cp rows in total (rows are stored compactly, so a control point may
    // hold more than its share), at most what the 160 KiB of a workgroup allow; never more than the 27 (N-1) that exist.
    int per_cp = c->cfg.max_rows_per_cp > 0 ? c->cfg.max_rows_per_cp : 64;
    if (per_cp > N - 1) per_cp = N - 1;
    if (per_cp < 1) per_cp = 1;
    constexpr int NBR = NCP - 3;              // control points that carry rows: 27 for M = 5
    int cap = NBR * per_cp;
    while (cap > NBR && plan_smem_bytes(c->hm.m.n_terms, c->hm.m.n_entries, cap) > 160 * 1024) cap -= NBR;
    c->cap = cap;
    c->hm.m.cap = cap;
    // Throughput build (256 lanes, two workgroups per CU): used when the shard has more agents than the GPU has CUs; its
    // capacity is what fits half a CU's LDS.  (Pointless -- and slower per agent -- for a shard that fits the chip.)
    c->cap_tp = 0; c->smem_tp = 0;
    if (c->cfg.max_rows_per_cp == 0) {
        int ct = NBR;
        while (plan_smem_bytes(c->hm.m.n_terms, c->hm.m.n_entries, ct + NBR, false) <= 80 * 1024 - 512) ct += NBR;
        if (ct < cap && plan_smem_bytes(c->hm.m.n_terms, c->hm.m.n_entries, ct, false) <= 80 * 1024 - 512) {
            c->cap_tp = ct;
            c->smem_tp = plan_smem_bytes(c->hm.m.n_terms, c->hm.m.n_entries, ct, false);
        }
    }
    HIPCHK(c, hipMemcpy(c->d_model, &c->hm.m, sizeof(Model), hipMemcpyHostToDevice));
    std::vector<double> r_obs(N), dw_obs(N);
    for (int i = 0; i < N; i++) { r_obs[i] = (double)(float)radius[i]; dw_obs[i] = (double)(float)downwash[i]; }
    auto up = [&](double **dst, const double *src, size_t n) -> hipError_t {
        hipError_t e = hipMalloc(dst, sizeof(double) * n);
        if (e != hipSuccess) return e;
        return hipMemcpy(*dst, src, sizeof(double) * n, hipMemcpyHostToDevice);
    };
    HIPCHK(c, up(&c->d_radius, radius, N));
    HIPCHK(c, up(&c->d_radius_obs, r_obs.data(), N));
    HIPCHK(c, up(&c->d_downwash, downwash, N));
    HIPCHK(c, up(&c->d_downwash_obs, dw_obs.data(), N));
    HIPCHK(c, up(&c->d_vmax, max_vel, 3 * (size_t)N));
    HIPCHK(c, up(&c->d_amax, max_acc, 3 * (size_t)N));
    HIPCHK(c, up(&c->d_vnom, nominal_vel, N));
    HIPCHK(c, hipMalloc(&c->d_stale, sizeof(float) * NV * (size_t)N));
    HIPCHK(c, hipMemset(c->d_stale, 0, sizeof(float) * NV * (size_t)N));   // TrajOptimizer::trajectory starts at (0,0,0)
    if (c->cfg.world_dimension == 2) {
        // Planar world: an agent whose FIRST solve fails keeps this trajectory (src/traj_planner.cpp:1553-1584).  The reference
        // leaves its z at 0 and overrides the agent's own z with world/z_2d on the next state callback (src/traj_planner.cpp:304-314);
        // here the z block starts at z_2d, so that the stale plan -- and the state propagated from it -- stay in the plane the
        // kernels and the host-buffer ticks' input check rely on (an out-of-plane stale plan used to make every later
        // host-buffer tick of the whole swarm return LSC_EINVAL).
        std::vector<float> init((size_t)NV * N, 0.0f);
        const float z = (float)c->cfg.world_z_2d;
        for (int q = 0; q < N; q++)
            for (int j = 0; j < SEGV; j++) init[(size_t)q * NV + 2 * SEGV + j] = z;
        HIPCHK(c, hipMemcpy(c->d_stale, init.data(), sizeof(float) * init.size(), hipMemcpyHostToDevice));
    }
    HIPCHK(c, hipMalloc(&c->d_sfc, sizeof(float) * M * 6 * (size_t)N));
    HIPCHK(c, hipMemset(c->d_sfc, 0, sizeof(float) * M * 6 * (size_t)N));
    HIPCHK(c, hipMalloc(&c->d_goal_cur, sizeof(float) * 3 * Np));
    HIPCHK(c, hipMemset(c->d_goal_cur, 0, sizeof(float) * 3 * Np));
    HIPCHK(c, hipMalloc(&c->d_sfc_init, sizeof(int) * (size_t)N));
    HIPCHK(c, hipMalloc(&c->d_sfc_err, sizeof(int) * (size_t)N));
    HIPCHK(c, hipMemset(c->d_sfc_err, 0, sizeof(int) * (size_t)N));
    {
        std::vector<int> ones(N, 1);                        // flag_initialize_sfc = true (src/traj_planner.cpp:48)
        HIPCHK(c, hipMemcpy(c->d_sfc_init, ones.data(), sizeof(int) * (size_t)N, hipMemcpyHostToDevice));
    }
    c->h_radius.assign(radius, radius + N);
    if (cap < NBR * (N - 1)) {
        // An agent can carry more rows than the LDS capacity holds: those agents are re-planned by a second pass with
        // their rows in HBM (the reference never drops a row, src/traj_optimizer.cpp:437-466).  One workspace per
        // persistent workgroup; 256 = one per CU.
        c->spill_stride = plan_spill_bytes(N);
        c->spill_slots = std::min(N, 256);
        HIPCHK(c, hipMalloc(&c->d_spill, c->spill_stride * (size_t)c->spill_slots));
    }
    {
        // alternate modes: persistent "was seen off its plan" flags, and -- when such a QP can occur at all -- the HBM
        // workspaces of lsc_general_kernel (one per persistent workgroup)
        HIPCHK(c, hipMalloc(&c->d_ever, (size_t)N));
        HIPCHK(c, hipMemset(c->d_ever, 0, (size_t)N));
        c->h_ever.assign(N, 0);
        const bool general_all = c->cfg.planner_mode == 1 || c->cfg.slack_mode != 0;
        if (general_all || c->cfg.reset_threshold > 0.0) {
            c->gen_stride = general_ws_bytes(N);
            c->gen_slots = std::min(N, 256);
            HIPCHK(c, hipMalloc(&c->d_gen_ws, c->gen_stride * (size_t)c->gen_slots));
        }
    }
    HIPCHK(c, hipMalloc(&c->d_nrows, sizeof(int) * (size_t)N));
    HIPCHK(c, hipMalloc(&c->d_bmax, sizeof(int) * (size_t)N));
    HIPCHK(c, hipMemset(c->d_bmax, 0, sizeof(int) * (size_t)N));
    HIPCHK(c, hipMalloc(&c->d_order, sizeof(int) * (size_t)N));
    HIPCHK(c, hipMalloc(&c->d_obs_bound, sizeof(float) * 4 * (size_t)N));
    HIPCHK(c, hipMemset(c->d_nrows, 0, sizeof(int) * (size_t)N));
    if (N >= NEIGH_MIN_AGENTS && (N - 1) * M <= 0xffff && c->cfg.prune == 1 && !getenv("LSC_NO_NEIGHBOUR_LISTS")) {
        // Neighbour lists (lsc_neigh.hip).  Cell size: what an agent at its velocity limit covers over the horizon + two diameters (1.6 m with
        // the shipped parameters: a query visits ~9 x 9 cells, one lane each, and a bucket's twelve slots hold a crowd four times as dense as
        // the 1024-agent benchmark's); any size is correct, the size only decides how many cells a query visits and how many agents share a
        // bucket.  LSC_NEIGH_CELL overrides it (measurements).
        NeighArgs &g = c->neigh;
        c->neigh_always = getenv("LSC_NEIGH_ALWAYS") != nullptr;
        double vm = 0.0, rm = 0.0, dmin = 1e300, dmax = 0.0;
        for (int i = 0; i < N; i++) {
            for (int k = 0; k < 3; k++) vm = std::max(vm, max_vel[3 * i + k]);
            rm = std::max(rm, radius[i]); dmin = std::min(dmin, dw_obs[i]); dmin = std::min(dmin, downwash[i]);
            dmax = std::max(dmax, dw_obs[i]); dmax = std::max(dmax, downwash[i]);
        }
        double cell = vm * M * c->cfg.dt + 4.0 * rm;
        if (const char *e = getenv("LSC_NEIGH_CELL")) { const double v = atof(e); if (v > 0.0) cell = v; }
        if (!(cell > 1e-3)) cell = 1e-3;
        g.sc_max = std::max(1.0, 1.0 / dmin); g.zscale = std::max(1.0, dmax);
        g.inv_cell = 1.0 / cell; g.inv_cell_z = 1.0 / (cell * g.zscale);
        unsigned H = 1024;
        while (H < 4u * (unsigned)N) H <<= 1;
        g.hmask = H - 1; g.ovf_cap = 4096; g.list_cap = 1024; g.plist_cap = NEIGH_PRIO_CAP; g.tag = 0;
        // (capacities the tests shrink to drive the overflow paths: an agent without a list culls by itself, results do not change)
        if (const char *e = getenv("LSC_NEIGH_OVF_CAP")) { const int v = atoi(e); if (v >= 1 && v <= 65536) g.ovf_cap = v; }
        if (const char *e = getenv("LSC_NEIGH_LIST_CAP")) { const int v = atoi(e); if (v >= 1 && v <= 65536) g.list_cap = v; }
        const size_t b_seg = sizeof(float) * 4 * M * (size_t)N, b_reach = sizeof(float) * M * (size_t)N, b_cells = 32 * (size_t)H, b_glob = 128,
                     b_ovf = sizeof(unsigned short) * (size_t)g.ovf_cap, b_list = sizeof(unsigned short) * (size_t)g.list_cap * N, b_cnt = sizeof(int) * (size_t)N,
                     b_plist = sizeof(unsigned short) * (size_t)g.plist_cap * N, b_view = sizeof(NeighView);
        auto al16 = [](size_t b) { return (b + 15) & ~(size_t)15; };
        const size_t total = al16(b_seg) + al16(b_reach) + al16(b_cells) + al16(b_glob) + al16(b_ovf) + al16(b_list) + 2 * al16(b_cnt) + al16(b_plist) + al16(b_view);
        HIPCHK(c, hipMalloc(&c->d_neigh, total));
        HIPCHK(c, hipMemset(c->d_neigh, 0, total));         // tag 0 everywhere: the first tick's tag is 1
        unsigned char *p = static_cast<unsigned char *>(c->d_neigh);
        g.seg_bound = reinterpret_cast<float *>(p); p += al16(b_seg);
        g.reach = reinterpret_cast<float *>(p); p += al16(b_reach);
        g.cells = reinterpret_cast<unsigned long long *>(p); p += al16(b_cells);
        g.glob = reinterpret_cast<unsigned long long *>(p); p += al16(b_glob);
        g.ovf = reinterpret_cast<unsigned short *>(p); p += al16(b_ovf);
        g.list = reinterpret_cast<unsigned short *>(p); p += al16(b_list);
        g.cnt = reinterpret_cast<int *>(p); p += al16(b_cnt);
        g.pcnt = reinterpret_cast<int *>(p); p += al16(b_cnt);
        g.plist = reinterpret_cast<unsigned short *>(p); p += al16(b_plist);
        NeighView v;
        v.list = g.list; v.cnt = g.cnt; v.plist = g.plist; v.pcnt = g.pcnt; v.cap = g.list_cap; v.pcap = g.plist_cap;
        HIPCHK(c, hipMemcpy(p, &v, sizeof(v), hipMemcpyHostToDevice));
        g.view = reinterpret_cast<const NeighView *>(p);
        g.prof = nullptr;
        if (getenv("LSC_NEIGH_PROFILE")) {        // stage stamps of the query kernel, printed by lsc_destroy (diagnostics)
            HIPCHK(c, hipMalloc(&g.prof, sizeof(long long) * 8 * (size_t)N));
            HIPCHK(c, hipMemset(g.prof, 0, sizeof(long long) * 8 * (size_t)N));
        }
    }
    HIPCHK(c, hipMalloc(&c->d_iters_acc, sizeof(long long) * (6 * (size_t)N)));       // [N] iterations, [N] iterations x LSC rows, [N][4] counters of the active-set solve (per agent: no workgroup shares an address)
    HIPCHK(c, hipMalloc(&c->d_prof, sizeof(long long) * 2 * PROF_PHASES * (size_t)N));          // [N] plan kernel, [N] general kernel
    HIPCHK(c, hipMemset(c->d_prof, 0, sizeof(long long) * 2 * PROF_PHASES * (size_t)N));
    HIPCHK(c, hipMalloc(&c->d_dbg, sizeof(double) * 4 * (size_t)N));
    HIPCHK(c, hipMemset(c->d_dbg, 0, sizeof(double) * 4 * (size_t)N));
    HIPCHK(c, hipMemset(c->d_iters_acc, 0, sizeof(long long) * (6 * (size_t)N)));
    {
        const size_t n = (size_t)N;
        float *in = nullptr;
        HIPCHK(c, hipMalloc(&in, sizeof(float) * (9 + 3 + NV) * n));
        c->d_state = in; c->d_goal = in + 9 * n; c->d_prev = in + 12 * n;
        // outputs: padded to table_rows so that the multi-GPU form gathers them in place
        unsigned char *out = nullptr;
        HIPCHK(c, hipMalloc(&out, (sizeof(double) + sizeof(float) * NV + 2 * sizeof(int)) * Np));
        HIPCHK(c, hipMemset(out, 0, (sizeof(double) + sizeof(float) * NV + 2 * sizeof(int)) * Np));
        c->d_cost = reinterpret_cast<double *>(out);
        c->d_next = reinterpret_cast<float *>(out + sizeof(double) * Np);
        c->d_status = reinterpret_cast<int *>(out + (sizeof(double) + sizeof(float) * NV) * Np);
        c->d_iters = c->d_status + Np;
        HIPCHK(c, hipHostMalloc(&c->h_in, sizeof(float) * (9 + 3 + NV) * n));
        HIPCHK(c, hipHostMalloc(&c->h_out, (sizeof(double) + sizeof(float) * NV + 2 * sizeof(int)) * Np));
    }
    return build_integrals(c);
}

int lsc_set_shard(lsc_ctx *c, int first, int count)
{
    if (!c || c->N == 0) return LSC_ESTATE;
    if (first < 0 || count < 0 || first + count > c->N) return LSC_EINVAL;   // count 0: a rank without agents
    c->first = first; c->count = count;
    return LSC_OK;
}

// Key32 table of the goal search: word d = floor(sqrt d) << rb | rank of frac(sqrt d) among the distinct fractional parts of
// sqrt(0 .. words - 1).  Returns false (table empty) when the ranks need more than 16 bits or the smallest gap between two distinct
// fractional parts -- the wrap-around gap 1 - max included -- is not orders of magnitude above the rounding of the reference's F.
// (The argument why such keys order and tie exactly like the reference's doubles is spelled out in build_goal_grid.)
static bool goal_key_table(int words, std::vector<uint32_t> &table, int &rank_bits)
{
    table.clear();
    rank_bits = 0;
    if (words < 1 || words > (1 << 16)) return false;
    std::vector<std::pair<long double, int>> fr(words);
    std::vector<uint32_t> q(words);
    for (int d = 0; d < words; d++) {
        uint32_t r = (uint32_t)std::sqrt((double)d);
        while ((long long)r * r > d) r--;
        while ((long long)(r + 1) * (r + 1) <= d) r++;
        q[d] = r;
        fr[d] = {(long long)r * r == d ? 0.0L : sqrtl((long double)d) - (long double)r, d};     // exact 0 for perfect squares
    }
    std::sort(fr.begin(), fr.end());
    std::vector<uint32_t> rank(words);
    long double min_gap = 1.0L - fr.back().first;
    uint32_t rk = 0;
    for (int i = 0; i < words; i++) {
        if (i > 0 && fr[i].first != fr[i - 1].first) { rk++; min_gap = std::min(min_gap, fr[i].first - fr[i - 1].first); }
        rank[fr[i].second] = rk;
    }
    int rb = 1;
    while ((1u << rb) <= rk) rb++;
    // gaps are ~1 / D^2 (>= 1e-10 for D < 65536); long double resolves 1e-16 here; the reference's F carries < 1e-9 of rounding
    // at most (ulp(2^19) = 6e-11, three roundings) against 10 x gap
    const bool safe = min_gap > 1e-12L && 10.0L * min_gap > 50.0L * 6e-11L && rb <= 16;
    // (with rb <= 16 and steps <= G_MAX = 32767, (steps + q) << rb stays below 2^32 - 1 = NONE)
    if (!safe) return false;
    table.resize(words);
    for (int d = 0; d < words; d++) table[d] = (q[d] << rb) | rank[d];
    rank_bits = rb;
    return true;
}

// test hook (host only, no device needed): the key table for squared distances 0 .. words - 1
int lsc_goal_key_table(int words, unsigned int *out, int *rank_bits)
{
    if (!out || !rank_bits) return LSC_EINVAL;
    std::vector<uint32_t> t;
    int rb = 0;
    if (!goal_key_table(words, t, rb)) return LSC_ESTATE;
    std::memcpy(out, t.data(), sizeof(uint32_t) * t.size());
    *rank_bits = rb;
    return LSC_OK;
}

// Goal planner inputs derived from the map (GridBasedPlanner::updateGridInfo / updateGridMap, distmap part,
// src/grid_based_planner.cpp:72-130): grid geometry, one static occupancy grid per distinct radius (indexed by the
// search key H*W*z + W*i + j), the distance field itself for castRay, and the bucket-count sequence of the
// std::unordered_map this build's libstdc++ provides (the reference's OPEN rows are such maps; see lsc_goal.hip).
static int build_goal_grid(lsc_ctx *c, const std::vector<double> &radii)
{
    if (c->cfg.goal_mode != 1 || !c->cfg.use_octomap) return LSC_OK;
    const double res = c->cfg.grid_resolution;
    if (!(res > 0)) { c->err = "grid_resolution must be positive"; return LSC_EINVAL; }
    int dim[3];
    for (int a = 0; a < 3; a++) {
        c->grid_min[a] = -std::floor((-(double)c->cfg.world_min[a] + 1e-9) / res) * res;
        const double gmax = std::floor(((double)c->cfg.world_max[a] + 1e-9) / res) * res;
        if (a == 2 && c->cfg.world_dimension == 2) {          // planar world: one layer at z = world/z_2d (:82-85)
            c->grid_min[a] = c->cfg.world_z_2d;
            dim[a] = 1;
        } else {
            dim[a] = (int)std::round((gmax - c->grid_min[a]) / res) + 1;
        }
        c->grid_dims[a] = dim[a];
    }
    const int H = dim[0], W = dim[1], A = dim[2];
    const size_t C = (size_t)H * W * A;
    if (C > 131071) { c->err = "goal planner: search grid has more than 131071 cells"; return LSC_EINVAL; }
    // OPEN row capacity: what fits the 160 KiB of LDS next to the per-cell state
    int cap = W * A;
    while (cap > 16 && goal_smem_bytes(H, W, A, cap) > 158 * 1024) cap--;
    if (goal_smem_bytes(H, W, A, cap) > 158 * 1024) { c->err = "goal planner: search grid does not fit LDS"; return LSC_EINVAL; }
    // ---- Key32: the search key as one 32-bit word (lsc_goal.hip).  The A* key is F = 10 (g + sqrt(d2)), g = steps, d2 = squared cell
    // distance to the goal: integers.  The search only ever ORDERS keys and tests them for EQUALITY, so any map of (g, d2) that
    // preserves both is as good as the double the reference computes.  Write sqrt(d2) = q + f, q = floor.  Then
    //     key = ((g + q) << rb) | rank(f),      rank = position of f among the distinct fractional parts of sqrt(0 .. D)
    // orders like g + sqrt(d2): integer parts first (f < 1), fractional parts by rank; and key equality <=> (g + q, f) equal <=>
    // the reals are equal.  The reference's doubles order the same way: equal reals mean equal d2 (same double) or two perfect
    // squares (every operation exact), and distinct reals differ by at least the smallest gap between distinct fractional parts,
    // which is CHECKED below to exceed the rounding error of the reference's three roundings (a few ulp of F < 2^19) by orders
    // of magnitude.  If the check fails, or the table does not fit next to a useful row capacity, Key64 (the double itself) is used.
    c->h_fcode.clear();
    c->fcode_rb = 0;
    {
        const long long D = (long long)(H - 1) * (H - 1) + (long long)(W - 1) * (W - 1) + (long long)(A - 1) * (A - 1);
        int cap32 = W * A;
        const int words = D < (1 << 16) ? (int)D + 1 : 0;
        if (words > 0) {
            while (cap32 > 16 && goal_smem_bytes(H, W, A, cap32, words) > 158 * 1024) cap32--;
            if (goal_smem_bytes(H, W, A, cap32, words) > 158 * 1024 || cap32 < std::min(W * A, 192)) cap32 = 0;
        }
        int jb_unused = 0;
        const bool fast = c->cfg.goal_search != 1 && goal_fast_slots(H, W, A, &jb_unused) != 0;     // else the table would be dead weight in LDS
        if (words > 0 && cap32 > 0 && c->cfg.goal_search != 2 && fast) {
            int rb = 0;
            if (goal_key_table(words, c->h_fcode, rb)) { c->fcode_rb = rb; cap = cap32; }
        }
    }
    if (c->cfg.goal_row_cap > 0) cap = std::max(4, std::min(cap, c->cfg.goal_row_cap));   // explicit smaller capacity (tests force the overflow path)
    c->grid_row_cap = cap;
    // bucket counts of a growing std::unordered_map<uint_least32_t, T> (identity hash), from the container itself
    {
        std::unordered_map<uint_least32_t, int> probe;
        c->nb_seq.clear();
        size_t last = probe.bucket_count();
        for (uint_least32_t k = 0; k < (uint_least32_t)cap + 1 && c->nb_seq.size() < 16; k++) {
            probe[k] = 0;
            if (probe.bucket_count() != last) { last = probe.bucket_count(); c->nb_seq.push_back((int)last); }
        }
    }
    const int nx = c->edt_dims[0], ny = c->edt_dims[1], nz = c->edt_dims[2];
    const double rf = 1.0 / c->edt_res;
    auto edt_at = [&](const float p[3]) -> float {
        int cc[3];
        const int dims[3] = {nx, ny, nz};
        for (int a = 0; a < 3; a++) {
            cc[a] = (int)std::floor(rf * (double)p[a]) + 32768 - c->edt_kmin[a];
            if (cc[a] < 0 || cc[a] >= dims[a]) return -1.0f;
        }
        return c->h_edt[((size_t)cc[0] * ny + cc[1]) * nz + cc[2]];
    };
    std::vector<unsigned char> occ(C * radii.size(), 0);
    const float margin = (float)c->cfg.grid_margin;
    for (size_t r = 0; r < radii.size(); r++)
        for (int i = 0; i < H; i++)
            for (int j = 0; j < W; j++)
                for (int k = 0; k < A; k++) {
                    const float p[3] = {(float)(c->grid_min[0] + i * res), (float)(c->grid_min[1] + j * res),
                                        (float)(c->grid_min[2] + k * res)};
                    if ((double)edt_at(p) < radii[r] + (double)margin) occ[r * C + (size_t)H * W * k + (size_t)W * i + j] = 1;
                }
    void *old[] = {c->d_edt, c->d_occ_static, c->d_goal_planned, c->d_goal_err, c->d_goal_flags, c->d_goal_exp, c->d_ray_stack, c->d_fcode};
    for (void *p : old) if (p) (void)hipFree(p);
    c->d_edt = c->d_goal_planned = c->d_ray_stack = nullptr; c->d_occ_static = nullptr;
    c->d_goal_err = c->d_goal_flags = c->d_goal_exp = nullptr; c->d_fcode = nullptr;
    if (!c->h_fcode.empty()) {
        HIPCHK(c, hipMalloc(&c->d_fcode, sizeof(uint32_t) * c->h_fcode.size()));
        HIPCHK(c, hipMemcpy(c->d_fcode, c->h_fcode.data(), sizeof(uint32_t) * c->h_fcode.size(), hipMemcpyHostToDevice));
    }
    HIPCHK(c, hipMalloc(&c->d_edt, sizeof(float) * c->h_edt.size()));
    HIPCHK(c, hipMemcpy(c->d_edt, c->h_edt.data(), sizeof(float) * c->h_edt.size(), hipMemcpyHostToDevice));
    HIPCHK(c, hipMalloc(&c->d_occ_static, occ.size()));
    HIPCHK(c, hipMemcpy(c->d_occ_static, occ.data(), occ.size(), hipMemcpyHostToDevice));
    const size_t N = (size_t)c->N;
    HIPCHK(c, hipMalloc(&c->d_goal_planned, sizeof(float) * 3 * N));
    HIPCHK(c, hipMalloc(&c->d_goal_err, sizeof(int) * N));
    HIPCHK(c, hipMalloc(&c->d_goal_flags, sizeof(int) * N));
    HIPCHK(c, hipMalloc(&c->d_goal_exp, sizeof(int) * N));
    HIPCHK(c, hipMemset(c->d_goal_err, 0, sizeof(int) * N));
    HIPCHK(c, hipMalloc(&c->d_ray_stack, sizeof(float) * N * 64 * 24 * 6));
    return LSC_OK;
}

// blocked-cell integral images, one per distinct agent radius:  blocked = EDT < r + res/2 - 1e-5
// (include/corridor_constructor.hpp:114)
static int build_integrals(lsc_ctx *c)
{
    if (c->h_edt.empty() || c->N == 0) return LSC_OK;
    const int nx = c->edt_dims[0], ny = c->edt_dims[1], nz = c->edt_dims[2];
    std::vector<double> radii;
    std::vector<int> img(c->N);
    for (int q = 0; q < c->N; q++) {
        auto it = std::find(radii.begin(), radii.end(), c->h_radius[q]);
        if (it == radii.end()) { radii.push_back(c->h_radius[q]); img[q] = (int)radii.size() - 1; }
        else img[q] = (int)(it - radii.begin());
    }
    const size_t isz = (size_t)(nx + 1) * (ny + 1) * (nz + 1);
    std::vector<int> I(isz * radii.size(), 0);
    for (size_t r = 0; r < radii.size(); r++) {
        const double thr = radii[r] + 0.5 * c->cfg.world_resolution - 1e-5;
        int *J = I.data() + r * isz;
        auto at = [&](int x, int y, int z) -> int & { return J[((size_t)x * (ny + 1) + y) * (nz + 1) + z]; };
        for (int x = 1; x <= nx; x++)
            for (int y = 1; y <= ny; y++)
                for (int z = 1; z <= nz; z++) {
                    const int b = (double)c->h_edt[((size_t)(x - 1) * ny + (y - 1)) * nz + (z - 1)] < thr ? 1 : 0;
                    at(x, y, z) = b + at(x - 1, y, z) + at(x, y - 1, z) + at(x, y, z - 1) - at(x - 1, y - 1, z) - at(x - 1, y, z - 1) -
                                  at(x, y - 1, z - 1) + at(x - 1, y - 1, z - 1);
                }
    }
    if (c->d_integral) { (void)hipFree(c->d_integral); c->d_integral = nullptr; }
    if (c->d_img_of_agent) { (void)hipFree(c->d_img_of_agent); c->d_img_of_agent = nullptr; }
    HIPCHK(c, hipMalloc(&c->d_integral, sizeof(int) * I.size()));
    HIPCHK(c, hipMemcpy(c->d_integral, I.data(), sizeof(int) * I.size(), hipMemcpyHostToDevice));
    HIPCHK(c, hipMalloc(&c->d_img_of_agent, sizeof(int) * (size_t)c->N));
    HIPCHK(c, hipMemcpy(c->d_img_of_agent, img.data(), sizeof(int) * (size_t)c->N, hipMemcpyHostToDevice));
    return build_goal_grid(c, radii);
}

int lsc_set_distmap(lsc_ctx *c, const float *edt, int nx, int ny, int nz, const int key_min[3], double res)
{
    if (!c || !edt || nx < 1 || ny < 1 || nz < 1 || !key_min || !(res > 0)) return LSC_EINVAL;
    // The corridor kernel visits the reference's lattice samples (spacing world/resolution) as contiguous cell ranges of
    // the distance field: that is only the same set when both resolutions agree (they do in every shipped launch file)
    if (std::fabs(res - c->cfg.world_resolution) > 1e-9 * res) {
        c->err = "lsc_set_distmap: map resolution differs from world_resolution";
        return LSC_EINVAL;
    }
    HIPCHK(c, hipSetDevice(c->cfg.device));
    c->h_edt.assign(edt, edt + (size_t)nx * ny * nz);
    c->edt_dims[0] = nx; c->edt_dims[1] = ny; c->edt_dims[2] = nz;
    for (int k = 0; k < 3; k++) c->edt_kmin[k] = key_min[k];
    c->edt_res = res;
    return build_integrals(c);
}

// goalPlanning(): fused into phase A of the plan kernel on maps without a distance field; with one it is a launch of
// its own (lsc_goal.hip) whose output replaces the goal input of the SFC and plan kernels
static int run_goal(lsc_ctx *c, const float *d_state, const float *&d_goal, const float *d_prev, int seq, hipStream_t st)
{
    if (!(c->cfg.goal_mode == 1 && c->cfg.use_octomap)) return LSC_OK;
    if (!c->d_occ_static) { c->err = "goal_mode prior_based with use_octomap: lsc_set_distmap was not called"; return LSC_ESTATE; }
    GoalArgs g;
    g.N = c->N; g.first = c->first; g.count = c->count; g.planner_seq = seq; g.dtf = (float)c->cfg.dt;
    g.state = d_state; g.goal = d_goal; g.traj_prev = d_prev;
    g.radius = c->d_radius; g.downwash = c->d_downwash; g.radius_obs = c->d_radius_obs; g.downwash_obs = c->d_downwash_obs;
    g.goal_threshold = c->cfg.goal_threshold; g.priority_dist_threshold = c->cfg.priority_dist_threshold;
    g.goal_radius = c->cfg.goal_radius;
    g.H = c->grid_dims[0]; g.W = c->grid_dims[1]; g.A = c->grid_dims[2]; g.dim2 = c->cfg.world_dimension == 2 ? 1 : 0;
    for (int k = 0; k < 3; k++) { g.gmin[k] = c->grid_min[k]; g.key_min[k] = c->edt_kmin[k]; }
    g.gres = c->cfg.grid_resolution;
    g.occ_static = c->d_occ_static; g.img_of_agent = c->d_img_of_agent;
    g.edt = c->d_edt; g.nx = c->edt_dims[0]; g.ny = c->edt_dims[1]; g.nz = c->edt_dims[2];
    g.rf = 1.0 / c->edt_res; g.wres = c->cfg.world_resolution;
    g.n_nb = (int)c->nb_seq.size();
    for (int k = 0; k < 16; k++) {
        g.nb_seq[k] = k < g.n_nb ? c->nb_seq[k] : 0;
        g.nb_magic[k] = k < g.n_nb ? (uint32_t)(0x100000000ull / (uint32_t)c->nb_seq[k]) : 0u;
    }
    g.row_cap = c->grid_row_cap;
    g.variant = c->cfg.goal_search == 1 ? 0 : goal_fast_slots(g.H, g.W, g.A, &g.jbits);
    if (g.variant == 0) g.jbits = 0;
    g.fcode = nullptr; g.fcode_n = 0; g.fcode_rb = 0;
    if (g.variant != 0 && c->d_fcode) { g.variant |= 8; g.fcode = c->d_fcode; g.fcode_n = (int)c->h_fcode.size(); g.fcode_rb = c->fcode_rb; }
    g.goal_out = c->d_goal_planned; g.err = c->d_goal_err; g.flags = c->d_goal_flags; g.expansions = c->d_goal_exp;
    g.path_out = c->d_goal_path; g.path_cap = c->goal_path_cap; g.path_len = c->d_goal_plen;
    g.ray_stack = c->d_ray_stack;
    g.reset_thr = c->cfg.planner_mode == 0 ? c->cfg.reset_threshold : 0.0; g.ever = c->d_ever;
    g.prof = c->goal_profiling ? c->d_goal_prof : nullptr;
    hipEvent_t e1 = nullptr;
    if (c->timing && timing_begin(c, 3, st, &e1) != LSC_OK) return LSC_EHIP;
    HIPCHK(c, launch_goal(g, st));
    if (c->timing) HIPCHK(c, hipEventRecord(e1, st));
    d_goal = c->d_goal_planned;
    return LSC_OK;
}

static int run_sfc(lsc_ctx *c, const float *d_state, const float *d_goal, const float *d_prev, int seq, hipStream_t st)
{
    if (!c->cfg.use_octomap) return LSC_OK;
    if (!c->d_integral) { c->err = "use_octomap is set but lsc_set_distmap was not called"; return LSC_ESTATE; }
    SfcArgs s;
    s.N = c->N; s.first = c->first; s.count = c->count;
    s.state = d_state; s.goal = d_goal; s.traj_prev = d_prev;
    s.radius = c->d_radius; s.img_of_agent = c->d_img_of_agent; s.integral = c->d_integral;
    s.nx = c->edt_dims[0]; s.ny = c->edt_dims[1]; s.nz = c->edt_dims[2];
    for (int k = 0; k < 3; k++) { s.key_min[k] = c->edt_kmin[k]; s.world_min[k] = c->cfg.world_min[k]; s.world_max[k] = c->cfg.world_max[k]; }
    s.rf = 1.0 / c->edt_res; s.wres = c->cfg.world_resolution;
    s.sfc = c->d_sfc; s.init_flag = c->d_sfc_init; s.err = c->d_sfc_err;
    double ext = 0.0;
    for (int k = 0; k < 3; k++) ext = std::max(ext, (double)c->cfg.world_max[k] - (double)c->cfg.world_min[k]);
    s.table_len = (int)std::ceil(ext / c->cfg.world_resolution) + 8;
    s.planner_seq = seq; s.reset_thr = c->cfg.planner_mode == 0 ? c->cfg.reset_threshold : 0.0;
    if (sizeof(double) * 6 * (size_t)s.table_len > 160 * 1024) {
        c->err = "world extent / world_resolution too large for the SFC face tables (limit 3400 steps per axis)";
        return LSC_EINVAL;
    }
    hipEvent_t e1 = nullptr;
    if (c->timing && timing_begin(c, 4, st, &e1) != LSC_OK) return LSC_EHIP;
    HIPCHK(c, launch_sfc(s, st));
    if (c->timing) HIPCHK(c, hipEventRecord(e1, st));
    return LSC_OK;
}

static int fill_plan_args(lsc_ctx *c, PlanArgs &a, const float *d_state, const float *d_goal, const float *d_prev, int seq,
                          float *d_next, double *d_cost, int *d_status, int *d_iters)
{
    if (c->N == 0) return LSC_ESTATE;
    a.model = c->d_model; a.terms = c->d_terms; a.entries = c->d_entries;
    a.N = c->N; a.first = c->first; a.count = c->count; a.planner_seq = seq; a.cap = c->cap;
    a.dim2 = c->cfg.world_dimension == 2 ? 1 : 0;
    a.solver = c->cfg.solver;
    a.solver_stats = c->d_iters_acc ? c->d_iters_acc + 2 * (size_t)c->N : nullptr;      // ([N][4] counters behind the two per-agent blocks)
    a.cap_tp = (c->count > c->n_cu) ? c->cap_tp : 0; a.smem_tp = c->smem_tp;
    a.order = (c->count > 2 * c->n_cu) ? c->d_order : nullptr;   // more than one round of throughput workgroups
    a.obs_bound = c->d_obs_bound;                                // obstacle-level pre-cull (throughput build; latency build of large swarms: launch_plan decides)
    a.neigh = c->neigh.cnt ? &c->neigh : nullptr;                // neighbour lists of a large swarm: run_plan launches their kernels and sets the three fields below
    a.nv = nullptr;
    a.state = d_state; a.goal = d_goal; a.traj_prev = d_prev;
    a.radius = c->d_radius; a.radius_obs = c->d_radius_obs; a.downwash = c->d_downwash; a.downwash_obs = c->d_downwash_obs;
    a.vmax = c->d_vmax; a.amax = c->d_amax; a.vnom = c->d_vnom;
    a.traj_next = d_next; a.cost = d_cost; a.status = d_status; a.iters = d_iters; a.nrows = c->d_nrows; a.bucket_max = c->d_bmax; a.iters_acc = c->d_iters_acc;
    a.stale = c->d_stale; a.sfc = c->cfg.use_octomap ? c->d_sfc : nullptr;
    a.sfc_err = c->cfg.use_octomap ? c->d_sfc_err : nullptr;
    const bool planned = c->cfg.goal_mode == 1 && c->cfg.use_octomap;   // goals come from lsc_goal_kernel
    a.goal_err = planned ? c->d_goal_err : nullptr;
    a.out_normal = nullptr; a.out_d = nullptr;
    a.goal_mode = planned ? 0 : c->cfg.goal_mode; a.goal_threshold = c->cfg.goal_threshold;
    a.priority_dist_threshold = c->cfg.priority_dist_threshold; a.goal_radius = c->cfg.goal_radius;
    a.goal_out = c->d_goal_cur; a.state_next = nullptr; a.finv = (float)std::pow(c->cfg.dt, -1);
    a.dbg = c->d_dbg; a.prof = c->profiling ? c->d_prof : nullptr;
    a.trace = c->trace_agent >= 0 ? c->d_trace : nullptr; a.trace_agent = c->trace_agent;
    a.spill_ws = c->d_spill; a.spill_stride = c->spill_stride;
    a.gmodel = c->d_gmodel; a.planner_mode = c->cfg.planner_mode; a.slack_mode = c->cfg.slack_mode;
    a.ncs = c->cfg.n_constraint_segments; a.general_all = (c->cfg.planner_mode == 1 || c->cfg.slack_mode != 0) ? 1 : 0;
    a.slack_w = c->cfg.slack_collision_weight; a.reset_thr = c->cfg.planner_mode == 0 ? c->cfg.reset_threshold : 0.0;
    a.ever = c->d_ever; a.gen_ws = c->d_gen_ws; a.gen_stride = c->gen_stride;
    return LSC_OK;
}

// Whether lsc_general_kernel has to run in this tick.  The alternate-mode kernel follows the plan kernel when a
// configured mode sends every agent there (BVC, slack modes).  With the disturbance checks on, a host-buffer tick knows
// the answer exactly (same float32 test on the host copy of the inputs, general_hint 0 / 1); a device-resident tick does
// not see the states (hint -1) and launches the kernel, whose workgroups leave at once when nobody was flagged.
static bool want_general(const lsc_ctx *c, int general_hint)
{
    if (!c->d_gen_ws) return false;
    if (c->cfg.planner_mode == 1 || c->cfg.slack_mode != 0) return true;
    return general_hint != 0;
}

// obstaclePredictionCheck / initialTrajPlanningCheck on host copies (src/traj_planner.cpp:866-878, 1047-1061); keeps the
// host mirror of the persistent flags in step with the device's
static int host_disturbance_hint(lsc_ctx *c, const float *state, const float *prev_traj, int planner_seq)
{
    if (!(c->cfg.reset_threshold > 0.0) || c->cfg.planner_mode != 0) return 0;
    if (c->h_ever_stale) {
        // device-resident ticks have run since the mirror was last in step: they flag agents on the device only
        if (c->d_ever && !c->h_ever.empty() &&
            (hipDeviceSynchronize() != hipSuccess ||
             hipMemcpy(c->h_ever.data(), c->d_ever, c->h_ever.size(), hipMemcpyDeviceToHost) != hipSuccess)) return 1;   // when in doubt, launch
        c->h_ever_stale = false;
    }
    int any = 0;
    for (int q = 0; q < c->N; q++) {
        if (planner_seq >= 2) {
            const float *t = prev_traj + (size_t)q * NV + NC, *s = state + 9 * q;
            const float dx = t[0] - s[0], dy = t[SEGV] - s[1], dz = t[2 * SEGV] - s[2];
            const float n2 = dx * dx + dy * dy + dz * dz;
            if (std::sqrt((double)n2) > c->cfg.reset_threshold) c->h_ever[q] = 1;
        }
        any |= c->h_ever[q];
    }
    return any;
}

// Planar worlds (world/dimension == 2).  The reference reads the planning agent's own position at z = world/z_2d whatever it is told
// (TrajPlanner::currentStateCallback, src/traj_planner.cpp:304-314), stores z = z_2d in every control point it plans
// (src/traj_optimizer.cpp:87-90) and drops the z term of every collision row (:450) -- which is only meaningful when the whole swarm
// sits in that plane, as it does in the reference's simulator (starts at z_2d, src/mission.cpp:88-93; ideal states evaluated from
// planar plans).  The kernels rely on exactly that, so the host-buffer ticks check it instead of planning something else silently.
static int planar_inputs_ok(lsc_ctx *c, const float *state, const float *prev_traj, int planner_seq)
{
    if (c->cfg.world_dimension != 2) return LSC_OK;
    const float z = (float)c->cfg.world_z_2d;
    for (int q = 0; q < c->N; q++) {
        bool ok = state[9 * (size_t)q + 2] == z;
        if (planner_seq >= 2) {
            const float *t = prev_traj + (size_t)q * NV + 2 * SEGV;
            for (int j = 0; j < SEGV; j++) ok = ok && t[j] == z;
        }
        if (!ok) {
            c->err = "planar world (world_dimension 2): agent " + std::to_string(q) + " is not at z = world_z_2d (state / previous plan)";
            return LSC_EINVAL;
        }
    }
    return LSC_OK;
}

static int run_plan(lsc_ctx *c, const PlanArgs &a_in, hipStream_t st, int general_hint = -1)
{
    const size_t smem = plan_smem_bytes(c->hm.m.n_terms, c->hm.m.n_entries, c->cap);
    hipEvent_t e1 = nullptr;
    if (c->timing && timing_begin(c, 0, st, &e1) != LSC_OK) return LSC_EHIP;
    PlanArgs a = a_in;
    // Neighbour lists cost two launches (~25 us at 1024 agents) and save every workgroup its walks over all N agents (cull: ~5 us per agent at
    // N = 1024, priority rule: ~4 us; both grow with N).  Worth it when the shard takes more than one round of workgroups or the swarm is
    // large; a one-round shard of a 1024-agent swarm is faster with the in-kernel walks (profiles/r06_neighbour_lists.log).
    const bool lists_pay = a.N >= 2048 || (a.cap_tp > 0 && a.count > 2 * c->n_cu) || c->neigh_always;
    if (a.neigh && lists_pay && a.count > 0 && !a.out_normal && !a.general_all && !a.trace) {
        // large swarm: bounds of every agent + the grid, then the shard's unit lists (two small launches on the tick's stream, inside the timed region)
        NeighArgs &g = *a.neigh;
        if (++g.tag == 0) { HIPCHK(c, hipMemsetAsync(g.cells, 0, 32 * ((size_t)g.hmask + 1), st)); HIPCHK(c, hipMemsetAsync(g.glob, 0, 128, st)); g.tag = 1; }
        g.N = a.N; g.first = a.first; g.count = a.count; g.planner_seq = a.planner_seq; g.dtf = (float)c->hm.m.dt; g.dim2 = a.dim2;
        g.hv_scale = c->hm.m.hv_scale; g.ha_scale = c->hm.m.ha_scale; g.z2d = c->hm.m.z2d;
        g.state = a.state; g.traj_prev = a.traj_prev;
        g.radius = a.radius; g.radius_obs = a.radius_obs; g.downwash = a.downwash; g.downwash_obs = a.downwash_obs; g.vmax = a.vmax; g.amax = a.amax;
        g.order = (a.cap_tp > 0) ? a.order : nullptr; g.iters = a.iters; g.nrows = a.nrows; g.obs_bound = a.obs_bound;
        // phase A's walks over all agents: candidates of the priority rule by position, the disturbance checks once per agent
        const bool alt = a.general_all || (a.reset_thr > 0.0 && a.ever);
        g.goal_mode = a.goal_mode; g.prio_thr = a.priority_dist_threshold;
        g.checks = (alt && a.reset_thr > 0.0 && a.planner_seq >= 2 && a.planner_mode == 0 && a.ever != nullptr) ? 1 : 0;
        g.reset_thr = a.reset_thr; g.ever = a.ever;
        HIPCHK(c, launch_neigh(g, st));
        a.nv = g.view;
    }
    HIPCHK(c, launch_plan(a, smem, st));
    if (c->d_spill) HIPCHK(c, launch_plan_spill(a, c->spill_slots, plan_smem_bytes(c->hm.m.n_terms, c->hm.m.n_entries, 0), st));
    if (want_general(c, general_hint)) HIPCHK(c, launch_general(a, c->gen_slots, st));
    if (c->timing) HIPCHK(c, hipEventRecord(e1, st));
    return LSC_OK;
}

int lsc_tick_device(lsc_ctx *c, const float *d_state, const float *d_goal, const float *d_traj_prev, int planner_seq,
                    float *d_traj_next, double *d_cost, int *d_status, int *d_iters, void *hip_stream)
{
    if (!c || !d_state || !d_goal || !d_traj_prev || !d_traj_next || !d_cost || !d_status || !d_iters) return LSC_EINVAL;
    PlanArgs a;
    if (c->N == 0) return LSC_ESTATE;
    int rc = run_goal(c, d_state, d_goal, d_traj_prev, planner_seq, (hipStream_t)hip_stream);
    if (rc) return rc;
    rc = fill_plan_args(c, a, d_state, d_goal, d_traj_prev, planner_seq, d_traj_next, d_cost, d_status, d_iters);
    if (rc) return rc;
    a.state_next = c->fused_state_next;
    rc = run_sfc(c, d_state, d_goal, d_traj_prev, planner_seq, (hipStream_t)hip_stream);
    if (rc) return rc;
    if (c->cfg.reset_threshold > 0.0) c->h_ever_stale = true;   // the device may flag agents the host mirror does not see
    return run_plan(c, a, (hipStream_t)hip_stream);
}

int lsc_tick_device_fused(lsc_ctx *c, const float *d_state, const float *d_goal, const float *d_traj_prev, int planner_seq,
                          float *d_traj_next, float *d_state_next, double *d_cost, int *d_status, int *d_iters, void *hip_stream)
{
    if (!c || !d_state_next) return LSC_EINVAL;
    c->fused_state_next = d_state_next;
    const int rc = lsc_tick_device(c, d_state, d_goal, d_traj_prev, planner_seq, d_traj_next, d_cost, d_status, d_iters, hip_stream);
    c->fused_state_next = nullptr;
    return rc;
}

// One tick of several independent swarms in ONE launch (lsc_plan_batch_kernel, blockIdx.y = swarm): the reference's mission list
// (src/multi_sync_simulator_node.cpp:43-70, src/param.cpp:106-122) as a batch axis.  Every context plans exactly what
// lsc_tick_device_fused would plan for it -- the same instantiation of the planning code reads the context's own argument block --
// so the results are the same bits; what changes is that a 64-agent swarm no longer has a 256-CU chip to itself.
int lsc_tick_device_fused_batch(lsc_ctx *const *ctx, int n, const float *const *d_state, const float *const *d_goal,
                                const float *const *d_traj_prev, const int *planner_seq, float *const *d_traj_next,
                                float *const *d_state_next, double *const *d_cost, int *const *d_status, int *const *d_iters,
                                void *hip_stream)
{
    if (!ctx || n < 1 || !ctx[0]) return LSC_EINVAL;
    lsc_ctx *c0 = ctx[0];
    if (n > PLAN_BATCH_MAX) { c0->err = "lsc_tick_device_fused_batch: at most " + std::to_string(PLAN_BATCH_MAX) + " swarms per launch"; return LSC_EINVAL; }
    if (!d_state || !d_goal || !d_traj_prev || !planner_seq || !d_traj_next || !d_state_next || !d_cost || !d_status || !d_iters) return LSC_EINVAL;
    PlanArgs a[PLAN_BATCH_MAX];
    size_t smem = 0;
    int slots = 0x7fffffff;
    bool general = false;
    for (int i = 0; i < n; i++) {
        lsc_ctx *c = ctx[i];
        if (!c || !d_state[i] || !d_goal[i] || !d_traj_prev[i] || !d_traj_next[i] || !d_state_next[i] || !d_cost[i] || !d_status[i] || !d_iters[i]) return LSC_EINVAL;
        if (c->N == 0) { c0->err = "lsc_tick_device_fused_batch: context " + std::to_string(i) + " has no agents"; return LSC_ESTATE; }
        // what one launch can hold: swarms on this device that take the latency build with their rows in LDS and nothing between
        // their launches (maps with a distance field run the goal search and the corridor kernel first; sharded swarms exchange)
        const char *why = nullptr;
        if (c->cfg.device != c0->cfg.device) why = "is on another device";
        else if (c->cfg.use_octomap) why = "has a distance field (goal search and corridor launches precede its plan kernel)";
        else if (c->comm || c->world != 1) why = "is a rank of a sharded swarm";
        else if (c->count > c->n_cu) why = "has more agents than the GPU has CUs (the throughput build is not batched)";
        else if (c->d_spill) why = "needs the second pass (row capacity below 27 (N - 1))";
        else if (c->profiling || c->trace_agent >= 0) why = "is being profiled / traced";
        else if ((c->cfg.solver >= 1) != (c0->cfg.solver >= 1)) why = "has another QP solver than the first (one instantiation per launch)";
        for (int j = 0; j < i && !why; j++) if (ctx[j] == c) why = "appears twice (its stale-plan and hand-over buffers belong to ONE swarm of the launch)";
        if (why) { c0->err = "lsc_tick_device_fused_batch: context " + std::to_string(i) + " " + why; return LSC_EINVAL; }
        const int rc = fill_plan_args(c, a[i], d_state[i], d_goal[i], d_traj_prev[i], planner_seq[i], d_traj_next[i], d_cost[i], d_status[i], d_iters[i]);
        if (rc) return rc;
        a[i].state_next = d_state_next[i];
        const size_t sm = plan_smem_bytes(c->hm.m.n_terms, c->hm.m.n_entries, c->cap);
        smem = sm > smem ? sm : smem;
        if (c->cfg.reset_threshold > 0.0) c->h_ever_stale = true;
        if (want_general(c, -1)) { general = true; slots = c->gen_slots < slots ? c->gen_slots : slots; }
    }
    // (all or none: the alternate-mode hooks are one instantiation per launch, and the hand-over launch covers every swarm of the batch)
    for (int i = 0; i < n; i++)
        if (want_general(ctx[i], -1) != general) { c0->err = "lsc_tick_device_fused_batch: contexts with and without alternate-mode hooks in one batch"; return LSC_EINVAL; }
    hipStream_t st = (hipStream_t)hip_stream;
    hipEvent_t e1 = nullptr;
    if (c0->timing && timing_begin(c0, 0, st, &e1) != LSC_OK) return LSC_EHIP;      // (one launch: timed on the first context)
    if (launch_plan_batch(a, n, smem, st) != hipSuccess) { c0->err = "lsc_tick_device_fused_batch: launch failed (contexts of different planar / alternate-mode classes?)"; return LSC_EHIP; }
    if (general) HIPCHK(c0, launch_general_batch(a, n, slots, st));
    if (c0->timing) HIPCHK(c0, hipEventRecord(e1, st));
    return LSC_OK;
}

int lsc_replan_tick(lsc_ctx *c, const float *state, const float *goal, const float *prev_traj, int planner_seq,
                    float *out_traj, double *out_cost, int *out_status, int *out_iters, float *out_lsc_normal,
                    double *out_lsc_d, float *out_sfc)
{
    if (!c || !state || !goal || !prev_traj || !out_traj || !out_cost || !out_status) return LSC_EINVAL;
    if (c->N == 0) return LSC_ESTATE;
    const auto t_entry = std::chrono::steady_clock::now();
    if (int prc = planar_inputs_ok(c, state, prev_traj, planner_seq)) return prc;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    const size_t N = c->N, cnt = c->count, first = c->first, nobs = N - 1;
    hipStream_t st = c->stream;
    c->last_host_seq = planner_seq;
    std::memcpy(c->h_in, state, sizeof(float) * 9 * N);
    std::memcpy(c->h_in + 9 * N, goal, sizeof(float) * 3 * N);
    std::memcpy(c->h_in + 12 * N, prev_traj, sizeof(float) * NV * N);
    HIPCHK(c, hipMemcpyAsync(c->d_state, c->h_in, sizeof(float) * (9 + 3 + NV) * N, hipMemcpyHostToDevice, st));
    PlanArgs a;
    const float *d_goal_in = c->d_goal;
    int rc = run_goal(c, c->d_state, d_goal_in, c->d_prev, planner_seq, st);
    if (rc) return rc;
    rc = fill_plan_args(c, a, c->d_state, d_goal_in, c->d_prev, planner_seq, c->d_next, c->d_cost, c->d_status, c->d_iters);
    if (rc) return rc;
    if (out_lsc_normal || out_lsc_d) {
        if (!c->d_onormal) {
            HIPCHK(c, hipMalloc(&c->d_onormal, sizeof(float) * 3 * M * nobs * N + 16));
            HIPCHK(c, hipMalloc(&c->d_od, sizeof(double) * NC * M * nobs * N + 16));
        }
        a.out_normal = c->d_onormal; a.out_d = c->d_od;
    }
    rc = run_sfc(c, c->d_state, d_goal_in, c->d_prev, planner_seq, st);
    if (rc) return rc;
    rc = run_plan(c, a, st, host_disturbance_hint(c, state, prev_traj, planner_seq));
    if (rc) return rc;
    const size_t Np = (size_t)c->table_rows;
    const size_t out_bytes = (sizeof(double) + sizeof(float) * NV + 2 * sizeof(int)) * Np;
    HIPCHK(c, hipMemcpyAsync(c->h_out, c->d_cost, out_bytes, hipMemcpyDeviceToHost, st));
    if (out_lsc_normal) HIPCHK(c, hipMemcpyAsync(out_lsc_normal, c->d_onormal, sizeof(float) * 3 * M * nobs * cnt, hipMemcpyDeviceToHost, st));
    if (out_lsc_d) HIPCHK(c, hipMemcpyAsync(out_lsc_d, c->d_od, sizeof(double) * NC * M * nobs * cnt, hipMemcpyDeviceToHost, st));
    if (out_sfc) HIPCHK(c, hipMemcpyAsync(out_sfc, c->d_sfc + first * M * 6, sizeof(float) * M * 6 * cnt, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    {
        const unsigned char *o = c->h_out;
        std::memcpy(out_cost, o + sizeof(double) * first, sizeof(double) * cnt);
        std::memcpy(out_traj, o + sizeof(double) * Np + sizeof(float) * NV * first, sizeof(float) * NV * cnt);
        const unsigned char *si = o + (sizeof(double) + sizeof(float) * NV) * Np;
        std::memcpy(out_status, si + sizeof(int) * first, sizeof(int) * cnt);
        if (out_iters) std::memcpy(out_iters, si + sizeof(int) * (Np + first), sizeof(int) * cnt);
        for (size_t q = 0; q < cnt; q++)
            if (out_status[q] == LSC_STATUS_GENERAL_K) {      // handed to the general kernel and never solved: an internal error, not a report
                c->err = "internal: an agent was handed to the alternate-mode kernel, which did not run";
                return LSC_ESTATE;
            }
    }
    c->next_has_all_rows = c->count == c->N;
    if (c->timing)
        c->host_tick_ms.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_entry).count());
    return LSC_OK;
}

// ---------------------------------------------------------------------------------------------------
// Agent-sharded multi-GPU.  The reference hands every agent every other agent's previous trajectory in
// MultiSyncSimulator::update (src/multi_sync_simulator.cpp:297-303); with the swarm partitioned over ranks that
// hand-over is the one exchange of a tick: an in-place RCCL all-gather of the freshly planned rows.
// ---------------------------------------------------------------------------------------------------
int lsc_comm_unique_id(unsigned char id[LSC_COMM_ID_BYTES])
{
    static_assert(LSC_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    const RcclApi *api = rccl_api();
    if (!api || !id) return LSC_ECOMM;
    ncclUniqueId u;
    if (api->GetUniqueId(&u) != ncclSuccess) return LSC_ECOMM;
    std::memcpy(id, u.internal, LSC_COMM_ID_BYTES);
    return LSC_OK;
}

int lsc_comm_init(lsc_ctx *c, int world_size, int rank, const unsigned char id[LSC_COMM_ID_BYTES])
{
    if (!c || world_size < 1 || rank < 0 || rank >= world_size || !id) return LSC_EINVAL;
    if (c->N != 0) { c->err = "lsc_comm_init must precede lsc_set_agents (the tables are padded to the world size)"; return LSC_ESTATE; }
    if (c->comm) { c->err = "lsc_comm_init: communicator already initialised"; return LSC_ESTATE; }
    const RcclApi *api = rccl_api();
    if (!api) { c->err = "librccl.so could not be loaded"; return LSC_ECOMM; }
    HIPCHK(c, hipSetDevice(c->cfg.device));
    ncclUniqueId u;
    std::memcpy(u.internal, id, LSC_COMM_ID_BYTES);
    NCCLCHK(c, api, api->CommInitRank(&c->comm, world_size, u, rank));
    c->world = world_size; c->rank = rank;
    return LSC_OK;
}

int lsc_comm_info(const lsc_ctx *c, int *world_size, int *rank, int *shard_rows, int *table_rows)
{
    if (!c) return LSC_EINVAL;
    if (world_size) *world_size = c->world;
    if (rank) *rank = c->rank;
    if (shard_rows) *shard_rows = c->shard_rows;
    if (table_rows) *table_rows = c->table_rows;
    return LSC_OK;
}

// in-place all-gather of `rows_elems` elements per agent row of a table padded to table_rows rows
static int exchange_rows(lsc_ctx *c, const RcclApi *api, void *table, size_t row_bytes, hipStream_t st)
{
    const size_t cnt = (size_t)c->shard_rows * row_bytes;
    NCCLCHK(c, api, api->AllGather(static_cast<unsigned char *>(table) + (size_t)c->rank * cnt, table, cnt, ncclUint8, c->comm, st));
    return LSC_OK;
}

int lsc_tick_device_sharded(lsc_ctx *c, float *d_state, const float *d_goal, const float *d_traj_prev, int planner_seq,
                            float *d_traj_next, double *d_cost, int *d_status, int *d_iters, void *hip_stream)
{
    if (!c || !d_state) return LSC_EINVAL;
    if (!c->comm) { c->err = "lsc_tick_device_sharded: lsc_comm_init was not called"; return LSC_ESTATE; }
    const RcclApi *api = rccl_api();
    hipStream_t st = (hipStream_t)hip_stream;
    int rc = lsc_tick_device(c, d_state, d_goal, d_traj_prev, planner_seq, d_traj_next, d_cost, d_status, d_iters, hip_stream);
    if (rc) return rc;
    hipEvent_t e1 = nullptr;
    if (c->timing && timing_begin(c, 2, st, &e1) != LSC_OK) return LSC_EHIP;
    rc = exchange_rows(c, api, d_traj_next, sizeof(float) * NV, st);
    if (rc) return rc;
    if (c->timing) HIPCHK(c, hipEventRecord(e1, st));
    // MultiSyncSimulator::update of the next tick: every rank needs every agent's ideal state
    HIPCHK(c, launch_propagate(d_traj_next, d_state, c->N, c->cfg.dt, st));
    return LSC_OK;
}

int lsc_replan_tick_all(lsc_ctx *c, const float *state, const float *goal, const float *prev_traj, int planner_seq,
                        float *out_traj, double *out_cost, int *out_status, int *out_iters, float *out_goal)
{
    if (!c || !state || !goal || !prev_traj || !out_traj || !out_cost || !out_status) return LSC_EINVAL;
    if (c->N == 0) return LSC_ESTATE;
    if (!c->comm) { c->err = "lsc_replan_tick_all: lsc_comm_init was not called"; return LSC_ESTATE; }
    if (int prc = planar_inputs_ok(c, state, prev_traj, planner_seq)) return prc;
    const RcclApi *api = rccl_api();
    HIPCHK(c, hipSetDevice(c->cfg.device));
    const size_t N = c->N, Np = (size_t)c->table_rows;
    hipStream_t st = c->stream;
    c->last_host_seq = planner_seq;
    std::memcpy(c->h_in, state, sizeof(float) * 9 * N);
    std::memcpy(c->h_in + 9 * N, goal, sizeof(float) * 3 * N);
    std::memcpy(c->h_in + 12 * N, prev_traj, sizeof(float) * NV * N);
    HIPCHK(c, hipMemcpyAsync(c->d_state, c->h_in, sizeof(float) * (9 + 3 + NV) * N, hipMemcpyHostToDevice, st));
    PlanArgs a;
    const float *d_goal_in = c->d_goal;
    int rc = run_goal(c, c->d_state, d_goal_in, c->d_prev, planner_seq, st);
    if (rc) return rc;
    rc = fill_plan_args(c, a, c->d_state, d_goal_in, c->d_prev, planner_seq, c->d_next, c->d_cost, c->d_status, c->d_iters);
    if (rc) return rc;
    rc = run_sfc(c, c->d_state, d_goal_in, c->d_prev, planner_seq, st);
    if (rc) return rc;
    rc = run_plan(c, a, st, host_disturbance_hint(c, state, prev_traj, planner_seq));
    if (rc) return rc;
    hipEvent_t e1 = nullptr;
    if (c->timing && timing_begin(c, 2, st, &e1) != LSC_OK) return LSC_EHIP;
    NCCLCHK(c, api, api->GroupStart());
    {   // the group is closed on every path: a communicator left inside an open group is unusable
        rc = exchange_rows(c, api, c->d_next, sizeof(float) * NV, st);
        if (!rc) rc = exchange_rows(c, api, c->d_cost, sizeof(double), st);
        if (!rc) rc = exchange_rows(c, api, c->d_status, sizeof(int), st);
        if (!rc) rc = exchange_rows(c, api, c->d_iters, sizeof(int), st);
        if (!rc) rc = exchange_rows(c, api, c->d_goal_cur, sizeof(float) * 3, st);
        const ncclResult_t ge = api->GroupEnd();
        if (rc) return rc;
        if (ge != ncclSuccess) { c->err = std::string("ncclGroupEnd: ") + api->GetErrorString(ge); return LSC_ECOMM; }
    }
    if (c->timing) HIPCHK(c, hipEventRecord(e1, st));
    const size_t out_bytes = (sizeof(double) + sizeof(float) * NV + 2 * sizeof(int)) * Np;
    HIPCHK(c, hipMemcpyAsync(c->h_out, c->d_cost, out_bytes, hipMemcpyDeviceToHost, st));
    if (out_goal) HIPCHK(c, hipMemcpyAsync(out_goal, c->d_goal_cur, sizeof(float) * 3 * N, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    const unsigned char *o = c->h_out;
    std::memcpy(out_cost, o, sizeof(double) * N);
    std::memcpy(out_traj, o + sizeof(double) * Np, sizeof(float) * NV * N);
    const unsigned char *si = o + (sizeof(double) + sizeof(float) * NV) * Np;
    std::memcpy(out_status, si, sizeof(int) * N);
    if (out_iters) std::memcpy(out_iters, si + sizeof(int) * Np, sizeof(int) * N);
    for (size_t q = 0; q < N; q++)
        if (out_status[q] == LSC_STATUS_GENERAL_K) {
            c->err = "internal: an agent was handed to the alternate-mode kernel, which did not run";
            return LSC_ESTATE;
        }
    c->next_has_all_rows = true;
    return LSC_OK;
}

// MultiSyncSimulator::savePlanningResult's agent-agent accounting (src/multi_sync_simulator.cpp:446-503) for the plans of the last
// host-buffer tick.  Sample time t falls into segment m = (int)(t / dt) at local parameter t / dt - m (getStateFromControlPoints,
// include/polynomial.hpp); the Bernstein weights are computed here, on the host, with the reference's own expression
// (nChoosek * pow(t, i) * pow(1 - t, n - i)) so that the device only repeats its multiply-adds.
int lsc_safety_ratio(lsc_ctx *c, const double *times, int n_times, double *out_ratio, int *out_partner, double *out_min)
{
    if (!c || !times || n_times < 1 || n_times > 64 || !out_min) return LSC_EINVAL;
    if (c->N < 1 || c->last_host_seq < 1) { if (c) c->err = "lsc_safety_ratio: no host-buffer tick was planned yet"; return LSC_ESTATE; }
    if (!c->next_has_all_rows) {
        // the partners of every pair are read from the context's own table: a sharded lsc_replan_tick leaves the other ranks' rows stale
        c->err = "lsc_safety_ratio: the last tick planned a shard only (use lsc_replan_tick_all, or a context that owns the whole swarm)";
        return LSC_ESTATE;
    }
    HIPCHK(c, hipSetDevice(c->cfg.device));
    const int N = c->N, cnt = c->count;
    std::vector<double> w((size_t)n_times * NC);
    std::vector<int> seg(n_times);
    auto choose = [](int n, int k) { if (k * 2 > n) k = n - k; if (k == 0) return 1; int r = n; for (int i = 2; i <= k; i++) { r *= (n - i + 1); r /= i; } return r; };
    for (int ti = 0; ti < n_times; ti++) {
        const double t = times[ti];
        int m = (int)(t / c->cfg.dt);
        if (m == M && t < M * c->cfg.dt + 1e-9) m = M - 1;
        else if (m >= M || t < 0.0) { c->err = "lsc_safety_ratio: sample time outside the planning horizon"; return LSC_EINVAL; }
        seg[ti] = m;
        const double tl = t / c->cfg.dt - m;
        for (int i = 0; i < NC; i++) w[(size_t)ti * NC + i] = choose(DEG, i) * std::pow(tl, i) * std::pow(1 - tl, DEG - i);
    }
    {
        // (sized for THIS call's shard and sample count: lsc_set_shard may have enlarged the shard since the last call, and the
        // carve-up below depends on both)
        const size_t bytes = (size_t)n_times * (sizeof(double) * NC + sizeof(int) * 2 + sizeof(float) * 3 * (size_t)N + (sizeof(double) + sizeof(int)) * (size_t)cnt) + 64;
        if (!c->d_safety || c->safety_bytes < bytes) {
            if (c->d_safety) (void)hipFree(c->d_safety);
            c->d_safety = nullptr; c->safety_bytes = 0;
            HIPCHK(c, hipMalloc(&c->d_safety, bytes));
            c->safety_bytes = bytes;
        }
    }
    // carve-up: ratios, weights, global minimum | positions | partners, segments
    double *d_ratio = reinterpret_cast<double *>(c->d_safety);
    double *d_w = d_ratio + (size_t)n_times * cnt;
    double *d_min = d_w + (size_t)n_times * NC;
    float *d_pos = reinterpret_cast<float *>(d_min + 1);
    int *d_partner = reinterpret_cast<int *>(d_pos + (size_t)n_times * 3 * N);
    int *d_seg = d_partner + (size_t)n_times * cnt;
    hipStream_t st = c->stream;
    HIPCHK(c, hipMemcpyAsync(d_w, w.data(), sizeof(double) * w.size(), hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(d_seg, seg.data(), sizeof(int) * seg.size(), hipMemcpyHostToDevice, st));
    HIPCHK(c, launch_safety(c->d_next, d_w, d_seg, n_times, N, c->first, cnt, c->d_radius, c->d_downwash, d_pos, d_ratio, d_partner, st));
    std::vector<double> ratio((size_t)n_times * cnt);
    std::vector<int> partner((size_t)n_times * cnt);
    HIPCHK(c, hipMemcpyAsync(ratio.data(), d_ratio, sizeof(double) * ratio.size(), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(partner.data(), d_partner, sizeof(int) * partner.size(), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    double mn = 1e300;
    for (double r : ratio) mn = r < mn ? r : mn;
    if (c->comm) {                                  // one all-reduce brings the swarm's minimum to every rank
        const RcclApi *api = rccl_api();
        HIPCHK(c, hipMemcpyAsync(d_min, &mn, sizeof(double), hipMemcpyHostToDevice, st));
        NCCLCHK(c, api, api->AllReduce(d_min, d_min, 1, ncclDouble, ncclMin, c->comm, st));
        HIPCHK(c, hipMemcpyAsync(&mn, d_min, sizeof(double), hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
    }
    if (out_ratio) std::memcpy(out_ratio, ratio.data(), sizeof(double) * ratio.size());
    if (out_partner) std::memcpy(out_partner, partner.data(), sizeof(int) * partner.size());
    *out_min = mn;
    return LSC_OK;
}

int lsc_propagate_device(lsc_ctx *c, const float *d_traj, float *d_state, void *hip_stream)
{
    if (!c || !d_traj || !d_state) return LSC_EINVAL;
    if (c->N == 0) return LSC_ESTATE;
    HIPCHK(c, launch_propagate(d_traj, d_state, c->N, c->cfg.dt, (hipStream_t)hip_stream));
    return LSC_OK;
}

static int sweep_device(lsc_ctx *c, const float *d_state, const float *d_traj_prev, int planner_seq, float *d_normal, double *d_d,
                        float *d_d32, void *hip_stream)
{
    if (!c || !d_state || !d_traj_prev || !d_normal || (!d_d && !d_d32)) return LSC_EINVAL;
    if (c->N < 2) return LSC_ESTATE;
    SweepArgs a;
    a.N = c->N; a.first = c->first; a.count = c->count; a.planner_seq = planner_seq; a.dtf = (float)c->cfg.dt;
    a.state = d_state; a.traj_prev = d_traj_prev;
    a.radius = c->d_radius; a.radius_obs = c->d_radius_obs; a.downwash = c->d_downwash; a.downwash_obs = c->d_downwash_obs;
    a.out_normal = d_normal; a.out_d = d_d; a.out_d32 = d_d32;
    hipStream_t st = (hipStream_t)hip_stream;
    hipEvent_t e1 = nullptr;
    if (c->timing && timing_begin(c, 1, st, &e1) != LSC_OK) return LSC_EHIP;
    HIPCHK(c, launch_sweep(a, st));
    if (c->timing) HIPCHK(c, hipEventRecord(e1, st));
    return LSC_OK;
}

int lsc_sweep_device(lsc_ctx *c, const float *d_state, const float *d_traj_prev, int planner_seq, float *d_normal,
                     double *d_d, void *hip_stream)
{
    return sweep_device(c, d_state, d_traj_prev, planner_seq, d_normal, d_d, nullptr, hip_stream);
}

int lsc_sweep_device_f32(lsc_ctx *c, const float *d_state, const float *d_traj_prev, int planner_seq, float *d_normal,
                         float *d_d32, void *hip_stream)
{
    return sweep_device(c, d_state, d_traj_prev, planner_seq, d_normal, nullptr, d_d32, hip_stream);
}

int lsc_gjk_batch(lsc_ctx *c, const double *pts, int count, double *v, double *dist)
{
    if (!c || !pts || count < 1 || !v || !dist) return LSC_EINVAL;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    double *d_p = nullptr, *d_v = nullptr, *d_d = nullptr;
    HIPCHK(c, hipMalloc(&d_p, sizeof(double) * 18 * (size_t)count));
    HIPCHK(c, hipMalloc(&d_v, sizeof(double) * 3 * (size_t)count));
    HIPCHK(c, hipMalloc(&d_d, sizeof(double) * (size_t)count));
    HIPCHK(c, hipMemcpy(d_p, pts, sizeof(double) * 18 * (size_t)count, hipMemcpyHostToDevice));
    HIPCHK(c, launch_gjk(d_p, count, d_v, d_d, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(v, d_v, sizeof(double) * 3 * (size_t)count, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(dist, d_d, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost));
    (void)hipFree(d_p); (void)hipFree(d_v); (void)hipFree(d_d);
    return LSC_OK;
}

int lsc_set_timing(lsc_ctx *c, int enabled)
{
    if (!c) return LSC_EINVAL;
    c->timing = enabled != 0;
    for (int w = 0; w < 5; w++) c->ev_used[w] = 0;
    c->host_tick_ms.clear();
    return LSC_OK;
}

int lsc_kernel_time_ms(lsc_ctx *c, int which, double *avg_ms, long *launches)
{
    if (!c || which < 0 || which > 5 || !avg_ms) return LSC_EINVAL;
    if (which == 5) {
        double t = 0;
        for (double v : c->host_tick_ms) t += v;
        *avg_ms = c->host_tick_ms.empty() ? 0.0 : t / (double)c->host_tick_ms.size();
        if (launches) *launches = (long)c->host_tick_ms.size();
        return LSC_OK;
    }
    double tot = 0;
    for (size_t i = 0; i < c->ev_used[which]; i++) {
        auto &p = c->ev_pool[which][i];
        HIPCHK(c, hipEventSynchronize(p.second));
        float ms = 0;
        HIPCHK(c, hipEventElapsedTime(&ms, p.first, p.second));
        tot += ms;
    }
    *avg_ms = c->ev_used[which] ? tot / (double)c->ev_used[which] : 0.0;
    if (launches) *launches = (long)c->ev_used[which];
    return LSC_OK;
}

int lsc_kernel_times_ms(lsc_ctx *c, int which, double *out_ms, long capacity, long *launches)
{
    if (!c || which < 0 || which > 5 || (capacity > 0 && !out_ms)) return LSC_EINVAL;
    if (which == 5) {
        const long nh = (long)c->host_tick_ms.size();
        for (long i = 0; i < nh && i < capacity; i++) out_ms[i] = c->host_tick_ms[(size_t)i];
        if (launches) *launches = nh;
        return LSC_OK;
    }
    const long n = (long)c->ev_used[which];
    for (long i = 0; i < n && i < capacity; i++) {
        auto &p = c->ev_pool[which][(size_t)i];
        HIPCHK(c, hipEventSynchronize(p.second));
        float ms = 0;
        HIPCHK(c, hipEventElapsedTime(&ms, p.first, p.second));
        out_ms[i] = ms;
    }
    if (launches) *launches = n;
    return LSC_OK;
}

// Diagnostics: phase profile of the plan kernel.  enable=1 switches to the instrumented kernel variant and clears
// the counters; out (may be null) receives [N][12] cycle counts (100 MHz wall clock) accumulated since then.
int lsc_phase_profile(lsc_ctx *c, int enable, long long *out)
{
    if (!c || c->N == 0) return LSC_EINVAL;
    // (checked before anything is copied: a refused call leaves `out` untouched)
    if (enable > 0 && c->cfg.world_dimension == 2) { c->err = "lsc_phase_profile: the instrumented plan kernel exists for 3-D worlds only"; return LSC_EINVAL; }
    HIPCHK(c, hipDeviceSynchronize());
    if (out) HIPCHK(c, hipMemcpy(out, c->d_prof, sizeof(long long) * PROF_PHASES * (size_t)c->N, hipMemcpyDeviceToHost));
    if (enable >= 0) {
        c->profiling = enable != 0;
        HIPCHK(c, hipMemset(c->d_prof, 0, sizeof(long long) * 2 * PROF_PHASES * (size_t)c->N));
    }
    return LSC_OK;
}

// Diagnostics: section profile of lsc_general_kernel (the alternate planner modes), collected while lsc_phase_profile is
// enabled and cleared with it.  out receives [N][16] shader cycles: set-up, start, then per interior-point section (residual
// pass, row reduction, assembly, factorization, both solves, affine pass, corrector right-hand side, its reduction and
// assembly, step), the iteration count and the number of solves of the agent.
int lsc_general_profile(lsc_ctx *c, long long *out)
{
    if (!c || c->N == 0 || !out) return LSC_EINVAL;
    HIPCHK(c, hipDeviceSynchronize());
    HIPCHK(c, hipMemcpy(out, c->d_prof + PROF_PHASES * (size_t)c->N, sizeof(long long) * PROF_PHASES * (size_t)c->N, hipMemcpyDeviceToHost));
    return LSC_OK;
}

// Diagnostics: section profile of the goal planner's register-resident search.  enable = 1 switches to the instrumented
// kernel and clears the counters, 0 switches back, < 0 only reads; out (may be null) receives [N][8] shader cycles accumulated
// since then: prologue (priority / retreat rule), grid set-up, search, path + line-of-sight goal; of the search: findMin,
// deleteMin, neighbour screening, insertions.
int lsc_goal_profile(lsc_ctx *c, int enable, long long *out)
{
    if (!c || c->N == 0) return LSC_EINVAL;
    HIPCHK(c, hipDeviceSynchronize());
    if (!c->d_goal_prof) {
        HIPCHK(c, hipMalloc(&c->d_goal_prof, sizeof(long long) * 16 * (size_t)c->N));
        HIPCHK(c, hipMemset(c->d_goal_prof, 0, sizeof(long long) * 16 * (size_t)c->N));
    }
    if (out) HIPCHK(c, hipMemcpy(out, c->d_goal_prof, sizeof(long long) * 16 * (size_t)c->N, hipMemcpyDeviceToHost));
    if (enable >= 0) {
        c->goal_profiling = enable != 0;
        HIPCHK(c, hipMemset(c->d_goal_prof, 0, sizeof(long long) * 16 * (size_t)c->N));
    }
    return LSC_OK;
}

// ---------------------------------------------------------------------------------------------------
// QP failure forensics.  On a solver failure the reference exports the model it could not solve (log/QPmodel.lp, CPLEX LP
// format, src/traj_optimizer.cpp:99-153); lsc_dump_qp writes the same file for one agent of the LAST host-buffer tick: variables
// x_m_i / y_m_i / z_m_i, rows c1.. in populatebyrow's order (:394-536: 15 equalities per axis, corridor rows, LSC rows of
// every obstacle, velocity / acceleration rows per axis, stop-at-horizon equalities), bounds (:274-303), objective as
// "linear + [ quadratic ] / 2 + constant".  Numbers are what the kernels used: LSC normals / margins from the dense sweep
// kernel on the tick's inputs, corridor boxes and planned goal from the context's device state.  LSC mode without slack
// rows (an agent whose rows carry slack variables after a disturbance has an alternate-mode QP: LSC_EINVAL).
int lsc_dump_qp(lsc_ctx *c, int agent, const char *path)
{
    if (!c || !path || c->N < 1 || agent < 0 || agent >= c->N) return LSC_EINVAL;
    if (agent < c->first || agent >= c->first + c->count) {
        // corridor boxes and plan inputs are only valid for the context's own shard: the owning rank writes the dump
        c->err = "lsc_dump_qp: the agent is not in this context's shard";
        return LSC_EINVAL;
    }
    if (c->last_host_seq < 1 || !c->h_in) { c->err = "lsc_dump_qp: no host-buffer tick has run on this context"; return LSC_ESTATE; }
    if (c->cfg.planner_mode != 0) { c->err = "lsc_dump_qp: LSC mode only"; return LSC_EINVAL; }
    const int N = c->N, nobs = N - 1, seq = c->last_host_seq;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    HIPCHK(c, hipDeviceSynchronize());
    if (c->cfg.reset_threshold > 0.0 && c->d_ever) {
        std::vector<unsigned char> ever(N);
        HIPCHK(c, hipMemcpy(ever.data(), c->d_ever, N, hipMemcpyDeviceToHost));
        for (int q = 0; q < N; q++)
            if (ever[q]) { c->err = "lsc_dump_qp: the swarm carries slack rows of a disturbance reset (alternate-mode QP)"; return LSC_EINVAL; }
    }
    const float *state = c->h_in, *prev = c->h_in + 12 * (size_t)N;
    // the agent's LSC (generateLSC) from the sweep kernel, its corridor and planned goal from the context
    std::vector<float> normal((size_t)std::max(nobs, 1) * M * 3), sfc(M * 6), goal(3);
    std::vector<double> dd((size_t)std::max(nobs, 1) * M * NC), vmax(3), amax(3);
    double vnom = 1.0;
    if (nobs > 0) {
        float *d_n = nullptr;
        double *d_d = nullptr;
        HIPCHK(c, hipMalloc(&d_n, sizeof(float) * normal.size()));
        HIPCHK(c, hipMalloc(&d_d, sizeof(double) * dd.size()));
        SweepArgs s;
        s.N = N; s.first = agent; s.count = 1; s.planner_seq = seq; s.dtf = (float)c->cfg.dt;
        s.state = c->d_state; s.traj_prev = c->d_prev;
        s.radius = c->d_radius; s.radius_obs = c->d_radius_obs; s.downwash = c->d_downwash; s.downwash_obs = c->d_downwash_obs;
        s.out_normal = d_n; s.out_d = d_d;
        hipError_t e = launch_sweep(s, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e == hipSuccess) e = hipMemcpy(normal.data(), d_n, sizeof(float) * normal.size(), hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(dd.data(), d_d, sizeof(double) * dd.size(), hipMemcpyDeviceToHost);
        (void)hipFree(d_n); (void)hipFree(d_d);
        if (e != hipSuccess) { c->err = std::string("lsc_dump_qp: ") + hipGetErrorString(e); return LSC_EHIP; }
    }
    HIPCHK(c, hipMemcpy(goal.data(), c->d_goal_cur + 3 * (size_t)agent, sizeof(float) * 3, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(vmax.data(), c->d_vmax + 3 * (size_t)agent, sizeof(double) * 3, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(amax.data(), c->d_amax + 3 * (size_t)agent, sizeof(double) * 3, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(&vnom, c->d_vnom + agent, sizeof(double), hipMemcpyDeviceToHost));
    if (c->cfg.use_octomap) HIPCHK(c, hipMemcpy(sfc.data(), c->d_sfc + (size_t)agent * M * 6, sizeof(float) * M * 6, hipMemcpyDeviceToHost));
    FILE *f = std::fopen(path, "w");
    if (!f) { c->err = std::string("lsc_dump_qp: cannot write ") + path; return LSC_EINVAL; }
    const char ax[3] = {'x', 'y', 'z'};
    const int n = DEG, phi = 3;
    const int dim = c->cfg.world_dimension == 2 ? 2 : 3;      // dim = param.world_dimension: a planar world has 60 variables (:8, 264-266)
    const double dt = c->cfg.dt, wc = c->cfg.control_weight, wt = c->cfg.terminal_weight;
    auto var = [&](int k, int m, int i) { char b[32]; std::snprintf(b, sizeof b, "%c_%d_%d", ax[k], m, i); return std::string(b); };
    struct Term { double v; std::string name; };
    auto emit = [&](const std::vector<Term> &t) {       // " - 25 x_0_0 + 25 x_0_1"; unit coefficients are left out like CPLEX does
        bool first = true;
        for (const Term &q : t) {
            if (q.v == 0.0) continue;
            const double m = std::fabs(q.v);
            std::fprintf(f, "%s", q.v < 0 ? (first ? " - " : " - ") : (first ? " " : " + "));
            if (m != 1.0) std::fprintf(f, "%.15g ", m);
            std::fprintf(f, "%s", q.name.c_str());
            first = false;
        }
        if (first) std::fprintf(f, " 0 %s", var(0, 0, 0).c_str());
    };
    // ---- objective: w_c sum x^T Q_base x + w_t sum_{m >= M - T} |c_{m,n} - goal|^2   (:329-372)
    const float *sa = state + 9 * (size_t)agent;
    int T;
    {
        const float dx = goal[0] - sa[0], dy = goal[1] - sa[1], dz = goal[2] - sa[2];
        const float n2 = dx * dx + dy * dy + dz * dz;
        const double ideal = std::sqrt((double)n2) / vnom;       // octomath norm: float32 sum of squares, double sqrt
        T = std::max((int)((M * dt - ideal + 1e-9) / dt), 1);
    }
    double Q[NC * NC];
    build_qbase(dt, Q);
    std::fprintf(f, "\\ENCODING=ISO-8859-1\n\\Problem name: lsc_planner_amd (agent %d, planner_seq %d)\n\nMinimize\n obj1:", agent, seq);
    {
        std::vector<Term> lin;
        for (int k = dim - 1; k >= 0; k--)
            for (int m = M - 1; m >= M - T; m--) lin.push_back({-2.0 * wt * (double)goal[k], var(k, m, n)});
        emit(lin);
        std::fprintf(f, " + [");
        bool first = true;
        for (int k = dim - 1; k >= 0; k--)
            for (int m = M - 1; m >= 0; m--)
                for (int i = n; i >= 0; i--)
                    for (int j = i; j >= 0; j--) {
                        // "[ ... ] / 2": the bracket holds twice the objective's quadratic coefficients
                        double v = i == j ? 2.0 * wc * Q[i * NC + i] : 2.0 * wc * (Q[i * NC + j] + Q[j * NC + i]);
                        if (i == j && i == n && m >= M - T) v += 2.0 * wt;
                        if (v == 0.0) continue;
                        std::fprintf(f, "%s%.15g %s", v < 0 ? " - " : (first ? " " : " + "), std::fabs(v), var(k, m, i).c_str());
                        if (i == j) std::fprintf(f, " ^2"); else std::fprintf(f, " * %s", var(k, m, j).c_str());
                        first = false;
                    }
        double cst = 0.0;
        for (int k = 0; k < dim; k++) cst += wt * T * (double)goal[k] * (double)goal[k];
        std::fprintf(f, " ] / 2 + %.15g\nSubject To\n", cst);
    }
    int row = 0;
    auto begin_row = [&]() { std::fprintf(f, " c%d:", ++row); };
    // ---- equalities (:394-405, Aeq_base :186-236, deq :239-259)
    const double A0[3][3] = {{1, 0, 0}, {-1, 1, 0}, {1, -2, 1}};       // A_0 rows 0..2, columns 0..2
    const double AT[3][3] = {{0, 0, 1}, {0, -1, 1}, {1, -2, 1}};       // A_T rows 0..2, columns n-2..n
    for (int k = 0; k < dim; k++) {
        int nn = 1;
        for (int i = 0; i < phi; i++) {
            std::vector<Term> t;
            for (int j = 0; j < 3; j++) t.push_back({std::pow(dt, -i) * nn * A0[i][j], var(k, 0, j)});
            begin_row(); emit(t);
            std::fprintf(f, " = %.15g\n", (double)sa[3 * i + k]);
            nn *= n - i;
        }
        for (int m = 1; m < M; m++) {
            nn = 1;
            for (int j = 0; j < phi; j++) {
                std::vector<Term> t;
                for (int q = 0; q < 3; q++) t.push_back({-std::pow(dt, -j) * nn * A0[j][q], var(k, m, q)});
                for (int q = 2; q >= 0; q--) t.push_back({std::pow(dt, -j) * nn * AT[j][q], var(k, m - 1, n - 2 + q)});
                begin_row(); emit(t);
                std::fprintf(f, " = 0\n");
                nn *= n - j;
            }
        }
    }
    const int ncs = c->cfg.n_constraint_segments > 0 ? std::min(c->cfg.n_constraint_segments, (int)M) : (int)M;
    // ---- corridor rows (:409-434; Box::convertToLSCs src/collision_constraints.cpp:37-59)
    if (c->cfg.use_octomap)
        for (int m = 0; m < ncs; m++)
            for (int k = 0; k < dim; k++)                         // Box::convertToLSCs(world_dimension): 2 dim half-spaces
                for (int side = 0; side < 2; side++)
                    for (int j = 0; j <= n; j++) {
                        if (m == 0 && j < phi) continue;
                        begin_row();
                        emit({{side == 0 ? 1.0 : -1.0, var(k, m, j)}});
                        std::fprintf(f, " >= %.15g\n", side == 0 ? (double)sfc[m * 6 + k] : -(double)sfc[m * 6 + 3 + k]);
                    }
    // ---- LSC rows (:437-466): n . (c_{m,i} - q_i) - d_i >= 0, q = the obstacle's predicted control point
    for (int oi = 0; oi < nobs; oi++) {
        const int qj = oi < agent ? oi : oi + 1;
        for (int m = 0; m < ncs; m++)
            for (int i = 0; i <= n; i++) {
                if (m == 0 && i < phi) continue;
                float q[3];
                if (seq < 2) {
                    const float *s = state + 9 * (size_t)qj;
                    const float mi = (float)((double)m + (double)i / (double)n), dtf = (float)dt;
                    for (int k = 0; k < 3; k++) { const float a1 = s[3 + k] * mi; const float a2 = a1 * dtf; q[k] = s[k] + a2; }
                } else {
                    const float *t = prev + (size_t)qj * NV;
                    const int cc = m < M - 1 ? (m + 1) * NC + i : (M - 1) * NC + n;
                    for (int k = 0; k < 3; k++) q[k] = t[k * SEGV + cc];
                }
                const float *nv = normal.data() + ((size_t)oi * M + m) * 3;
                double rhs = dd[((size_t)oi * M + m) * NC + i];
                std::vector<Term> t;
                for (int k = dim - 1; k >= 0; k--) { t.push_back({(double)nv[k], var(k, m, i)}); }
                for (int k = 0; k < dim; k++) rhs += (double)nv[k] * (double)q[k];   // (the z term only `if (dim == 3)`, :450)
                begin_row(); emit(t);
                std::fprintf(f, " >= %.15g\n", rhs);
            }
    }
    // ---- velocity / acceleration rows (:469-523)
    for (int k = 0; k < dim; k++)
        for (int m = 0; m < M; m++) {
            const double cv = std::pow(dt, -1) * n, ca = std::pow(dt, -2) * n * (n - 1);
            for (int i = 0; i < n; i++) {
                if (m == 0 && (i == 0 || i == 1)) continue;
                for (int sg = 1; sg >= -1; sg -= 2) {
                    begin_row(); emit({{sg * cv, var(k, m, i + 1)}, {-sg * cv, var(k, m, i)}});
                    std::fprintf(f, " <= %.15g\n", vmax[k]);
                }
            }
            for (int i = 0; i < n - 1; i++) {
                if (m == 0 && i == 0) continue;
                for (int sg = 1; sg >= -1; sg -= 2) {
                    begin_row(); emit({{sg * ca, var(k, m, i + 2)}, {-2.0 * sg * ca, var(k, m, i + 1)}, {sg * ca, var(k, m, i)}});
                    std::fprintf(f, " <= %.15g\n", amax[k]);
                }
            }
        }
    // ---- stop at the horizon (:529-536)
    for (int k = 0; k < dim; k++)
        for (int i = 1; i < phi; i++) {
            begin_row(); emit({{1.0, var(k, M - 1, n)}, {-1.0, var(k, M - 1, n - i)}});
            std::fprintf(f, " = 0\n");
        }
    // ---- bounds (:274-303)
    std::fprintf(f, "Bounds\n");
    for (int k = dim - 1; k >= 0; k--)
        for (int m = M - 1; m >= 0; m--)
            for (int i = n; i >= 0; i--) {
                if (m == 0 && i < 3) continue;
                std::fprintf(f, " %.15g <= %s <= %.15g\n", (double)c->cfg.world_min[k], var(k, m, i).c_str(), (double)c->cfg.world_max[k]);
            }
    for (int k = dim - 1; k >= 0; k--)
        for (int i = 0; i < 3; i++) std::fprintf(f, "      %s Free\n", var(k, 0, i).c_str());
    std::fprintf(f, "End\n");
    std::fclose(f);
    return LSC_OK;
}

// per-iteration solver trace of one agent: [64][8] = gap, |rp|, objective, affine step, sigma, step, |dx_aff|, mu
int lsc_solver_trace(lsc_ctx *c, int agent, double *out)
{
    if (!c || c->N == 0) return LSC_EINVAL;
    HIPCHK(c, hipDeviceSynchronize());
    if (!c->d_trace) { HIPCHK(c, hipMalloc(&c->d_trace, sizeof(double) * (64 * 8 + NY * KLD + W_SIZE + 512))); }
    if (out) HIPCHK(c, hipMemcpy(out, c->d_trace, sizeof(double) * (64 * 8 + NY * KLD + W_SIZE + 512), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemset(c->d_trace, 0, sizeof(double) * (64 * 8 + NY * KLD + W_SIZE + 512)));
    c->trace_agent = agent;
    return LSC_OK;
}

// last (gap, |rp|, |rd|, objective) of every agent's solve, [N][4]
int lsc_solver_residuals(lsc_ctx *c, double *out)
{
    if (!c || !out || c->N == 0) return LSC_EINVAL;
    HIPCHK(c, hipDeviceSynchronize());
    HIPCHK(c, hipMemcpy(out, c->d_dbg, sizeof(double) * 4 * (size_t)c->N, hipMemcpyDeviceToHost));
    return LSC_OK;
}

// sum over agents of the interior-point iterations run since the last reset (bench flop accounting)
int lsc_iterations_total(lsc_ctx *c, long long *total, int reset)
{
    if (!c || !total || c->N == 0) return LSC_EINVAL;
    std::vector<long long> h(c->N);
    HIPCHK(c, hipDeviceSynchronize());
    HIPCHK(c, hipMemcpy(h.data(), c->d_iters_acc, sizeof(long long) * (size_t)c->N, hipMemcpyDeviceToHost));
    long long t = 0;
    for (long long v : h) t += v;
    *total = t;
    if (reset) HIPCHK(c, hipMemset(c->d_iters_acc, 0, sizeof(long long) * (6 * (size_t)c->N)));
    return LSC_OK;
}

// sum over agents of (interior-point iterations x LSC rows the agent's QP carried) since the last reset of lsc_iterations_total:
// the row passes the kernels executed, against the 27 (N - 1) rows per iteration the reference's model holds
int lsc_row_iterations_total(lsc_ctx *c, long long *total)
{
    if (!c || !total || c->N == 0) return LSC_EINVAL;
    std::vector<long long> h(c->N);
    HIPCHK(c, hipDeviceSynchronize());
    HIPCHK(c, hipMemcpy(h.data(), c->d_iters_acc + c->N, sizeof(long long) * (size_t)c->N, hipMemcpyDeviceToHost));
    long long t = 0;
    for (long long v : h) t += v;
    *total = t;
    return LSC_OK;
}

// counters of the active-set solve since the last reset of lsc_iterations_total: agent-replans it finished, agent-replans it handed to
// the interior point, changes of the working set, interior-point iterations of the handed-over agents
int lsc_solver_stats(lsc_ctx *c, long long out[4])
{
    if (!c || !out || c->N == 0) return LSC_EINVAL;
    HIPCHK(c, hipDeviceSynchronize());
    // (kept per agent on the device -- four counters every workgroup of a tick added to were 0.6 us at the end of a 64-agent tick -- and summed here)
    std::vector<long long> h(4 * (size_t)c->N);
    HIPCHK(c, hipMemcpy(h.data(), c->d_iters_acc + 2 * (size_t)c->N, sizeof(long long) * h.size(), hipMemcpyDeviceToHost));
    for (int k = 0; k < 4; k++) out[k] = 0;
    for (int q = 0; q < c->N; q++) for (int k = 0; k < 4; k++) out[k] += h[4 * (size_t)q + k];
    return LSC_OK;
}

int lsc_set_goal_trace(lsc_ctx *c, int path_cap)
{
    if (!c || c->N == 0 || path_cap < 0) return LSC_EINVAL;
    if (c->d_goal_path) { (void)hipFree(c->d_goal_path); c->d_goal_path = nullptr; }
    if (c->d_goal_plen) { (void)hipFree(c->d_goal_plen); c->d_goal_plen = nullptr; }
    c->goal_path_cap = path_cap;
    if (path_cap > 0) {
        HIPCHK(c, hipMalloc(&c->d_goal_path, sizeof(int) * (size_t)c->N * path_cap));
        HIPCHK(c, hipMalloc(&c->d_goal_plen, sizeof(int) * (size_t)c->N));
        HIPCHK(c, hipMemset(c->d_goal_plen, 0, sizeof(int) * (size_t)c->N));
    }
    return LSC_OK;
}

int lsc_get_goal_trace(lsc_ctx *c, int *path_cells, int *path_len, int *flags, int *expansions, int grid_dims[3],
                       double grid_min[3])
{
    if (!c || c->N == 0) return LSC_EINVAL;
    if (!c->d_goal_flags) { c->err = "goal trace: the goal planner is not active (goal_mode 1 + use_octomap + distmap)"; return LSC_ESTATE; }
    HIPCHK(c, hipDeviceSynchronize());
    const size_t cnt = c->count, first = c->first;
    if (flags) HIPCHK(c, hipMemcpy(flags, c->d_goal_flags + first, sizeof(int) * cnt, hipMemcpyDeviceToHost));
    if (expansions) HIPCHK(c, hipMemcpy(expansions, c->d_goal_exp + first, sizeof(int) * cnt, hipMemcpyDeviceToHost));
    if (grid_dims) for (int k = 0; k < 3; k++) grid_dims[k] = c->grid_dims[k];
    if (grid_min) for (int k = 0; k < 3; k++) grid_min[k] = c->grid_min[k];
    if (path_len || path_cells) {
        if (!c->d_goal_path) { c->err = "goal trace: call lsc_set_goal_trace(ctx, path_cap) first"; return LSC_ESTATE; }
        std::vector<int> len(cnt), keys(cnt * (size_t)c->goal_path_cap);
        HIPCHK(c, hipMemcpy(len.data(), c->d_goal_plen, sizeof(int) * cnt, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(keys.data(), c->d_goal_path, sizeof(int) * keys.size(), hipMemcpyDeviceToHost));
        const int H = c->grid_dims[0], W = c->grid_dims[1];
        for (size_t q = 0; q < cnt; q++) {
            if (path_len) path_len[q] = len[q];
            if (!path_cells) continue;
            for (int t = 0; t < len[q] && t < c->goal_path_cap; t++) {
                const int key = keys[q * c->goal_path_cap + t];
                const int z = key / (H * W), rem = key % (H * W);
                int *o = path_cells + (q * c->goal_path_cap + t) * 3;
                o[0] = rem / W; o[1] = rem % W; o[2] = z;
            }
        }
    }
    return LSC_OK;
}

int lsc_last_goals(lsc_ctx *c, float *goals)
{
    if (!c || !goals || c->N == 0) return LSC_EINVAL;
    HIPCHK(c, hipDeviceSynchronize());
    HIPCHK(c, hipMemcpy(goals, c->d_goal_cur, sizeof(float) * 3 * (size_t)c->N, hipMemcpyDeviceToHost));
    return LSC_OK;
}

// LSC rows an agent may carry before it is handed to the pass with its rows in HBM: latency build (512 lanes, shards that fit
// the chip) and throughput build (256 lanes, two workgroups per CU; 0 when that build is not available for this context)
int lsc_row_capacity(const lsc_ctx *c, int *lds_rows, int *throughput_rows)
{
    if (!c || c->N == 0) return LSC_EINVAL;
    if (lds_rows) *lds_rows = c->cap;
    if (throughput_rows) *throughput_rows = c->cap_tp;
    return LSC_OK;
}

// rows of the fullest control-point bucket of every agent in the last tick (what the LDS row capacity has to hold)
int lsc_last_bucket_max(lsc_ctx *c, int *rows /*[N]*/)
{
    if (!c || !rows || c->N == 0) return LSC_EINVAL;
    HIPCHK(c, hipMemcpy(rows, c->d_bmax, sizeof(int) * (size_t)c->N, hipMemcpyDeviceToHost));
    return LSC_OK;
}

// read-back of per-agent active LSC row counts of the last tick (diagnostics for bench / tests)
int lsc_last_row_counts(lsc_ctx *c, int *rows /*[N]*/)
{
    if (!c || !rows || c->N == 0) return LSC_EINVAL;
    HIPCHK(c, hipMemcpy(rows, c->d_nrows, sizeof(int) * (size_t)c->N, hipMemcpyDeviceToHost));
    return LSC_OK;
}

// read-back of the neighbour-list lengths of the last tick (diagnostics)
int lsc_neighbour_counts(lsc_ctx *c, int *units /*[N]*/, int *priority_candidates /*[N] or null*/)
{
    if (!c || !units || c->N == 0) return LSC_EINVAL;
    if (!c->neigh.cnt) { c->err = "this context builds no neighbour lists (fewer than 512 agents, prune != 1, or LSC_NO_NEIGHBOUR_LISTS)"; return LSC_ESTATE; }
    HIPCHK(c, hipMemcpy(units, c->neigh.cnt, sizeof(int) * (size_t)c->N, hipMemcpyDeviceToHost));
    if (priority_candidates) {
        HIPCHK(c, hipMemcpy(priority_candidates, c->neigh.pcnt, sizeof(int) * (size_t)c->N, hipMemcpyDeviceToHost));
        for (int q = 0; q < c->N; q++) if (priority_candidates[q] >= 0) priority_candidates[q] &= 0xffff;      // (bit 30: the swarm's disturbance bit)
    }
    return LSC_OK;
}

}  // extern "C"
